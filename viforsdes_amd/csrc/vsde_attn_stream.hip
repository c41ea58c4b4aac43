// Self-attention core for sequences that do not fit the LDS-resident kernels of vsde_attn.hip (N > 544 tokens) or for
// head_dim 128 -- the synthetic stress configuration (1001 grid tokens, encoder 512 / 4 heads).  Same math, same token-major
// layout [B][N][H][D] and the same lane = token orientation of the products (see vsde_attn.hip), but the "other side" of the
// product streams through LDS in stages of TR tokens (double-buffered, one barrier per stage; TR = 32, or 128 = four 32-token
// sub-tiles for the 8-wave forward / dq kernels) and the forward keeps a running maximum (online softmax; the rescale of the
// accumulators is skipped while the maximum grows by < 2^8, so it runs once or twice per query).  D in {64, 128}; reference:
// F.scaled_dot_product_attention at primitives/attn.py:104-106.  The blocks of one (batch, head) pair share an XCD (as_block_map).
//   forward   workgroup = (batch, head, 256 queries): 8 waves x 32 queries (128 / 4 waves with VSDE_ATTN_STREAM_NT=256); K, V stream.
//   dq        same ownership; K and V tiles stream; also emits delta_i = <dO_i, O_i>.
//   dk / dv   workgroup = (batch, head, 256 keys); Q and dO tiles (+ their lse, delta) stream; head_dim 128 keeps the owned V rows
//             in LDS (no registers left for their fragments).
// Deterministic: every output element has one owner, no atomics.
#include <stdlib.h>

#include "vsde_common.h"

namespace vsde {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4s __attribute__((ext_vector_type(4)));
typedef __bf16 hbf16x2 __attribute__((ext_vector_type(2)));
typedef float f32v2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// workgroup = NT threads = NT / 64 waves x 32 owned tokens (NT / 2 tokens): 512 (8 waves) by default, 256 for A/B runs -- every
// streamed 32-token tile then serves 8 waves, half the L2 -> LDS traffic of the other side's re-reads per owned token

template <int D, int NT = 256> struct ASCfg {
    static constexpr int KS = D / 16;            // k-steps of a product contracting over the channels
    static constexpr int DB = D / 32;            // 32-channel blocks of an accumulator
    static constexpr int LD = D == 64 ? 72 : 144;   // LDS row stride (bf16): conflict-free ds_read_b64_tr_b16, <= 2-way ds_read_b128
    static constexpr int CH = D / 8;             // 16-byte chunks per row
};

struct ASParams {
    const uint16_t *q, *k, *v, *o, *dout;   // [B][N][H][D] bf16
    uint16_t *out, *dq, *dk, *dv;
    float *lse;                             // [B][H][N]
    const float *lse_in;
    float *delta;                           // [B][H][N]
    int N, H, ntile;
    float scale, scale_log2e;
};

__device__ __forceinline__ uint32_t as_pack(float a, float b) {
    const f32v2 f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, hbf16x2));
}
__device__ __forceinline__ void as_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ uint2 as_read_tr(const uint16_t *ptr) {
    bf16x4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4s *)ptr);
    return *(uint2 *)&r;
}

// [32 rows][D] tile (row stride LD) times the B fragments "column = owned token, k = channel": T[row][token]
template <int D>
__device__ __forceinline__ f32x16 as_product(const uint16_t *arow, const bf16x8 (&bfrag)[D / 16]) {
    f32x16 t = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8 *)(arow + ks * 16), bfrag[ks], t, 0, 0, 0);
    return t;
}

// acc[db] += X^T B for a row-major LDS tile X [32 tokens][D]: A operand (row = channel, contraction over the tile's tokens in
// the order B's registers hold them: 4 h2 + {0..3, 8..11 | 16..19, 24..27}) read with the hardware transpose.
template <int D>
__device__ __forceinline__ void as_accumulate_t(const uint16_t *tile, int lane, const bf16x8 &b0, const bf16x8 &b1, f32x16 (&acc)[D / 32]) {
    constexpr int LD = ASCfg<D>::LD;
    const int h2 = lane >> 5, m = lane & 15;
    const uint16_t *src = tile + (4 * h2 + (m >> 2)) * LD + ((lane >> 4) & 1) * 16 + (m & 3) * 4;
#pragma unroll
    for (int db = 0; db < D / 32; ++db) {
        const uint2 a0 = as_read_tr(src + db * 32), a1 = as_read_tr(src + db * 32 + 8 * LD);
        const uint2 a2 = as_read_tr(src + db * 32 + 16 * LD), a3 = as_read_tr(src + db * 32 + 24 * LD);
        uint4 w0 = make_uint4(a0.x, a0.y, a1.x, a1.y), w1 = make_uint4(a2.x, a2.y, a3.x, a3.y);
        acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8 *)&w0, b0, acc[db], 0, 0, 0);
        acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8 *)&w1, b1, acc[db], 0, 0, 0);
    }
}

__device__ __forceinline__ void as_pack_tile(const float (&x)[16], bf16x8 &b0, bf16x8 &b1) {
    uint4 w0 = make_uint4(as_pack(x[0], x[1]), as_pack(x[2], x[3]), as_pack(x[4], x[5]), as_pack(x[6], x[7]));
    uint4 w1 = make_uint4(as_pack(x[8], x[9]), as_pack(x[10], x[11]), as_pack(x[12], x[13]), as_pack(x[14], x[15]));
    b0 = *(bf16x8 *)&w0; b1 = *(bf16x8 *)&w1;
}

// B fragments of one owned token row (zero when the token is padding)
template <int D>
__device__ __forceinline__ void as_load_frag(const uint16_t *base, int64_t ts, int row, bool ok, int h2, bf16x8 (&f)[D / 16]) {
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
        uint4 t = make_uint4(0, 0, 0, 0);
        if (ok) t = *(const uint4 *)(base + row * ts + ks * 16 + h2 * 8);
        f[ks] = *(bf16x8 *)&t;
    }
}

// accumulator blocks (rows = channels, column = owned token) -> that token's bf16 row, scaled
template <int D>
__device__ __forceinline__ void as_store_t(uint16_t *row, int h2, const f32x16 (&a)[D / 32], float mul) {
#pragma unroll
    for (int db = 0; db < D / 32; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *(uint2 *)(row + db * 32 + 8 * g + 4 * h2) = make_uint2(as_pack(a[db][4 * g] * mul, a[db][4 * g + 1] * mul),
                                                                    as_pack(a[db][4 * g + 2] * mul, a[db][4 * g + 3] * mul));
}

// tile `t` (32 token rows, zero beyond N) of two row-major operands: global -> registers / registers -> LDS
template <int D, int NT, int TR>
__device__ __forceinline__ void as_tile_load(u32x4 (&ra)[TR * (D / 8) / NT], u32x4 (&rb)[TR * (D / 8) / NT], const uint16_t *a, const uint16_t *b,
                                             int64_t ts, int t, int N, int tid) {
    constexpr int CH = ASCfg<D>::CH;
#pragma unroll
    for (int i = 0; i < TR * CH / NT; ++i) {
        const int idx = tid + NT * i, row = idx / CH, c = idx % CH, n = t * TR + row;
        const u32x4 z = {0u, 0u, 0u, 0u};
        ra[i] = n < N ? *(const u32x4 *)(a + n * ts + c * 8) : z;
        rb[i] = n < N ? *(const u32x4 *)(b + n * ts + c * 8) : z;
    }
}
template <int D, int NT, int TR>
__device__ __forceinline__ void as_tile_store(const u32x4 (&ra)[TR * (D / 8) / NT], const u32x4 (&rb)[TR * (D / 8) / NT], uint16_t *sa, uint16_t *sb, int tid) {
    constexpr int CH = ASCfg<D>::CH, LD = ASCfg<D>::LD;
#pragma unroll
    for (int i = 0; i < TR * CH / NT; ++i) {
        const int idx = tid + NT * i, row = idx / CH, c = idx % CH;
        *(u32x4 *)(sa + row * LD + c * 8) = ra[i];
        *(u32x4 *)(sb + row * LD + c * 8) = rb[i];
    }
}


// Workgroup -> ((batch, head) pair, owned-token block).  The hardware deals consecutive workgroup ids round-robin over the 8 XCDs
// (one L2 each); the blocks of one pair all stream the same K / V (or Q / dO) rows, so they are given consecutive slots of ONE
// XCD: they run at the same time on the same L2 and the pair's rows come from HBM once instead of once per block.
__device__ __forceinline__ void as_block_map(int &pair, int &blk) {
    const int nblk = gridDim.y, npair = gridDim.x;
    if (npair & 7) { pair = blockIdx.x; blk = blockIdx.y; return; }
    const int L = blockIdx.x + npair * blockIdx.y, xcd = L & 7, slot = L >> 3;
    blk = slot % nblk;
    pair = (slot / nblk) * 8 + xcd;
}

// ------------------------------------------------------------------------------------------------------ forward
template <int D, int NT, int TR>
__global__ void __launch_bounds__(NT) attn_fwd_stream_kernel(ASParams p) {
    constexpr int LD = ASCfg<D>::LD, DB = ASCfg<D>::DB, TILE = TR * LD, SUBS = TR / 32;   // TR streamed rows per stage (one barrier)
    __shared__ __attribute__((aligned(16))) uint16_t Ks[2 * TILE], Vs[2 * TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 31, h2 = lane >> 5;
    int pair, blk;
    as_block_map(pair, blk);
    const int b = pair / p.H, hh = pair - b * p.H, N = p.N;
    const int64_t ts = (int64_t)p.H * D, base = ((int64_t)b * N * p.H + hh) * D;
    const uint16_t *kb = p.k + base, *vb = p.v + base;
    const int query = blk * (NT / 2) + wave * 32 + fr;
    const bool qok = query < N;
    bf16x8 qf[D / 16];
    as_load_frag<D>(p.q + base, ts, query, qok, h2, qf);
    u32x4 rk[TR * (D / 8) / NT], rv[TR * (D / 8) / NT];
    as_tile_load<D, NT, TR>(rk, rv, kb, vb, ts, 0, N, tid);
    as_tile_store<D, NT, TR>(rk, rv, Ks, Vs, tid);
    as_barrier();
    const int nstage = (N + TR - 1) / TR;
    if (nstage > 1) as_tile_load<D, NT, TR>(rk, rv, kb, vb, ts, 1, N, tid);
    const float c2 = p.scale_log2e;
    float m = -INFINITY, lsum = 0.f;   // running maximum (log2 units, equal in the two lanes of a query) and this lane's share of the sum
    f32x16 o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[db][e] = 0.f;
    const bool ragged = (N & 31) != 0;
    for (int st = 0; st < nstage; ++st) {
        // Software pipeline inside the stage: the score product of sub-tile i + 1 is issued before the exponentials of sub-tile i,
        // so the matrix pipe works on it while the VALU runs the softmax (tools/probes/overlap_probe.hip: the two only overlap when
        // independent work of both kinds is in flight -- waves in lock step after a barrier otherwise run phase after phase).
        f32x16 s = as_product<D>(Ks + (st & 1) * TILE + fr * LD + h2 * 8, qf);   // S^T [key][query] of the stage's first sub-tile
#pragma unroll
      for (int sub = 0; sub < SUBS; ++sub) {
        const int kt = st * SUBS + sub;
        if (kt >= p.ntile) break;   // the last stage's tail lies beyond the sequence (workgroup-uniform)
        const uint16_t *vt_ = Vs + (st & 1) * TILE + sub * 32 * LD;
        if (ragged && kt == p.ntile - 1) {   // wave-uniform branch: only the last tile pays for the mask
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2 >= N) s[r] = -INFINITY;
            asm volatile("" ::: "memory");   // keeps the branch (if-converted, the 48 mask instructions would run for every tile)
        }
        float mx = fmaxf(s[0], s[1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) mx = fmaxf(fmaxf(s[r], s[r + 1]), mx);   // v_max3_f32
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c2;   // scale > 0
        if (__any(mx > m + 8.0f)) {   // wave-uniform: rescale every accumulator of the wave (factor 1 where the maximum held)
            const float mn = fmaxf(m, mx), alpha = fast_exp2(m - mn);   // m = -inf: alpha = 0
            lsum *= alpha;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
            m = mn;
        }
        // the next sub-tile's score product shares a basic block with the exponentials (no branch between them)
        f32x16 sn = s;
        // (unconditional inside a stage: beyond the sequence it multiplies stale rows and nobody reads the result)
        if (sub + 1 < SUBS) sn = as_product<D>(Ks + (st & 1) * TILE + (sub + 1) * 32 * LD + fr * LD + h2 * 8, qf);
        float pr[16], l2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            pr[r] = fast_exp2(fmaf(s[r], c2, -m)); lsum += pr[r];
            pr[r + 1] = fast_exp2(fmaf(s[r + 1], c2, -m)); l2 += pr[r + 1];
        }
        lsum += l2;
        bf16x8 pb0, pb1;
        as_pack_tile(pr, pb0, pb1);
        as_accumulate_t<D>(vt_, lane, pb0, pb1, o);   // O^T += V^T P^T
        s = sn;
      }
        if (st + 1 < nstage) as_tile_store<D, NT, TR>(rk, rv, Ks + ((st + 1) & 1) * TILE, Vs + ((st + 1) & 1) * TILE, tid);
        as_barrier();
        if (st + 2 < nstage) as_tile_load<D, NT, TR>(rk, rv, kb, vb, ts, st + 2, N, tid);
    }
    lsum += __shfl_xor(lsum, 32, 64);
    if (qok) {
        as_store_t<D>(p.out + base + query * ts, h2, o, 1.0f / lsum);
        if (h2 == 0) p.lse[((int64_t)b * p.H + hh) * N + query] = (m + __log2f(lsum)) * 0.6931471805599453f;
    }
}

// ------------------------------------------------------------------------------------------------------ backward: dq
template <int D, int NT, int TR>
__global__ void __launch_bounds__(NT) attn_bwd_dq_stream_kernel(ASParams p) {
    constexpr int LD = ASCfg<D>::LD, DB = ASCfg<D>::DB, TILE = TR * LD, SUBS = TR / 32;   // TR streamed rows per stage (one barrier)
    __shared__ __attribute__((aligned(16))) uint16_t Ks[2 * TILE], Vs[2 * TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 31, h2 = lane >> 5;
    int pair, blk;
    as_block_map(pair, blk);
    const int b = pair / p.H, hh = pair - b * p.H, N = p.N;
    const int64_t ts = (int64_t)p.H * D, base = ((int64_t)b * N * p.H + hh) * D, srow = ((int64_t)b * p.H + hh) * N;
    const uint16_t *kb = p.k + base, *vb = p.v + base;
    const int query = blk * (NT / 2) + wave * 32 + fr;
    const bool qok = query < N;
    bf16x8 qf[D / 16], dof[D / 16];
    float dsum = 0.f;   // delta_i = <dO_i, O_i>: this lane holds half of the channels of its query
    {
        bf16x8 of[D / 16];
        as_load_frag<D>(p.q + base, ts, query, qok, h2, qf);
        as_load_frag<D>(p.dout + base, ts, query, qok, h2, dof);
        as_load_frag<D>(p.o + base, ts, query, qok, h2, of);
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                dsum = fmaf(__uint_as_float(((uint32_t)(uint16_t)dof[ks][e]) << 16), __uint_as_float(((uint32_t)(uint16_t)of[ks][e]) << 16), dsum);
    }
    dsum += __shfl_xor(dsum, 32, 64);
    if (qok && h2 == 0) p.delta[srow + query] = dsum;
    const float lse2 = (qok ? p.lse_in[srow + query] : INFINITY) * 1.4426950408889634f;   // padded queries: P = 0
    u32x4 rk[TR * (D / 8) / NT], rv[TR * (D / 8) / NT];
    as_tile_load<D, NT, TR>(rk, rv, kb, vb, ts, 0, N, tid);
    as_tile_store<D, NT, TR>(rk, rv, Ks, Vs, tid);
    as_barrier();
    const int nstage = (N + TR - 1) / TR;
    if (nstage > 1) as_tile_load<D, NT, TR>(rk, rv, kb, vb, ts, 1, N, tid);
    const float c2 = p.scale_log2e;
    const bool ragged = (N & 31) != 0;
    f32x16 acc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[db][e] = 0.f;
    for (int st = 0; st < nstage; ++st) {
        // dQ^T += K^T dS^T of sub-tile i - 1 is issued together with the exponentials of sub-tile i (one basic block: the matrix
        // pipe and the VALU then overlap inside the wave, see the forward kernel); the last one of a stage follows the loop
        bf16x8 pb0 = {0, 0, 0, 0, 0, 0, 0, 0}, pb1 = pb0;
        int done = 0;
#pragma unroll
      for (int sub = 0; sub < SUBS; ++sub) {
        const int kt = st * SUBS + sub;
        if (kt >= p.ntile) break;
        const uint16_t *kt_ = Ks + (st & 1) * TILE + sub * 32 * LD, *vt_ = Vs + (st & 1) * TILE + sub * 32 * LD;
        f32x16 stl = as_product<D>(kt_ + fr * LD + h2 * 8, qf);          // S^T  [key][query]
        const f32x16 dpt = as_product<D>(vt_ + fr * LD + h2 * 8, dof);   // dP^T [key][query]
        if (ragged && kt == p.ntile - 1) {   // wave-uniform branch: zero rows of K have P != 0 -- score -inf makes it 0
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2 >= N) stl[r] = -INFINITY;
            asm volatile("" ::: "memory");   // keeps the branch (if-converted, the mask instructions would run for every tile)
        }
        if (sub > 0) as_accumulate_t<D>(kt_ - 32 * LD, lane, pb0, pb1, acc);   // sub-tile i - 1 (zero fragments would also be harmless)
        float ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) ds[r] = fast_exp2(fmaf(stl[r], c2, -lse2)) * (dpt[r] - dsum);
        as_pack_tile(ds, pb0, pb1);
        done = sub + 1;
      }
        as_accumulate_t<D>(Ks + (st & 1) * TILE + (done - 1) * 32 * LD, lane, pb0, pb1, acc);   // dQ^T += K^T dS^T, last sub-tile
        if (st + 1 < nstage) as_tile_store<D, NT, TR>(rk, rv, Ks + ((st + 1) & 1) * TILE, Vs + ((st + 1) & 1) * TILE, tid);
        as_barrier();
        if (st + 2 < nstage) as_tile_load<D, NT, TR>(rk, rv, kb, vb, ts, st + 2, N, tid);
    }
    if (qok) as_store_t<D>(p.dq + base + query * ts, h2, acc, p.scale);
}

// ------------------------------------------------------------------------------------------------- backward: dk, dv
template <int D, int NT, int TR, bool VLDS>
__global__ void __launch_bounds__(NT) attn_bwd_dkv_stream_kernel(ASParams p) {
    constexpr int LD = ASCfg<D>::LD, DB = ASCfg<D>::DB, TILE = TR * LD, SUBS = TR / 32;   // TR streamed rows per stage (one barrier)
    __shared__ __attribute__((aligned(16))) uint16_t Qs[2 * TILE], Os[2 * TILE];
    // VLDS: the owned V rows wait in LDS instead of 32 registers per lane (the B fragments of dP = dO V^T are read per tile)
    __shared__ __attribute__((aligned(16))) uint16_t Vown[VLDS ? (NT / 2) * LD : 8];
    __shared__ __attribute__((aligned(16))) float lse2s[2 * TR], dels[2 * TR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 31, h2 = lane >> 5;
    int pair, blk;
    as_block_map(pair, blk);
    const int b = pair / p.H, hh = pair - b * p.H, N = p.N;
    const int64_t ts = (int64_t)p.H * D, base = ((int64_t)b * N * p.H + hh) * D, srow = ((int64_t)b * p.H + hh) * N;
    const uint16_t *qb = p.q + base, *dob = p.dout + base;
    const int key = blk * (NT / 2) + wave * 32 + fr;
    const bool kok = key < N;
    bf16x8 kf[D / 16], vf[VLDS ? 1 : D / 16];
    as_load_frag<D>(p.k + base, ts, key, kok, h2, kf);
    if constexpr (VLDS) {
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) {
            uint4 t = make_uint4(0, 0, 0, 0);
            if (kok) t = *(const uint4 *)(p.v + base + key * ts + ks * 16 + h2 * 8);
            *(uint4 *)(Vown + (wave * 32 + fr) * LD + ks * 16 + h2 * 8) = t;   // read back by this wave only
        }
    } else {
        as_load_frag<D>(p.v + base, ts, key, kok, h2, *(bf16x8(*)[D / 16]) & vf);
    }
    u32x4 rq[TR * (D / 8) / NT], rdo[TR * (D / 8) / NT];
    float rl = 0.f, rd = 0.f;   // per-query statistics of the tile in flight (threads 0..31)
    auto stat_load = [&](int t) {
        if (tid < TR) {
            const int n = t * TR + tid;
            rl = n < N ? p.lse_in[srow + n] * 1.4426950408889634f : INFINITY;   // padded queries: P = 0
            rd = n < N ? p.delta[srow + n] : 0.f;
        }
    };
    auto stat_store = [&](int buf) { if (tid < TR) { lse2s[buf * TR + tid] = rl; dels[buf * TR + tid] = rd; } };
    as_tile_load<D, NT, TR>(rq, rdo, qb, dob, ts, 0, N, tid); stat_load(0);
    as_tile_store<D, NT, TR>(rq, rdo, Qs, Os, tid); stat_store(0);
    as_barrier();
    const int nstage = (N + TR - 1) / TR;
    if (nstage > 1) { as_tile_load<D, NT, TR>(rq, rdo, qb, dob, ts, 1, N, tid); stat_load(1); }
    const float c2 = p.scale_log2e;
    f32x16 dk[DB], dv[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dk[db][e] = 0.f; dv[db][e] = 0.f; }
    for (int st = 0; st < nstage; ++st) {
        const int buf = st & 1;
#pragma unroll
      for (int sub = 0; sub < SUBS; ++sub) {
        if (st * SUBS + sub >= p.ntile) break;
        const uint16_t *qt_ = Qs + buf * TILE + sub * 32 * LD, *dot_ = Os + buf * TILE + sub * 32 * LD;
        const f32x16 sc = as_product<D>(qt_ + fr * LD + h2 * 8, kf);     // S  [query][key]
        f32x16 dp;                                                        // dP [query][key]
        if constexpr (VLDS) {
            dp = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            const uint16_t *arow = dot_ + fr * LD + h2 * 8, *brow = Vown + (wave * 32 + fr) * LD + h2 * 8;
#pragma unroll
            for (int ks = 0; ks < D / 16; ++ks)
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8 *)(arow + ks * 16), *(const bf16x8 *)(brow + ks * 16), dp, 0, 0, 0);
        } else {
            dp = as_product<D>(dot_ + fr * LD + h2 * 8, *(const bf16x8(*)[D / 16]) & vf);
        }
        float pr[16], ds[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {   // registers 4g..4g+3 are queries 8g + 4h2 + 0..3 of the tile
            const float4 l4 = *(const float4 *)(lse2s + buf * TR + sub * 32 + 8 * g + 4 * h2);
            const float4 d4 = *(const float4 *)(dels + buf * TR + sub * 32 + 8 * g + 4 * h2);
            const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                pr[r] = fast_exp2(fmaf(sc[r], c2, -lv[e]));
                ds[r] = pr[r] * (dp[r] - dl[e]);
            }
        }
        bf16x8 p0, p1, s0, s1;
        as_pack_tile(pr, p0, p1);
        as_pack_tile(ds, s0, s1);
        as_accumulate_t<D>(dot_, lane, p0, p1, dv);   // dV^T += dO^T P
        as_accumulate_t<D>(qt_, lane, s0, s1, dk);    // dK^T += Q^T dS
      }
        if (st + 1 < nstage) { as_tile_store<D, NT, TR>(rq, rdo, Qs + (1 - buf) * TILE, Os + (1 - buf) * TILE, tid); stat_store(1 - buf); }
        as_barrier();
        if (st + 2 < nstage) { as_tile_load<D, NT, TR>(rq, rdo, qb, dob, ts, st + 2, N, tid); stat_load(st + 2); }
    }
    if (kok) {
        as_store_t<D>(p.dk + base + key * ts, h2, dk, p.scale);
        as_store_t<D>(p.dv + base + key * ts, h2, dv, 1.0f);
    }
}

template <int D, int NT, int TR>
static int as_forward(const ASParams &p, int64_t BH, hipStream_t s) {
    hipLaunchKernelGGL((attn_fwd_stream_kernel<D, NT, TR>), dim3((unsigned)BH, (p.N + NT / 2 - 1) / (NT / 2)), dim3(NT), 0, s, p);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}
template <int D, int NT, int TR>
static int as_backward(const ASParams &p, int64_t BH, hipStream_t s) {
    const dim3 grid((unsigned)BH, (p.N + NT / 2 - 1) / (NT / 2));
    hipLaunchKernelGGL((attn_bwd_dq_stream_kernel<D, NT, TR>), grid, dim3(NT), 0, s, p);    // also writes delta, read by the next kernel
    // head_dim 128: no registers for wider stages, V rows in LDS; head_dim 64 has both to spare
    hipLaunchKernelGGL((attn_bwd_dkv_stream_kernel<D, NT, (D == 64 && NT == 512) ? 128 : 32, (D == 128 && NT == 512)>), grid, dim3(NT), 0, s, p);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}
constexpr int AS_WIDE_TR = 128;   // streamed rows per stage of the 8-wave kernels
// VSDE_ATTN_STREAM_NT=256: four-wave workgroups with 32-token stages (A/B runs)
static bool as_wide() {
    static int v = -1;
    if (v < 0) v = vsde_knob("VSDE_ATTN_STREAM_NT", 512) == 512 ? 1 : 0;
    return v != 0;
}

int launch_attention_stream_fwd(const void *q, const void *k, const void *v, void *o, float *lse, int64_t B, int N, int H, int D,
                                double scale, hipStream_t s) {
    ASParams p = {};
    p.q = (const uint16_t *)q; p.k = (const uint16_t *)k; p.v = (const uint16_t *)v; p.out = (uint16_t *)o; p.lse = lse;
    p.N = N; p.H = H; p.ntile = (N + 31) / 32; p.scale = (float)scale; p.scale_log2e = (float)(scale * 1.4426950408889634);
    if (D == 64) return as_wide() ? as_forward<64, 512, AS_WIDE_TR>(p, B * H, s) : as_forward<64, 256, 32>(p, B * H, s);
    return as_wide() ? as_forward<128, 512, AS_WIDE_TR>(p, B * H, s) : as_forward<128, 256, 32>(p, B * H, s);
}

int launch_attention_stream_bwd(const void *dout, const void *q, const void *k, const void *v, const void *o, const float *lse, void *dq,
                                void *dk, void *dv, float *delta, int64_t B, int N, int H, int D, double scale, hipStream_t s) {
    ASParams p = {};
    p.q = (const uint16_t *)q; p.k = (const uint16_t *)k; p.v = (const uint16_t *)v; p.o = (const uint16_t *)o; p.dout = (const uint16_t *)dout;
    p.lse_in = lse; p.delta = delta; p.dq = (uint16_t *)dq; p.dk = (uint16_t *)dk; p.dv = (uint16_t *)dv;
    p.N = N; p.H = H; p.ntile = (N + 31) / 32; p.scale = (float)scale; p.scale_log2e = (float)(scale * 1.4426950408889634);
    if (D == 64) return as_wide() ? as_backward<64, 512, AS_WIDE_TR>(p, B * H, s) : as_backward<64, 256, 32>(p, B * H, s);
    return as_wide() ? as_backward<128, 512, AS_WIDE_TR>(p, B * H, s) : as_backward<128, 256, 32>(p, B * H, s);
}

}  // namespace vsde

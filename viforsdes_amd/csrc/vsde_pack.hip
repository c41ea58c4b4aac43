// Refresh of the encoder's cached bf16 GEMM operands after an optimizer step (gfx950).
//
// The reference re-casts every nn.Linear weight to bf16 inside each forward under autocast (primitives/attn.py:46-54,
// primitives/mlp.py:41-54: one cast kernel per Linear per step, plus the cat / pad this build's merged projections would
// need).  Here the bf16 operands -- concatenated ([q | k | v | gate]), zero-padded and 16-row interleaved (SwiGLU) packs and
// their transposes for the input-gradient GEMMs -- are persistent buffers; after the optimizer step ONE launch re-fills all
// of them from the fp32 parameters.  The work is a table of tiles (built once by the host, viforsdes_amd/primitives/fused.py):
// tile = up to 16 consecutive rows x all columns of one row block of one parameter.  A thread owns a column: it reads its 16
// fp32 values (coalesced across the workgroup), stores them as bf16 into the pack's rows (coalesced) and as one 32-byte run
// into the transposed copy.  ~8 M parameters: 33 MB read + 33 MB written.
#include "vsde_common.h"

namespace vsde {

struct PackTile {       // 64 bytes; all pointers device pointers
    const float *src;   // first element of the tile's first row in the fp32 parameter
    uint16_t *dst;      // same position in the packed bf16 operand
    uint16_t *dst_t;    // dst_t + col * pitch_t = position of (this row run, col) in the transposed copy, or nullptr
    int64_t src_pitch;  // row pitches in elements
    int64_t dst_pitch;
    int64_t pitch_t;
    int32_t rows;       // 1..16
    int32_t cols;
    int64_t reserved;
};
static_assert(sizeof(PackTile) == 64, "the host builds the table as int64 [n_tiles][8]");

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }

__global__ void __launch_bounds__(256) pack_refresh_kernel(const PackTile *tiles) {
    const PackTile t = tiles[blockIdx.x];
    for (int c = threadIdx.x; c < t.cols; c += 256) {
        uint16_t v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = r < t.rows ? f32_to_bf16_rne(t.src[(int64_t)r * t.src_pitch + c]) : (uint16_t)0;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (r < t.rows) t.dst[(int64_t)r * t.dst_pitch + c] = v[r];
        if (t.dst_t != nullptr) {
            uint16_t *o = t.dst_t + (int64_t)c * t.pitch_t;
            if (t.rows == 16 && (((uintptr_t)o) & 15) == 0) {
                u32x4 lo, hi;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo[e] = (uint32_t)v[2 * e] | ((uint32_t)v[2 * e + 1] << 16);
                    hi[e] = (uint32_t)v[8 + 2 * e] | ((uint32_t)v[8 + 2 * e + 1] << 16);
                }
                *(u32x4 *)o = lo; *(u32x4 *)(o + 8) = hi;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (r < t.rows) o[r] = v[r];
            }
        }
    }
}

}  // namespace vsde

using namespace vsde;

extern "C" int vsde_pack_tile_bytes(void) { return (int)sizeof(PackTile); }

extern "C" int vsde_pack_refresh(const void *tiles, int n_tiles, void *stream) {
    VSDE_CHECK_ARG(n_tiles >= 0 && (n_tiles == 0 || tiles != nullptr), VSDE_E_BADARG, "bad pack-refresh table");
    if (n_tiles == 0) return 0;
    hipLaunchKernelGGL(pack_refresh_kernel, dim3((unsigned)n_tiles), dim3(256), 0, (hipStream_t)stream, (const PackTile *)tiles);
    VSDE_CHECK_HIP(hipGetLastError());
    return 0;
}

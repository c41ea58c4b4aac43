"""GPU: the encoder's bf16 MFMA GEMM kernels (csrc/vsde_linear.hip) against an fp32 torch reference of the same op.

y = x W^T + b with bf16 inputs, fp32 accumulation, bf16 output: tolerance 1e-2 relative to the output's max magnitude
(one bf16 rounding of the result, |y| <= ~40 here); the SwiGLU epilogues are checked against the unfused chain of
primitives/mlp.py:21-24 evaluated in fp32 from the same bf16-rounded intermediates (2e-2)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rand(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(torch.bfloat16)


def _rel(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-30))


# (M, N, K): the LV encoder's forward and input-gradient shapes, the C=128 fixture's, the synthetic C=512 ones, ragged M
SHAPES = [(4264, 448, 128), (4264, 128, 128), (4264, 128, 384), (4264, 128, 448), (4264, 128, 768),
          (20000, 832, 256), (20000, 256, 256), (20000, 256, 768), (20000, 256, 832), (20000, 256, 1536),
          (3000, 1664 + 128, 512), (3000, 512, 512), (9000, 1664, 512), (3000, 512, 1408), (3000, 512, 1664), (3000, 512, 2816),
          (257, 256, 256), (31, 128, 64),
          # more than one round of resident workgroups: the stripes of the last, partly filled round run in finer column chunks
          # (133,000 rows = 520 stripes of 256 on 512 slots), K = 512 on eight-wave workgroups with a ragged last stripe
          (133000, 256, 256), (133000, 832, 256), (66000 + 77, 512, 512)]
# the persistent deep-reduction kernel (opt-in: VSDE_DEEP_GEMM=1, M >= 32768): LV and config-5 shapes, ragged M, a row range that
# is not a multiple of the 256-row tile, M just past a tile boundary
DEEP_SHAPES = [(40000, 256, 768), (33001, 256, 832), (65536 + 32, 256, 1536), (35001, 512, 1408), (32768, 512, 512), (70000, 256, 1664)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_plain_linear_matches_fp32_reference(M, N, K):
    from viforsdes_amd import _hip
    assert _hip.linear_supported(M, N, K)
    x, w, b = _rand(M, K, seed=1), _rand(N, K, scale=K ** -0.5, seed=2), _rand(N, seed=3)
    ref = x.float() @ w.float().t() + b.float()
    y = _hip.linear_bf16(x, w, b)
    assert _rel(y, ref) < 1e-2
    y2 = _hip.linear_bf16(x, w, None)
    assert _rel(y2, x.float() @ w.float().t()) < 1e-2
    # row-pitched input (a column range of a wider buffer) and a row-pitched output
    wide = _rand(M, K + 64, seed=4)
    out = torch.zeros(M, N + 8, device=DEV, dtype=torch.bfloat16)
    _hip.linear_bf16(wide[:, 32:32 + K] if (32 * 2) % 16 == 0 else wide[:, :K], w, b, out=out[:, :N])
    assert _rel(out[:, :N], wide[:, 32:32 + K].float() @ w.float().t() + b.float()) < 1e-2 and float(out[:, N:].abs().max()) == 0.0


def _interleave(w_a, w_b):
    """Packed row order of the SwiGLU input projection: blocks of 16 rows, a then b (see primitives/fused.py)."""
    H = w_a.shape[0]
    return torch.stack([w_a.reshape(H // 16, 16, -1), w_b.reshape(H // 16, 16, -1)], dim=1).reshape(2 * H, -1)


@pytest.mark.parametrize("M,H,K", [(20000, 768, 256), (4264, 384, 128), (300, 64, 128), (9000, 1408, 512), (200, 128, 512),
                                   # >= 512 stripes: three uneven column chunks (704 = 11 tile pairs), tail chunks; K = 512 on eight waves
                                   (133000 + 5, 704, 256), (66000 + 77, 1408, 512), (205312, 704, 256)])   # the last: the LV benchmark's shape
def test_swiglu_epilogues(M, H, K):
    from viforsdes_amd import _hip
    x = _rand(M, K, seed=5)
    w_a, w_b = _rand(H, K, scale=K ** -0.5, seed=6), _rand(H, K, scale=K ** -0.5, seed=7)
    b_a, b_b = _rand(H, seed=8), _rand(H, seed=9)
    w = _interleave(w_a, w_b).contiguous()
    bias = _interleave(b_a[:, None], b_b[:, None]).reshape(-1).contiguous()
    u, s = _hip.linear_swiglu_bf16(x, w, bias)
    a_ref = (x.float() @ w_a.float().t() + b_a.float()).to(torch.bfloat16).float()
    b_ref = (x.float() @ w_b.float().t() + b_b.float()).to(torch.bfloat16).float()
    u_ref = _interleave(a_ref.t(), b_ref.t()).t()
    assert _rel(u, u_ref) < 1e-2
    # s from the kernel's own u (isolates the epilogue arithmetic from the GEMM rounding)
    ua = u.float().reshape(M, H // 16, 2, 16)[:, :, 0].reshape(M, H)
    ub = u.float().reshape(M, H // 16, 2, 16)[:, :, 1].reshape(M, H)
    s_ref = (ua * torch.sigmoid(ua)).to(torch.bfloat16).float() * ub
    assert _rel(s, s_ref) < 1e-2
    _, s_only = _hip.linear_swiglu_bf16(x, w, bias, want_u=False)
    assert torch.equal(s_only, s)
    # backward epilogue: du = swiglu'(u) * (dy W2) with W2 [K2, H] the output projection
    K2 = K
    dy, w2 = _rand(M, K2, seed=10), _rand(K2, H, scale=H ** -0.5, seed=11)
    du = _hip.linear_swiglu_bwd_bf16(dy, w2.t().contiguous(), u)
    ds = dy.float() @ w2.float()
    sg = torch.sigmoid(ua)
    da, db = ds * ub * sg * (1 + ua * (1 - sg)), ds * ua * sg
    du_ref = _interleave(da.t(), db.t()).t()
    assert _rel(du, du_ref) < 2e-2


@pytest.mark.gpu
def test_pack_refresh_kernel_refills_every_pack_kind():
    """vsde_pack_refresh (one launch over a tile table) against the torch piece copies: plain, row-stacked ([qkv | gate]),
    zero-padded and 16-row interleaved SwiGLU packs with odd widths (682 -> 768), biases, and the transposed copies the
    input-gradient GEMMs read.  Reference semantics: the bf16 cast of each Linear weight under autocast
    (primitives/attn.py:46-54, primitives/mlp.py:41-54)."""
    import torch
    from viforsdes_amd.primitives import fused
    dev = "cuda:0"
    g = torch.Generator().manual_seed(1)
    P = lambda *s: torch.nn.Parameter(torch.randn(*s, generator=g).to(dev))
    qkv_w, qkv_b, gate_w, gate_b = P(768, 256), P(768), P(64, 256), P(64)
    w_in, b_in, w_out, b_out = P(2 * 682, 256), P(2 * 682), P(256, 682), P(256)
    odd_w, odd_b = P(40, 24), P(40)
    packs = [fused.row_pack([qkv_w, gate_w], [qkv_b, gate_b]), fused.plain_pack(odd_w, odd_b)]
    packs += list(fused.swiglu_packs(w_in, b_in, w_out, b_out, 768, interleave=True))
    packs += list(fused.swiglu_packs(w_in, None, w_out, None, 768, interleave=False))
    for pk in packs:
        pk.operands()
        pk.transposed()
    with torch.no_grad():   # an update that does not bump Tensor._version, like the fused AdamW kernel
        for p in (qkv_w, qkv_b, gate_w, gate_b, w_in, b_in, w_out, b_out, odd_w, odd_b):
            torch._foreach_add_([p.data], 0.37)
    expect = []
    for pk in packs:
        w = torch.zeros_like(pk.weight)
        b = None if pk.bias is None else torch.zeros_like(pk.bias)
        for p, s0, n, d0 in pk.weight_pieces:
            w[d0:d0 + n, :p.shape[1]] = p.detach()[s0:s0 + n].to(torch.bfloat16)
        for p, s0, n, d0 in pk.bias_pieces:
            b[d0:d0 + n] = p.detach()[s0:s0 + n].to(torch.bfloat16)
        expect.append((w, b))
    assert not torch.equal(packs[0].weight, expect[0][0])
    fused.PackedWeight.refresh_all(force=True)
    torch.cuda.synchronize()
    for pk, (w, b) in zip(packs, expect):
        assert torch.equal(pk.weight, w) and (b is None or torch.equal(pk.bias, b))
        assert torch.equal(pk.weight_t, w.t().contiguous()) and torch.equal(pk.transposed(), w.t().contiguous())
        assert not pk.stale()


@pytest.mark.ablation_build
@pytest.mark.gpu
def test_persistent_deep_reduction_kernel():
    """lin_deep_kernel (persistent workgroups, both operands through LDS by global_load_lds, counted waits) is opt-in
    (VSDE_DEEP_GEMM=1, read once per process): its shapes run in a child process with the switch on."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "from viforsdes_amd import _hip\n"
        "for M, N, K in %r:\n"
        "    assert _hip.linear_variant(M, N, K) == 3, (M, N, K)\n"
        "    g = torch.Generator().manual_seed(M)\n"
        "    x = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16)\n"
        "    w = (torch.randn(N, K, generator=g) * K ** -0.5).cuda().to(torch.bfloat16)\n"
        "    b = torch.randn(N, generator=g).cuda().to(torch.bfloat16)\n"
        "    ref = x.float() @ w.float().t() + b.float()\n"
        "    y = _hip.linear_bf16(x, w, b)\n"
        "    err = float((y.float() - ref).abs().max() / ref.abs().max())\n"
        "    assert err < 1e-2, (M, N, K, err)\n"
        "    wide = torch.randn(M, K + 64, generator=g).cuda().to(torch.bfloat16)\n"
        "    out = torch.zeros(M, N + 8, device='cuda', dtype=torch.bfloat16)\n"
        "    _hip.linear_bf16(wide[:, 32:32 + K], w, None, out=out[:, :N])\n"
        "    ref2 = wide[:, 32:32 + K].float() @ w.float().t()\n"
        "    assert float((out[:, :N].float() - ref2).abs().max() / ref2.abs().max()) < 1e-2 and float(out[:, N:].abs().max()) == 0.0\n"
        "print('deep ok')\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), DEEP_SHAPES)
    res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VSDE_DEEP_GEMM="1"), capture_output=True, text=True)
    assert res.returncode == 0 and "deep ok" in res.stdout, res.stdout + res.stderr


@pytest.mark.gpu
def test_grouped_weight_gradients_are_bit_identical_to_single_launches():
    """``vsde_linear_wgrad_group_bf16`` plans every problem exactly as ``vsde_linear_wgrad_bf16_rows`` does and sums its partial
    tiles in the same fixed order: the results of a mixed group (both tile widths, with and without bias, a row map, few and many
    rows, more problems than one launch takes) must be bit-identical to one call per problem."""
    from viforsdes_amd import _hip
    g = torch.Generator().manual_seed(7)
    R = lambda *s: torch.randn(*s, generator=g).to("cuda", torch.bfloat16)
    shapes = [(12928, 832, 256, True), (12928, 256, 704, True), (12928, 1408, 256, True), (12928, 256, 256, False),
              (4096, 128, 128, True), (700, 64, 256, False), (33, 8, 64, True), (205312, 256, 256, True)]
    shapes = shapes + [(3000 + 64 * i, 256, 256, bool(i & 1)) for i in range(24)]      # 32 problems: more than one group holds
    problems, expect = [], []
    for i, (M, N, K, bias) in enumerate(shapes):
        dy, x = R(M, N), R(M, K)
        row_map, rows = None, None
        if i == 2:   # the interleaved SwiGLU pack: rows permuted, a few dropped
            perm = torch.randperm(N, generator=g)
            row_map = torch.where(perm < N - 26, perm, torch.full_like(perm, -1)).to("cuda", torch.int32)
            rows = N - 26
        problems.append((dy, x, bias, row_map, rows))
        expect.append(_hip.linear_wgrad(dy, x, bias, row_map, rows))
    got = _hip.linear_wgrad_group(problems)
    planned = _hip.linear_wgrad_group(problems, group_plan=True)   # split counts chosen for the group: another summation order
    torch.cuda.synchronize()
    for i, ((dW, db), (eW, eb)) in enumerate(zip(planned, expect)):
        sel = slice(None) if problems[i][3] is None else problems[i][3][problems[i][3] >= 0].long()
        scale = float(eW[sel].abs().max()) + 1e-30
        assert float((dW[sel] - eW[sel]).abs().max()) <= 2e-5 * scale, (i, shapes[i])
        if eb is not None:
            assert float((db[sel] - eb[sel]).abs().max()) <= 2e-5 * (float(eb[sel].abs().max()) + 1e-30), (i, shapes[i])
    for i, ((dW, db), (eW, eb)) in enumerate(zip(got, expect)):
        if problems[i][3] is not None:   # dropped rows are not written by either path: compare the mapped rows only
            keep = problems[i][3][problems[i][3] >= 0].long()
            assert torch.equal(dW[keep], eW[keep]) and torch.equal(db[keep], eb[keep]), i
        else:
            assert torch.equal(dW, eW), (i, shapes[i])
            assert (db is None) == (eb is None) and (db is None or torch.equal(db, eb)), (i, shapes[i])

"""GPU: the fused SwiGLU MLP kernels (csrc/vsde_mlp.hip) against the unfused chain of primitives/mlp.py:50-54 under autocast,
evaluated in fp32 from the same bf16-rounded intermediates: u = bf16(x W_in^T + b_in), s = bf16(bf16(silu(a)) * b),
y = bf16(s W_out^T + b_out).  Tolerances: s 1e-2 of its max (one bf16 rounding of u flips the last bit of some s), y 1e-2."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rand(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(torch.bfloat16)


def _rel(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-30))


def _packs(C, H, hreal, seed=0, interleave=False):
    from viforsdes_amd.primitives import fused
    g = torch.Generator().manual_seed(seed)
    w_in = torch.nn.Parameter((torch.randn(2 * hreal, C, generator=g) * C ** -0.5).to(DEV))
    b_in = torch.nn.Parameter((torch.randn(2 * hreal, generator=g)).to(DEV))
    w_out = torch.nn.Parameter((torch.randn(C, hreal, generator=g) * hreal ** -0.5).to(DEV))
    b_out = torch.nn.Parameter((torch.randn(C, generator=g)).to(DEV))
    pin, pout = fused.swiglu_packs(w_in, b_in, w_out, b_out, H, interleave=interleave)
    return (w_in, b_in, w_out, b_out), pin, pout, fused.MlpImages(pin, pout, H)


def _reference(x, w_in, b_in, w_out, b_out):
    bf = lambda t: t.to(torch.bfloat16).float()
    h = w_out.shape[1]
    u = bf(x.float() @ bf(w_in).t() + bf(b_in))
    a, b = u[:, :h], u[:, h:]
    s = bf(bf(a * torch.sigmoid(a)) * b)
    return s, s @ bf(w_out).t() + bf(b_out)


# (M, C, H padded, H real): the LV encoder's MLP (682 -> 704), the C = 128 fixture width, ragged M, a single partial stripe,
# more than one round of workgroups, the benchmark's own shape
@pytest.mark.parametrize("M,C,H,hreal", [(20000, 256, 704, 682), (4264, 128, 384, 341), (300, 128, 64, 64), (77, 256, 128, 100),
                                         (133000 + 5, 256, 704, 682), (205312, 256, 704, 682)])
@pytest.mark.parametrize("interleave", [False, True])
def test_fused_mlp_forward(M, C, H, hreal, interleave):
    from viforsdes_amd import _hip
    params, pin, pout, img = _packs(C, H, hreal, interleave=interleave)
    x = _rand(M, C, seed=5)
    s_ref, y_ref = _reference(x, *[q.detach() for q in params])
    w1, w2, b1 = img.operands()
    y, s = _hip.mlp_fwd(x, w1, w2, b1, pout.bias, H, want_s=True)
    assert _rel(s[:, :hreal], s_ref) < 1e-2 and (hreal == H or float(s[:, hreal:].abs().max()) == 0.0)
    assert _rel(y, y_ref) < 1e-2
    # y from the kernel's own s isolates the second product from the rounding of u
    y_own = s.float() @ pout.weight.float().t() + pout.bias.float()
    assert _rel(y, y_own) < 6e-3
    y2, none = _hip.mlp_fwd(x, w1, w2, b1, pout.bias, H, want_s=False)
    assert none is None and torch.equal(y2, y)
    # a row-pitched input (column range of a wider buffer)
    wide = _rand(M, C + 64, seed=6)
    y3, _ = _hip.mlp_fwd(wide[:, 32:32 + C], w1, w2, b1, pout.bias, H)
    y3c, _ = _hip.mlp_fwd(wide[:, 32:32 + C].contiguous(), w1, w2, b1, pout.bias, H)
    assert torch.equal(y3, y3c)


def test_images_follow_a_pack_refresh():
    from viforsdes_amd import _hip
    from viforsdes_amd.primitives import fused
    params, pin, pout, img = _packs(256, 704, 682)
    x = _rand(5000, 256, seed=7)
    w1, w2, b1 = img.operands()
    y0, _ = _hip.mlp_fwd(x, w1, w2, b1, pout.bias, 704)
    with torch.no_grad():
        for q in params:
            q.mul_(0.5)
    fused.note_parameters_changed()
    ids = {id(q) for q in params}
    fused.PackedWeight.refresh_all(force=True, params=ids)
    # an EAGER forced refresh only marks the derived images dirty (the training step never reads them: no launches for them per step);
    # whoever reads them without going through operands() -- a captured sampling call before its replay -- rebuilds the dirty ones first
    assert img._key is None
    fused.PackedWeight.refresh_dirty_derived(ids)
    assert img._key is not None
    y1, _ = _hip.mlp_fwd(x, img.w1, img.w2, img.b1, pout.bias, 704)   # no operands() call: what a captured sampling call replays
    _, y_ref = _reference(x, *[q.detach() for q in params])
    assert _rel(y1, y_ref) < 1e-2 and _rel(y0, y_ref) > 0.1
    # inside a stream capture the forced refresh rebuilds them itself (the captured optimizer step's tail): replaying that graph after
    # another parameter change brings packs AND images up to date on the device
    with torch.no_grad():
        for q in params:
            q.mul_(2.0)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fused.PackedWeight.refresh_all(force=True, params=ids)   # (the refresh table is built outside the capture)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fused.PackedWeight.refresh_all(force=True, params=ids)
    with torch.no_grad():
        for q in params:
            q.mul_(0.25)
    g.replay()
    y2, _ = _hip.mlp_fwd(x, img.w1, img.w2, img.b1, pout.bias, 704)
    _, y_ref2 = _reference(x, *[q.detach() for q in params])
    assert _rel(y2, y_ref2) < 1e-2


@pytest.mark.parametrize("B,N,C,H,hreal,last", [(40, 401, 256, 704, 682, False), (40, 401, 256, 704, 682, True), (33, 129, 128, 384, 341, False), (128, 101, 256, 704, 682, False),
                                               (512, 401, 256, 704, 682, False)])
def test_block_form_matches_the_unfused_chain(B, N, C, H, hreal, last):
    """vsde_mlp_block_fwd_bf16 against the kernels it replaces: residual_ln_fwd -> fused MLP -> residual_ln_fwd / gated_residual_fwd
    (same rounding points; the row statistics are summed in another order: one bf16 ulp of the outputs, |out| <= ~8 here)."""
    from viforsdes_amd import _hip
    params, pin, pout, img = _packs(C, H, hreal)
    w1, w2, b1 = img.operands()
    x, yin = _rand(B, N, C, seed=11), _rand(B, N, C, seed=12)
    allm = _rand(B, 8 * C + 64, scale=0.5, seed=13)   # one buffer, six column ranges: the row pitch is shared
    ga, sc, sh, gm, sn, hs = [allm[:, i * C:(i + 1) * C] for i in range(6)]
    eps = 1e-5
    x1, h2, _, _ = _hip.residual_ln_fwd(x, yin, ga, sc, sh, eps)
    m, _ = _hip.mlp_fwd(h2.reshape(B * N, C), w1, w2, b1, pout.bias, H)
    m = m.reshape(B, N, C)
    if last:
        tok_ref, h_ref = _hip.gated_residual_fwd(x1, m, gm), None
    else:
        tok_ref, h_ref, _, _ = _hip.residual_ln_fwd(x1, m, gm, sn, hs, eps)
    tok, hn = _hip.mlp_block_fwd(x, yin, ga, sc, sh, gm, None if last else sn, None if last else hs, eps, eps, w1, w2, b1, pout.bias, H)
    assert _rel(tok, tok_ref) < 1e-2
    assert (hn is None) == last
    if not last:
        assert _rel(hn, h_ref) < 1.5e-2


@pytest.mark.parametrize("B,N,C,H,hreal,last", [(40, 401, 256, 704, 682, False), (40, 401, 256, 704, 682, True), (33, 129, 128, 384, 341, False),
                                               (128, 101, 256, 704, 682, False), (512, 401, 256, 704, 682, False)])
def test_block_form_with_the_out_projection_in_front(B, N, C, H, hreal, last):
    """vsde_mlp_attn_block_fwd_bf16 = vsde_linear_gated_bf16 (gate + out projection of the attention branch, attn.py:107-110) followed
    by vsde_mlp_block_fwd_bf16: same operands, same rounding points, the same k order in the out projection -> the two routes differ by
    the MFMA summation inside a k-step at most (one bf16 ulp of a few outputs)."""
    from viforsdes_amd import _hip
    from viforsdes_amd.primitives import fused
    params, pin, pout, img = _packs(C, H, hreal)
    w1, w2, b1 = img.operands()
    g = torch.Generator(device="cpu").manual_seed(41)
    wo = torch.nn.Parameter((torch.randn(C, C, generator=g) / C ** 0.5).cuda())
    bo = torch.nn.Parameter((0.1 * torch.randn(C, generator=g)).cuda())
    po = fused.plain_pack(wo, bo)
    x, attn = _rand(B, N, C, seed=11), _rand(B, N, C, seed=12)
    gbuf = _rand(B * N, 96, scale=2.0, seed=14)
    glog = gbuf[:, 16:80]                               # a column range of a wider buffer: the row pitch is an argument
    allm = _rand(B, 8 * C + 64, scale=0.5, seed=13)
    ga, sc, sh, gm, sn, hs = [allm[:, i * C:(i + 1) * C] for i in range(6)]
    eps = 1e-5
    wo_b, bo_b = po.operands()
    yin = _hip.linear_gated_bf16(attn.view(B * N, C), glog, wo_b, bo_b).view(B, N, C)
    tok_ref, h_ref = _hip.mlp_block_fwd(x, yin, ga, sc, sh, gm, None if last else sn, None if last else hs, eps, eps, w1, w2, b1, pout.bias, H)
    oimg = fused.OutProjImage(po)
    tok, hn = _hip.mlp_attn_block_fwd(x, attn, glog, oimg.operand(), bo_b, ga, sc, sh, gm, None if last else sn, None if last else hs, eps, eps,
                                      w1, w2, b1, pout.bias, H)
    assert _rel(tok, tok_ref) < 4e-3
    assert (hn is None) == last
    if not last:
        assert _rel(hn, h_ref) < 8e-3
    # and against the plain float32 statement of the out projection
    og = (attn.float() * torch.sigmoid(glog.float()).bfloat16().float().repeat(1, C // 64).view(B, N, C)).bfloat16().float()
    yin32 = (og.view(B * N, C) @ wo_b.float().t() + bo_b.float()).view(B, N, C)
    assert _rel(yin, yin32) < 1e-2


@pytest.mark.parametrize("M,C,H,hreal", [(20000, 256, 704, 682), (4264, 128, 384, 341), (300, 128, 64, 64), (77, 256, 128, 100),
                                         (133000 + 5, 256, 704, 682), (205312, 256, 704, 682)])
@pytest.mark.ablation_build
def test_fused_mlp_backward(M, C, H, hreal):
    """vsde_mlp_bwd_bf16 against the two kernels it replaces (rows kernel with the SwiGLU derivative in its epilogue, then the dx
    GEMM over du) and against the fp32 formulas evaluated from the same bf16 u and dy: du to 2e-2 of its max (the derivative is
    computed in fp32 from bf16 inputs and rounded once), dx to 1e-2."""
    from viforsdes_amd import _hip
    from viforsdes_amd.primitives import fused
    params, pin, pout, _ = _packs(C, H, hreal, interleave=True)
    img = fused.MlpBwdImages(pin, pout, H).operand()
    x, dy = _rand(M, C, seed=21), _rand(M, C, seed=22)
    w1, b1 = pin.operands()
    u, _ = _hip.linear_swiglu_bf16(x, w1, b1, want_u=True)
    du, dx = _hip.mlp_bwd(dy, u, img, H)
    du_old = _hip.linear_swiglu_bwd_bf16(dy, pout.transposed(), u)
    assert _rel(du, du_old) < 1e-2
    ua = u.float().reshape(M, H // 16, 2, 16)[:, :, 0].reshape(M, H)
    ub = u.float().reshape(M, H // 16, 2, 16)[:, :, 1].reshape(M, H)
    ds = dy.float() @ pout.weight.float()
    sg = torch.sigmoid(ua)
    da, db = ds * ub * sg * (1 + ua * (1 - sg)), ds * ua * sg
    du_ref = torch.stack([da.reshape(M, H // 16, 16), db.reshape(M, H // 16, 16)], dim=2).reshape(M, 2 * H)
    assert _rel(du, du_ref) < 2e-2
    dx_own = du.float() @ pin.weight.float()        # from the kernel's own du: isolates the second product
    assert _rel(dx, dx_own) < 6e-3
    assert _rel(dx, du_ref @ pin.weight.float()) < 1e-2


@pytest.mark.ablation_build
def test_opt_in_fused_backward_gives_the_same_gradients():
    """``fused.FUSED_MLP_BWD`` (VSDE_FUSED_MLP_BWD=1) swaps the two backward launches of ``_SwiGLUMLP`` for mlp_bwd_kernel: input and
    parameter gradients of the module-level op must agree with the default route (bf16 du: 1e-2 of each gradient's max)."""
    from viforsdes_amd.primitives import fused
    M, C, H, hreal = 40000, 256, 704, 682
    params, pin, pout, _ = _packs(C, H, hreal, interleave=True)
    x0 = _rand(M, C, seed=31)
    dy = _rand(M, C, seed=32)
    out = {}
    for flag in (False, True):
        fused.FUSED_MLP_BWD = flag
        try:
            for q in params:
                q.grad = None
            x = x0.clone().requires_grad_(True)
            y = fused.swiglu_mlp(x, pin, pout)
            y.backward(dy)
            out[flag] = [x.grad.float()] + [q.grad.float().clone() for q in params]
        finally:
            fused.FUSED_MLP_BWD = False
    for a, b in zip(out[False], out[True]):
        assert _rel(b, a) < 1e-2


@pytest.mark.parametrize("M,K,bias,transposed", [(20000, 704, True, False), (205312, 1408, False, True), (40000 + 17, 832, False, True),
                                                 (300, 256, True, False), (77, 512, False, False)])
@pytest.mark.ablation_build
def test_deep_reduction_gemm(M, K, bias, transposed):
    """vsde_linear_deep256_bf16 (y [M, 256] = x [M, K] W^T + b, the products the library ran until round 5) against the float32 product
    of the same bf16 operands: fp32 accumulation, one rounding of the output (1e-2 of max is two bf16 ulps of the largest output);
    weights as [256, K] and as the transposed view of a [K, 256] pack (the input-gradient GEMMs), ragged M, the shallowest K."""
    from viforsdes_amd import _hip
    from viforsdes_amd.primitives import fused
    g = torch.Generator().manual_seed(K + M)
    shape = (K, 256) if transposed else (256, K)
    w = torch.nn.Parameter((torch.randn(*shape, generator=g) * K ** -0.5).to(DEV))
    b = torch.nn.Parameter(torch.randn(256, generator=g).to(DEV)) if bias else None
    pack = fused.plain_pack(w, b)
    x = _rand(M, K, seed=3)
    wide = _rand(M, K + 64, seed=4)
    for xin in (x, wide[:, 32:32 + K]):   # a column range of a wider buffer: the row pitch is an argument
        y = fused.deep256(xin, pack, transposed, pack.bias)
        wb = pack.weight.float()
        ref = xin.float() @ (wb if transposed else wb.t()) + (pack.bias.float() if bias else 0.0)
        assert y.shape == (M, 256) and torch.isfinite(y).all()
        assert _rel(y, ref) < 1e-2
    # the image follows a refresh of the pack
    with torch.no_grad():
        w.mul_(0.5)
    fused.note_parameters_changed()
    fused.PackedWeight.refresh_all(force=True, params={id(w)})
    y2 = fused.deep256(x, pack, transposed, pack.bias)
    wb = pack.weight.float()
    ref2 = x.float() @ (wb if transposed else wb.t()) + (pack.bias.float() if bias else 0.0)
    assert _rel(y2, ref2) < 1e-2

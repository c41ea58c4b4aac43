"""``torch.ops.vsde.sde_fwd`` / ``sde_bwd``: dispatcher registration of the fused head on top of the C ABI."""
import numpy as np
import pytest
import torch

from helpers import W_NAMES, load_head_case


def test_operators_are_registered_with_the_reference_argument_order():
    import viforsdes_amd.torch_ops  # noqa: F401  (registers on import)
    fwd, bwd = torch.ops.vsde.sde_fwd.default, torch.ops.vsde.sde_bwd.default
    names = [a.name for a in fwd._schema.arguments]
    assert names == ["x0", "context", "sde_parameters", "eps", "weights", "time_step", "save_activations"]
    assert len(fwd._schema.returns) == 5
    assert [a.name for a in bwd._schema.arguments][:6] == ["grad_paths", "grad_means", "grad_cholesky", "context",
                                                           "sde_parameters", "eps"]


@pytest.mark.gpu
def test_operators_match_the_golden_case():
    import viforsdes_amd.torch_ops  # noqa: F401
    d = load_head_case("tiny_l2")
    dev = "cuda:0"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ws = [t(d["w_" + n]) for n in W_NAMES]
    ctx = t(d["context_full"])[:, :-1]
    paths, means, chol, raw, acts = torch.ops.vsde.sde_fwd(t(d["x0"]), ctx, t(d["sde_parameters"]), t(d["eps"]), ws, float(d["dt"]), True)
    for got, key in ((paths, "paths"), (means, "means"), (chol, "chol")):
        ref = d["o1f64_" + key]   # the reference's eager head in float64 (tests/golden/make_golden.py)
        assert np.abs(got.cpu().numpy() - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), key
    grads = torch.ops.vsde.sde_bwd(t(d["g_paths"]), t(d["g_means"]), t(d["g_chol"]), ctx, t(d["sde_parameters"]), t(d["eps"]),
                                   paths, raw, acts, ws, float(d["dt"]))
    assert len(grads) == 13
    from helpers import G_NAMES, rel_err
    for g, name in zip(grads, G_NAMES):
        ref = d["o1f64_grad_" + name]
        if ref.size:
            assert rel_err(g.cpu().numpy(), ref) < 5e-5, name
    # shapes also come out of the fake (meta) implementations: torch.compile / tracing can see through the operators
    with torch._subclasses.FakeTensorMode():
        fx0 = torch.empty(4, 2, device=dev); fctx = torch.empty(4, 7, 16, device=dev)
        fw = [torch.empty(w.shape, device=dev) for w in ws]
        out = torch.ops.vsde.sde_fwd(fx0, fctx, torch.empty(4, 3, device=dev), torch.empty(4, 7, 2, device=dev), fw, 0.1, False)
        assert out[0].shape == (4, 8, 2) and out[3].numel() == 0

"""vsde_attention_fwd_bf16 vs torch SDPA (memory-efficient backend): correctness and time at the LV encoder shape."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F
from viforsdes_amd import _hip

dev = torch.device("cuda:0")
def run(B, N, H, reps=20):
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(B, N, H, 64, generator=g).to(dev, torch.bfloat16) for _ in range(3))
    o, lse = _hip.attention_fwd(q, k, v, 0.125)
    qf, kf, vf = (t.float().transpose(1, 2) for t in (q, k, v))
    s = (qf @ kf.transpose(-1, -2)) * 0.125
    ref = (torch.softmax(s, -1) @ vf).transpose(1, 2)
    ref_lse = torch.logsumexp(s, -1)
    err = (o.float() - ref).abs().max().item() / ref.abs().max().item()
    lerr = (lse - ref_lse).abs().max().item()
    def t(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    t_own = t(lambda: _hip.attention_fwd(q, k, v, 0.125))
    qt, kt, vt = (x.transpose(1, 2) for x in (q, k, v))
    from torch.nn.attention import SDPBackend, sdpa_kernel
    with sdpa_kernel([SDPBackend.EFFICIENT_ATTENTION]):
        t_ref = t(lambda: F.scaled_dot_product_attention(qt, kt, vt))
    go = torch.randn_like(q)
    dq, dk, dv = _hip.attention_bwd(go, q, k, v, o, lse, 0.125)
    qr, kr, vr = (t.float().transpose(1, 2).requires_grad_() for t in (q, k, v))
    sr = (qr @ kr.transpose(-1, -2)) * 0.125
    rr = (torch.softmax(sr, -1) @ vr).transpose(1, 2)
    gq, gk, gv = torch.autograd.grad((rr * go.float()).sum(), [qr, kr, vr])
    berr = [((a.float() - g.transpose(1, 2)).abs().max() / g.abs().max()).item() for a, g in ((dq, gq), (dk, gk), (dv, gv))]
    t_bwd = t(lambda: _hip.attention_bwd(go, q, k, v, o, lse, 0.125))
    zero = torch.zeros((), dtype=torch.int64)
    t_bref = t(lambda: torch.ops.aten._efficient_attention_backward(go, q, k, v, None, o, None, None, N, N, lse, 0.0, zero, zero, 0, False, scale=0.125))
    print(f"   bwd rel err dq {berr[0]:.2e} dk {berr[1]:.2e} dv {berr[2]:.2e} | own {t_bwd:7.1f} us  library {t_bref:7.1f} us")
    fl = 4.0 * B * H * N * N * 64
    print(f"B={B} N={N} H={H}: rel err {err:.2e} lse err {lerr:.2e} | own {t_own:7.1f} us ({fl/t_own/1e6:6.1f} TF/s)  sdpa {t_ref:7.1f} us ({fl/t_ref/1e6:6.1f} TF/s)")

run(2, 37, 4, reps=3)
run(3, 101, 4, reps=3)
run(512, 401, 4)
run(128, 101, 4)
run(64, 544, 4)

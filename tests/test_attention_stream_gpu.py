"""GPU: the streamed attention kernels (csrc/vsde_attn_stream.hip: K / V tiles through LDS, online softmax) vs an fp32
softmax(scale q k^T) v of the same bf16 inputs, forward, log-sum-exp and all three gradients.

Shapes: the synthetic stress configuration's (1001 tokens, head_dim 128), the first head_dim-64 length beyond the LDS-resident
kernels (545), a short head_dim-128 sequence, ragged / tiny lengths.  Tolerances as for the resident kernels
(tests/test_encoder_fused_gpu.py): output 1e-2 of the max, gradients 2e-2 of the max + 1e-2 absolute (bf16 P and dS operands).
The rescale branch of the online softmax is data dependent and rare on random inputs, so one case FORCES it: a key late in the
sequence whose score towers over everything before it (rule: a rare branch needs its own test)."""
import numpy as np
import pytest
import torch

from helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _reference(q, k, v, go, scale):
    qf, kf, vf = (t.detach().float().transpose(1, 2).requires_grad_() for t in (q, k, v))
    s = (qf @ kf.transpose(-1, -2)) * scale
    ref = (torch.softmax(s, -1) @ vf).transpose(1, 2)
    grads = torch.autograd.grad((ref * go.float()).sum(), [qf, kf, vf])
    return ref.detach(), torch.logsumexp(s.detach(), -1), [g.transpose(1, 2) for g in grads]


def _check(q, k, v, go, scale):
    from viforsdes_amd import _hip
    from viforsdes_amd.primitives import fused
    assert fused.attention_usable(q)
    qg, kg, vg = (t.clone().requires_grad_() for t in (q, k, v))
    o = fused.attention(qg, kg, vg, scale)
    dq, dk, dv = torch.autograd.grad((o.float() * go.float()).sum(), [qg, kg, vg])
    ref, lse_ref, (rq, rk, rv) = _reference(q, k, v, go, scale)
    _, lse = _hip.attention_fwd(q, k, v, scale)
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all()
    assert rel_err(o.detach().float().cpu().numpy(), ref.cpu().numpy()) < 1e-2
    assert torch.allclose(lse, lse_ref, atol=2e-3, rtol=1e-5)
    for name, a, r in (("dq", dq, rq), ("dk", dk, rk), ("dv", dv, rv)):
        a, r = a.float().cpu().numpy(), r.cpu().numpy()
        assert np.abs(a - r).max() <= 2e-2 * np.abs(r).max() + 1e-2, name


@pytest.mark.parametrize("B,N,H,D", [(3, 1001, 4, 128), (2, 545, 2, 64), (2, 100, 2, 128), (1, 33, 1, 128), (2, 1, 2, 128),
                                      (1, 640, 1, 64), (2, 257, 3, 128),
                                      # pair counts that are multiples of 8 take the XCD-aware (pair, block) mapping
                                      (4, 1001, 4, 128), (8, 300, 2, 128), (2, 513, 4, 128), (4, 600, 2, 64), (8, 31, 1, 128)])
def test_streamed_attention_vs_fp32_softmax(B, N, H, D):
    g = torch.Generator().manual_seed(N + D)
    q, k, v, go = (torch.randn(B, N, H, D, generator=g).to(DEV, torch.bfloat16) for _ in range(4))
    _check(q, k, v, go, D ** -0.5)


def test_streamed_attention_forced_rescale():
    """Scores grow along the sequence: keys 0..N-1 scaled by an increasing factor, plus one key near the end aligned with a
    query at 40x -- the running maximum jumps by far more than the 2^8 deferral threshold several tiles into the loop."""
    g = torch.Generator().manual_seed(5)
    B, N, H, D = 2, 300, 2, 128
    q, k, v, go = (torch.randn(B, N, H, D, generator=g) for _ in range(4))
    k = k * torch.linspace(0.2, 3.0, N).view(1, N, 1, 1)
    k[:, 250] = q[:, 7] * 3.0          # query 7 (and its neighbours in norm) meets a towering score at key 250
    q[:, 7] = q[:, 7] * 1.5
    q, k, v, go = (t.to(DEV, torch.bfloat16) for t in (q, k, v, go))
    _check(q, k, v, go, D ** -0.5)


def test_encoder_module_at_head_dim_128_uses_own_attention():
    """hidden 256 / 2 heads = head_dim 128, 600 tokens: fused route (own streamed attention) vs the unfused torch chain."""
    from viforsdes_amd import EncoderConfig
    from viforsdes_amd.models.encoder import ObservationContextEncoder
    from viforsdes_amd.primitives import fused
    torch.manual_seed(0)
    enc = ObservationContextEncoder(2, 3, EncoderConfig(hidden_dim=256, cond_dim=32, num_heads=2, depth=2)).to(DEV)
    with torch.no_grad():
        for n, p in enc.named_parameters():
            if p.requires_grad and p.abs().sum() == 0:
                p.normal_(0, 0.05)
    obs_t = torch.tensor([0.0, 1.0, 2.0], device=DEV); obs_v = torch.randn(3, 2, device=DEV)
    theta = torch.rand(8, 3, device=DEV) + 0.2
    outs = []
    for flag in (True, False):
        fused.ENABLED = flag
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                outs.append(enc(obs_v, obs_t, theta, 5.99, 0.01).float())
        finally:
            fused.ENABLED = True
    assert outs[0].shape == (8, 600, 256)
    assert rel_err(outs[0].detach().cpu().numpy(), outs[1].detach().cpu().numpy()) < 3e-2

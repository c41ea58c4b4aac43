import sys; sys.path.insert(0, "/root/repo")
import torch
from viforsdes_amd import _hip
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, N, H = 4, 401, 4
q, k, v = (torch.randn(B, N, H, 64, generator=g).to(dev, torch.bfloat16) for _ in range(3))
out, lse, seed, off, mq, mk = torch.ops.aten._efficient_attention_forward(q, k, v, None, None, None, None, None, 0.0, 0, True, scale=0.125)
print("out", out.shape, out.stride(), "lse", lse.shape, lse.dtype, "seed", seed.shape, seed.dtype, seed, "off", off, mq, mk)
o2, lse2 = _hip.attention_fwd(q, k, v, 0.125)
print("lse diff", (lse[..., :N] - lse2).abs().max().item(), "out diff", (out.float() - o2.float()).abs().max().item())
go = torch.randn_like(out)
r1 = torch.ops.aten._efficient_attention_backward(go, q, k, v, None, out, None, None, N, N, lse, 0.0, seed, off, 0, False, scale=0.125)
lse_pad = lse2 if lse.shape[-1] == N else torch.nn.functional.pad(lse2, (0, lse.shape[-1] - N))
r2 = torch.ops.aten._efficient_attention_backward(go, q, k, v, None, o2, None, None, N, N, lse_pad.contiguous(), 0.0, seed, off, 0, False, scale=0.125)
for a, b, n in zip(r1[:3], r2[:3], "qkv"):
    print("d" + n, a.shape, a.stride(), (a.float() - b.float()).abs().max().item(), a.float().abs().max().item())

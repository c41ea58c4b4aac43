#!/usr/bin/env python3
"""Where does the bf16 deviation of the OU / LV optimizer-step trajectory come from?  Same initial state and injected
noise, K optimizer steps on the GPU in four arithmetic modes: fp32 fused, bf16 fused (the benchmark's route), bf16 with the
fused encoder operators off (torch autocast chain + HIP head/ELBO) and fp32 unfused; prints the ELBO per step, the
posterior means and the cosine / relative distance of the first step's gradient to the fp32-unfused one.
    python tools/parity_probe.py [ou|lv] [batch] [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_trainer  # noqa: E402
from viforsdes_amd.examples.sdes import lv_problem, ou_problem  # noqa: E402
from viforsdes_amd.primitives import fused  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "ou"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
problem = ou_problem() if name == "ou" else lv_problem()
sde, obs, like, prior, horizon, dt, *_ = problem
T, S, P = int(round(horizon / dt)), sde.state_dim, sde.sde_param_dim
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(99)
teps = [torch.randn(batch, P, generator=g).to(dev) for _ in range(steps)]
noise = [torch.randn(batch, T, S, generator=g).to(dev) for _ in range(steps)]
enc = dict(enc_hidden=256, enc_depth=8)

ref = build_trainer(problem, batch, dev, False, seed=4321, **enc)
init = {k: v.clone() for k, v in ref.ctx.model.state_dict().items()}
del ref


def run(mp, fused_on):
    fused.ENABLED = fused_on
    try:
        tr = build_trainer(problem, batch, dev, mp, seed=4321, **enc)
        tr.ctx.model.load_state_dict(init)
        tr.ctx.ema._init_shadow()
        elbos, g0 = [], None
        for k in range(steps):
            if k == 0:   # gradient of the first step (unscaled), before the optimizer touches anything
                m = tr.ctx.model
                tr._forward_backward(m, teps[0], noise[0])
                sc = float(tr.ctx.scaler.get_scale()) if tr.ctx.scaler.is_enabled() else 1.0
                named = [(n, p.grad.detach().float().flatten() / sc) for n, p in m.named_parameters() if p.grad is not None]
                g0 = torch.cat([v for _, v in named])
                groups = {}
                for n, v in named:
                    key = n.split(".")[0] + ("." + n.split(".")[-2] if n.startswith("encoder.sit.blocks") else "")
                    groups.setdefault(key, []).append(v)
                gparts = {k2: torch.cat(v) for k2, v in groups.items()}
            r = tr._train_step(tr.ctx.model, theta_eps=teps[k], path_noise=noise[k])
            elbos.append(float(r.elbo_result.evidence_lower_bound))
        ev = tr.ctx.model.sde_parameter_posterior.expected_value.detach().cpu().tolist()
        worst_pack = 0.0
        for pk in list(fused.PackedWeight._live):   # every cached bf16 operand against its fp32 sources
            if not any(any(q is p_ for p_ in tr.ctx.model.parameters()) for q in pk.params):
                continue
            dst, src = pk._copy_lists()
            for d_, s_ in zip(dst, src):
                worst_pack = max(worst_pack, float((d_.float() - s_.to(torch.bfloat16).float()).abs().max()))
            if pk.weight_t is not None:
                worst_pack = max(worst_pack, float((pk.weight_t.float() - pk.weight.t().float()).abs().max()))
        print(f"   [{'bf16' if mp else 'fp32'} {'fused' if fused_on else 'unfused'}] live packs of this model: max |cached - param| = {worst_pack:.3e}")
        final = {n: p_.detach().clone() for n, p_ in tr.ctx.model.named_parameters()}
        scale = float(tr.ctx.scaler.get_scale()) if tr.ctx.scaler.is_enabled() else None
        return elbos, ev, g0, scale, gparts, final
    finally:
        fused.ENABLED = True


modes = [("fp32 unfused", False, False), ("fp32 fused", False, True), ("bf16 unfused", True, False), ("bf16 fused", True, True)]
res = {n: run(mp, f) for n, mp, f in modes}
base = res["fp32 unfused"]
for n, (e, ev, g0, sc, gp, fin) in res.items():
    line = f"{n:13s} elbo {['%.4f' % v for v in e]} E[theta] {['%.5f' % v for v in ev]} scaler {sc}"
    if g0 is not None and base[2] is not None:
        b = base[2]
        cos = float(torch.dot(g0, b) / (g0.norm() * b.norm()))
        line += f" | grad cos {cos:.6f} rel {float((g0 - b).norm() / b.norm()):.3e} |g| {float(g0.norm()):.4e}"
    print(line)
    if n != "fp32 unfused":
        dpar = sorted(((float((fin[k2] - base[5][k2]).abs().max()), k2) for k2 in fin), reverse=True)[:4]
        print("      largest parameter differences after the steps:", ", ".join(f"{k2} {v:.2e}" for v, k2 in dpar))
        worst = sorted(((float((gp[k2] - base[4][k2]).norm() / (base[4][k2].norm() + 1e-30)), k2) for k2 in gp), reverse=True)[:6]
        print("      worst parameter groups (rel L2 to fp32 unfused):", ", ".join(f"{k2} {v:.2e}" for v, k2 in worst))

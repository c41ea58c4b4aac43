"""Wall-clock split of one LV training step into its phases (GPU events, eager)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from torch import nn
from viforsdes_amd.examples.sdes import lv_problem
from viforsdes_amd.inference.diffusion_path_sampler import sample_diffusion_paths
from viforsdes_amd.inference.evidence_lower_bound import compute_evidence_lower_bound

dev = torch.device("cuda:0")
tr = bench.build_trainer(lv_problem(), 512, dev, True, seed=1234)
model, ctx, cfg = tr.ctx.model, tr.ctx, tr.config
for _ in range(5):
    tr._train_step(model)
names = ["zero+rsample", "encoder+head fwd", "elbo fwd", "backward", "allreduce+unscale+clip", "optimizer+refresh"]
acc = [0.0] * len(names)
R = 10
for _ in range(R):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
    ev[0].record()
    ctx.grad_sync.zero_grad()
    theta = model.sde_parameter_posterior.rsample(cfg.batch_size)
    ev[1].record()
    with torch.autocast(device_type="cuda", dtype=cfg.amp_dtype.value, enabled=ctx.scaler.is_enabled()):
        sample = sample_diffusion_paths(model.encoder, model.head, ctx.observations, theta, ctx.x0_buffer,
                                        tr.time_horizon, cfg.time_step, tr.state_space)
        ev[2].record()
        res = compute_evidence_lower_bound(tr.sde, ctx.observations, tr.observation_likelihood, tr.prior,
                                           model.sde_parameter_posterior, theta, sample, cfg.time_step)
    ev[3].record()
    ctx.scaler.scale(-res.evidence_lower_bound).backward()
    ev[4].record()
    ctx.grad_sync.all_reduce()
    ctx.scaler.unscale_(ctx.optimizer)
    nn.utils.clip_grad_norm_(model.parameters(), cfg.grad_clip_norm)
    ev[5].record()
    ctx.scaler.step(ctx.optimizer); ctx.scaler.update()
    from viforsdes_amd.primitives import fused
    fused.PackedWeight.refresh_all()
    ctx.ema.update()
    ev[6].record()
    torch.cuda.synchronize()
    for i in range(len(names)):
        acc[i] += ev[i].elapsed_time(ev[i + 1])
for n, a in zip(names, acc):
    print(f"{n:28s} {a / R:7.2f} ms")
print(f"{'total':28s} {sum(acc) / R:7.2f} ms")

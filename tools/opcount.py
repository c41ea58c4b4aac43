"""Kernel launches of ONE LV training step by phase (GPU only): what is left outside the hand-written kernels, by count and by
device time.  Phases are profiled one after the other on the same trainer."""
import os, sys
from collections import defaultdict
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from viforsdes_amd.examples.sdes import lv_problem


def count(fn, label, top=0):
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        out = fn()
        torch.cuda.synchronize()
    kern = defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA:
            k = kern[ev.name[:110]]
            k[0] += 1; k[1] += ev.device_time
    small = {k: v for k, v in kern.items() if "vsde" not in k and "Cijk" not in k}
    print(f"{label:34s} kernels {sum(v[0] for v in kern.values()):4d} = {sum(v[1] for v in kern.values()) / 1e3:7.2f} ms;"
          f"  not vsde/hipBLASLt {sum(v[0] for v in small.values()):4d} = {sum(v[1] for v in small.values()) / 1e3:6.2f} ms")
    for k, v in sorted(small.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"      {v[0]:4d} {v[1]:8.1f} us  {k}")
    return out


def main():
    device = torch.device("cuda:0")
    tr = bench.build_trainer(lv_problem(), 512, device, True, seed=1234)
    model, ctx, cfg = tr.ctx.model, tr.ctx, tr.config
    from viforsdes_amd.inference.diffusion_path_sampler import sample_diffusion_paths
    from viforsdes_amd.inference.evidence_lower_bound import compute_evidence_lower_bound

    def step():
        tr._train_step(model)
        ctx.ema.update()

    for _ in range(3):
        step()
    count(step, "whole step", top=12)
    ctx.grad_sync.zero_grad()
    theta = count(lambda: model.sde_parameter_posterior.rsample(cfg.batch_size), "theta ~ q")
    ac = torch.autocast(device_type="cuda", dtype=cfg.amp_dtype.value, enabled=ctx.scaler.is_enabled())
    with ac:
        context = count(lambda: model.encoder(ctx.observations.values, ctx.observations.times, theta, tr.time_horizon, cfg.time_step),
                        "encoder forward", top=10)
        sample = count(lambda: sample_diffusion_paths(lambda *a: context, model.head, ctx.observations, theta, ctx.x0_buffer,
                                                      tr.time_horizon, cfg.time_step, tr.state_space), "head forward (+ noise)", top=6)
        res = count(lambda: compute_evidence_lower_bound(tr.sde, ctx.observations, tr.observation_likelihood, tr.prior,
                                                         model.sde_parameter_posterior, theta, sample, cfg.time_step), "ELBO forward", top=12)
    count(lambda: ctx.scaler.scale(-res.evidence_lower_bound).backward(), "backward (all)", top=40)
    count(tr._optimizer_step, "optimizer step (+ weight refresh)", top=10)
    count(ctx.ema.update, "EMA", top=3)


if __name__ == "__main__":
    main()

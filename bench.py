#!/usr/bin/env python3
"""Headline benchmark: Lotka-Volterra (S=2, T=400, batch 512 per GPU), full ELBO gradient step.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one complete optimizer iteration of the reference's trainer._train_step
(inference/trainer.py:166-206): theta ~ q -> SiT encoder (bf16 autocast) -> fused HIP GRU path
sampler -> fused HIP ELBO -> backward (fused HIP BPTT + encoder autograd) -> [RCCL gradient
all-reduce] -> unscale -> clip -> AdamW -> EMA.  Nets as in the reference's example
(encoder 256 x 8 layers x 4 heads, GRU 64 x 2); random-init weights with the emission matrix
randomised (the default all-zero init makes the GRU irrelevant), synthetic inputs = the example's
observations.  Weak scaling: every rank draws its own 512 paths.

Rank 0 prints ONE JSON line.  value = sample paths pushed through a full ELBO step per second over
all ranks (= global_batch * ELBO-iters/s); ELBO-iters/s and the no-grad sampled-paths/s (encoder +
head, the VariationalPosterior.sample path) are extra fields.  `roofline` describes the dominant
hand-written kernel (the serial GRU time-stepping forward, training variant), timed with HIP
events on its launch stream; `cpu_baseline` times the same step with the CPU oracle standing in
for the HIP kernels on a bounded sample (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X datasheet HBM3E bandwidth (MI355X_MICROARCH.md)


def build_trainer(problem, batch, device, mixed_precision, seed, enc_hidden=256, enc_depth=8, heads=4, head_hidden=64,
                  head_layers=2):
    from viforsdes_amd import EncoderConfig, HeadConfig, TrainingConfig
    from viforsdes_amd.console import Console
    from viforsdes_amd.inference.trainer import VariationalInferenceTrainer
    sde, obs, like, prior, horizon, dt, state_pos, theta_pos = problem
    tr = VariationalInferenceTrainer(
        sde=sde, observations=obs, observation_likelihood=like, prior=prior, time_horizon=horizon,
        config=TrainingConfig(time_step=dt, batch_size=batch, n_iterations=1, learning_rate=1e-4, sde_param_lr=1e-3),
        encoder_config=EncoderConfig(hidden_dim=enc_hidden, num_heads=heads, depth=enc_depth),
        head_config=HeadConfig(hidden_dim=head_hidden, num_layers=head_layers), state_positive_dims=state_pos,
        sde_param_positive_dims=theta_pos, device=device, mixed_precision=mixed_precision, console=Console(enabled=False),
        seed=seed)
    g = torch.Generator(device="cpu").manual_seed(7)
    with torch.no_grad():  # synthetic weights: make the emission depend on the GRU state
        w = tr.ctx.model.head.out_proj.weight
        w.copy_((torch.randn(w.shape, generator=g) * 0.1).to(w.device))
        lam = tr.ctx.model.sde_parameter_posterior
        lam.log_std.fill_(-1.0)  # keep LV draws in a numerically sane range for a throughput run
    tr.ctx.ema._init_shadow()
    tr.ctx.model.train()
    return tr


def pmc_traffic_bytes(workload, batch):
    """HBM bytes per launch of the serial forward kernel from the committed rocprofv3 --pmc passes
    (profiles/r01_pmc_head_lv_v2.txt: FETCH_SIZE and WRITE_SIZE in KiB, separate passes, LV B=512).  The
    kernel reads with 4-byte loads, for which the guide gives no FETCH_SIZE correction, so the raw
    counter is used; null for other workloads."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_head_lv_v2.txt")
    if workload != "lv" or batch != 512 or not os.path.exists(path):
        return None
    vals = {}
    for line in open(path):
        if "head_fwd_v2_kernel" in line:
            parts = line.split()
            ctr = [p for p in parts if p in ("FETCH_SIZE", "WRITE_SIZE")]
            if ctr:
                vals[ctr[0]] = float(line.split("mean=")[1])
    if len(vals) != 2:
        return None
    return (vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0


def sync(device):
    if device.type == "cuda":
        torch.cuda.synchronize(device)


def barrier(distributed):
    if distributed:
        dist.barrier()


def timed(fn, steps, warmup, device, distributed):
    for _ in range(warmup):
        fn()
    sync(device); barrier(distributed); sync(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync(device); barrier(distributed); sync(device)
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def cpu_baseline(problem, sample_batch, steps):
    """The same ELBO step on the host CPUs: torch-CPU encoder + the C oracle for head/ELBO ops."""
    from oracle.torch_backend import OracleBackend
    from viforsdes_amd.kernels.backend import set_backend
    set_backend(OracleBackend())
    try:
        tr = build_trainer(problem, sample_batch, torch.device("cpu"), False, seed=1234)
        tr._train_step(tr.ctx.model)  # warm-up (thread pools, allocator)
        t0 = time.perf_counter()
        for _ in range(steps):
            tr._train_step(tr.ctx.model)
            tr.ctx.ema.update()
        dt = time.perf_counter() - t0
    finally:
        set_backend(None)
    return {"value": sample_batch * steps / dt, "unit": "paths/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"LV T=400 S=2 full ELBO step, batch {sample_batch} x {steps} steps, fp32, torch-CPU encoder + "
                      f"C oracle head/ELBO ({os.cpu_count()} logical CPUs on the host)",
            "elbo_iters_per_sec_at_sample_batch": steps / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="sample paths per GPU")
    ap.add_argument("--workload", default="lv", choices=["lv", "ou", "synthetic"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hip-graph", action="store_true", help="step eagerly instead of replaying a captured HIP graph")
    ap.add_argument("--hip-graph", action="store_true", help="always replay the captured HIP graph (default: whichever of "
                    "eager / replay is faster in a 3-step probe before the warm-up; same kernels and work either way)")
    ap.add_argument("--cpu-sample-batch", type=int, default=16)
    ap.add_argument("--cpu-steps", type=int, default=2)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    distributed = world > 1
    if args.gpus != world and distributed:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if args.gpus > 1 and not distributed:
        raise SystemExit("launch multi-GPU runs with `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the fused kernels have no CPU fallback)")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    if distributed and not dist.is_initialized():
        dist.init_process_group(backend="nccl")

    from viforsdes_amd import _hip
    from viforsdes_amd.examples.sdes import lv_problem, ou_problem, synthetic_problem
    from viforsdes_amd.inference.diffusion_path_sampler import sample_diffusion_paths

    if args.workload == "lv":
        problem, enc = lv_problem(), dict(enc_hidden=256, enc_depth=8)
    elif args.workload == "ou":
        problem, enc = ou_problem(), dict(enc_hidden=256, enc_depth=8)
        if args.batch == 512:
            args.batch = 128
    else:
        problem, enc = synthetic_problem(8), dict(enc_hidden=512, enc_depth=12)
        if args.batch == 512:
            args.batch = 256
    sde, obs, like, prior, horizon, dt, state_pos, theta_pos = problem
    T, S = int(round(horizon / dt)), sde.state_dim
    tr = build_trainer(problem, args.batch, device, True, seed=1234, **enc)
    model, ctx = tr.ctx.model, tr.ctx

    def train_step():
        tr._train_step(model)
        ctx.ema.update()

    # single-process runs may replay the whole step (same kernels, same work) from a HIP graph: a win when the step is
    # launch-bound (OU), a small loss when it is GPU-bound (graph nodes dispatch with a larger gap than a busy eager queue)
    graph_mode = False
    if not distributed and not args.no_hip_graph:
        def probe(fn, n=3):
            fn(); sync(device)
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            sync(device)
            return (time.perf_counter() - t0) / n
        for _ in range(3):
            train_step()
        t_eager = probe(train_step)
        replay = tr.capture_step_graph(warmup=3)
        if replay is not None and (args.hip_graph or probe(replay) < t_eager):
            train_step, graph_mode = replay, True
        elif replay is not None:  # drop the graph and its private memory pool
            del replay
            tr._graph = None
            import gc
            gc.collect()
            torch.cuda.empty_cache()

    elapsed = timed(train_step, args.steps, args.warmup, device, distributed)
    iters_per_sec = args.steps / elapsed
    global_batch = args.batch * world

    # no-grad sampling call: theta rsample -> encoder -> head (eval kernel), as VariationalPosterior.sample
    model.eval()

    @torch.no_grad()
    def sample_step():
        theta = model.sde_parameter_posterior.rsample(args.batch)
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            sample_diffusion_paths(model.encoder, model.head, ctx.observations, theta, ctx.x0_buffer, horizon, dt,
                                   tr.state_space)
    s_elapsed = timed(sample_step, args.steps, max(2, args.warmup // 2), device, distributed)
    model.train()

    # per-kernel timing of the dominant hand-written kernel (HIP events on the launch stream)
    H, L, C = model.head.hidden_dim, model.head.num_layers, model.head.context_dim
    ntril = S * (S + 1) // 2
    _hip.profile_enable(True)
    fwd_ms, bwd_ms = [], []
    for _ in range(5):
        tr._train_step(model)
        fwd_ms.append(_hip.profile_elapsed_ms(0))
        bwd_ms.append(_hip.profile_elapsed_ms(1))
    _hip.profile_enable(False)
    fwd_ms_avg = sum(fwd_ms) / len(fwd_ms)
    bwd_ms_avg = sum(bwd_ms) / len(bwd_ms)
    # algorithmic bytes per path-step of the serial forward (training variant), SURVEY 8(d) with the
    # context term replaced by the 3H-float projected record this kernel actually consumes:
    #   4 * [3H + S + (2S + S^2 + n_tril) + 5 L H]
    fwd_bytes_step = 4 * (3 * H + S + (2 * S + S * S + ntril) + 5 * L * H)
    bwd_bytes_step = 4 * (4 * S + S * S + ntril + 5 * L * H + 4 * L * H + (S + ntril))
    fwd_bytes = fwd_bytes_step * args.batch * T
    achieved = fwd_bytes / (fwd_ms_avg * 1e-3) / 1e9

    out = {
        "metric": "sampled-paths/sec + ELBO-iters/sec (full ELBO gradient step; value = global_batch * ELBO-iters/s)",
        "value": global_batch * iters_per_sec, "unit": "paths/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 (GRU/ELBO kernels) + bf16-autocast encoder", "data": "synthetic",
        "config": {"workload": f"{args.workload}: state_dim={S}, T={T} Euler steps (dt={dt}), batch={args.batch}/GPU, "
                               f"encoder {enc['enc_hidden']}x{enc['enc_depth']}x4 heads, GRU {H}x{L}",
                   "global_batch": global_batch, "parallelism": f"dp{world}", "hip_graph": graph_mode},
        "elbo_iters_per_sec": iters_per_sec,
        "sampled_paths_per_sec": global_batch * args.steps / s_elapsed,
        "roofline": {"kernel": f"vsde::head_fwd_v2_kernel<{L}, true, ...> (serial GRU time-stepping forward, training variant)",
                     "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic_bytes(args.workload, args.batch),
                     "avg_ms": fwd_ms_avg, "algorithmic_bytes": fwd_bytes, "bytes_per_path_step": fwd_bytes_step,
                     # SURVEY 8(d) prices the forward with the raw C-float context read (4*[C + S + 2S + S^2 + ntril + 5LH]); this
                     # kernel streams the 3H-float projected record instead (the context itself is read by the projection GEMM)
                     "survey_bytes_per_path_step": 4 * (C + S + (2 * S + S * S + ntril) + 5 * L * H),
                     "frac_with_survey_bytes": 4 * (C + S + (2 * S + S * S + ntril) + 5 * L * H) * args.batch * T
                                               / (fwd_ms_avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "bwd_kernel_avg_ms": bwd_ms_avg,
                     "bwd_achieved_GBs": bwd_bytes_step * args.batch * T / (bwd_ms_avg * 1e-3) / 1e9},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "lv":
        out["cpu_baseline"] = cpu_baseline(problem, args.cpu_sample_batch, args.cpu_steps)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

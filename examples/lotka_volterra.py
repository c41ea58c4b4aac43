"""Lotka-Volterra example: same problem and network sizes as the reference's examples/lotka_volterra.py
(5 observations over horizon 40, dt=0.1 -> 400 Euler steps, softplus state transform, LogNormal prior).
The reference trains with batch 24; the benchmark configuration of this repository uses 512 (--batch).

    python examples/lotka_volterra.py [--iterations 30000] [--batch 24]
"""
from __future__ import annotations

import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from viforsdes_amd import EncoderConfig, HeadConfig, InferenceConfig, PretrainConfig, TrainingConfig, infer
from viforsdes_amd.console import Console
from viforsdes_amd.examples.sdes import lv_problem


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=30000)
    ap.add_argument("--batch", type=int, default=24)
    ap.add_argument("--no-pretrain", action="store_true")
    ap.add_argument("--out", default="lotka_volterra_posterior.pt")
    args = ap.parse_args()
    sde, obs, like, prior, horizon, dt, state_pos, theta_pos = lv_problem()
    names = ["theta1", "theta2", "theta3"]
    console = Console()
    posterior = infer(
        sde=sde, observations=obs, observation_likelihood=like, prior=prior, time_horizon=horizon,
        config=InferenceConfig(
            training=TrainingConfig(time_step=dt, batch_size=args.batch, n_iterations=args.iterations, learning_rate=1e-4,
                                    sde_param_lr=1e-3, grad_clip_norm=1.0),
            encoder=EncoderConfig(hidden_dim=256, num_heads=4, depth=8), head=HeadConfig(hidden_dim=64, num_layers=2),
            state_positive_dims=state_pos, sde_param_positive_dims=theta_pos, console=console, param_names=names,
            pretrain=False if args.no_pretrain else PretrainConfig()))
    console.summary_table(posterior.summary(n_samples=500), posterior.diagnostics(), param_names=names)
    posterior.save(args.out)


if __name__ == "__main__":
    main()

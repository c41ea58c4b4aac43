"""Ornstein-Uhlenbeck example: same problem and network sizes as the reference's
examples/ornstein_uhlenbeck.py (6 observations, dt=0.05, batch 128, encoder 256x8x4, GRU 64x2).

    python examples/ornstein_uhlenbeck.py [--iterations 20000] [--no-pretrain]
"""
from __future__ import annotations

import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from viforsdes_amd import EncoderConfig, HeadConfig, InferenceConfig, PretrainConfig, TrainingConfig, infer
from viforsdes_amd.console import Console
from viforsdes_amd.examples.sdes import ou_problem


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=20000)
    ap.add_argument("--no-pretrain", action="store_true")
    ap.add_argument("--out", default="ou_posterior.pt")
    args = ap.parse_args()
    sde, obs, like, prior, horizon, dt, state_pos, theta_pos = ou_problem()
    names = ["kappa", "mu", "sigma"]
    console = Console()
    posterior = infer(
        sde=sde, observations=obs, observation_likelihood=like, prior=prior, time_horizon=horizon,
        config=InferenceConfig(
            training=TrainingConfig(time_step=dt, batch_size=128, n_iterations=args.iterations, learning_rate=1e-4,
                                    sde_param_lr=1e-3, grad_clip_norm=1.0),
            encoder=EncoderConfig(hidden_dim=256, num_heads=4, depth=8), head=HeadConfig(hidden_dim=64, num_layers=2),
            sde_param_positive_dims=theta_pos, console=console, param_names=names,
            pretrain=False if args.no_pretrain else PretrainConfig()))
    console.summary_table(posterior.summary(n_samples=500), posterior.diagnostics(), param_names=names)
    posterior.save(args.out)


if __name__ == "__main__":
    main()

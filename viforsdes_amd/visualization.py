"""Optional matplotlib plot of posterior trajectories (reference: visualization.py; plotting is out
of the hot-path scope, kept so ``VariationalPosterior.plot`` exists)."""
from __future__ import annotations


def plot_posterior(samples, observations, time_horizon: float, show: bool = True):
    import matplotlib.pyplot as plt
    import torch
    paths = samples.diffusion_paths.detach().cpu()
    n_steps, S = paths.shape[1], paths.shape[2]
    grid = torch.linspace(0, time_horizon, n_steps)
    fig, axes = plt.subplots(1, S, figsize=(5 * S, 3.5), squeeze=False)
    for d in range(S):
        ax = axes[0][d]
        ax.plot(grid, paths[:, :, d].T, color="C0", alpha=0.25, lw=0.8)
        if d < observations.values.shape[-1]:
            ax.scatter(observations.times.cpu(), observations.values[:, d].cpu(), color="k", zorder=3)
        ax.set_xlabel("t"); ax.set_title(f"state {d}")
    if show:
        plt.show()
    return fig

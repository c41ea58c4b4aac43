"""Minimal progress/metrics sink with the method names the trainer and ``InferenceConfig.console``
use (reference: console.py:106-142 ``TrainingProgress.update``).  The reference's rich-based live
UI is out of scope; this prints a line every ``update_interval`` steps."""
from __future__ import annotations

import sys
import time
from contextlib import contextmanager
from typing import Any, Iterator, Optional


class TrainingProgress:
    def __init__(self, console: "Console", total: int, update_interval: int, param_names: Optional[list[str]]) -> None:
        self.console, self.total, self.update_interval, self.param_names = console, total, update_interval, param_names
        self._t0 = time.perf_counter()
        self.last: dict[str, Any] = {}

    def update(self, step: int, loss: float, elbo: float, best_elbo: float, components: Any = None,
               grad_norm: Optional[float] = None, param_means: Any = None) -> None:
        self.last = dict(step=step, loss=loss, elbo=elbo, best_elbo=best_elbo, grad_norm=grad_norm)
        if not self.console.enabled or (step + 1) % self.update_interval and step + 1 != self.total:
            return
        rate = (step + 1) / max(time.perf_counter() - self._t0, 1e-9)
        msg = f"[{step + 1}/{self.total}] loss {loss:.4f} elbo {elbo:.4f} best {best_elbo:.4f} {rate:.2f} it/s"
        if grad_norm is not None:
            msg += f" |g| {grad_norm:.3g}"
        if param_means is not None:
            vals = [float(v) for v in param_means.detach().flatten().tolist()]
            names = self.param_names or [f"p{i}" for i in range(len(vals))]
            msg += " " + " ".join(f"{n}={v:.4g}" for n, v in zip(names, vals))
        print(msg, file=self.console.stream, flush=True)


class PretrainProgress:
    def __init__(self, console: "Console", total: int) -> None:
        self.console, self.total = console, total

    def update(self, step: int, mse: float, best_mse: float, sigma_median: float) -> None:
        if self.console.enabled and ((step + 1) % 100 == 0 or step + 1 == self.total):
            print(f"[pretrain {step + 1}/{self.total}] mse {mse:.5g} best {best_mse:.5g} sigma~{sigma_median:.3g}",
                  file=self.console.stream, flush=True)


class Console:
    def __init__(self, enabled: bool = True, stream=None) -> None:
        self.enabled, self.stream = enabled, stream or sys.stderr

    def config_panel(self, config: Any) -> None:
        if self.enabled:
            print(f"config: {config}", file=self.stream, flush=True)

    @contextmanager
    def training_progress(self, total: int, update_interval: int = 10, param_names: Optional[list[str]] = None
                          ) -> Iterator[TrainingProgress]:
        yield TrainingProgress(self, total, update_interval, param_names)

    @contextmanager
    def pretrain_progress(self, total: int) -> Iterator[PretrainProgress]:
        yield PretrainProgress(self, total)

    def summary_table(self, summary: Any, diagnostics: Any, param_names: Optional[list[str]] = None) -> None:
        if not self.enabled:
            return
        means = summary.sde_parameter_mean.tolist()
        stds = summary.sde_parameter_std.tolist()
        names = param_names or [f"p{i}" for i in range(len(means))]
        for n, m, s in zip(names, means, stds):
            print(f"{n}: {m:.4f} +- {s:.4f}", file=self.stream)
        print(f"final ELBO {diagnostics.final_evidence_lower_bound:.4f} after {diagnostics.n_iterations} iterations",
              file=self.stream, flush=True)

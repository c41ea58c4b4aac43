"""Progress / metrics sink with the method names the trainer and ``InferenceConfig.console`` use.

``TrainingProgress.update`` keeps the same quantities as the reference's live panel
(reference: console.py:106-142 update, :144-228 display): iterations/s as an exponential moving
average (0.9 / 0.1 over the instantaneous rate between updates), elapsed / ETA, loss, ELBO, best
ELBO, gradient norm, the posterior parameter means, the five ELBO components and the device memory
in use.  The reference renders them with ``rich`` (UI, out of scope); here they are a plain
``metrics()`` dict, one text line per ``update_interval`` steps, and optionally one JSON object per
line appended to ``Console(metrics_path=...)`` for dashboards / regression tracking."""
from __future__ import annotations

import json
import sys
import time
from contextlib import contextmanager
from typing import Any, Iterator, Optional

COMPONENT_FIELDS = (("observation", "observation_log_prob"), ("sde", "sde_log_prob"), ("generative", "generative_log_prob"),
                    ("prior", "prior_log_prob"), ("posterior", "posterior_log_prob"))


def _scalar(v: Any) -> Optional[float]:
    if v is None:
        return None
    return float(v.item()) if hasattr(v, "item") else float(v)


class TrainingProgress:
    def __init__(self, console: "Console", total: int, update_interval: int, param_names: Optional[list[str]]) -> None:
        self.console, self.total, self.update_interval, self.param_names = console, total, max(1, update_interval), param_names
        self._start_time = self._last_time = time.perf_counter()
        self._last_step = -1
        self.iter_per_sec = 0.0
        self.last: dict[str, Any] = {}

    def update(self, step: int, loss: float, elbo: float, best_elbo: float, components: Any = None,
               grad_norm: Optional[float] = None, param_means: Any = None) -> None:
        now = time.perf_counter()
        if step > self._last_step:
            dt = now - self._last_time
            if dt > 0:
                instant = (step - self._last_step) / dt
                # same smoothing as the reference (console.py:120-122); seeded with the first measurement instead of 0
                self.iter_per_sec = instant if self.iter_per_sec == 0.0 else 0.9 * self.iter_per_sec + 0.1 * instant
            self._last_step, self._last_time = step, now
        emit = self.console.enabled and ((step + 1) % self.update_interval == 0 or step + 1 == self.total)
        self.last = dict(step=step, loss=loss, elbo=elbo, best_elbo=best_elbo, grad_norm=grad_norm,
                         iter_per_sec=self.iter_per_sec, elapsed_s=now - self._start_time)
        if not emit:
            return
        m = self.metrics(components, param_means)
        msg = (f"[{step + 1}/{self.total}] loss {m['loss']:.4f} elbo {m['elbo']:.4f} best {m['best_elbo']:.4f} "
               f"{m['iter_per_sec']:.2f} it/s eta {m['eta_s']:.0f}s")
        if m["grad_norm"] is not None:
            msg += f" |g| {m['grad_norm']:.3g}"
        if "components" in m:
            msg += " " + " ".join(f"{k[:3]}={v:+.2f}" for k, v in m["components"].items())
        if "param_means" in m:
            msg += " " + " ".join(f"{n}={v:.4g}" for n, v in m["param_means"].items())
        if m.get("memory_allocated_gb") is not None:
            msg += f" mem {m['memory_allocated_gb']:.2f}GB"
        print(msg, file=self.console.stream, flush=True)
        self.console.write_metrics(m)

    def metrics(self, components: Any = None, param_means: Any = None) -> dict[str, Any]:
        """Everything the reference's panel shows, as plain numbers (device scalars are synchronised here, i.e. only
        when a line is actually emitted)."""
        m = dict(self.last)
        for key in ("loss", "elbo", "best_elbo", "grad_norm"):   # device scalars: the only place they are synchronised
            if m.get(key) is not None:
                m[key] = _scalar(m[key])
        step = m.get("step", -1)
        m["eta_s"] = (self.total - step - 1) / max(self.iter_per_sec, 0.01)
        if components is not None:
            m["components"] = {label: _scalar(getattr(components, attr)) for label, attr in COMPONENT_FIELDS}
        if param_means is not None:
            vals = [float(v) for v in param_means.detach().flatten().tolist()]
            names = self.param_names or [f"theta[{i}]" for i in range(len(vals))]
            m["param_means"] = dict(zip(names, vals))
        try:
            import torch
            m["memory_allocated_gb"] = torch.cuda.memory_allocated() / 1024 ** 3 if torch.cuda.is_available() else None
        except Exception:
            m["memory_allocated_gb"] = None
        return m


class PretrainProgress:
    def __init__(self, console: "Console", total: int) -> None:
        self.console, self.total = console, total

    def update(self, step: int, mse: float, best_mse: float, sigma_median: float) -> None:
        if self.console.enabled and ((step + 1) % 100 == 0 or step + 1 == self.total):
            print(f"[pretrain {step + 1}/{self.total}] mse {mse:.5g} best {best_mse:.5g} sigma~{sigma_median:.3g}",
                  file=self.console.stream, flush=True)


class Console:
    def __init__(self, enabled: bool = True, stream=None, metrics_path: Optional[str] = None) -> None:
        self.enabled, self.stream, self.metrics_path = enabled, stream or sys.stderr, metrics_path

    def write_metrics(self, record: dict[str, Any]) -> None:
        if self.metrics_path:
            with open(self.metrics_path, "a") as f:
                f.write(json.dumps(record) + "\n")

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

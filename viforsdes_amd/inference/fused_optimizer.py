"""The optimizer step of the ELBO trainer as two HIP launches (csrc/vsde_optim.hip).

Reference sequence (inference/trainer.py:197-204, inference/exponential_moving_average.py:27-32)::

    scaler.unscale_(optimizer); clip_grad_norm_(parameters, max_norm); scaler.step(optimizer); scaler.update(); ema.update()

= five multi-tensor passes over ~200 parameter tensors (~390 us at the LV model).  ``FusedOptimizerStep`` keeps the
``torch.optim.AdamW`` object as the owner of the hyper-parameters and of the moment tensors (``state_dict`` / ``load_state_dict``
work as before) and the ``GradScaler`` as the owner of the loss scale, and replaces the arithmetic by ``_hip.optim_step``:
one pass for the global gradient norm / non-finite check, one pass for unscale x clip, AdamW and the EMA lerp.  The scaler is
told the outcome (``found_inf``) the way ``GradScaler.step`` would, so ``scaler.update()`` adjusts the scale as usual.

Used on CUDA for fp32, contiguous parameters with plain AdamW (no amsgrad / maximize); ``VSDE_FUSED_OPTIMIZER=0`` keeps the torch
sequence.  Deterministic (fixed-order reductions), graph-capture safe (step count and outcome stay on the device).

Gradients after the step: ``p.grad`` keeps what autograd wrote (loss-SCALED and unclipped) -- the kernel reads the gradients, it does
not rewrite them the way ``unscale_`` / ``clip_grad_norm_`` do in place.  Code that inspects gradients after ``_optimizer_step`` must
divide by ``scaler.get_scale()`` itself (nothing in the package does).

EMA contract: the kernel applies this step's EMA lerp and sets ``ema.fused_step_done``; the ``ema.update()`` call that the
trainer's loops (and the reference's) issue after every step then returns at once.  A caller that steps WITHOUT updating the EMA
must construct the step with ``ema=None`` (``VariationalInferenceTrainer.fuse_ema = False``).
"""
from __future__ import annotations

import os
from typing import Optional

import torch
from torch import Tensor

from .. import _hip

ENABLED = os.environ.get("VSDE_FUSED_OPTIMIZER", "1") != "0"


class FusedOptimizerStep:
    def __init__(self, optimizer: torch.optim.Optimizer, ema, scaler, max_norm: float) -> None:
        self.optimizer, self.ema, self.scaler, self.max_norm = optimizer, ema, scaler, float(max_norm)
        self._key = None
        self._disabled = False
        self._captured_ptr_hosts: list[Tensor] = []   # pinned address tables of captured steps (alive as long as this object)
        self._hooks = [optimizer.register_state_dict_pre_hook(lambda opt: self._publish_steps()),
                       optimizer.register_load_state_dict_post_hook(lambda opt: self._invalidate())]

    # ------------------------------------------------------------------------------------------------ eligibility
    @staticmethod
    def usable(optimizer: torch.optim.Optimizer, scaler=None) -> bool:
        if not (ENABLED and isinstance(optimizer, torch.optim.AdamW)):
            return False
        if scaler is not None and scaler.is_enabled():
            # the outcome of the step is handed to GradScaler through the two attributes its own step() fills
            try:
                from torch.amp.grad_scaler import OptState  # noqa: F401
            except ImportError:
                return False
            if not (hasattr(scaler, "_per_optimizer_states") and hasattr(scaler, "_scale")):
                return False
        for g in optimizer.param_groups:
            if g.get("amsgrad", False) or g.get("maximize", False) or isinstance(g["lr"], Tensor):
                return False
            for p in g["params"]:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    return False
        return True

    def _invalidate(self) -> None:
        self._key = None

    # ------------------------------------------------------------------------------------------------ tables
    def _params(self) -> list[Tensor]:
        return [p for g in self.optimizer.param_groups for p in g["params"] if p.requires_grad]

    def _build(self, params: list[Tensor]) -> bool:
        opt, dev = self.optimizer, params[0].device
        chunk = _hip.optim_chunk_elems()
        shadow_of = {}
        if self.ema is not None:
            shadow_of = {id(p): self.ema.shadow[n] for n, p in self.ema.model.named_parameters()}
        rows, steps = [], []
        index = {id(p): i for i, p in enumerate(params)}
        for gi, g in enumerate(opt.param_groups):
            for p in g["params"]:
                if not p.requires_grad:
                    continue
                st = opt.state[p]
                if "exp_avg" not in st:   # as torch's lazy state initialisation (fused / capturable: the step count is a device float)
                    st["step"] = torch.zeros((), dtype=torch.float32, device=dev)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                steps.append(float(st["step"]))
                m, v = st["exp_avg"], st["exp_avg_sq"]
                sh = shadow_of.get(id(p))
                if not (m.is_contiguous() and v.is_contiguous() and (sh is None or (sh.is_contiguous() and sh.dtype == torch.float32))):
                    raise RuntimeError("optimizer state is not contiguous fp32")
                for off in range(0, p.numel(), chunk):
                    n = min(chunk, p.numel() - off)
                    rows.append((p.data_ptr() + 4 * off, m.data_ptr() + 4 * off, v.data_ptr() + 4 * off,
                                 0 if sh is None else sh.data_ptr() + 4 * off, index[id(p)] | (n << 32), off, gi, 0))
        if len(set(steps)) > 1:
            # e.g. a parameter had no gradient in a step the torch sequence ran: torch counts steps per parameter, the kernel has
            # ONE count -- leave this optimizer to the torch sequence from here on
            self._disabled = True
            return False
        self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
        self.partials = torch.zeros(len(rows), device=dev, dtype=torch.float32)
        self.tstate = torch.tensor([steps[0], steps[0]], dtype=torch.float32).to(dev)
        self.out = torch.zeros(2, device=dev, dtype=torch.float32)
        # the gradients' addresses travel through pinned host buffers; the CPU runs ahead of the GPU, so a buffer is only rewritten
        # once the copy that read it has completed (ring of 4, one event each)
        self.ptr_host = [torch.zeros(len(params), dtype=torch.int64).pin_memory() for _ in range(4)]
        self.ptr_np = [t.numpy() for t in self.ptr_host]
        self.ptr_events: list = [None] * 4
        self.ptr_slot = 0
        self.ptr_dev = torch.zeros(len(params), dtype=torch.int64, device=dev)
        self._capture_pool = [torch.zeros(len(params), dtype=torch.int64).pin_memory() for _ in range(8)]
        self.has_ema = self.ema is not None and all(id(p) in shadow_of for p in params)
        # hyper-parameters: ONE device tensor for the life of these tables (a captured launch keeps its address), refilled in
        # place through a pinned staging buffer when a group's lr / betas / eps / weight decay changes
        self._groups_key = None
        self.groups = torch.zeros(len(opt.param_groups), 5, dtype=torch.float64, device=dev)
        self.groups_host = torch.zeros(len(opt.param_groups), 5, dtype=torch.float64).pin_memory()
        self.groups_event = None
        return True

    def _hyper(self) -> Tensor:
        key = tuple((float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]))
                    for g in self.optimizer.param_groups)
        if key != self._groups_key:
            if self.groups_event is not None:
                self.groups_event.synchronize()
            self.groups_host.copy_(torch.tensor(key, dtype=torch.float64))
            self.groups.copy_(self.groups_host, non_blocking=True)
            if not torch.cuda.is_current_stream_capturing():
                self.groups_event = torch.cuda.Event()
                self.groups_event.record()
            self._groups_key = key
        return self.groups

    def _publish_steps(self) -> None:
        """``optimizer.state_dict()`` sees the per-parameter step tensors torch keeps: fill them from the shared device count."""
        if self._key is None:
            return
        for p in self._params():
            st = self.optimizer.state.get(p)
            if st is not None and "step" in st:
                st["step"].copy_(self.tstate[1])

    # ------------------------------------------------------------------------------------------------ the step
    @torch.no_grad()
    def step(self) -> Optional[Tensor]:
        """Runs the step and returns the global gradient norm (0-dim device tensor), or None when this step cannot take the
        fused route (a parameter without gradient, a non-contiguous gradient): the caller then runs the torch sequence."""
        if self._disabled:
            return None
        params = self._params()
        grads = [p.grad for p in params]
        if not params or any(g is None or g.dtype != torch.float32 or not g.is_contiguous() for g in grads):
            # the torch sequence runs this step: hand it the current step counts, and re-read them from its state next time
            self._publish_steps()
            self._invalidate()
            return None
        ema_v = getattr(self.ema, "version", 0) if self.ema is not None else 0
        # identity AND storage: model.to(...) / p.data = ... keep the Parameter object and move what the table points at
        key = (tuple((id(p), p.data_ptr()) for p in params), ema_v)
        if key != self._key:
            self._publish_steps()   # a rebuild (new EMA shadows) must not lose the step count
            if not self._build(params):
                return None
            self._key = key
        if torch.cuda.is_current_stream_capturing():
            # a captured copy re-reads its host buffer at EVERY replay: it gets a pinned buffer of its own that nothing else ever
            # writes (the gradients of a captured step live in the graph's pool: these addresses stay valid for the graph's life)
            # (taken from a pool pinned outside of any capture: hipHostMalloc is not a legal call on a capturing thread)
            host = self._capture_pool.pop() if self._capture_pool else torch.zeros(len(params), dtype=torch.int64).pin_memory()
            host.numpy()[:] = [g.data_ptr() for g in grads]
            self._captured_ptr_hosts.append(host)
            self.ptr_dev.copy_(host, non_blocking=True)
        else:
            slot = self.ptr_slot = (self.ptr_slot + 1) % 4
            if self.ptr_events[slot] is not None:
                self.ptr_events[slot].synchronize()
            self.ptr_np[slot][:] = [g.data_ptr() for g in grads]   # the gradients are new allocations every step
            self.ptr_dev.copy_(self.ptr_host[slot], non_blocking=True)
            ev = self.ptr_events[slot] = torch.cuda.Event()
            ev.record()
        scaler = self.scaler
        scale = scaler._scale if (scaler is not None and scaler.is_enabled()) else None
        ema_w = (1.0 - self.ema.decay) if self.has_ema else -1.0
        _hip.optim_step(self.table, self.ptr_dev, scale, self.partials, self.tstate, self._hyper(), self.max_norm, ema_w, self.out)
        if scale is not None:   # what GradScaler.unscale_ / .step record for .update()
            from torch.amp.grad_scaler import OptState
            st = scaler._per_optimizer_states[id(self.optimizer)]
            st["found_inf_per_device"] = {self.out.device: self.out[1]}
            st["stage"] = OptState.STEPPED
        if self.has_ema:
            self.ema.fused_step_done = True
        # Tensor._version is what cached views of the parameters (primitives/fused.py::PackedWeight) compare: the kernel wrote
        # the parameters behind autograd's back, so say so (host-side counters; under capture this runs once -- a replayed step
        # re-fills the packs in its own captured tail, and the trainer's replay() checks for packs someone else rewrote)
        torch.autograd.graph.increment_version(params)
        from ..primitives import fused
        fused.note_parameters_changed()
        # a fresh scalar per eager step (clip_grad_norm_ returns one too); a captured step hands out the static output slot
        return self.out[0] if torch.cuda.is_current_stream_capturing() else self.out[0].clone()

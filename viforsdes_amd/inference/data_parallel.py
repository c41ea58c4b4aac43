"""Data parallelism over Monte-Carlo sample paths: one process per GPU, RCCL over xGMI.

Every path (and its theta draw) is independent through encoder, head and ELBO; the only coupling
is the mean over the batch in the loss, i.e. a gradient average.  Parameters are replicated
(8.3 M fp32 = 33 MB at the example configs), each rank draws its own ``batch_size`` samples with
seed ``base + rank`` and the gradients are averaged once per optimizer step, before unscale/clip
so that the clipping norm is global.

All gradients live in ONE flat fp32 buffer (``p.grad`` are views into it), so the exchange is a
single large all-reduce per step (split into at most ``max_buckets`` launches): with 8 GPUs fully
connected by point-to-point xGMI links a few large messages beat many small ones, and at 33 MB the
step is latency-, not bandwidth-, bound.

The reference wraps the model in DDP but never calls the wrapper's forward, so its reducer is
never armed and gradients are not synchronised; this module implements the
intended semantics instead of copying that behaviour."""
from __future__ import annotations

import os
from typing import Iterable, Optional

import torch
import torch.distributed as dist
from torch import Tensor, nn


def env_rank_info() -> tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment."""
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", os.environ.get("LOCAL_RANK", 0)))
    local = int(os.environ.get("LOCAL_RANK", 0))
    return rank, local, world


def init_process_group_if_needed(device_type: str) -> bool:
    """Initialise torch.distributed from the environment. ``nccl`` on ROCm *is* RCCL."""
    _, _, world = env_rank_info()
    if world <= 1:
        return False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC, needed by RCCL on this driver) is launcher environment: it is read when
        # HSA initialises, long before this point; viforsdes_amd/__init__.py defaults it at import time
        dist.init_process_group(backend="nccl" if device_type == "cuda" else "gloo")
    return True


def _deferred_ids(pending_only: bool) -> set:
    """ids of the parameters whose weight gradients the encoder's deferred path writes at the end of the backward pass
    (``pending_only``: only those with a product queued right now)."""
    try:
        from ..primitives import fused
    except Exception:   # (the module is importable without the encoder's HIP operators)
        return set()
    return fused.deferred_parameter_ids(pending_only)


class FlatGradientAllReduce:
    """Averages the gradients of ``params`` across ranks through ONE flat fp32 buffer.

    Single process: nothing is copied or reduced -- ``zero_grad`` just drops the ``.grad`` tensors, so autograd writes
    each gradient once instead of accumulating into a pre-zeroed buffer (one fewer kernel per parameter and step).
    Multi process: after backward the gradients are packed into the flat buffer with one multi-tensor copy, the buffer
    is all-reduced in at most ``max_buckets`` pieces and ``.grad`` is re-pointed at its views (what unscale / clip / the
    optimizer then read)."""

    def __init__(self, params: Iterable[nn.Parameter], max_buckets: int = 2, force_buffer: bool = False,
                 overlap: Optional[bool] = None) -> None:
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        self.world_size = dist.get_world_size() if dist.is_initialized() else 1
        self.max_buckets = max_buckets
        self.flat: Optional[Tensor] = None
        self._views: list[Tensor] = []
        self.buckets: list[Tensor] = []
        self.active = self.world_size > 1 or force_buffer  # force_buffer: exercise the packed path on one rank (tests)
        # Overlap (round 5): the gradients that are finished EARLY in the backward pass (head, theta posterior, the last encoder
        # blocks) travel while the rest of the backward still runs.  The arrival order is recorded by post-accumulate hooks during
        # the first step; from the second step on the flat buffer is laid out [early | late] and the early bucket's all-reduce
        # is issued from the hook of its last arrival.  Same sums, element by element: results are bit-identical to the plain
        # path (tests/test_data_parallel.py).  VSDE_DP_OVERLAP=0 switches it off; VSDE_DP_OVERLAP_MIN: minimum payload (elements).
        env = os.environ.get("VSDE_DP_OVERLAP")
        self.overlap = (env != "0") if overlap is None else overlap
        self.overlap_min = int(os.environ.get("VSDE_DP_OVERLAP_MIN", str(1 << 20)))
        self.early_launches = 0                       # how many steps sent their early bucket from inside the backward
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._recording: Optional[list[int]] = None   # arrival order of the step being recorded
        self._early: list[int] = []                   # parameter indices of the early bucket (empty: no overlap)
        self._early_set: set[int] = set()
        self._arrived = 0
        self._early_handle = None
        self._early_sent = False
        self._laid_out = False
        self._hooks: list = []
        if self.active:
            self._allocate()
            if self.overlap:
                self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def close(self) -> None:
        """Remove the gradient hooks (an instance that is dropped while its parameters live on: tools, the bench's probe)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def _allocate(self, order: Optional[list[int]] = None, n_early: int = 0) -> None:
        """Flat buffer and the per-parameter views in ``order`` (default: parameter order); with ``n_early`` > 0 the first
        bucket is exactly the first ``n_early`` parameters of the order."""
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        if self.flat is None:
            self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        order = list(range(len(self.params))) if order is None else order
        views: list[Optional[Tensor]] = [None] * len(self.params)
        off, early_end = 0, 0
        for k, i in enumerate(order):
            p = self.params[i]
            views[i] = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
            if k + 1 == n_early:
                early_end = off
        self._views = views  # type: ignore[assignment]
        if n_early > 0 and 0 < early_end < total:
            self.buckets = [self.flat[:early_end], self.flat[early_end:]]
        else:
            n = max(1, min(self.max_buckets, total // (1 << 20) or 1))
            cuts = [round(i * total / n) for i in range(n + 1)]
            self.buckets = [self.flat[cuts[i]:cuts[i + 1]] for i in range(n)]

    # ---- overlap of the early bucket with the rest of the backward pass
    def _on_grad(self, p: Tensor) -> None:
        i = self._index.get(id(p))
        if i is None:
            return
        if self._recording is not None:
            self._recording.append(i)
            return
        if not self._early or self._early_sent or i not in self._early_set:
            return
        self._arrived += 1
        if self._arrived == len(self._early):
            if p.is_cuda and torch.cuda.is_current_stream_capturing():
                return   # a captured backward: the collective stays outside the graph (trainer.capture_step_graph)
            self._send_early()

    @torch.no_grad()
    def _send_early(self) -> None:
        if _deferred_ids(pending_only=True) & {id(self.params[i]) for i in self._early}:
            return   # a queued (deferred) weight-gradient product will still add to an early parameter at the end of the backward
        have = [(self._views[i], self.params[i].grad) for i in self._early
                if self.params[i].grad is not None and self.params[i].grad is not self._views[i]]
        if len(have) + sum(1 for i in self._early if self.params[i].grad is self._views[i]) != len(self._early):
            return   # a gradient of the early set is absent this step: the plain path handles everything after the backward
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        if self.world_size > 1:
            self._early_handle = dist.all_reduce(self.buckets[0], op=dist.ReduceOp.SUM, async_op=True)
        self._early_sent = True
        self.early_launches += 1

    def _propose_early(self) -> list[int]:
        """This rank's proposal for the early bucket from the arrival order it recorded (empty: no overlap)."""
        seen: set[int] = set()
        arrived = [i for i in (self._recording or []) if not (i in seen or seen.add(i))]
        # parameters the deferred weight-gradient path (primitives/fused.py) has ever written are complete only at the END of the
        # backward, whatever their hooks said earlier: never early
        late_ids = _deferred_ids(pending_only=False)
        arrived = [i for i in arrived if id(self.params[i]) not in late_ids]
        total = sum(p.numel() for p in self.params)
        if not self.overlap or total < self.overlap_min or len(arrived) < 2:
            return []
        early, acc = [], 0
        for i in arrived[:-1]:              # at least one arrival stays late: the hook of the last early one fires mid-backward
            early.append(i)
            acc += self.params[i].numel()
            if acc * 2 >= total:
                break
        return early

    def _finish_recording(self) -> None:
        """First step done: lay the flat buffer out as [early | late].  The layout is RANK 0's: every rank proposes from its own
        arrival order and environment (VSDE_DP_OVERLAP*, deferred weight gradients), then rank 0's index list is broadcast and
        adopted by all (what DDP does when it rebuilds its buckets) -- ranks whose order or switches differ would otherwise
        all-reduce buffers whose elements belong to different parameters, silently.  A rank may still decline to SEND its early
        bucket from inside the backward (overlap off locally, an absent gradient): it then issues the same two collectives in
        the same order after the backward."""
        early = self._propose_early()
        self._recording = None
        self._laid_out = True
        if self.world_size > 1:
            n = len(self.params)
            plan = torch.full((n + 1,), -1, dtype=torch.int64)
            plan[0] = len(early)
            if early:
                plan[1:1 + len(early)] = torch.tensor(early, dtype=torch.int64)
            plan = plan.to(self.params[0].device)
            dist.broadcast(plan, src=0)
            vals = plan.tolist()
            early = [int(v) for v in vals[1:1 + int(vals[0])]]
            if any(not (0 <= i < n) for i in early) or len(set(early)) != len(early):   # a different model on rank 0
                raise RuntimeError("data-parallel ranks disagree on the parameter list (early-bucket plan out of range)")
        if not early:
            return
        rest = [i for i in range(len(self.params)) if i not in set(early)]
        self._early, self._early_set = early, set(early)
        self._allocate(order=early + rest, n_early=len(early))

    def early_fraction(self) -> float:
        """Share of the payload that travels from inside the backward pass (0 until the [early | late] layout exists)."""
        if not self._early or self.flat is None:
            return 0.0
        return float(self.buckets[0].numel()) / float(self.flat.numel())

    def flat_gradients(self) -> Tensor:
        """Copy of all gradients as one fp32 vector in parameter order (zeros where a gradient is absent)."""
        return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).detach().float().reshape(-1)
                          for p in self.params])

    def zero_grad(self) -> None:
        """Replaces ``optimizer.zero_grad(set_to_none=True)``."""
        for p in self.params:
            p.grad = None
        self._arrived, self._early_sent, self._early_handle = 0, False, None
        if self.active and not self._laid_out and self._recording is None and (self.overlap or self.world_size > 1):
            self._recording = []   # (multi-rank: every rank takes part in the layout agreement, whatever its own switches say)

    @torch.no_grad()
    def pack(self) -> None:
        """Gather every ``p.grad`` into the flat buffer (one multi-tensor copy; absent gradients become zeros)."""
        skip = self._early_set if self._early_sent else ()
        have = [(v, p.grad) for i, (p, v) in enumerate(zip(self.params, self._views))
                if i not in skip and p.grad is not None and p.grad is not v]
        missing = [v for i, (p, v) in enumerate(zip(self.params, self._views)) if i not in skip and p.grad is None]
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        for v in missing:
            v.zero_()

    @torch.no_grad()
    def reduce(self) -> None:
        """Average the flat buffer over the ranks: at most ``max_buckets`` RCCL all-reduces, then one scale."""
        if self.world_size > 1:
            todo = self.buckets[1:] if self._early_sent else self.buckets
            handles = [dist.all_reduce(b, op=dist.ReduceOp.SUM, async_op=True) for b in todo]
            if self._early_handle is not None:
                handles.insert(0, self._early_handle)
                self._early_handle = None
            for h in handles:
                h.wait()
            self.flat.mul_(1.0 / self.world_size)
        # the early bucket of THIS backward is accounted for: a later reduce() without a zero_grad() in between (the split-graph
        # replay calls reduce() directly) must send every bucket again
        self._early_sent, self._early_handle, self._arrived = False, None, 0

    def attach(self) -> None:
        """Point ``p.grad`` at the views of the flat buffer (what unscale / clip / the optimizer read)."""
        for p, v in zip(self.params, self._views):
            p.grad = v

    @torch.no_grad()
    def all_reduce(self) -> None:
        if not self.active:
            return
        self.pack()
        self.reduce()
        if self._recording is not None:   # the first step's arrival order is known now: [early | late] layout from the next step on
            grads = [p.grad for p in self.params]
            vals = [None if g is None else v.clone() for g, v in zip(grads, self._views)]
            self._finish_recording()
            for v_new, val in zip(self._views, vals):   # (the reduced values move with their parameters)
                if val is not None:
                    v_new.copy_(val)
                else:
                    v_new.zero_()
        self.attach()


@torch.no_grad()
def broadcast_module_state(module: nn.Module, src: int = 0) -> None:
    """Make every rank start from rank ``src``'s parameters and buffers (what DDP's constructor does)."""
    if not dist.is_initialized() or dist.get_world_size() <= 1:
        return
    tensors = [t for t in list(module.parameters()) + list(module.buffers())]
    real = [t for t in tensors if not t.is_complex()]
    by_dtype: dict[torch.dtype, list[Tensor]] = {}
    for t in real:
        by_dtype.setdefault(t.dtype, []).append(t)
    for dtype, group in by_dtype.items():
        send_dtype = torch.uint8 if dtype == torch.bool else dtype
        flat = torch.cat([t.detach().to(send_dtype).reshape(-1) for t in group])
        dist.broadcast(flat, src=src)
        off = 0
        for t in group:
            t.copy_(flat[off:off + t.numel()].view_as(t).to(dtype))
            off += t.numel()
    for t in tensors:
        if t.is_complex():
            r = torch.view_as_real(t).contiguous()
            dist.broadcast(r, src=src)
            t.copy_(torch.view_as_complex(r))


def all_reduce_mean_(values: Tensor) -> Tensor:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(values, op=dist.ReduceOp.SUM)
        values.div_(dist.get_world_size())
    return values

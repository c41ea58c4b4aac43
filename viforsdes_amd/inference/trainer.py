"""Training loop (reference: inference/trainer.py:49-262).

One optimizer step = theta ~ q(theta) -> encoder -> fused head -> ELBO -> backward ->
[gradient all-reduce] -> unscale -> clip -> AdamW -> EMA.  Differences from the reference that do
not change results: the ELBO scalars stay on the device and are copied to the host every
``update_interval`` steps instead of a ``.item()`` sync per step; under torchrun gradients are
actually averaged across ranks (see data_parallel.py)."""
from __future__ import annotations

import os

from dataclasses import dataclass
from typing import TYPE_CHECKING, Callable, Optional

import torch
from torch import Tensor, nn

from ..accelerate import suppress_torch_compile_output
from ..config import EncoderConfig, HeadConfig, PretrainConfig, TrainingConfig
from ..console import Console
from ..core.euler_maruyama import euler_maruyama
from ..core.observations import ObservationLikelihood, Observations
from ..core.priors import Prior
from ..core.sde import SDE
from .constants import LOSS_EMA_DECAY
from ..primitives import fused
from .data_parallel import all_reduce_mean_
from .diffusion_path_sampler import sample_diffusion_paths
from .evidence_lower_bound import compute_evidence_lower_bound
from .exponential_moving_average import ExponentialMovingAverage
from .fused_optimizer import FusedOptimizerStep
from .state_space import StateSpace
from .training_context import TrainingContext
from .types import EvidenceLowerBoundResult
from ..models.variational_sde_posterior import VariationalSDEPosterior

if TYPE_CHECKING:
    from ..accelerate import Accelerator


@dataclass
class TrainingState:
    step: int
    evidence_lower_bound_history: list[float]
    best_evidence_lower_bound: float
    model: VariationalSDEPosterior
    exponential_moving_average: ExponentialMovingAverage


@dataclass(frozen=True)
class TrainStepResult:
    elbo_result: EvidenceLowerBoundResult
    grad_norm: Tensor  # 0-dim tensor on the training device (call .item() to sync)



def _head_range_exceeded() -> bool:
    """The sticky f16-range flag of the GRU head's MFMA kernels (``_hip.head_mfma_range_exceeded``); False without the GPU library."""
    try:
        from .. import _hip
        return _hip.head_mfma_range_exceeded()
    except Exception:
        return False


class VariationalInferenceTrainer:
    def __init__(self, sde: SDE, observations: Observations, observation_likelihood: ObservationLikelihood,
                 prior: Prior, time_horizon: float, config: TrainingConfig, encoder_config: EncoderConfig,
                 head_config: HeadConfig, state_positive_dims: list[int], sde_param_positive_dims: list[int],
                 device: torch.device | str = "cuda", mixed_precision: bool = True, console: Optional[Console] = None,
                 param_names: Optional[list[str]] = None, accelerator: "Optional[Accelerator]" = None,
                 sde_param_init_mean: Optional[Tensor] = None, seed: Optional[int] = None) -> None:
        self.sde, self.param_names = sde, param_names
        self.observation_likelihood, self.prior = observation_likelihood, prior
        self.time_horizon, self.config = time_horizon, config
        self.state_space = StateSpace(sde.state_dim, state_positive_dims)
        self.sde_param_positive_dims = sde_param_positive_dims
        self.console = console if console is not None else Console()
        self.ctx = TrainingContext.create(
            observations=observations, state_dim=sde.state_dim, sde_param_dim=sde.sde_param_dim, config=config,
            encoder_config=encoder_config, head_config=head_config, sde_param_positive_dims=sde_param_positive_dims,
            device=device, mixed_precision=mixed_precision, accelerator=accelerator,
            sde_param_init_mean=sde_param_init_mean, seed=seed)
        self.step = 0
        self.evidence_lower_bound_history: list[float] = []
        self.best_evidence_lower_bound = float("-inf")

    @property
    def device(self) -> torch.device:
        return self.ctx.device

    # ------------------------------------------------------------------------------ one step
    def _forward_backward(self, model: VariationalSDEPosterior, theta_eps: Optional[Tensor] = None,
                          path_noise: Optional[Tensor] = None) -> EvidenceLowerBoundResult:
        """theta ~ q -> encoder -> head -> ELBO -> backward; leaves the (loss-scaled) gradients in ``p.grad``."""
        ctx, cfg = self.ctx, self.config
        ctx.grad_sync.zero_grad()
        sde_parameters = model.sde_parameter_posterior.rsample(cfg.batch_size, eps=theta_eps)
        with torch.autocast(device_type=ctx.device.type, dtype=cfg.amp_dtype.value, enabled=ctx.scaler.is_enabled()):
            sample = sample_diffusion_paths(model.encoder, model.head, ctx.observations, sde_parameters, ctx.x0_buffer,
                                            self.time_horizon, cfg.time_step, self.state_space, noise=path_noise)
            result = compute_evidence_lower_bound(self.sde, ctx.observations, self.observation_likelihood, self.prior,
                                                  model.sde_parameter_posterior, sde_parameters, sample, cfg.time_step)
        # the encoder's small weight-gradient products are collected over the backward pass and issued together (primitives/fused.py)
        with fused.deferred_weight_grads():
            ctx.scaler.scale(-result.evidence_lower_bound).backward()
        # hand out detached scalars: nothing the caller keeps may hold this step's autograd graph (and its AccumulateGrad
        # nodes) alive across steps or across a HIP-graph capture; the reference returns host floats (trainer.py:199-206)
        c = result.components
        return EvidenceLowerBoundResult(
            evidence_lower_bound=result.evidence_lower_bound.detach(),
            components=type(c)(**{f: getattr(c, f).detach() for f in c.__dataclass_fields__}))

    def _optimizer_step(self) -> Tensor:
        """unscale -> clip (global norm) -> AdamW -> scaler update -> refresh of the cached bf16 GEMM operands."""
        ctx, cfg = self.ctx, self.config
        grad_norm = None
        # a flag left over from an earlier fused step whose caller never ran ema.update() must not swallow the update that
        # follows THIS step (which may take the torch sequence)
        ctx.ema.fused_step_done = False
        if ctx.device.type == "cuda":
            # one pass for the global norm / non-finite check, one for unscale x clip + AdamW + EMA (inference/fused_optimizer.py)
            fs = getattr(self, "_fused_opt", None)
            if fs is None and FusedOptimizerStep.usable(ctx.optimizer, ctx.scaler):
                fs = self._fused_opt = FusedOptimizerStep(ctx.optimizer, ctx.ema if getattr(self, "fuse_ema", True) else None,
                                                          ctx.scaler, cfg.grad_clip_norm)
            if fs is not None:
                grad_norm = fs.step()
        if grad_norm is None:
            ctx.scaler.unscale_(ctx.optimizer)
            grad_norm = nn.utils.clip_grad_norm_(ctx.model.parameters(), cfg.grad_clip_norm)
            ctx.scaler.step(ctx.optimizer)
        ctx.scaler.update()
        if ctx.device.type == "cuda":
            # bf16 GEMM operands of the encoder follow the updated parameters: ONE kernel over all packs now, instead of a
            # staleness check + copies per pack inside the next forward.  (An eager forward would also notice by itself: the optimizer
            # step advances fused._param_epoch and the parameters' version counters.  A CAPTURED forward does not check anything --
            # capture_step_graph's replay() looks for stale packs before every replay.)
            ids = getattr(self, "_param_ids", None)
            if ids is None:
                ids = self._param_ids = {id(q) for q in ctx.model.parameters()}
            fused.PackedWeight.refresh_all(force=True, params=ids)
        return grad_norm.detach()

    def _train_step(self, model: VariationalSDEPosterior, theta_eps: Optional[Tensor] = None,
                    path_noise: Optional[Tensor] = None) -> TrainStepResult:
        result = self._forward_backward(model, theta_eps, path_noise)
        self.ctx.grad_sync.all_reduce()     # mean over ranks of the (still loss-scaled) gradients
        return TrainStepResult(elbo_result=result, grad_norm=self._optimizer_step())

    # ------------------------------------------------------------------------------ HIP graph
    def capture_step_graph(self, warmup: int = 3, warm_results: Optional[list] = None
                           ) -> Optional[Callable[[], TrainStepResult]]:
        """Capture one full optimizer step (+ EMA update) into HIP graph(s) and return a ``replay()`` callable.

        The OU-size step is launch-bound (~900 small kernels): replaying a graph removes the per-kernel host
        cost (28.7 -> 9.6 ms/step measured); at LV size the GPU is busy either way and the eager queue is ~3 %
        faster than graph replay (44.8 vs 46.1 ms), so callers should prefer eager stepping there.
        Single process: ONE graph holds the whole step.  Data parallel (or ``grad_sync.active``): TWO graphs -- [theta draw
        ... backward, gradients packed into the flat buffer] and [unscale, clip, AdamW, EMA] -- with the RCCL all-reduce
        of the flat buffer issued eagerly between them, so the collective never sits inside a captured region and a
        small-batch replica is not launch-bound either.
        ``warmup`` eager steps run first (optimizer state, allocator pools, lazily built caches); they are real
        training steps (their results are appended to ``warm_results`` when given).  Returns ``None`` when
        capture is not applicable (CPU) or fails, in which case the caller keeps stepping eagerly.  Each replay draws
        fresh noise (graph-safe Philox offsets)."""
        ctx = self.ctx
        if ctx.device.type != "cuda":
            return None
        model = ctx.model
        split = ctx.grad_sync.active
        try:
            side = torch.cuda.Stream(device=ctx.device)
            side.wait_stream(torch.cuda.current_stream(ctx.device))
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    r = self._train_step(model)
                    ctx.ema.update()
                    if warm_results is not None:
                        warm_results.append(r)
            torch.cuda.current_stream(ctx.device).wait_stream(side)
            if not split:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    static = self._train_step(model)
                    ctx.ema.update()
                graphs = (graph,)
            else:
                g_fb, g_opt = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_fb):
                    elbo = self._forward_backward(model)
                    ctx.grad_sync.pack()                 # gradients -> flat buffer (static addresses)
                ctx.grad_sync.reduce()                   # eager RCCL all-reduce of the flat buffer (a real step)
                ctx.grad_sync.attach()                   # p.grad = views of the flat buffer: what graph 2 reads
                with torch.cuda.graph(g_opt, pool=g_fb.pool()):
                    gnorm = self._optimizer_step()
                    ctx.ema.update()
                static = TrainStepResult(elbo_result=elbo, grad_norm=gnorm)
                graphs = (g_fb, g_opt)
        except Exception as err:  # capture is an optimisation, never a requirement
            if os.environ.get("VSDE_DEBUG_CAPTURE"):   # where it failed (the console only shows the message)
                import traceback
                traceback.print_exc()
            torch.cuda.synchronize(ctx.device)
            self.console.config_panel(f"HIP graph capture unavailable ({type(err).__name__}: {err}); running eagerly")
            return None
        self._graph = graphs  # keep alive

        # The captured forward holds no operand refresh (the packs were fresh at capture, ``PackedWeight.operands()`` was a no-op);
        # only the tail of the captured optimizer step re-fills them.  Anything that rewrites the packs BETWEEN replays -- an EMA
        # swap around ``VariationalPosterior.sample()`` / ``summary()`` from a callback, another sampler's forced refresh -- would
        # otherwise make the next replay run its forward on those weights.  Same check as ``CapturedPathSampler.__call__``: when a
        # pack of this model is stale (versions / parameter epoch moved since it was filled), re-fill all of them from the live
        # parameters with one kernel before replaying.
        from ..primitives.fused import PackedWeight
        ids = {id(q) for q in model.parameters()}

        def refresh_if_stale() -> None:
            if any(pk.stale() for pk in PackedWeight._live if any(id(q) in ids for q in pk.params)):
                PackedWeight.refresh_all(force=True, params=ids)

        # The replayed optimizer step rewrites parameters and packs on the device; no host counter moves.  Operands DERIVED from the
        # packs (the tile images of the no-grad block kernels) that were built after the capture are not in the graph: they are
        # marked dirty after every replay, so the next no-grad forward (or captured sampler replay) rebuilds them from the packs.
        def after_replay() -> None:
            PackedWeight.invalidate_derived(ids)

        if not split:
            def replay() -> TrainStepResult:
                refresh_if_stale()
                graphs[0].replay()
                after_replay()
                return static
        else:
            def replay() -> TrainStepResult:
                refresh_if_stale()
                graphs[0].replay()
                ctx.grad_sync.reduce()   # sends every bucket: reduce() leaves no early-bucket state behind (an eager step in between)
                graphs[1].replay()
                after_replay()
                return static
        return replay

    # ----------------------------------------------------------------------------- main loop
    def train(self, callback: Optional[Callable[[int, float], None]] = None, update_interval: int = 10,
              hip_graph: bool = True, graph_warmup: int = 3) -> TrainingState:
        ctx, n_iter = self.ctx, self.config.n_iterations
        model = ctx.model
        model.train()
        if ctx.is_main:
            self.console.config_panel(self.config)
        if callback is not None:
            update_interval = 1
        pending: list[Tensor] = []
        pending_first = 0
        last: Optional[TrainStepResult] = None
        loss_ema = 0.0

        def flush(progress) -> None:
            nonlocal pending, pending_first, loss_ema
            if not pending:
                return
            vals = all_reduce_mean_(torch.stack(pending).float()) if ctx.is_distributed else torch.stack(pending)
            for offset, elbo in enumerate(vals.tolist()):
                step = pending_first + offset
                loss_ema = -elbo if step == 0 else LOSS_EMA_DECAY * loss_ema + (1 - LOSS_EMA_DECAY) * (-elbo)
                smoothed = loss_ema / (1 - LOSS_EMA_DECAY ** (step + 1))
                self.evidence_lower_bound_history.append(elbo)
                self.best_evidence_lower_bound = max(self.best_evidence_lower_bound, elbo)
                if callback is not None and ctx.is_main:
                    callback(step, elbo)
            if ctx.is_main and last is not None:
                progress.update(step=pending_first + len(pending) - 1, loss=smoothed, elbo=elbo,
                                best_elbo=self.best_evidence_lower_bound, components=last.elbo_result.components,
                                grad_norm=float(last.grad_norm), param_means=model.sde_parameter_posterior.expected_value)
            pending_first += len(pending)
            pending = []

        with suppress_torch_compile_output(), self.console.training_progress(
                n_iter, update_interval=update_interval, param_names=self.param_names) as progress:
            replay = None
            step = 0
            if hip_graph and n_iter > graph_warmup + 1:
                warm: list[TrainStepResult] = []
                replay = self.capture_step_graph(graph_warmup, warm)   # the warm-up steps are iterations 0..k-1
                for r in warm:
                    last = r
                    pending.append(r.elbo_result.evidence_lower_bound.detach().clone())
                    step += 1
                    if len(pending) >= update_interval:
                        flush(progress)
            while step < n_iter:
                self.step = step
                if replay is not None:
                    last = replay()
                    pending.append(last.elbo_result.evidence_lower_bound.detach().clone())  # static output tensor
                    if _head_range_exceeded():
                        # the captured step keeps the multi-path MFMA kernels it was captured with; a GRU weight has left their f16
                        # range (that step's ELBO is non-finite): eager steps from here on take the fp32 kernels
                        replay = None
                        self.console.config_panel("a GRU head weight left the f16 range of the MFMA kernels: HIP graph dropped, "
                                                  "stepping eagerly on the fp32 kernels")
                else:
                    last = self._train_step(model)
                    ctx.ema.update()
                    pending.append(last.elbo_result.evidence_lower_bound.detach())
                step += 1
                if len(pending) >= update_interval or step == n_iter:
                    flush(progress)
        return TrainingState(step=self.step, evidence_lower_bound_history=self.evidence_lower_bound_history,
                             best_evidence_lower_bound=self.best_evidence_lower_bound, model=ctx.model,
                             exponential_moving_average=ctx.ema)

    # --------------------------------------------------------------------------- pre-training
    def pretrain_sde_parameters(self, config: Optional[PretrainConfig] = None) -> Tensor:
        """Fit a Gaussian over (log-)theta by matching simulated paths of the MODEL SDE to the
        observations (reference: trainer.py:208-259). Returns the best mean found."""
        cfg = config or PretrainConfig()
        d, dev = self.sde.sde_param_dim, self.device
        pos = list(self.sde_param_positive_dims)
        free = [i for i in range(d) if i not in pos]
        mu = nn.Parameter(torch.zeros(d, device=dev))
        if free:
            mu.data[free] = cfg.init_scale * torch.randn(len(free), device=dev)
        log_sigma = nn.Parameter(torch.zeros(d, device=dev))
        self._pretrain_opt = opt = torch.optim.Adam([mu, log_sigma], lr=cfg.learning_rate, capturable=dev.type == "cuda")
        best_mu, best_mse = mu.detach().clone(), float("inf")
        obs = self.ctx.observations
        obs_idx = (obs.times / self.config.time_step).round().long()
        x0 = obs.values[0].unsqueeze(0).expand(cfg.batch_size, -1)
        pos_host = torch.zeros(d, dtype=torch.bool)
        pos_host[pos] = True
        pos_mask = pos_host.to(dev)

        def simulate():
            """One stochastic evaluation of the matching loss for the current (mu, log_sigma)."""
            sigma = log_sigma.exp()
            log_theta = mu + sigma * torch.randn(cfg.batch_size, d, device=dev)
            theta = torch.where(pos_mask, log_theta.exp(), log_theta) if pos else log_theta
            paths = euler_maruyama(self.sde, x0, theta, self.time_horizon, self.config.time_step,
                                   self.state_space.positive_dims)
            return ((paths[:, obs_idx] - obs.values) ** 2).mean(), sigma.median()

        replay = self._capture_pretrain_step(simulate, mu, log_sigma, cfg) if dev.type == "cuda" else None
        with self.console.pretrain_progress(cfg.n_iterations) as progress:
            for step in range(cfg.n_iterations):
                if replay is not None:
                    # the whole iteration (simulation, backward, clip, Adam) replays from a HIP graph: the T-step Python loop
                    # is ~3 kernels per Euler step and entirely launch-bound.  The graph always applies the update; a
                    # non-finite loss is undone from the snapshot it took first (same semantics as the eager branch below).
                    value, finite, med, mu_before = replay()
                    if finite and value < best_mse:
                        best_mu, best_mse = mu_before, value
                    progress.update(step, value, best_mse, med)
                    continue
                opt.zero_grad()
                mse, med_t = simulate()
                finite = bool(torch.isfinite(mse))
                value = mse.item()
                if finite and value < best_mse:
                    best_mu, best_mse = mu.detach().clone(), value
                if finite:
                    mse.backward()
                    nn.utils.clip_grad_norm_([mu, log_sigma], 1.0)
                    opt.step()
                progress.update(step, value, best_mse, med_t.item())
        return best_mu

    def _capture_pretrain_step(self, simulate, mu: nn.Parameter, log_sigma: nn.Parameter, cfg: PretrainConfig):
        """HIP graph of one pre-training iteration; returns ``replay() -> (loss, finite, median sigma, mu before the update)``
        or ``None`` when capture is not possible (the caller then runs the eager loop)."""
        opt = self._pretrain_opt
        dev = mu.device
        params = [mu, log_sigma]

        def iteration():
            opt.zero_grad(set_to_none=True)
            mse, med = simulate()
            mse.backward()
            nn.utils.clip_grad_norm_(params, 1.0)
            opt.step()
            return mse.detach(), med.detach()

        try:
            state0 = [p.detach().clone() for p in params]
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                rng = torch.cuda.get_rng_state(dev)
                iteration()  # warm-up: creates the optimizer state, allocator pools ...
                torch.cuda.set_rng_state(rng, dev)
            torch.cuda.current_stream(dev).wait_stream(side)
            with torch.no_grad():  # ... and is undone: the captured iterations start from the initial state
                for p, s0 in zip(params, state0):
                    p.copy_(s0)
                for st in opt.state.values():
                    for v in st.values():
                        if torch.is_tensor(v):
                            v.zero_()
            opt_tensors = [v for st in opt.state.values() for v in st.values() if torch.is_tensor(v)]
            snap_p = [torch.empty_like(p) for p in params]
            snap_o = [torch.empty_like(v) for v in opt_tensors]
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                with torch.no_grad():
                    torch._foreach_copy_(snap_p, [p.detach() for p in params])
                    torch._foreach_copy_(snap_o, opt_tensors)
                s_mse, s_med = iteration()
                s_out = torch.stack([s_mse.float(), s_med.float()])
        except Exception as err:  # capture is an optimisation, never a requirement
            torch.cuda.synchronize(dev)
            self.console.config_panel(f"pretrain HIP graph unavailable ({type(err).__name__}: {err}); running eagerly")
            return None
        self._pretrain_graph = graph

        def replay():
            graph.replay()
            value, med = s_out.tolist()
            finite = value == value and abs(value) != float("inf")
            if not finite:  # the reference skips the update: restore what the graph changed
                with torch.no_grad():
                    torch._foreach_copy_([p.detach() for p in params], snap_p)
                    torch._foreach_copy_(opt_tensors, snap_o)
            return value, finite, med, snap_p[0].clone()
        return replay

    # ------------------------------------------------------------------------- resumable state
    def training_state_dict(self) -> dict:
        """Everything needed to continue an interrupted run bit-for-bit on the same hardware: parameters, EMA shadow, AdamW
        moments, loss-scaler state and the RNG streams (additive: the reference checkpoints parameters and EMA only,
        ``posterior/variational_posterior.py:150-192``)."""
        ctx = self.ctx
        state = {"model": ctx.model.state_dict(), "ema": ctx.ema.state_dict(), "optimizer": ctx.optimizer.state_dict(),
                 "scaler": ctx.scaler.state_dict(), "rng_cpu": torch.get_rng_state()}
        if ctx.device.type == "cuda":
            state["rng_cuda"] = torch.cuda.get_rng_state(ctx.device)
        return state

    def load_training_state_dict(self, state: dict) -> None:
        ctx = self.ctx
        ctx.model.load_state_dict(state["model"])
        for name, t in state["ema"].items():
            ctx.ema.shadow[name].copy_(t)
        ctx.optimizer.load_state_dict(state["optimizer"])
        ctx.scaler.load_state_dict(state["scaler"])
        torch.set_rng_state(state["rng_cpu"])
        if ctx.device.type == "cuda" and "rng_cuda" in state:
            torch.cuda.set_rng_state(state["rng_cuda"], ctx.device)

    def cleanup(self) -> None:
        self.ctx.cleanup()

"""Host-side guards around the HIP fast paths (CPU tests).

* ``builtin_sde_kind``: only the exact closed forms the library implements are routed to the built-in kernels; a user
  subclass that overrides ``drift`` / ``diffusion`` of an example SDE (reference core/sde.py:8-14 is a Protocol, so
  subclassing an example is the natural way to tweak it) keeps its Python callables.
* ``SDEParameterPosterior._positive_dims`` follows the ``positive_mask`` buffer (reference
  models/sde_parameter_posterior.py:28-33) through ``load_state_dict`` / in-place edits.
* ``TrainingProgress.update`` + ``Console(metrics_path=...)``: the JSONL record carries the quantities of the reference's
  live panel (reference console.py:106-228) and device scalars are only synchronised when a line is emitted."""
import io
import json

import torch

from viforsdes_amd.console import COMPONENT_FIELDS, Console
from viforsdes_amd.core.sde import builtin_sde_kind, make_sde
from viforsdes_amd.examples.sdes import LinearDiagonalSDE, LotkaVolterra, OrnsteinUhlenbeck
from viforsdes_amd.inference.types import EvidenceLowerBoundComponents
from viforsdes_amd.models.sde_parameter_posterior import SDEParameterPosterior


def test_exact_example_classes_are_builtin():
    assert builtin_sde_kind(OrnsteinUhlenbeck()) == "ornstein_uhlenbeck"
    assert builtin_sde_kind(LotkaVolterra()) == "lotka_volterra"
    assert builtin_sde_kind(LinearDiagonalSDE(8)) == "linear_diagonal"
    assert builtin_sde_kind(make_sde(lambda x, t: x, lambda x, t: x, 1, 1)) is None


def test_subclass_overriding_a_coefficient_is_not_builtin():
    class DampedLV(LotkaVolterra):
        def drift(self, x, sde_parameters):
            return 0.5 * super().drift(x, sde_parameters)

    class NoisyOU(OrnsteinUhlenbeck):
        def diffusion(self, x, sde_parameters):
            return 2.0 * super().diffusion(x, sde_parameters)

    class RenamedLV(LotkaVolterra):   # nothing overridden: still the closed form
        pass

    assert builtin_sde_kind(DampedLV()) is None
    assert builtin_sde_kind(NoisyOU()) is None
    assert builtin_sde_kind(RenamedLV()) == "lotka_volterra"
    shadowed = LotkaVolterra()
    shadowed.drift = lambda x, t: x            # instance attribute shadows the method
    assert builtin_sde_kind(shadowed) is None


def test_linear_diagonal_beyond_kernel_limit_falls_back():
    assert builtin_sde_kind(LinearDiagonalSDE(32)) == "linear_diagonal"
    assert builtin_sde_kind(LinearDiagonalSDE(33)) is None


def test_overridden_drift_is_used_by_the_cpu_simulator():
    from viforsdes_amd.core.euler_maruyama import euler_maruyama

    class FrozenOU(OrnsteinUhlenbeck):
        def drift(self, x, sde_parameters):
            return torch.zeros_like(x)

    x0, theta = torch.ones(3, 1), torch.tensor([[1.0, 0.0, 0.0]]).repeat(3, 1)   # sigma = 0: the path never moves
    traj = euler_maruyama(FrozenOU(), x0, theta, 1.0, 0.1, noise=torch.randn(3, 10, 1))
    assert torch.equal(traj, torch.ones(3, 11, 1))


def test_positive_dims_follow_the_mask_buffer():
    a = SDEParameterPosterior(4, [0, 2])
    assert a._positive_dims == (0, 2)
    b = SDEParameterPosterior(4, [1])
    a.load_state_dict(b.state_dict())
    assert a._positive_dims == (1,)
    a.positive_mask[3] = True
    assert a._positive_dims == (1, 3)
    theta = a.rsample(5)
    assert bool((theta[:, [1, 3]] > 0).all())


class _CountingScalar:
    """Stands in for a device scalar: counts host synchronisations (``item`` calls)."""

    def __init__(self, v):
        self.v, self.items = v, 0

    def item(self):
        self.items += 1
        return self.v


def test_metrics_sink_writes_jsonl_records_and_syncs_only_on_emit(tmp_path):
    path = tmp_path / "metrics.jsonl"
    stream = io.StringIO()
    console = Console(enabled=True, stream=stream, metrics_path=str(path))
    comps = EvidenceLowerBoundComponents(*[torch.tensor(float(i)) for i in range(5)])
    gnorm = _CountingScalar(3.5)
    with console.training_progress(total=10, update_interval=5, param_names=["kappa", "mu"]) as prog:
        for step in range(10):
            prog.update(step, loss=1.0 + step, elbo=-1.0 - step, best_elbo=-1.0, components=comps, grad_norm=gnorm,
                        param_means=torch.tensor([0.5, 2.0]))
    assert gnorm.items == 2                     # steps 5 and 10 emit; the other eight updates never touch the device scalar
    records = [json.loads(line) for line in path.read_text().splitlines()]
    assert [r["step"] for r in records] == [4, 9]
    last = records[-1]
    for key in ("step", "loss", "elbo", "best_elbo", "grad_norm", "iter_per_sec", "elapsed_s", "eta_s", "components",
                "param_means", "memory_allocated_gb"):
        assert key in last, key
    assert last["grad_norm"] == 3.5 and last["loss"] == 10.0 and last["elbo"] == -10.0
    assert list(last["components"]) == [label for label, _ in COMPONENT_FIELDS]
    assert last["components"]["sde"] == 1.0 and last["param_means"] == {"kappa": 0.5, "mu": 2.0}
    assert last["iter_per_sec"] > 0 and last["eta_s"] == 0.0
    assert stream.getvalue().count("\n") == 2 and "|g| 3.5" in stream.getvalue()


def test_disabled_console_emits_nothing(tmp_path):
    path = tmp_path / "metrics.jsonl"
    console = Console(enabled=False, metrics_path=str(path))
    with console.training_progress(total=3, update_interval=1) as prog:
        for step in range(3):
            prog.update(step, 1.0, -1.0, -1.0, grad_norm=_CountingScalar(1.0))
    assert not path.exists()


def test_packed_operands_follow_a_fused_optimizer_step():
    """The cached bf16 GEMM operands must follow the parameters through ANY optimizer step.  torch's fused AdamW kernel (and the
    trainer's own optimizer kernel) update the parameters in place WITHOUT bumping ``Tensor._version``, so a version check
    alone kept serving the first step's weights (the round-3 defect).  Since round 4 every ``Optimizer.step`` also advances
    ``fused._param_epoch`` (global step post-hook; the own kernel bumps versions + epoch itself): the pack notices by itself, no
    ``refresh_all(force=True)`` needed (reference semantics: ``nn.Linear`` under autocast re-casts its weight every forward,
    primitives/mlp.py:50-54).  The forced refresh the trainer still issues is checked below too."""
    from viforsdes_amd.primitives.fused import PackedWeight, plain_pack
    torch.manual_seed(0)
    w = torch.nn.Parameter(torch.randn(16, 8))
    b = torch.nn.Parameter(torch.randn(16))
    pack = plain_pack(w, b)
    wb, bb = pack.operands()
    assert torch.equal(wb, w.detach().to(torch.bfloat16))
    opt = torch.optim.AdamW([w, b], lr=0.1, fused=True)
    w.grad, b.grad = torch.randn_like(w), torch.randn_like(b)
    v0 = w._version
    opt.step()
    assert w._version == v0, "torch's fused AdamW started bumping Tensor._version: the epoch hook is no longer the only guard"
    moved = not torch.equal(wb, w.detach().to(torch.bfloat16))
    assert moved, "the step must have changed the bf16 image of the weight for this test to mean anything"
    assert pack.stale()
    wb2, bb2 = pack.operands()                 # no forced refresh: the pack saw the optimizer step
    assert torch.equal(wb2, w.detach().to(torch.bfloat16)) and torch.equal(bb2, b.detach().to(torch.bfloat16))
    tr = pack.transposed()
    w.grad = torch.randn_like(w)
    opt.step()
    assert torch.equal(pack.transposed(), w.detach().to(torch.bfloat16).t()) and tr is pack.transposed()
    w.grad = torch.randn_like(w)
    opt.step()
    PackedWeight.refresh_all(force=True)       # what the trainer does: one pass over every live pack
    assert not pack.stale()
    assert torch.equal(pack.transposed(), w.detach().to(torch.bfloat16).t()) and tr is pack.transposed()


def test_forced_refresh_is_scoped_to_the_callers_parameters():
    """A trainer's forced refresh (and a HIP graph captured from its step, which replays with raw addresses) must only touch
    the packs built from ITS parameters: the registry of live packs is process-wide, and a pack of another model that happens
    to be alive may be freed later.  Packs outside the scope stay covered by their own staleness check."""
    from viforsdes_amd.primitives.fused import PackedWeight, plain_pack
    torch.manual_seed(1)
    wa, wb_ = torch.nn.Parameter(torch.randn(16, 8)), torch.nn.Parameter(torch.randn(16, 8))
    pa, pb = plain_pack(wa, None), plain_pack(wb_, None)
    a0, b0 = pa.operands()[0].clone(), pb.operands()[0].clone()
    wa.data.add_(1.0); wb_.data.add_(1.0)      # in place through .data: no version bump, no optimizer hook -- only `force` sees it
    assert not pa.stale() and not pb.stale()
    PackedWeight.refresh_all(force=True, params={id(wa)})
    assert torch.equal(pa.weight, wa.detach().to(torch.bfloat16)) and not torch.equal(pa.weight, a0)
    assert torch.equal(pb.weight, b0), "a pack outside the caller's parameters was rewritten"
    PackedWeight.refresh_all(force=True)       # unscoped: every live pack
    assert torch.equal(pb.weight, wb_.detach().to(torch.bfloat16))

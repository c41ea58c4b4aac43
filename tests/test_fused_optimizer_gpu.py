"""GPU: the two-launch optimizer step (csrc/vsde_optim.hip, inference/fused_optimizer.py) against the torch sequence it replaces
(reference inference/trainer.py:197-204 + exponential_moving_average.py:27-32):

    scaler.unscale_ -> clip_grad_norm_ -> scaler.step(AdamW fused, capturable) -> scaler.update -> EMA lerp

on identical parameters / gradients over several steps, with and without a loss scale, with a step whose gradient holds an inf
(both must skip it and halve the scale) and with a norm below and above the clipping threshold.  Tolerance: 2e-6 relative to the
tensor's max (the arithmetic is the same operation sequence; the global norm is summed in a different order)."""
import copy

import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _Toy(nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        shapes = {"a": (5,), "b": (33, 7), "c": (4097,), "d": (3, 4101), "e": (256, 256), "theta": (3,), "frozen": (4,)}
        for k, s in shapes.items():
            self.register_parameter(k, nn.Parameter(torch.randn(*s, generator=g)))
        self.frozen.requires_grad_(False)


def _setup(scaled):
    from torch.amp import GradScaler
    from viforsdes_amd.inference.exponential_moving_average import ExponentialMovingAverage
    model = _Toy().to(DEV)
    groups = lambda m: [{"params": [p for n, p in m.named_parameters() if n != "theta"], "lr": 1e-3},
                        {"params": [m.theta], "lr": 1e-2}]
    opt = torch.optim.AdamW(groups(model), fused=True, capturable=True)
    scaler = GradScaler("cuda", enabled=scaled, init_scale=1024.0, growth_interval=3)
    return model, opt, scaler, ExponentialMovingAverage(model, decay=0.99)


def _grads(model, step, scale, poison):
    g = torch.Generator().manual_seed(100 + step)
    mag = 10.0 if step % 2 else 0.01   # global norm above / below the threshold
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        p.grad = (torch.randn(p.shape, generator=g) * mag).to(DEV) * scale
        if poison and n == "d":
            p.grad[1, 17] = float("inf")


@pytest.mark.parametrize("scaled", [True, False])
def test_fused_step_matches_the_torch_sequence(scaled):
    from viforsdes_amd.inference.fused_optimizer import FusedOptimizerStep
    m_ref, o_ref, s_ref, e_ref = _setup(scaled)
    m_own, o_own, s_own, e_own = _setup(scaled)
    assert FusedOptimizerStep.usable(o_own)
    fs = FusedOptimizerStep(o_own, e_own, s_own, max_norm=1.0)
    for step in range(7):
        poison = scaled and step == 3
        for model, scaler in ((m_ref, s_ref), (m_own, s_own)):
            if scaled:
                scaler.scale(torch.zeros((), device=DEV))   # creates / keeps the scale tensor the way the trainer's backward does
            _grads(model, step, float(scaler.get_scale()) if scaled else 1.0, poison)
        # reference sequence
        s_ref.unscale_(o_ref)
        n_ref = nn.utils.clip_grad_norm_(m_ref.parameters(), 1.0)
        s_ref.step(o_ref); s_ref.update(); e_ref.update()
        # fused
        n_own = fs.step()
        assert n_own is not None
        s_own.update(); e_own.update()
        assert not e_own.fused_step_done
        if not poison:
            assert abs(float(n_own) - float(n_ref)) <= 2e-6 * float(n_ref), (step, float(n_own), float(n_ref))
        else:
            assert not torch.isfinite(n_own)
        assert s_own.get_scale() == s_ref.get_scale(), step
        for (name, p), q in zip(m_ref.named_parameters(), m_own.parameters()):
            tol = 2e-6 * float(p.detach().abs().max())
            assert float((p - q).abs().max()) <= tol, (step, name)
            if p.requires_grad:
                for k in ("exp_avg", "exp_avg_sq"):
                    a, b = o_ref.state[p][k], o_own.state[q][k]
                    assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max()) + 1e-30, (step, name, k)
            assert float((e_ref.shadow[name] - e_own.shadow[name]).abs().max()) <= tol, (step, name)
    # the step count is published to the per-parameter tensors torch keeps
    sd_ref, sd_own = o_ref.state_dict(), o_own.state_dict()
    for k in sd_ref["state"]:
        assert float(sd_ref["state"][k]["step"]) == float(sd_own["state"][k]["step"]) == (6.0 if scaled else 7.0)


def test_state_dict_round_trip_and_new_shadows():
    """load_state_dict (new moment tensors) and a rebuilt EMA shadow set must be picked up; the step count survives both."""
    from viforsdes_amd.inference.fused_optimizer import FusedOptimizerStep
    model, opt, scaler, ema = _setup(False)
    fs = FusedOptimizerStep(opt, ema, scaler, max_norm=1.0)
    for step in range(3):
        _grads(model, step, 1.0, False)
        assert fs.step() is not None
        ema.update()
    snap = copy.deepcopy(opt.state_dict())
    p_before = [p.detach().clone() for p in model.parameters()]
    _grads(model, 3, 1.0, False)
    fs.step(); ema.update()
    p_after = [p.detach().clone() for p in model.parameters()]
    with torch.no_grad():
        for p, q in zip(model.parameters(), p_before):
            p.copy_(q)
    opt.load_state_dict(snap)          # new exp_avg / exp_avg_sq / step tensors
    ema._init_shadow()                 # new shadow tensors
    _grads(model, 3, 1.0, False)
    fs.step(); ema.update()
    for p, q in zip(model.parameters(), p_after):
        assert torch.equal(p, q)       # the replayed step is bit-identical
    assert float(opt.state_dict()["state"][0]["step"]) == 4.0
    for (n, p), q0, q1 in zip(model.named_parameters(), p_before, p_after):
        if p.requires_grad:   # fresh shadows = the parameters before the step, then one lerp with weight 1 - decay = 0.01
            assert torch.allclose(ema.shadow[n], q0 + 0.01 * (q1 - q0), rtol=1e-5, atol=1e-7), n


def test_a_step_that_falls_back_to_torch_keeps_the_step_counts_in_sync():
    """A parameter without gradient sends one step to the torch sequence: the shared device step count must reach torch's
    per-parameter counters before it and be re-read from them afterwards (bias corrections depend on it)."""
    from viforsdes_amd.inference.fused_optimizer import FusedOptimizerStep
    m_ref, o_ref, s_ref, e_ref = _setup(False)
    m_own, o_own, s_own, e_own = _setup(False)
    fs = FusedOptimizerStep(o_own, e_own, s_own, max_norm=1.0)
    for step in range(5):
        for model in (m_ref, m_own):
            _grads(model, step, 1.0, False)
            if step == 2:
                model.b.grad = None
        nn.utils.clip_grad_norm_(m_ref.parameters(), 1.0); o_ref.step(); e_ref.update()
        if fs.step() is None:
            assert step >= 2   # from the fallback step on torch's per-parameter counts differ: the torch sequence keeps the optimizer
            nn.utils.clip_grad_norm_(m_own.parameters(), 1.0); o_own.step()
        e_own.update()
        for (name, p), q in zip(m_ref.named_parameters(), m_own.parameters()):
            assert float((p - q).abs().max()) <= 3e-6 * float(p.detach().abs().max()), (step, name)
    steps = {n: float(o_own.state_dict()["state"][i]["step"]) for i, n in enumerate(["a", "b", "c", "d", "e"])}
    assert steps["a"] == 5.0 and steps["b"] == 4.0, steps


def test_unscaled_nan_gradient_poisons_like_the_torch_sequence():
    """No loss scaling and a NaN in one gradient: ``clip_grad_norm_`` makes the clip coefficient NaN (``torch.clamp`` propagates
    it) and every gradient, hence every trainable parameter, becomes NaN -- a loud divergence.  The fused step must not apply a
    silent unclipped update instead."""
    from viforsdes_amd.inference.fused_optimizer import FusedOptimizerStep
    m_ref, o_ref, s_ref, e_ref = _setup(False)
    m_own, o_own, s_own, e_own = _setup(False)
    fs = FusedOptimizerStep(o_own, e_own, s_own, max_norm=1.0)
    for model in (m_ref, m_own):
        _grads(model, 1, 1.0, False)
        model.d.grad[1, 17] = float("nan")
    nn.utils.clip_grad_norm_(m_ref.parameters(), 1.0); o_ref.step()
    n_own = fs.step()
    assert n_own is not None and torch.isnan(n_own)
    for (name, p), q in zip(m_ref.named_parameters(), m_own.parameters()):
        if p.requires_grad:
            assert torch.isnan(p).all() and torch.isnan(q).all(), name
        else:
            assert torch.equal(p, q)


def test_grad_norms_of_successive_eager_steps_do_not_alias():
    from viforsdes_amd.inference.fused_optimizer import FusedOptimizerStep
    model, opt, scaler, ema = _setup(False)
    fs = FusedOptimizerStep(opt, ema, scaler, max_norm=1.0)
    norms = []
    for step in range(3):
        _grads(model, step, 1.0, False)
        norms.append(fs.step())
        ema.update()
    vals = [float(n) for n in norms]   # read AFTER all three steps: each must still hold its own step's norm
    assert vals[1] > 100 * vals[0] and vals[1] > 100 * vals[2] and len({n.data_ptr() for n in norms}) == 3, vals


def test_moved_parameter_storage_rebuilds_the_tables():
    """``p.data = ...`` keeps the Parameter object but moves its storage: the chunk table must follow."""
    from viforsdes_amd.inference.fused_optimizer import FusedOptimizerStep
    m_ref, o_ref, s_ref, e_ref = _setup(False)
    m_own, o_own, s_own, e_own = _setup(False)
    fs = FusedOptimizerStep(o_own, e_own, s_own, max_norm=1.0)
    for step in range(4):
        for model in (m_ref, m_own):
            _grads(model, step, 1.0, False)
        if step == 2:
            m_own.e.data = m_own.e.data.clone()      # new storage, same values
        nn.utils.clip_grad_norm_(m_ref.parameters(), 1.0); o_ref.step(); e_ref.update()
        assert fs.step() is not None
        e_own.update()
        for (name, p), q in zip(m_ref.named_parameters(), m_own.parameters()):
            assert float((p - q).abs().max()) <= 2e-6 * float(p.detach().abs().max()), (step, name)


def test_captured_step_survives_interleaved_eager_steps():
    """capture -> replay -> 5 eager steps -> replay: the captured H2D copy of the gradient-address table
    re-reads its pinned host buffer at every replay, so that buffer must never be a slot of the eager ring (5 eager steps
    overwrite all 4 slots with the addresses of freed eager gradients).  Compared against the torch sequence fed the same
    gradients: the captured step always consumes the graph's own static gradient tensors."""
    from viforsdes_amd.inference.fused_optimizer import FusedOptimizerStep
    m_ref, o_ref, s_ref, e_ref = _setup(False)
    m_own, o_own, s_own, e_own = _setup(False)
    fs = FusedOptimizerStep(o_own, e_own, s_own, max_norm=1.0)

    def ref_step(gr):
        for p, g in zip([p for p in m_ref.parameters() if p.requires_grad], gr):
            p.grad = g.clone()
        nn.utils.clip_grad_norm_(m_ref.parameters(), 1.0); o_ref.step(); e_ref.update()

    def rand_grads(seed):
        g = torch.Generator().manual_seed(seed)
        return [(torch.randn(p.shape, generator=g) * 0.3).to(DEV) for p in m_own.parameters() if p.requires_grad]

    train = [p for p in m_own.parameters() if p.requires_grad]
    static_src = [torch.zeros_like(p) for p in train]       # what the captured "backward" copies into its gradients
    # warm-up (creates optimizer state) on a side stream, as trainer.capture_step_graph does
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        gr = rand_grads(0)
        for p, g in zip(train, gr):
            p.grad = g.clone()
        fs.step(); e_own.update()
    torch.cuda.current_stream().wait_stream(side)
    ref_step(gr)
    graph = torch.cuda.CUDAGraph()
    for p in train:
        p.grad = None
    gr = rand_grads(1)
    for s_, g in zip(static_src, gr):
        s_.copy_(g)
    with torch.cuda.graph(graph):
        for p, s_ in zip(train, static_src):
            p.grad = s_ * 1.0                                # gradient tensors allocated in the graph's pool
        fs.step(); e_own.update()
    graph.replay()                                           # capture does not execute: this is the step for seed 1
    ref_step(gr)
    captured = [p.grad for p in train]
    for k in range(5):                                       # eager steps with fresh gradient allocations
        gr = rand_grads(10 + k)
        for p, g in zip(train, gr):
            p.grad = g.clone()
        assert fs.step() is not None
        e_own.update()
        ref_step(gr)
    del gr
    torch.cuda.empty_cache()                                 # the eager gradients' memory is really gone
    junk = [torch.full((1 << 20,), float("nan"), device=DEV) for _ in range(8)]   # and may be reused by anything
    gr = rand_grads(99)
    for s_, g in zip(static_src, gr):
        s_.copy_(g)
    for p, g in zip(train, captured):
        p.grad = g
    graph.replay()
    ref_step(gr)
    torch.cuda.synchronize()
    for (name, p), q in zip(m_ref.named_parameters(), m_own.parameters()):
        assert torch.isfinite(q).all(), name
        assert float((p - q).abs().max()) <= 5e-6 * float(p.detach().abs().max()), name
        if p.requires_grad:
            assert float((e_ref.shadow[name] - e_own.shadow[name]).abs().max()) <= 5e-6 * float(p.detach().abs().max()), name
    del junk

"""The callers either side of the hot path, pinned by reference-generated fixtures:

* ``VariationalPosterior.sample`` / ``.summary`` / ``.diagnostics`` under the EMA swap
  (reference posterior/variational_posterior.py:93-144, inference/exponential_moving_average.py:29-41):
  tests/golden/posterior_sample.npz = {live state_dict, EMA shadow, every torch.randn draw} -> {theta, paths, summary}.
* ``VariationalInferenceTrainer.pretrain_sde_parameters`` (reference inference/trainer.py:208-259):
  tests/golden/pretrain.npz = {draws} -> {per-iteration loss / best loss / median sigma, returned mean}; the LV case has one
  iteration whose loss is non-finite (the reference skips that update).

Both were made by ``tests/golden/make_golden.py posterior_sample pretrain`` (imports the reference).  The draws are injected by
replacing ``torch.randn`` with a feeder that hands out persistent buffers per shape: a HIP-graph replay reads the same buffers,
so the graph-replayed pre-training loop on the GPU sees exactly the recorded draws too."""
from contextlib import contextmanager
from pathlib import Path

import numpy as np
import pytest
import torch

from helpers import GOLDEN, rel_err
from viforsdes_amd import (EncoderConfig, GaussianObservationLikelihood, HeadConfig, Observations, PretrainConfig, Prior,
                           PriorType, TrainingConfig)
from viforsdes_amd.console import Console
from viforsdes_amd.examples.sdes import LotkaVolterra, OrnsteinUhlenbeck
from viforsdes_amd.inference.exponential_moving_average import ExponentialMovingAverage
from viforsdes_amd.inference.state_space import StateSpace
from viforsdes_amd.inference.trainer import VariationalInferenceTrainer
from viforsdes_amd.kernels.backend import set_backend
from viforsdes_amd.models.variational_sde_posterior import VariationalSDEPosterior
from viforsdes_amd.posterior.variational_posterior import VariationalPosterior


@pytest.fixture()
def oracle_backend():
    from oracle.torch_backend import OracleBackend
    set_backend(OracleBackend())
    yield
    set_backend(None)


class DrawFeeder:
    """Stands in for ``torch.randn``: every requested shape has ONE persistent buffer on the device (what a captured HIP graph
    keeps reading); ``load`` copies the next recorded draws into them."""

    def __init__(self, device):
        self.device, self.buffers, self.calls = torch.device(device), {}, 0

    def load(self, *draws):
        for d in draws:
            t = torch.as_tensor(np.asarray(d), dtype=torch.float32)
            buf = self.buffers.get(tuple(t.shape))
            if buf is None:
                self.buffers[tuple(t.shape)] = t.to(self.device).clone()
            else:
                buf.copy_(t)

    def __call__(self, *shape, device=None, dtype=None, **kw):
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else tuple(shape)
        assert shape in self.buffers, f"unexpected torch.randn{shape}; recorded shapes {list(self.buffers)}"
        assert device is None or torch.device(device).type == self.device.type
        self.calls += 1
        return self.buffers[shape]

    @contextmanager
    def installed(self):
        real = torch.randn
        torch.randn = self
        try:
            yield self
        finally:
            torch.randn = real


def _case(d, name):
    pre = name + "::"
    return {k[len(pre):]: v for k, v in d.items() if k.startswith(pre)}


def _state(c, prefix):
    sd = {}
    for k, v in c.items():
        if k.startswith(prefix):
            t = torch.from_numpy(v)
            if k.endswith("rope_freqs"):
                t = torch.view_as_complex(t.contiguous())
            sd[k[len(prefix):]] = t
    return sd


# ------------------------------------------------------------------------------------------------- posterior sample / summary
def _posterior(c, name, device):
    S, P, n, n_summary = (int(v) for v in c["cfg"])
    model = VariationalSDEPosterior(S, S, P, EncoderConfig(hidden_dim=32, cond_dim=16, num_heads=4, depth=2),
                                    HeadConfig(hidden_dim=16, num_layers=2), [int(i) for i in c["theta_pos"]])
    model.load_state_dict(_state(c, "init::"))
    model.to(device)
    ema = ExponentialMovingAverage(model)
    ema.load_state_dict({k: v.to(device) for k, v in _state(c, "ema::").items()})
    obs = Observations(times=torch.from_numpy(c["obs_times"]), values=torch.from_numpy(c["obs_values"]))
    prior = (Prior(type=PriorType.LOG_NORMAL, mean=0.0, std=1.5, dim=3) if name == "lv"
             else Prior(type=PriorType.NORMAL, mean=0.0, std=1.0, dim=3))
    vp = VariationalPosterior(model=model, exponential_moving_average=ema, prior=prior, observations=obs,
                              time_horizon=float(c["horizon"]), time_step=float(c["dt"]),
                              state_space=StateSpace(S, [int(i) for i in c["state_pos"]]),
                              evidence_lower_bound_history=[-3.0, -2.0], device=torch.device(device))
    return vp, n, n_summary


def _check_posterior(name, device, tol):
    c = _case(dict(np.load(f"{GOLDEN}/posterior_sample.npz")), name)
    vp, n, n_summary = _posterior(c, name, device)
    live = {k: v.clone() for k, v in vp.model.state_dict().items()}
    feeder = DrawFeeder(device)
    with feeder.installed():
        feeder.load(c["sample_theta_eps"], c["sample_noise"])
        s = vp.sample(n)
        assert feeder.calls == 2
        feeder.load(c["summary_theta_eps"], c["summary_noise"])
        summ = vp.summary(n_summary)
        assert feeder.calls == 4
    for k, v in vp.model.state_dict().items():          # the EMA swap was undone: live weights untouched, bit for bit
        assert torch.equal(v, live[k]), k
    assert not vp.model.training
    got = lambda t: t.detach().cpu().numpy()
    assert rel_err(got(s.sde_parameters), c["sde_parameters"]) < tol
    assert rel_err(got(s.diffusion_paths), c["diffusion_paths"]) < tol
    # sample() ran on the SHADOW: with the live weights the paths differ by far more than the tolerance
    with feeder.installed():
        feeder.load(c["sample_theta_eps"], c["sample_noise"])
        vp.exponential_moving_average.load_state_dict({n_: p.detach().clone() for n_, p in vp.model.named_parameters()})
        wrong = vp.sample(n)
    assert rel_err(got(wrong.diffusion_paths), c["diffusion_paths"]) > 50 * tol
    assert rel_err(got(summ.sde_parameter_mean), c["summary_mean"]) < tol
    assert rel_err(got(summ.sde_parameter_std), c["summary_std"]) < 5 * tol
    q = summ.sde_parameter_quantiles
    assert rel_err(got(torch.stack([q.q05, q.q25, q.q50, q.q75, q.q95])), c["summary_quantiles"]) < tol
    assert rel_err(got(summ.diffusion_path_mean), c["summary_path_mean"]) < tol
    assert rel_err(got(summ.diffusion_path_std), c["summary_path_std"]) < 10 * tol
    dg = vp.diagnostics()
    assert [dg.final_evidence_lower_bound, dg.n_iterations] == list(c["diagnostics"])


@pytest.mark.parametrize("name", ["lv", "ou"])
def test_posterior_sample_and_summary_match_reference(oracle_backend, name):
    _check_posterior(name, "cpu", 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["lv", "ou"])
def test_posterior_sample_and_summary_match_reference_on_gpu(name):
    """Same fixture through the HIP no-grad sampling kernel and the encoder's GPU route (fp32: mixed precision is a trainer
    setting, ``sample`` runs outside autocast as in the reference)."""
    _check_posterior(name, "cuda", 5e-5)


# ------------------------------------------------------------------------------------------------------------- pre-training
class _Sink:
    def __init__(self, feeder, c):
        self.feeder, self.c, self.log = feeder, c, []

    def update(self, step, mse, best, sigma_median):
        self.log.append([float(mse), float(best), float(sigma_median)])
        if step + 1 < len(self.c["theta_eps"]):          # the next iteration's draws go into the buffers the loop reads
            self.feeder.load(self.c["theta_eps"][step + 1], self.c["path_noise"][step + 1])


class _RecordingConsole(Console):
    def __init__(self, sink):
        super().__init__(enabled=False)
        self.sink = sink

    @contextmanager
    def pretrain_progress(self, total):
        yield self.sink


def _check_pretrain(name, device, tol):
    c = _case(dict(np.load(f"{GOLDEN}/pretrain.npz")), name)
    K, B, bad = (int(v) for v in c["cfg"])
    feeder = DrawFeeder(device)
    sink = _Sink(feeder, c)
    sde = LotkaVolterra() if name == "lv" else OrnsteinUhlenbeck()
    prior = (Prior(type=PriorType.LOG_NORMAL, mean=0.0, std=1.5, dim=3) if name == "lv"
             else Prior(type=PriorType.NORMAL, mean=0.0, std=1.0, dim=3))
    obs = Observations(times=torch.from_numpy(c["obs_times"]), values=torch.from_numpy(c["obs_values"]))
    tr = VariationalInferenceTrainer(
        sde=sde, observations=obs, observation_likelihood=GaussianObservationLikelihood(variance=float(c["var"])), prior=prior,
        time_horizon=float(c["horizon"]), config=TrainingConfig(time_step=float(c["dt"]), batch_size=4, n_iterations=2),
        encoder_config=EncoderConfig(hidden_dim=32, cond_dim=16, num_heads=4, depth=1),
        head_config=HeadConfig(hidden_dim=16, num_layers=1), state_positive_dims=[int(i) for i in c["state_pos"]],
        sde_param_positive_dims=[int(i) for i in c["theta_pos"]], device=device, mixed_precision=False,
        console=_RecordingConsole(sink))
    with feeder.installed():
        if "init_draw" in c:
            feeder.load(c["init_draw"])
        feeder.load(c["theta_eps"][0], c["path_noise"][0])
        best_mu = tr.pretrain_sde_parameters(PretrainConfig(n_iterations=K, batch_size=B, learning_rate=0.02))
    log, ref = np.array(sink.log), c["log"]
    assert log.shape == ref.shape == (K, 3)
    finite = np.isfinite(ref[:, 0])
    assert (np.isfinite(log[:, 0]) == finite).all()
    if bad >= 0:
        assert not finite[bad] and finite.sum() == K - 1
    assert np.allclose(log[finite, 0], ref[finite, 0], rtol=tol), (log[:, 0], ref[:, 0])           # per-iteration loss
    assert np.allclose(log[:, 1], ref[:, 1], rtol=tol)                                             # best loss so far
    assert np.allclose(log[:, 2], ref[:, 2], rtol=tol)                                             # median sigma (after updates)
    assert np.allclose(best_mu.detach().cpu().numpy(), c["best_mu"], rtol=tol, atol=tol)
    return tr


@pytest.mark.parametrize("name", ["ou", "lv"])
def test_pretraining_trajectory_matches_reference(name):
    _check_pretrain(name, "cpu", 2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ou", "lv"])
def test_pretraining_graph_replay_matches_reference_on_gpu(name):
    """The HIP-graph replayed loop (one graph = simulate with the HIP Euler-Maruyama kernels, backward, clip, Adam; a non-finite
    iteration undone from the graph's own snapshot) against the reference's eager CPU loop on the same draws."""
    tr = _check_pretrain(name, "cuda", 5e-4)
    assert getattr(tr, "_pretrain_graph", None) is not None, "the pre-training loop did not replay from a HIP graph"


# ---------------------------------------------------------------------------------------------- captured sampling call (GPU)
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["lv", "ou"])
def test_repeated_sample_replays_a_graph_and_matches_the_eager_call(name):
    """``sample(n)`` with a repeated ``n`` replays the call as one HIP graph from the second occurrence on: with the generator
    seeded alike the replay must give what the eager call gives (same kernels, same Philox offsets), the EMA swap must still be
    undone, new EMA weights must reach the replay (it reads the parameters in place), and two replays must differ (fresh draws)."""
    from viforsdes_amd.inference import diffusion_path_sampler as dps
    c = _case(dict(np.load(f"{GOLDEN}/posterior_sample.npz")), name)
    vp, n, _ = _posterior(c, name, "cuda")
    live = {k: v.clone() for k, v in vp.model.state_dict().items()}
    vp.sample(n)                                    # first call of this size: eager
    vp.sample(n)                                    # second: captured
    assert vp._captured.get((n, None)) is not None, "capture failed: sample() fell back to the eager call"

    def eager(seed):
        dps.SAMPLE_GRAPH = False
        try:
            torch.manual_seed(seed)
            return vp.sample(n)
        finally:
            dps.SAMPLE_GRAPH = True

    def replay(seed):
        torch.manual_seed(seed)
        return vp.sample(n)
    a, b = eager(11), replay(11)
    assert torch.allclose(a.sde_parameters, b.sde_parameters, rtol=1e-6, atol=1e-7)
    assert rel_err(b.diffusion_paths.cpu().numpy(), a.diffusion_paths.cpu().numpy()) < 1e-5
    b2 = vp.sample(n)                               # no reseed: the next draws
    assert not torch.equal(b2.diffusion_paths, b.diffusion_paths)
    for k, v in vp.model.state_dict().items():      # swap undone after a replay as well
        assert torch.equal(v, live[k]), k
    # new shadow weights (here: the live ones) must be what the next replay samples with
    vp.exponential_moving_average.load_state_dict({n_: p.detach().clone() for n_, p in vp.model.named_parameters()})
    a3, b3 = eager(5), replay(5)
    assert rel_err(b3.diffusion_paths.cpu().numpy(), a3.diffusion_paths.cpu().numpy()) < 1e-5
    assert rel_err(b3.diffusion_paths.cpu().numpy(), replay(5).diffusion_paths.cpu().numpy()) < 1e-7


@pytest.mark.gpu
def test_captured_sampler_under_autocast_follows_parameter_changes():
    """The bench's sampling leg: ``CapturedPathSampler`` under bf16 autocast at the OU benchmark's size (encoder 256 wide: own
    GEMM kernels reading bf16 operand packs).  A parameter changed between two replays must reach the second one -- the packs
    are refreshed outside the graph, before the replay."""
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    from bench import build_trainer
    from viforsdes_amd.examples.sdes import ou_problem
    from viforsdes_amd.inference.diffusion_path_sampler import CapturedPathSampler, sample_diffusion_paths
    problem = ou_problem()
    horizon, dt = problem[4], problem[5]
    n = 128
    tr = build_trainer(problem, n, torch.device("cuda:0"), True, seed=5, enc_hidden=256, enc_depth=2)
    model, ctx = tr.ctx.model.eval(), tr.ctx
    smp = CapturedPathSampler(model, ctx.observations, horizon, dt, tr.state_space, n, autocast_dtype=torch.bfloat16)

    @torch.no_grad()
    def eager(seed):
        torch.manual_seed(seed)
        theta = model.sde_parameter_posterior.rsample(n)
        x0 = ctx.observations.values[0].unsqueeze(0).expand(n, -1)
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            return sample_diffusion_paths(model.encoder, model.head, ctx.observations, theta, x0, horizon, dt, tr.state_space).x

    def replay(seed):
        torch.manual_seed(seed)
        return smp()[1].clone()
    assert rel_err(replay(3).cpu().numpy(), eager(3).cpu().numpy()) < 1e-5
    with torch.no_grad():
        for p in model.encoder.parameters():
            p.mul_(1.05)
    changed = eager(3)
    assert rel_err(changed.cpu().numpy(), eager(3).cpu().numpy()) == 0.0
    assert rel_err(replay(3).cpu().numpy(), changed.cpu().numpy()) < 1e-5


@pytest.mark.gpu
def test_mixed_precision_sample_is_the_fp32_sample_within_bf16_accuracy():
    """``sample(n, mixed_precision=True)``: encoder under bf16 autocast (the fused encoder kernels), same draws -- the paths agree
    with the fp32 call to bf16 accuracy, eagerly (first call) and replayed (third call), and the live weights come back."""
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    from bench import build_trainer
    from viforsdes_amd.examples.sdes import ou_problem
    problem = ou_problem()
    n = 128
    tr = build_trainer(problem, n, torch.device("cuda:0"), True, seed=9, enc_hidden=256, enc_depth=2)
    vp = VariationalPosterior(model=tr.ctx.model, exponential_moving_average=tr.ctx.ema, prior=problem[3], observations=problem[1],
                              time_horizon=problem[4], time_step=problem[5], state_space=tr.state_space,
                              evidence_lower_bound_history=[], device=torch.device("cuda:0"))
    live = {k: v.clone() for k, v in vp.model.state_dict().items()}

    def draw(mixed):
        torch.manual_seed(21)
        return vp.sample(n, mixed_precision=mixed)
    ref = draw(False)
    for call in range(3):                          # eager, capture + replay, replay
        got = draw(True)
        assert torch.equal(got.sde_parameters, ref.sde_parameters)
        err = rel_err(got.diffusion_paths.cpu().numpy(), ref.diffusion_paths.cpu().numpy())
        assert 0.0 < err < 3e-2, (call, err)
    assert vp._captured.get((n, torch.bfloat16)) is not None
    for k, v in vp.model.state_dict().items():
        assert torch.equal(v, live[k]), k

"""CPU, gloo, world_size 2: the data-parallel ELBO step (one process per rank).

Checks that (1) ranks start from identical parameters, (2) each rank draws different Monte-Carlo
samples, (3) after the step the gradients every rank applied are the MEAN of the per-rank gradients
(verified against a single-process recomputation of both ranks' losses), and (4) parameters stay
bit-identical across ranks after several optimizer steps."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_trainer(seed):
    from viforsdes_amd import EncoderConfig, HeadConfig, TrainingConfig
    from viforsdes_amd.console import Console
    from viforsdes_amd.examples.sdes import ou_problem
    from viforsdes_amd.inference.trainer import VariationalInferenceTrainer
    sde, obs, like, prior, horizon, dt, sp, tp = ou_problem()
    tr = VariationalInferenceTrainer(
        sde=sde, observations=obs, observation_likelihood=like, prior=prior, time_horizon=horizon,
        config=TrainingConfig(time_step=0.25, batch_size=4, n_iterations=3, learning_rate=1e-3, sde_param_lr=1e-2),
        encoder_config=EncoderConfig(hidden_dim=16, cond_dim=8, num_heads=2, depth=1),
        head_config=HeadConfig(hidden_dim=8, num_layers=2), state_positive_dims=sp, sde_param_positive_dims=tp,
        device="cpu", mixed_precision=False, console=Console(enabled=False), seed=seed)
    with torch.no_grad():
        g = torch.Generator().manual_seed(99)
        w = tr.ctx.model.head.out_proj.weight
        w.copy_(torch.randn(w.shape, generator=g) * 0.2)
    tr.ctx.model.train()
    return tr


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    # VSDE_TEST_RANK_ENV="<rank>:<NAME>=<value>[,...]": environment of ONE rank only (ranks that disagree on a switch)
    for item in filter(None, os.environ.get("VSDE_TEST_RANK_ENV", "").split(",")):
        r, kv = item.split(":", 1)
        if int(r) == rank:
            k, v = kv.split("=", 1)
            os.environ[k] = v
    from oracle.torch_backend import OracleBackend
    from viforsdes_amd.kernels.backend import set_backend
    set_backend(OracleBackend())
    tr = _make_trainer(seed=77)
    ctx, model = tr.ctx, tr.ctx.model
    assert ctx.is_distributed and ctx.world_size == world and dist.get_backend() == "gloo"
    if os.environ.get("VSDE_TEST_REVERSE_HOOKS") == str(rank):
        # this rank's gradients "arrive" in the opposite order (what a rank-dependent graph or deferral would do)
        gs = ctx.grad_sync
        real = gs._propose_early
        gs._propose_early = lambda: (setattr(gs, "_recording", list(reversed(gs._recording or []))), real())[1]
    flat0 = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.zeros_like(flat0) for _ in range(world)]
    dist.all_gather(gathered, flat0)
    assert all(torch.equal(gathered[0], g) for g in gathered), "ranks must start from identical parameters"
    B, P, T, S = 4, 3, 20, 1
    eps_theta, noise = torch.randn(B, P), torch.randn(B, T, S)  # per-rank RNG stream (seed + rank)
    both = [torch.zeros_like(eps_theta) for _ in range(world)]
    dist.all_gather(both, eps_theta)
    assert not torch.equal(both[0], both[1]), "ranks must draw different samples"
    tr._train_step(model, theta_eps=eps_theta, path_noise=noise)
    applied = ctx.grad_sync.flat_gradients()  # unscaled (no GradScaler on CPU), clipped in place
    torch.save({"eps": eps_theta, "noise": noise, "applied": applied}, os.path.join(out_dir, f"r{rank}.pt"))
    for _ in range(2):
        tr._train_step(model)
        ctx.ema.update()
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    dist.all_gather(gathered, flat)
    assert all(torch.equal(gathered[0], g) for g in gathered), "parameters diverged across ranks"
    # a reduce() that follows an eager step WITHOUT a zero_grad() in between (the split-graph replay, trainer.capture_step_graph)
    # must send every bucket again: fill the flat buffer with rank + 1, reduce, expect the mean everywhere
    gs = ctx.grad_sync
    gs.flat.fill_(float(rank + 1))
    gs.reduce()
    whole = bool(torch.all(gs.flat == (world + 1) / 2.0))
    torch.save({"final": flat, "early_launches": gs.early_launches, "early": list(gs._early), "order_ok": whole,
                "early_fraction": gs.early_fraction()}, os.path.join(out_dir, f"final{rank}.pt"))
    ctx.cleanup()


@pytest.mark.timeout(600)
def test_early_bucket_overlap_is_bit_identical(tmp_path, monkeypatch):
    """The early gradient bucket (sent from inside the backward pass from the second step on) against the plain path (everything
    after the backward): the same three optimizer steps on two ranks must end in bit-identical parameters, and the overlapped run
    must really have launched early (steps 2 and 3)."""
    finals = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("VSDE_DP_OVERLAP", mode)
        monkeypatch.setenv("VSDE_DP_OVERLAP_MIN", "0")
        d = tmp_path / f"overlap{mode}"
        d.mkdir()
        mp.spawn(_worker, args=(2, _free_port(), str(d)), nprocs=2, join=True)
        finals[mode] = torch.load(d / "final0.pt")
    assert finals["1"]["early_launches"] == 2 and finals["0"]["early_launches"] == 0
    assert torch.equal(finals["1"]["final"], finals["0"]["final"])


@pytest.mark.timeout(600)
def test_ranks_that_disagree_on_switches_and_arrival_order_share_rank0_layout(tmp_path, monkeypatch):
    """the [early | late] layout of the flat gradient buffer used to be built from each rank's OWN hook
    arrival order and environment.  Rank 1 here has the overlap switched off and reports its arrivals in reverse: both ranks must
    end up with rank 0's early list (broadcast after the first step), identical parameters after three steps -- the same as a run
    without any overlap -- and a reduce() issued without a zero_grad() in between must still average every bucket."""
    monkeypatch.setenv("VSDE_DP_OVERLAP_MIN", "0")
    d0 = tmp_path / "plain"; d0.mkdir()
    monkeypatch.setenv("VSDE_DP_OVERLAP", "0")
    mp.spawn(_worker, args=(2, _free_port(), str(d0)), nprocs=2, join=True)
    monkeypatch.setenv("VSDE_DP_OVERLAP", "1")
    monkeypatch.setenv("VSDE_TEST_RANK_ENV", "1:VSDE_DP_OVERLAP=0")
    d1 = tmp_path / "mixed"; d1.mkdir()
    mp.spawn(_worker, args=(2, _free_port(), str(d1)), nprocs=2, join=True)
    monkeypatch.delenv("VSDE_TEST_RANK_ENV")
    monkeypatch.setenv("VSDE_TEST_REVERSE_HOOKS", "1")
    d2 = tmp_path / "reversed"; d2.mkdir()
    mp.spawn(_worker, args=(2, _free_port(), str(d2)), nprocs=2, join=True)
    plain = torch.load(d0 / "final0.pt")
    for d in (d1, d2):
        r0, r1 = torch.load(d / "final0.pt"), torch.load(d / "final1.pt")
        assert r0["early"] and r0["early"] == r1["early"], (r0["early"], r1["early"])
        assert 0.0 < r0["early_fraction"] < 1.0
        assert r0["order_ok"] and r1["order_ok"]
        assert torch.equal(r0["final"], r1["final"]) and torch.equal(r0["final"], plain["final"])
    assert torch.load(d1 / "final0.pt")["early_launches"] == 2 and torch.load(d1 / "final1.pt")["early_launches"] == 0


@pytest.mark.timeout(600)
def test_two_rank_gradient_average(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    recs = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    assert torch.equal(recs[0]["applied"], recs[1]["applied"])
    # single-process recomputation: mean of the two ranks' gradients, then the same clipping
    from oracle.torch_backend import OracleBackend
    from viforsdes_amd.kernels.backend import set_backend
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    set_backend(OracleBackend())
    try:
        grads = []
        for r in range(world):
            tr = _make_trainer(seed=77)
            model, ctx = tr.ctx.model, tr.ctx
            ctx.grad_sync.zero_grad()
            theta = model.sde_parameter_posterior.rsample(4, eps=recs[r]["eps"])
            from viforsdes_amd.inference.diffusion_path_sampler import sample_diffusion_paths
            from viforsdes_amd.inference.evidence_lower_bound import compute_evidence_lower_bound
            sample = sample_diffusion_paths(model.encoder, model.head, ctx.observations, theta, ctx.x0_buffer, tr.time_horizon,
                                            tr.config.time_step, tr.state_space, noise=recs[r]["noise"])
            res = compute_evidence_lower_bound(tr.sde, ctx.observations, tr.observation_likelihood, tr.prior,
                                               model.sde_parameter_posterior, theta, sample, tr.config.time_step)
            (-res.evidence_lower_bound).backward()
            grads.append(ctx.grad_sync.flat_gradients())
        mean = (grads[0] + grads[1]) / 2
        norm = mean.norm()
        clipped = mean * min(1.0, float(1.0 / (norm + 1e-6)))
        assert torch.allclose(clipped, recs[0]["applied"], rtol=1e-5, atol=1e-7)
    finally:
        set_backend(None)


def test_training_state_resume_is_bit_exact():
    """3 steps + save + 3 steps  ==  restore into a fresh trainer + the same last 3 steps (parameters, EMA, optimizer moments
    and RNG streams all travel in training_state_dict)."""
    import copy
    from oracle.torch_backend import OracleBackend
    from viforsdes_amd.kernels.backend import set_backend
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    set_backend(OracleBackend())
    try:
        a = _make_trainer(seed=5)
        for _ in range(3):
            a._train_step(a.ctx.model); a.ctx.ema.update()
        state = copy.deepcopy(a.training_state_dict())
        for _ in range(3):
            a._train_step(a.ctx.model); a.ctx.ema.update()
        b = _make_trainer(seed=123)  # different initialisation: everything must come from the state
        b.load_training_state_dict(state)
        for _ in range(3):
            b._train_step(b.ctx.model); b.ctx.ema.update()
        for (n, p), (_, q) in zip(a.ctx.model.named_parameters(), b.ctx.model.named_parameters()):
            assert torch.equal(p, q), n
        for n in a.ctx.ema.shadow:
            assert torch.equal(a.ctx.ema.shadow[n], b.ctx.ema.shadow[n]), n
    finally:
        set_backend(None)


def _pretrain_worker(rank, world, port, out_dir):
    """infer(pretrain=True) on two ranks: each rank pre-trains on its own draws (seed + rank), so without the broadcast of the
    pre-trained mean the replicas would start -- and stay -- apart."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from oracle.torch_backend import OracleBackend
    from viforsdes_amd import EncoderConfig, HeadConfig, InferenceConfig, PretrainConfig, TrainingConfig, infer
    from viforsdes_amd.console import Console
    from viforsdes_amd.examples.sdes import ou_problem
    from viforsdes_amd.kernels.backend import set_backend
    set_backend(OracleBackend())
    sde, obs, like, prior, horizon, dt, sp, tp = ou_problem()
    captured = {}
    from viforsdes_amd.inference import trainer as infer_mod
    real_cleanup = infer_mod.VariationalInferenceTrainer.cleanup

    def gather_then_cleanup(self):  # compare the replicas while the process group is still up
        flat = torch.cat([p.detach().reshape(-1) for p in self.ctx.model.parameters()])
        both = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        captured["same"] = all(torch.equal(both[0], b) for b in both)
        ema = self.ctx.ema.shadow["sde_parameter_posterior.mean"].clone()
        emas = [torch.zeros_like(ema) for _ in range(world)]
        dist.all_gather(emas, ema)
        captured["ema_same"] = all(torch.equal(emas[0], e) for e in emas)
        real_cleanup(self)

    infer_mod.VariationalInferenceTrainer.cleanup = gather_then_cleanup
    cfg = InferenceConfig(training=TrainingConfig(time_step=0.25, batch_size=4, n_iterations=3),
                          encoder=EncoderConfig(hidden_dim=16, cond_dim=8, num_heads=2, depth=1),
                          head=HeadConfig(hidden_dim=8, num_layers=2), sde_param_positive_dims=tp, device="cpu",
                          mixed_precision=False, console=Console(enabled=False), seed=5,
                          pretrain=PretrainConfig(n_iterations=5, batch_size=32))
    infer(sde, obs, like, prior, horizon, cfg)
    assert captured["same"], "replicas diverged: the pre-trained mean was not broadcast"
    assert captured["ema_same"]


@pytest.mark.timeout(600)
def test_two_rank_pretrain_keeps_replicas_identical(tmp_path):
    mp.spawn(_pretrain_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)


def _split_worker(rank, world, port, out_dir):
    """The step in the order the GPU's two-graph replay runs it -- [forward/backward + pack] | all-reduce | [attach is fixed at
    capture: p.grad ARE the flat views] optimizer -- against the plain ``_train_step`` on a twin trainer with the same draws."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from oracle.torch_backend import OracleBackend
    from viforsdes_amd.kernels.backend import set_backend
    set_backend(OracleBackend())
    ref, two = _make_trainer(seed=31), _make_trainer(seed=31)
    assert ref.ctx.grad_sync.active and two.ctx.grad_sync.active
    g = torch.Generator().manual_seed(1000 + rank)
    draws = [(torch.randn(4, 3, generator=g), torch.randn(4, 20, 1, generator=g)) for _ in range(3)]
    for k, (eps, noise) in enumerate(draws):
        ref._train_step(ref.ctx.model, theta_eps=eps, path_noise=noise)
        ref.ctx.ema.update()
        # --- phase 1 (graph 1 on the GPU): forward/backward, gradients packed into the flat buffer
        two._forward_backward(two.ctx.model, eps, noise)
        gs = two.ctx.grad_sync
        gs.pack()
        # --- between the graphs: the eager all-reduce of the flat buffer
        gs.reduce()
        if k == 0:
            gs.attach()          # captured once: from here on p.grad are the flat buffer's views and STAY so across replays
        else:
            # a replay does not run Python: the optimizer graph keeps reading the views attached at capture time
            for p, v in zip(gs.params, gs._views):
                p.grad = v
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(gs.params, gs._views))
        # --- phase 2 (graph 2): unscale, clip, AdamW, EMA
        two._optimizer_step()
        two.ctx.ema.update()
        for a, b in zip(ref.ctx.model.parameters(), two.ctx.model.parameters()):
            assert torch.equal(a, b), "split two-phase step diverged from the plain step"
    flat = torch.cat([p.detach().reshape(-1) for p in two.ctx.model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert all(torch.equal(gathered[0], g_) for g_ in gathered), "parameters diverged across ranks"
    shadow = torch.cat([v.reshape(-1) for v in two.ctx.ema.shadow.values()])
    sh = [torch.zeros_like(shadow) for _ in range(world)]
    dist.all_gather(sh, shadow)
    assert all(torch.equal(sh[0], s_) for s_ in sh), "EMA shadows diverged across ranks"
    torch.save({"ok": True}, os.path.join(out_dir, f"split{rank}.pt"))
    ref.ctx.cleanup()


@pytest.mark.timeout(600)
def test_split_two_phase_step_matches_plain_step_on_two_ranks(tmp_path):
    """pack -> all-reduce -> (attach) -> optimizer as separate phases (the ordering of the two-HIP-graph replay under data
    parallelism, trainer.capture_step_graph) gives bit-identical parameters to ``_train_step`` on both ranks, and the ranks stay
    identical (reference semantics: one gradient average per optimizer step before unscale / clip, trainer.py:128-131,
    training_context.py:59-68)."""
    world, port = 2, _free_port()
    mp.spawn(_split_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"split{r}.pt").exists() for r in range(world))

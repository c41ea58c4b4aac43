"""CPU: ``bench.py --gpus N`` self-launch logic (N fresh child ranks with the torchrun environment, rank 0's line relayed,
non-zero exit when any rank fails).  The children here are a stub script, not the benchmark (no GPU in this container)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, body, n=3):
    stub = tmp_path / "stub.py"
    stub.write_text(textwrap.dedent(body))
    driver = tmp_path / "driver.py"
    driver.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import torch
        torch.cuda.device_count = lambda: {n}       # the parent only counts devices, it never initialises the GPU
        import bench
        bench.self_launch({n}, ["--gpus", "{n}"], script={str(stub)!r})
    """))
    return subprocess.run([sys.executable, str(driver)], capture_output=True, text=True, timeout=120)


def test_self_launch_relays_rank0_and_sets_env(tmp_path):
    r = _run(tmp_path, """
        import json, os, sys
        assert sys.argv[1:] == ["--gpus", "3"]
        env = {k: os.environ[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                          "HSA_ENABLE_IPC_MODE_LEGACY")}
        print(json.dumps(env))
    """)
    assert r.returncode == 0, r.stderr
    import json
    env = json.loads(r.stdout.strip())
    assert env["RANK"] == "0" and env["WORLD_SIZE"] == "3" and env["MASTER_ADDR"] == "127.0.0.1"
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and len(r.stdout.strip().splitlines()) == 1  # only rank 0's line


def test_self_launch_fails_when_a_rank_fails(tmp_path):
    r = _run(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(30)      # the surviving ranks would hang in a collective: the launcher must end them
    """)
    assert r.returncode != 0


def test_two_rank_line_carries_the_collective_fields(tmp_path):
    """A 2-rank launch whose children form a gloo group, all-reduce a buffer the way bench.py times the gradient exchange and let
    rank 0 print a line with the fields the driver reads (``n_gpus``, ``rccl_ranks``, ``allreduce_ms_per_step``): the launcher
    must hand every rank a working rendezvous environment and relay exactly rank 0's line."""
    r = _run(tmp_path, """
        import json, os, sys, time
        import torch, torch.distributed as dist
        assert sys.argv[1:] == ["--gpus", "2"]
        dist.init_process_group(backend="gloo")          # RANK / WORLD_SIZE / MASTER_* come from the launcher
        buf = torch.full((1 << 16,), float(dist.get_rank() + 1))
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(3):
            dist.all_reduce(buf)
        ms = (time.perf_counter() - t0) / 3 * 1e3
        assert float(buf[0]) == 12.0                      # 1 + 2 = 3, then doubled by each of the two further sums
        t = torch.tensor([ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if dist.get_rank() == 0:
            print(json.dumps({"n_gpus": dist.get_world_size(), "rccl_ranks": dist.get_world_size(),
                              "allreduce_ms_per_step": float(t), "scaling": "weak"}))
        dist.barrier()
        dist.destroy_process_group()
    """, n=2)
    assert r.returncode == 0, r.stderr
    import json
    line = json.loads(r.stdout.strip())
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["allreduce_ms_per_step"] > 0 and line["scaling"] == "weak"

"""CPU: the multi-GPU batch partition and the final gather, exercised with world_size 2 over gloo."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pailliercryptolib_python_amd import sharding


def test_shard_bounds_cover_the_batch_exactly():
    for n in (0, 1, 7, 8, 9, 1000, 1 << 20):
        for world in (1, 2, 3, 8):
            b = sharding.shard_bounds(n, world)
            assert len(b) == world and b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in b]
            assert max(sizes) - min(s for s in sizes if s or n == 0 or True) <= -(-n // world)


def test_library_shard_plan_agrees_with_the_python_partition():
    """pai_shard_plan (C ABI, no device needed) == sharding.shard_bounds for every rank."""
    from pailliercryptolib_python_amd import engine

    for n in (0, 1, 7, 8, 9, 1000, (1 << 20) + 3):
        for world in (1, 2, 3, 8):
            assert [(b, b + c) for b, c in engine.shard_plan(n, world)] == sharding.shard_bounds(n, world)


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = torch.arange(n_total * 4, dtype=torch.int32).reshape(n_total, 4)
    s, e = sharding.my_shard(n_total, rank, world)
    out = sharding.gather_rows(full[s:e].contiguous(), n_total)
    q.put((rank, bool(torch.equal(out, full))))
    dist.destroy_process_group()


def _worker_plan(rank, world, port, n_total, q):
    """One rank of the bench's strong-scaling arrangement on CPU tensors: the library's partition (pai_shard_plan), this
    rank's block, the final gather, and the round trip back to the blocks."""
    from pailliercryptolib_python_amd import engine

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = (torch.arange(n_total * 6, dtype=torch.int64).reshape(n_total, 6) * 2654435761 % (1 << 31)).to(torch.int32)
    plan = engine.shard_plan(n_total, world)
    begin, count = plan[rank]
    out = sharding.gather_rows(full[begin:begin + count].contiguous(), n_total)
    ok = bool(torch.equal(out, full)) and sum(c for _, c in plan) == n_total
    ok = ok and all(torch.equal(out[b:b + c], full[b:b + c]) for b, c in plan)      # every rank's block is where the plan says
    q.put((rank, ok, count))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [1003, 5, 64])
def test_shard_plan_and_gather_round_trip_world8_gloo(n_total):
    """World size 8 (BASELINE configs[3]/[4]) with a ragged tail (1003 = 7 x 126 + 121), with trailing ranks that own
    nothing (5 elements on 8 ranks) and with an exact split."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() + n_total) % 1000
    procs = [ctx.Process(target=_worker_plan, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert [r[0] for r in res] == list(range(world)) and all(r[1] for r in res)
    per = -(-n_total // world)
    assert [r[2] for r in res] == [max(0, min(per, n_total - g * per)) for g in range(world)]


@pytest.mark.parametrize("n_total", [10, 11, 1])
def test_gather_rows_world2_gloo(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n_total) % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_bench_refuses_to_run_fewer_ranks_than_asked():
    """`python bench.py --gpus N` launches N ranks itself; with fewer visible GPUs than ranks (none, in the CPU container)
    it must fail loudly instead of quietly measuring fewer devices — and a torchrun world that disagrees with --gpus too."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a box with fewer than two GPUs")
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PAI_BENCH_BACKEND")}
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                         timeout=300, env=env, cwd=str(root))
    assert res.returncode != 0 and "GPU(s) are visible" in res.stderr
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "4", "--steps", "1"], capture_output=True, text=True,
                         timeout=300, env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), cwd=str(root))
    assert res.returncode != 0 and "launcher started 2 rank(s)" in res.stderr

"""N > 1 path on CPU: world_size 2, gloo.  Each rank owns a contiguous env shard (harness backend,
no GPU), steps it with no data-path collective, and the done gather / counter reduction must equal
the single-process run of the whole batch."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, total, steps, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tetris_amd import VecTetris, _lib
    from tetris_amd.distributed import DoneGather, shard_range
    import harness_backend
    _lib._install_test_backend(harness_backend.binding())
    lo, hi = shard_range(total, rank, world)
    env = VecTetris(10, 20, hi - lo, device="cpu", auto_reset=True, seed=21, env_offset=lo)
    gather = DoneGather(hi - lo)
    idx_log = []
    for t in range(steps):
        env.step(env.random_actions())
        idx_log.append(gather.gather_indices(env.done))
    totals = gather.gather_counters(env.totals())
    if rank == 0:
        torch.save(dict(idx=idx_log, totals=totals), os.path.join(out_dir, "gathered.pt"))
    torch.save(dict(cols=env.cols, obs=env.obs), os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_shards_match_single_process(tmp_path, host_backend):
    total, steps, world = 512, 60, 2
    port = 29500 + (os.getpid() % 2000)
    mp.start_processes(_worker, args=(world, port, total, steps, str(tmp_path)), nprocs=world, join=True,
                       start_method="spawn")
    from tetris_amd import VecTetris
    whole = VecTetris(10, 20, total, device="cpu", auto_reset=True, seed=21)
    idx_ref = []
    for t in range(steps):
        whole.step(whole.random_actions())
        idx_ref.append(torch.nonzero(whole.done).flatten())
    g = torch.load(os.path.join(tmp_path, "gathered.pt"))
    for a, b in zip(g["idx"], idx_ref):
        assert torch.equal(a, b)
    assert torch.equal(g["totals"], whole.totals())
    parts = [torch.load(os.path.join(tmp_path, "rank%d.pt" % r)) for r in range(world)]
    assert torch.equal(torch.cat([p["cols"] for p in parts], dim=0), whole.cols)
    assert torch.equal(torch.cat([p["obs"] for p in parts]), whole.obs)
    assert int(g["totals"][1]) > 0

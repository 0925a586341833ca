"""bench.py's rank launcher and accounting, on CPU: `--gpus 2` started plainly must run TWO ranks
(gloo + the harness build of the lane logic stand in for RCCL + the HIP library) and print one
JSON line with n_gpus == 2; the byte accounting must follow the storage the library reports."""
import ctypes
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=280):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(TETRIS_BENCH_BACKEND="gloo")
    env.update(extra_env or {})
    # the CPU entry (tests/bench_cpu_entry.py) binds the harness build and runs bench.main() on CPU tensors
    return subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_cpu_entry.py")] + args, env=env,
                          capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(300)
def test_gpus_2_starts_two_ranks(host_backend):
    r = _run(["--gpus", "2", "--steps", "8", "--warmup", "4", "--batch", "512", "--gather-every", "4"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 8
    assert out["config"]["envs_per_gpu"] == 512
    assert out["done_gather"]["counter_gathers_in_timed_region"] == 2
    k = out["roofline"]["kernel_ms_per_rank"]
    assert 0 < k["min"] <= k["max"]
    # whole-job value: both ranks' envs
    assert abs(out["value"] - 2 * 512 * 8 / (out["ms_per_step"] * 8e-3)) < 1e-6 * out["value"]
    # what the collective layer itself saw: two ranks, two disjoint env ranges
    assert out["ranks"]["world_size_observed"] == 2
    assert [r["envs"] for r in out["ranks"]["per_rank"]] == [[0, 512], [512, 1024]]
    assert out["done_gather"]["bitmask_gathers_in_timed_region"] == 1


@pytest.mark.timeout(120)
def test_failing_rank_ends_the_job(host_backend):
    """One rank exits non-zero before the rendezvous: the parent must end its sibling (which would
    otherwise sit in the rendezvous until its timeout) and return 1."""
    import time
    t0 = time.time()
    r = _run(["--gpus", "2", "--steps", "8", "--warmup", "4", "--batch", "512"],
             extra_env=dict(TETRIS_TEST_FAIL_RANK="1"), timeout=100)
    assert r.returncode == 1, (r.returncode, r.stderr[-1000:])
    assert "rank(s) failed" in r.stderr and "rank 1 rc 3" in r.stderr
    assert time.time() - t0 < 60
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.timeout(120)
def test_parent_deadline(host_backend):
    """A job that outlives the parent's deadline is ended (rank hung in a collective)."""
    r = _run(["--gpus", "2", "--steps", "2000000", "--warmup", "0", "--batch", "64", "--gather-every", "1000000"],
             extra_env=dict(TETRIS_BENCH_DEADLINE_S="8"), timeout=100)
    assert r.returncode == 1 and "ending them" in r.stderr


@pytest.mark.timeout(300)
def test_single_rank_line_and_mismatch(host_backend):
    r = _run(["--gpus", "1", "--steps", "4", "--warmup", "2", "--batch", "256", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and "done_gather" not in out and out["config"]["device"] == "cpu"
    assert out["roofline"]["algorithmic_bytes_per_env_step"] == 111
    assert out["roofline"]["survey_bytes_per_env_step"] == 127
    assert out["roofline"]["traffic"] is None or out["roofline"]["traffic"] > 0
    # --gpus must agree with the world size torchrun set
    r = _run(["--gpus", "3", "--steps", "2", "--warmup", "0", "--batch", "64"],
             extra_env=dict(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "disagrees" in (r.stderr + r.stdout)


def test_algorithmic_bytes_follow_the_stored_planes():
    sys.path.insert(0, ROOT)
    import bench
    from tetris_amd import _lib
    old = _lib._install_test_backend(None)
    try:
        lib = _lib.load()
    finally:
        _lib._install_test_backend(old)
    ids = (ctypes.c_int32 * 2)(4, 3)
    for rows, want in ((20, 111), (40, 175), (21, 127)):  # 10x21 is not packed: ten u32 planes
        d = _lib.TetrisDesc()
        assert lib.desc_init(ctypes.byref(d), 10, rows, ids, 2, None) == 0
        planes = lib.n_planes(ctypes.byref(d))
        assert bench.algorithmic_bytes_per_env_step(planes, d.word_bytes) == want
        assert bench.algorithmic_bytes_per_env_step(planes, d.word_bytes, with_obs=False) == want - 32
    assert bench.survey_bytes_per_env_step(10, 4) == 127 and bench.survey_bytes_per_env_step(10, 8) == 207


def test_stale_traffic_profile_is_not_quoted(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    os.makedirs(tmp_path / "tetris_amd" / "csrc")
    (tmp_path / "tetris_amd" / "csrc" / "k.hip").write_text("// v1\n")
    prof = dict(envs=1 << 20, columns=10, rows=20, pieces="default", csrc_hash=bench.csrc_hash(),
                step_kernel_hbm_bytes_per_launch=123.0)
    (tmp_path / "profiles" / "pmc_traffic.json").write_text(json.dumps(prof))
    assert bench.load_traffic(10, 20, "default", 1 << 20) == 123.0
    assert bench.load_traffic(10, 40, "default", 1 << 20) is None       # other workload
    (tmp_path / "tetris_amd" / "csrc" / "k.hip").write_text("// v2\n")  # the kernels changed
    assert bench.load_traffic(10, 20, "default", 1 << 20) is None
    # what counts is the hash compiled into the LOADED library
    assert bench.load_traffic(10, 20, "default", 1 << 20, lib_hash=prof["csrc_hash"]) == 123.0
    assert bench.load_traffic(10, 20, "default", 1 << 20, lib_hash="0123456789abcdef") is None

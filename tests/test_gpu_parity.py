"""Parity tests proper: the HIP kernels (through the C-ABI, via tetris_amd)
against the CPU oracle and the golden fixtures.  Need a real MI355X."""
import pytest

import parity_cases as pc

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("C,R,pieces", [(10, 20, "default"), (10, 20, "standard7"), (10, 40, "default"),
                                        (10, 40, "standard7"), (6, 10, "standard7"), (8, 12, "default")])
def test_lockstep_small(orc, C, R, pieces):
    pc.lockstep_small(DEV, orc, C, R, pieces)


def test_cfg2_batch_65536_bit_exact(orc):
    """BASELINE config 2: 65,536 envs, 10x20, random actions, T = 512 steps, every output every step."""
    pc.cfg2_bit_exact(DEV, orc)


def test_no_auto_reset_and_invalid_actions(orc):
    pc.no_auto_reset_and_invalid_actions(DEV, orc)


def test_golden_trajectories_replay(orc, golden_dir):
    pc.golden_trajectories_replay(DEV, orc, golden_dir)


def test_golden_placements_afterstates(orc, golden_dir):
    pc.golden_placements_afterstates(DEV, orc, golden_dir)


def test_sharding_equals_single_batch():
    pc.sharding_equals_single_batch(DEV)


def test_full_size_properties():
    """BASELINE config 3 size: 1,048,576 envs."""
    pc.full_size_properties(DEV)


def test_tall_board_cfg5_properties():
    """BASELINE config 5: 10x40 (u64 columns) + auto-reset, size-independent invariants at 1M envs."""
    pc.full_size_properties(DEV, R=40, steps=100)


def test_mask_rescue_stress(orc):
    pc.mask_rescue_stress(DEV, orc, n_boards=1500)
    pc.mask_rescue_stress(DEV, orc, n_boards=500, R=40, seed=1)
    pc.mask_rescue_stress(DEV, orc, n_boards=400, R=20, C=12, seed=3)  # 12-bit level fields, 64-bit missing-cell words
    pc.mask_rescue_stress(DEV, orc, n_boards=200, R=40, C=11, seed=4)


def test_facade_golden_trajectories(golden_dir):
    """game.Tetris drop-in (batch of one) replaying recorded reference runs on the GPU."""
    import facade_cases as fc
    fc.golden_trajectory(DEV, golden_dir, "default", 20, 0)
    fc.golden_trajectory(DEV, golden_dir, "standard7", 40, 1, steps=120)
    fc.dtypes_and_directions(DEV, golden_dir)
    fc.reset_features(DEV, golden_dir)
    fc.best_policy(DEV, golden_dir)
    fc.rollouts_and_misc(DEV)


def test_edge_geometries(orc):
    pc.edge_geometries(DEV, orc)


def test_step_without_obs(orc):
    pc.step_without_obs(DEV, orc)


def test_greedy_policy(orc, golden_dir):
    pc.greedy_policy(DEV, orc, golden_dir)


def test_integration_md_stub_runs():
    """The ctypes stub printed in INTEGRATION.md is executable as written and agrees with VecTetris."""
    import os
    import re
    import torch
    from tetris_amd import VecTetris, build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(.*?)```", text, re.S).group(1)
    code = code.replace('ctypes.CDLL("libtetris_hip.so")', 'ctypes.CDLL(%r)' % build.SO_PATH)
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    ThreeL = type("ThreeL", (), {})
    ThreeLine = type("ThreeLine", (), {})
    ref_env = type("Env", (), dict(num_columns=10, num_rows=20, tetrominos=[ThreeL(), ThreeLine()]))()
    B = 1000
    hb = ns["HipBoards"](ref_env, B)
    hb.reset(seed=5)
    env = VecTetris(10, 20, B, device="cuda", auto_reset=True, seed=5)
    assert torch.equal(hb.meta, env.meta)
    for t in range(30):
        feats, nv = hb.get_after_states()
        f2, nv2 = env.get_after_states()
        assert torch.equal(feats, f2) and torch.equal(nv, nv2)
        a = env.random_actions().clone()
        hb.step(a, seed=5)
        env.step(a)
        assert torch.equal(hb.cols, env.cols) and torch.equal(hb.obs, env.obs) and torch.equal(hb.meta, env.meta)


def test_rollouts(orc):
    pc.rollouts(DEV, orc)


def test_step_many_equals_steps(orc):
    pc.step_many_equals_steps(DEV, orc)


def test_terminal_boards_are_refused():
    pc.terminal_boards_are_refused(DEV)


def test_golden_edges_through_kernels(orc, golden_dir):
    pc.golden_edges_through_kernels(DEV, orc, golden_dir)


def test_golden_placements_through_step(orc, golden_dir):
    pc.golden_placements_through_step(DEV, orc, golden_dir)


def test_feature_directions_in_kernels(orc, golden_dir):
    pc.feature_directions_in_kernels(DEV, orc, golden_dir)


def test_action_major_layout():
    pc.action_major_layout(DEV)


def test_state_dict_roundtrip():
    pc.state_dict_roundtrip(DEV)


def test_replay_stream_exhaustion():
    pc.replay_stream_exhaustion(DEV)


def test_device_bag_properties_1Mi():
    """1,048,576 envs (default set): permutation per bag, bag survives auto-reset and reset(mask)."""
    pc.device_bag_properties(DEV)


def test_rollouts_pinned_to_reference(orc, golden_dir):
    pc.rollouts_pinned_to_reference(DEV, orc, golden_dir)


def test_facade_rollout_script_and_render(golden_dir):
    import facade_cases as fc
    fc.rollout_script(DEV, golden_dir)
    fc.render_strings(DEV, golden_dir)


def test_cfg3_full_size_bit_exact(orc):
    """BASELINE config 3 (1,048,576 envs): every output of every step against the oracle."""
    pc.cfg3_full_size_bit_exact(DEV, orc)


def test_numpy_exact_bag_stream(orc, golden_dir):
    pc.numpy_exact_bag_stream(DEV, orc, golden_dir)


def test_graph_steps_equal_steps():
    pc.graph_steps_equal_steps(DEV)


def test_done_bit_packing():
    pc.done_bit_packing(DEV)


def test_cfg5_full_size_bit_exact(orc):
    """BASELINE config 5 at its full size -- 1,048,576 envs, 10x40 (u64 boards), in-kernel auto-reset --
    in lock-step with the oracle: every output of every step."""
    pc.cfg3_full_size_bit_exact(DEV, orc, R=40, steps=24, board_every=12)


def test_afterstate_family_full_size(orc):
    """get_after_states (both matrices) and get_best_policy of EVERY env of 262,144-env batches, whole-array
    bit-exact against the oracle, in steady-state play: 10x20 (u32 boards) and 10x40 (u64 boards)."""
    pc.afterstate_family_full_size(DEV, orc, B=1 << 18, R=20, steps=24, every=8)
    pc.afterstate_family_full_size(DEV, orc, B=1 << 18, R=40, pieces="standard7", steps=40, every=20)


@pytest.mark.parametrize("C", [5, 6, 7, 8, 9, 10, 11, 12])
def test_repeat_and_shard_consistency(C):
    """Every kernel of the afterstate family, 196,608 envs (several workgroups per compute unit): repeated
    launches agree bit for bit, the whole batch equals its 16,384-env shards, two copies stepping with the
    in-kernel greedy policy stay identical -- 32-bit and 64-bit boards, packed (x20, x40) and one plane per column
    (x24, x50: the NCH = 0 kernel variants), both piece sets."""
    for R, pieces in ((20, "default"), (20, "standard7"), (40, "default"), (40, "standard7"), (24, "default"), (50, "standard7")):
        pc.repeat_and_shard_consistency(DEV, C=C, R=R, pieces=pieces)


def test_rollouts_many_workgroups_per_compute_unit(orc):
    """perform_rollouts against the oracle on batches whose (env, action) lanes put several workgroups on every
    compute unit -- 32- and 64-bit boards, incl. the 9-column 64-bit kernel that had the last-VGPR hazard."""
    pc.rollouts(DEV, orc, B=12288, geometries=((10, "default", 20, 5), (9, "default", 40, 7), (12, "standard7", 40, 8)))


def test_bench_rccl_path_one_rank():
    """bench.py as the driver starts it, but with the collective path forced on in a world of ONE rank
    (TETRIS_BENCH_FORCE_DIST=1): process-group set-up over RCCL, the barrier, the bitmask all-gather on
    the side stream, the counter all-reduce and the max-over-ranks reduction all run on this GPU, so the
    N > 1 code path of BASELINE config 4 is exercised every round even without a multi-GPU node."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.update(TETRIS_BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                        "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["config"]["backend"] == "nccl" and out["config"]["device"] == "cuda"
    assert out["ranks"]["world_size_observed"] == 1 and out["ranks"]["backend_observed"] == "nccl"
    assert out["ranks"]["per_rank"][0]["envs"] == [0, 1 << 20]
    assert out["done_gather"]["bitmask_gathers_in_timed_region"] == 1 and out["done_gather"]["ms_per_gather"] > 0
    assert out["episodes"] > 0 and out["n_gpus"] == 1 and out["value"] > 1e9


def test_gather_payload_from_the_step():
    pc.gather_payload_from_the_step(DEV)


def test_rollouts_fed_pieces(orc, golden_dir):
    pc.rollouts_fed_pieces(DEV, orc, golden_dir)


def test_unsupported_widths_are_refused():
    import facade_cases as fc
    fc.unsupported_widths_are_refused(DEV)


def test_golden_wide_trajectories(orc, golden_dir):
    pc.golden_wide_trajectories(DEV, orc, golden_dir)

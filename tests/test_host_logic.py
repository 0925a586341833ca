"""Host logic + lane logic on CPU: tetris_amd bound to the g++ harness build of
csrc/tetris_core.hpp (tests/harness).  Same cases as tests/test_gpu_parity.py at
sizes that finish in seconds; no HIP code runs here."""
import pytest

import parity_cases as pc

DEV = "cpu"


@pytest.mark.parametrize("C,R,pieces", [(10, 20, "default"), (10, 20, "standard7"), (10, 40, "default"),
                                        (6, 10, "standard7"), (8, 12, "default")])
def test_lockstep_small(host_backend, orc, C, R, pieces):
    pc.lockstep_small(DEV, orc, C, R, pieces, B=257, steps=120)


def test_no_auto_reset_and_invalid_actions(host_backend, orc):
    pc.no_auto_reset_and_invalid_actions(DEV, orc, B=512)


def test_golden_trajectories_replay(host_backend, orc, golden_dir):
    pc.golden_trajectories_replay(DEV, orc, golden_dir)


def test_golden_placements_afterstates(host_backend, orc, golden_dir):
    pc.golden_placements_afterstates(DEV, orc, golden_dir)


def test_sharding_equals_single_batch(host_backend):
    pc.sharding_equals_single_batch(DEV, B=512)


def test_properties(host_backend):
    pc.full_size_properties(DEV, B=4096, steps=60)
    pc.full_size_properties(DEV, B=2048, R=40, steps=100)


def test_mask_rescue_stress(host_backend, orc):
    pc.mask_rescue_stress(DEV, orc, n_boards=600)
    pc.mask_rescue_stress(DEV, orc, n_boards=200, R=40, seed=1)
    pc.mask_rescue_stress(DEV, orc, n_boards=200, R=10, C=6, seed=2)
    pc.mask_rescue_stress(DEV, orc, n_boards=250, R=20, C=12, seed=3)  # 12-bit level fields, 64-bit missing-cell words
    pc.mask_rescue_stress(DEV, orc, n_boards=150, R=40, C=11, seed=4)


def test_edge_geometries(host_backend, orc):
    pc.edge_geometries(DEV, orc)


def test_step_without_obs(host_backend, orc):
    pc.step_without_obs(DEV, orc)


def test_greedy_policy(host_backend, orc, golden_dir):
    pc.greedy_policy(DEV, orc, golden_dir)


def test_afterstate_family_whole_batch(host_backend, orc):
    pc.afterstate_family_full_size(DEV, orc, B=1500, steps=40, every=8)
    pc.afterstate_family_full_size(DEV, orc, B=700, R=40, pieces="standard7", steps=60, every=20)
    pc.afterstate_family_full_size(DEV, orc, B=500, R=20, C=12, steps=40, every=10)


def test_selftest_runs_on_the_harness(host_backend):
    from tetris_amd import selftest
    assert selftest.run([(12, 20, "default"), (7, 40, "standard7"), (10, 24, "default")], device=DEV, verbose=False,
                        B=700, shard=256, warm=12) == []


def test_rollouts(host_backend, orc):
    pc.rollouts(DEV, orc)


def test_step_many_equals_steps(host_backend, orc):
    pc.step_many_equals_steps(DEV, orc)


def test_terminal_boards_are_refused(host_backend):
    pc.terminal_boards_are_refused("cpu")


def test_golden_edges_through_kernels(host_backend, orc, golden_dir):
    pc.golden_edges_through_kernels(DEV, orc, golden_dir)


def test_golden_placements_through_step(host_backend, orc, golden_dir):
    pc.golden_placements_through_step(DEV, orc, golden_dir)


def test_feature_directions_in_kernels(host_backend, orc, golden_dir):
    pc.feature_directions_in_kernels(DEV, orc, golden_dir, B=120)


def test_action_major_layout(host_backend):
    pc.action_major_layout(DEV, B=65)


def test_state_dict_roundtrip(host_backend):
    pc.state_dict_roundtrip(DEV, B=130)


def test_replay_stream_exhaustion(host_backend):
    pc.replay_stream_exhaustion(DEV)


def test_device_bag_properties(host_backend):
    pc.device_bag_properties(DEV, B=8192, steps=64)


def test_rollouts_pinned_to_reference(host_backend, orc, golden_dir):
    pc.rollouts_pinned_to_reference(DEV, orc, golden_dir)


def test_cfg3_lockstep_small(host_backend, orc):
    pc.cfg3_full_size_bit_exact(DEV, orc, B=6000, steps=60)


def test_numpy_exact_bag_stream(host_backend, orc, golden_dir):
    pc.numpy_exact_bag_stream(DEV, orc, golden_dir)


def test_graph_steps_equal_steps(host_backend):
    pc.graph_steps_equal_steps(DEV, B=300)


def test_done_bit_packing(host_backend):
    pc.done_bit_packing(DEV)


def test_gather_payload_from_the_step(host_backend):
    pc.gather_payload_from_the_step(DEV, B=64 * 5 + 11)


def test_rollouts_fed_pieces(host_backend, orc, golden_dir):
    pc.rollouts_fed_pieces(DEV, orc, golden_dir)


def test_golden_wide_trajectories(host_backend, orc, golden_dir):
    pc.golden_wide_trajectories(DEV, orc, golden_dir)

"""tetris_amd.Tetris facade on CPU (harness backend): host logic of game.py/state.py."""
import pytest

import facade_cases as fc


@pytest.mark.parametrize("tag,R,seed", [("default", 20, 0), ("default", 20, 3), ("standard7", 20, 1),
                                        ("default", 40, 2), ("standard7", 40, 0)])
def test_golden_trajectory(host_backend, golden_dir, tag, R, seed):
    fc.golden_trajectory("cpu", golden_dir, tag, R, seed, steps=150)


def test_dtypes_and_directions(host_backend, golden_dir):
    fc.dtypes_and_directions("cpu", golden_dir)


def test_reset_features(host_backend, golden_dir):
    fc.reset_features("cpu", golden_dir)


def test_best_policy(host_backend, golden_dir):
    fc.best_policy("cpu", golden_dir)


def test_rollouts_and_misc(host_backend):
    fc.rollouts_and_misc("cpu")


def test_rollout_script(host_backend, golden_dir):
    fc.rollout_script("cpu", golden_dir)


def test_render_strings(host_backend, golden_dir):
    fc.render_strings("cpu", golden_dir)


def test_unsupported_widths_are_refused(host_backend):
    fc.unsupported_widths_are_refused("cpu")

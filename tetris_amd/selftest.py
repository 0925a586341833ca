"""Self-test of the afterstate family (get_after_states, get_best_policy, rollouts, in-kernel greedy
steps) on THIS machine's GPU, driver and build of the library: no reference needed, a few seconds.

    python -m tetris_amd.selftest            # every board width, 32- and 64-bit boards
    python -m tetris_amd.selftest 10 20      # one geometry

The -m gpu test suite and __graft_entry__.smoke() run the same function."""
import sys

import torch


def afterstate_family_consistency(device="cuda", C=10, R=20, pieces="default", B=3 << 16, warm=30, shard=1 << 14, repeats=3,
                                 rollouts=True):
    """Every kernel of the afterstate family on a batch large enough that several workgroups share a
    compute unit: (a) repeated launches on the same state must agree bit for bit, (b) the whole-batch
    result must equal the same boards evaluated in shards of `shard` envs (one workgroup per compute
    unit), (c) two copies of the env stepping with the in-kernel greedy policy must stay identical.
    Needs no reference: it is the net for faults that depend on which waves share a SIMD -- round 3 found a gfx950
    hardware hazard of that kind (a 64-bit shift whose shift amount sits in the last allocated VGPR reads v0
    instead), and a kernel that has it fails this test on every launch.  Raises AssertionError on a mismatch."""
    from .vec_env import VecTetris
    env = VecTetris(C, R, B, device=device, pieces=pieces, auto_reset=True, seed=5)
    for t in range(warm):
        env.step()
    snap = env.state_dict()

    def greedy():
        ba, bv, fit = env.greedy_actions(include_fitness=True)
        return ba.clone(), bv.clone(), fit.clone()

    def matrix():
        f, nv, fa, na = env.get_after_states(include_terminal=True)
        return f.clone(), nv.clone(), fa.clone(), na.clone()

    g0, m0 = greedy(), matrix()
    for rep in range(repeats - 1):
        for a, b in zip(greedy(), g0):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), "get_best_policy differs between launches (rep %d)" % rep
        for a, b in zip(matrix(), m0):
            assert torch.equal(a, b), "get_after_states differs between launches (rep %d)" % rep
    if rollouts:
        r0 = env.rollouts(length=3, n=2, policy="greedy").clone()
        r1 = env.rollouts(length=3, n=2, policy="greedy")
        assert torch.equal(r0.view(torch.int64), r1.view(torch.int64)), "rollouts differ between launches"
    # (b) shards: exact copies of the state of `shard` envs (same seed, env offset and step index, so the rollouts
    # draw the same pieces), one workgroup per compute unit
    assert shard % 64 == 0
    for lo in range(0, B, shard):
        n = min(shard, B - lo)
        e2 = VecTetris(C, R, n, device=device, pieces=pieces, auto_reset=True, seed=5, env_offset=lo)
        e2.cols.copy_(env.cols[lo // 64:(lo + n + 63) // 64])
        e2.meta.copy_(env.meta[lo:lo + n])
        e2.piece.copy_(env.piece[lo:lo + n])
        e2.n_valid.copy_(env.n_valid[lo:lo + n])
        e2.step_idx = env.step_idx
        ba, bv, fit = e2.greedy_actions(include_fitness=True)
        assert torch.equal(ba, g0[0][lo:lo + n]), "best action: whole batch != shard at env %d" % lo
        assert torch.equal(fit.view(torch.int32), g0[2][lo:lo + n].view(torch.int32)), "fitness: whole batch != shard at env %d" % lo
        f, nv, fa, na = e2.get_after_states(include_terminal=True)
        assert torch.equal(fa, m0[2][lo:lo + n]) and torch.equal(f, m0[0][lo:lo + n]) and torch.equal(nv, m0[1][lo:lo + n]), \
            "get_after_states: whole batch != shard at env %d" % lo
        if rollouts:
            r2 = e2.rollouts(length=3, n=2, policy="greedy")
            assert torch.equal(r2.view(torch.int64), r0[lo:lo + n].view(torch.int64)), "rollouts: whole batch != shard at env %d" % lo
    # (c) two copies under the in-kernel greedy policy
    twin = VecTetris(C, R, B, device=device, pieces=pieces, auto_reset=True, seed=5)
    twin.load_state_dict(snap)
    env.load_state_dict(snap)
    o1 = env.step_many(6, policy="greedy")
    o2 = twin.step_many(6, policy="greedy")
    for k in ("obs", "reward", "_done", "lines", "action", "n_valid", "piece"):
        assert torch.equal(o1[k], o2[k]), "step_many(greedy): %s differs between two copies" % k
    assert torch.equal(env.cols, twin.cols) and torch.equal(env.meta, twin.meta)
    return True


GEOMETRIES = tuple((C, R, pieces) for C in range(5, 13)
                   for R, pieces in ((20, "default"), (20, "standard7"), (40, "default"), (24, "default"), (50, "standard7")))


def run(geometries=GEOMETRIES, device="cuda", verbose=True, **kw):
    """afterstate_family_consistency over `geometries`; returns the list of (geometry, message) that failed."""
    failed = []
    for C, R, pieces in geometries:
        try:
            afterstate_family_consistency(device, C=C, R=R, pieces=pieces, **kw)
            msg = "ok"
        except AssertionError as exc:
            msg = "FAILED: %s" % exc
            failed.append(((C, R, pieces), str(exc)))
        if verbose:
            print("%2d x %2d %-9s %s" % (C, R, pieces, msg), flush=True)
    return failed


if __name__ == "__main__":
    geo = GEOMETRIES
    if len(sys.argv) >= 3:
        geo = [(int(sys.argv[1]), int(sys.argv[2]), p) for p in ("default", "standard7")]
    bad = run(geo)
    print("self-test: %d of %d geometries failed" % (len(bad), len(geo)))
    sys.exit(1 if bad else 0)

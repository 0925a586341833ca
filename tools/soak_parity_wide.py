import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
os.environ["TETRIS_SOAK_PROGRESS"]="1"
import parity_cases as pc
from oracle import oracle as orc
orc.lib()
for name, kw in (("12x40, default pieces", dict(R=40, C=12, pieces="default", steps=128, every=64)),
                 ("11x40, standard-7 pieces", dict(R=40, C=11, pieces="standard7", steps=96, every=48)),
                 ("9x40, default pieces", dict(R=40, C=9, pieces="default", steps=96, every=48)),
                 ("12x24 (one plane per column), default pieces", dict(R=24, C=12, pieces="default", steps=96, every=48))):
    t0=time.perf_counter()
    n=pc.afterstate_family_full_size("cuda", orc, B=1<<20, **kw)
    print("afterstate family, %s: get_after_states (valid + include-terminal matrices) and get_best_policy of all 1,048,576 envs bit-exact at %d points of steady-state play; %.0f s" % (name, n, time.perf_counter()-t0), flush=True)

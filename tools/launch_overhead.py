import sys, time, torch
sys.path.insert(0, '.')
from tetris_amd import VecTetris
env = VecTetris(10, 20, 64, device="cuda", auto_reset=True, seed=0)
for _ in range(200): env.step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(5000): env.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("enqueue %.1f us per step() call, incl. drain %.1f us" % ((t1 - t0) / 5000 * 1e6, (t2 - t0) / 5000 * 1e6))

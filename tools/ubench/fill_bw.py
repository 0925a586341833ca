#!/usr/bin/env python3
"""Plain write / copy bandwidth of this GPU for buffers of the afterstate matrix's size (1.2 GB):
the ceiling the afterstates kernel's stores are priced against."""
import torch

n = 1048576 * 36 * 8
x = torch.empty(n, dtype=torch.float32, device="cuda")
y = torch.empty(n, dtype=torch.float32, device="cuda")


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


t = timeit(lambda: x.fill_(1.0))
print("fill  %.1f MB: %.3f ms = %.2f TB/s written" % (n * 4 / 1e6, t, n * 4 / t / 1e9))
t = timeit(lambda: y.copy_(x))
print("copy  %.1f MB: %.3f ms = %.2f TB/s read + %.2f TB/s written" % (n * 4 / 1e6, t, n * 4 / t / 1e9, n * 4 / t / 1e9))
# rows of 32 B written by different "lanes" 1152 B apart, as the env-major afterstate matrix is filled
v = x.view(1048576, 36, 8)
t = timeit(lambda: v[:, 0::2, :].fill_(2.0))
print("every other row (half the bytes): %.3f ms = %.2f TB/s written" % (t, n * 2 / t / 1e9))

#!/usr/bin/env python3
"""What does one cross-stream hand-off cost the STEPPING stream?  1,048,576 envs, 2,000 steps, one hand-off
after every step (worst case), against the plain loop:
  (a) event record (hipEventDisableTiming | hipEventReleaseToDevice) + hipStreamWaitEvent on the side stream
      (what tetris_hip_stream_link does),
  (b) hipStreamWriteValue32 on the stepping stream + hipStreamWaitValue32 on the side stream
      (memory from hipExtMallocWithFlags(hipMallocSignalMemory))."""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tetris_amd import VecTetris  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
env = VecTetris(10, 20, 1 << 20, device="cuda", auto_reset=True, seed=0)
for _ in range(200):
    env.step()
torch.cuda.synchronize()
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
ms, ss = ctypes.c_void_p(main.cuda_stream), ctypes.c_void_p(side.cuda_stream)
N = 2000


def loop(op):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(N):
        env.step()
        op(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6


def none(k):
    pass


def event(k):
    env._lib.stream_link(ms, ss)


sig = ctypes.c_void_p()
rc = hip.hipExtMallocWithFlags(ctypes.byref(sig), ctypes.c_size_t(8), ctypes.c_uint(0x2))  # hipMallocSignalMemory
print("hipExtMallocWithFlags(signal) rc", rc)
hip.hipMemset(sig, 0, ctypes.c_size_t(8))


def value(k):
    r1 = hip.hipStreamWriteValue32(ms, sig, ctypes.c_uint32(k + 1), ctypes.c_uint(0))
    r2 = hip.hipStreamWaitValue32(ss, sig, ctypes.c_uint32(k + 1), ctypes.c_uint(0), ctypes.c_uint32(0xFFFFFFFF))  # hipStreamWaitValueGte
    if (r1 or r2) and k == 0:
        print("write/wait value rc", r1, r2)


for name, op in (("plain loop", none), ("event record + wait (stream_link)", event), ("write value + wait value", value), ("plain loop", none)):
    print("%-36s %.2f us per step" % (name, loop(op)), flush=True)

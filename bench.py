#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the fused Tetris step on MI355X.

  python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over the whole batch: ONE launch of the
fused step kernel (uniform random valid action per env drawn in-kernel and written
out, decode, land, lock, clear, done, reward, BCTS observation, in-kernel
auto-reset), every output tensor written.  Workload at N = 1: BASELINE config 3 -- 1,048,576 envs, 10x20
board, default piece set; N > 1 is config 4 (weak scaling, 1,048,576 envs per
GPU, contiguous env shards, no data-path collective; RCCL only gathers the
done counters / done bitmask).

Prints ONE JSON line (rank 0).  `roofline` prices the step kernel alone against
HBM: algorithmic bytes per env-step (2*C*W + 47 = 127 B at 10x20,
207 B at 10x40) x envs per launch / HIP-event kernel time.  `cpu_baseline` times
the CPU oracle (a port, oracle/) on this host's cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides: 8.0 TB/s; ~6.3 TB/s achievable)


def algorithmic_bytes_per_env_step(C, word_bytes, with_obs=True):
    return 2 * C * word_bytes + (47 if with_obs else 15)  # 127/207 B, 95/175 B without obs


def cpu_baseline(C, R, pieces, seconds=12.0):
    """The oracle (CPU restatement, OpenMP over envs) on a bounded sample."""
    from oracle import oracle as orc
    import numpy as np
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, int(os.environ.get("TETRIS_BENCH_CPU_THREADS", "16")))  # a 1-GPU box's CPU share
    B = 2048 * cores
    env = orc.OracleVecEnv(C, R, B, pieces=pieces, auto_reset=True, seed=0, nthreads=cores)
    rng = np.random.default_rng(0)

    def one():
        a = (rng.random(B) * env.n_valid).astype(np.int32)
        env.step(a)

    for _ in range(3):
        one()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        one()
        n += 1
    dt = time.perf_counter() - t0
    return dict(value=B * n / dt, unit="env-steps/s", cores=cores, kind="port",
                sample="%d envs x %d steps (oracle/tetris_oracle.c, OpenMP over envs, auto-reset, random "
                       "valid actions)" % (B, n))


def load_traffic(columns, rows, pieces, envs):
    """HBM bytes per launch from the committed PMC profile (profiles/pmc_traffic.json, produced by
    tools/pmc_probe.py + tools/parse_pmc.py) when it was taken on this very workload, else None."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(p))
        if (d.get("columns", 10), d.get("rows", 20), d.get("pieces", "default"), d.get("envs")) == \
                (columns, rows, pieces, envs):
            return d.get("step_kernel_hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--batch", type=int, default=1 << 20, help="envs per GPU")
    ap.add_argument("--rows", type=int, default=20)
    ap.add_argument("--columns", type=int, default=10)
    ap.add_argument("--pieces", default="default")
    ap.add_argument("--gather-every", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-obs", action="store_true", help="skip the observation output (uses 95 B / 175 B per env-step)")
    ap.add_argument("--fuse", type=int, default=1,
                    help="env-steps per kernel launch (tetris_hip_step_many: boards stay in registers between the "
                         "fused steps, every step's outputs are still written); 1 = one launch per step (headline)")
    ap.add_argument("--streams", type=int, default=1,
                    help="hold the batch of one GPU as this many env shards, each stepped on its own HIP stream "
                         "(as the shards of several GPUs are): the tail of one shard's launch overlaps the ramp of "
                         "the next.  1 = one launch over the whole batch per step (headline)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("TETRIS_BENCH_BACKEND", "nccl")  # "gloo": rehearse N ranks on fewer GPUs
    dev_index = local_rank % max(1, torch.cuda.device_count()) if world > 1 else 0
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from tetris_amd import VecTetris
    from tetris_amd.distributed import DoneGather

    B = args.batch
    S = max(1, args.streams)
    if B % S:
        raise SystemExit("--batch must be a multiple of --streams")
    streams = [torch.cuda.current_stream(dev)] if S == 1 else [torch.cuda.Stream(dev) for _ in range(S)]
    envs = []
    for k in range(S):  # shard k = global envs [rank*B + k*B/S, ...): the same pieces as one big batch draws
        with torch.cuda.stream(streams[k]):
            envs.append(VecTetris(args.columns, args.rows, B // S, device=dev, pieces=args.pieces, auto_reset=True,
                                  seed=0, env_offset=rank * B + k * (B // S), compute_obs=not args.no_obs))
    env = envs[0]
    gather = DoneGather(B)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def all_totals():
        for st in streams[1:] if S > 1 else []:
            torch.cuda.current_stream(dev).wait_stream(st)
        if S > 1:
            torch.cuda.current_stream(dev).wait_stream(streams[0])
        return sum(e.totals() for e in envs)

    fuse = max(1, args.fuse)
    if args.steps % fuse or args.warmup % fuse:
        raise SystemExit("--steps and --warmup must be multiples of --fuse")
    traj = [None] * S

    def shard_step(k):
        if fuse == 1:
            envs[k].step()  # action = None: uniform random valid action drawn inside the step kernel
        else:  # `fuse` steps per launch, trajectory buffers reused
            traj[k] = envs[k].step_many(fuse, out=traj[k])

    def one_step(t):
        if S == 1:
            shard_step(0)
        else:
            for k in range(S):
                with torch.cuda.stream(streams[k]):
                    shard_step(k)
        if world > 1 and (t + 1) % max(1, args.gather_every // fuse) == 0:
            gather.gather_counters(all_totals())

    for t in range(args.warmup // fuse):
        one_step(t)
    barrier()
    t0 = time.perf_counter()
    for t in range(args.steps // fuse):
        one_step(t)
    if world > 1:
        for st in streams:
            torch.cuda.current_stream(dev).wait_stream(st)
        gather.gather_bits(env.done if S == 1 else torch.cat([e.done for e in envs]))  # the done/reset gather over RCCL
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    totals = gather.gather_counters(all_totals()).cpu().tolist()

    # step-kernel time alone: HIP events on the launch stream (torch's current stream) around
    # runs of 200 back-to-back launches; per-launch time = elapsed / launches (includes the ~1.5 us
    # inter-kernel gap, so it reads a few % above rocprofv3's kernel-only average)
    # (with --streams S > 1 the bracket sits on shard 0's stream: it reads the period at which that
    # stream's launches complete while the other shards' launches run beside them)
    n_rep, n_per = 5, 200  # (long runs: the idle-queue start-up of a bracket is amortised over 200 launches)
    k_ms = []
    for _ in range(n_rep):
        s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_ev.record(streams[0])
        for _ in range(n_per):
            one_step(-2)
        e_ev.record(streams[0])
        torch.cuda.synchronize(dev)
        k_ms.append(s_ev.elapsed_time(e_ev) / n_per)
    k_ms = sorted(k_ms)[len(k_ms) // 2]
    for e in envs:
        e.check()

    if rank == 0:
        alg = algorithmic_bytes_per_env_step(args.columns, env.desc.word_bytes, not args.no_obs)
        # S shard launches of B/S envs run beside each other during every period k_ms
        achieved = alg * B * fuse / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "env-steps/sec",
            "value": B * world * args.steps / dt,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32" if env.desc.word_bytes == 4 else "u64",
            "data": "synthetic",
            "config": {"workload": "%d envs/GPU x %d GPU, %dx%d board, pieces=%s, uniform random valid actions, "
                                   "in-kernel auto-reset, device bag seed 0" % (B, world, args.columns, args.rows,
                                                                                args.pieces),
                       "envs_per_gpu": B, "observation_output": not args.no_obs, "env_steps_per_launch": fuse,
                       "streams_per_gpu": S, "board": "%dx%d" % (args.columns, args.rows), "pieces": args.pieces,
                       "sharding": "env-index ranges, no data-path collective; RCCL gathers done counters every "
                                   "%d steps + done bitmask at the end" % args.gather_every},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None if (args.no_obs or fuse > 1 or S > 1) else load_traffic(args.columns, args.rows, args.pieces, B),
                         "kernel": "step_kernel" if fuse == 1 else "step_many_kernel (%d steps per launch)" % fuse, "kernel_ms": k_ms, "algorithmic_bytes_per_env_step": alg},
            "episodes": totals[1], "lines_cleared": totals[2],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.columns, args.rows, args.pieces, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the fused Tetris step on MI355X.

  python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over the whole batch: ONE launch of the
fused step kernel (uniform random valid action per env drawn in-kernel and written
out, decode, land, lock, clear, done, reward, BCTS observation, in-kernel
auto-reset), every output tensor written.  Workload at N = 1: BASELINE config 3 --
1,048,576 envs, 10x20 board, default piece set; N > 1 is config 4 (weak scaling,
1,048,576 envs per GPU, contiguous env shards, no data-path collective; RCCL only
gathers the done counters / done bitmask).

Ranks.  One process per GPU.  Under torchrun (WORLD_SIZE set) this process IS a rank.
Started plainly with --gpus N > 1 it is a PARENT that never touches the GPU: it
starts N fresh rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment), forwards rank 0's JSON
line and exits non-zero if any rank failed.

Prints ONE JSON line (rank 0).  `roofline` prices the step kernel alone against HBM:
algorithmic bytes per env-step = the board planes actually stored, read + written
(2 * n_planes * word: 8 planes at 10x20 / 10x40) + the 47 B of SURVEY 8(d)'s fixed
part = 111 B at 10x20, 175 B at 10x40 (SURVEY's own 127 / 207 B priced ten unpacked
column words; that figure is kept as `achieved_survey_bytes`), x envs per launch /
HIP-event kernel time.  `cpu_baseline` times the CPU oracle (a port, oracle/) on this
host's cores on a bounded sample, single-thread and all-core.
"""
import argparse
import ctypes
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s measured float4 copy)
FIXED_BYTES = 47       # action 4 + piece r/w 2 + bag r/w 2 + obs 32 + reward 4 + done/lines/n_valid 3
FIXED_BYTES_NO_OBS = 15


def algorithmic_bytes_per_env_step(n_planes, word_bytes, with_obs=True):
    """Board planes as stored (read + write) + the fixed per-env part of SURVEY section 8(d)."""
    return 2 * n_planes * word_bytes + (FIXED_BYTES if with_obs else FIXED_BYTES_NO_OBS)


def survey_bytes_per_env_step(C, word_bytes, with_obs=True):
    """SURVEY section 8(d) as written (one word per column): 127 B at 10x20, 207 B at 10x40."""
    return 2 * C * word_bytes + (FIXED_BYTES if with_obs else FIXED_BYTES_NO_OBS)


def csrc_hash():
    """Content hash of the kernel sources: a committed PMC profile is only quoted for the code it measured."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "tetris_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp", ".inc", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def load_traffic(columns, rows, pieces, envs, lib_hash=None):
    """HBM bytes per launch of the step kernel from the PMC passes of tools/profile_round.sh
    (profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, corrected on
    refresh_kernel as MI355X_MICROARCH.md's HBM section prescribes) -- only when that profile was taken
    on THIS workload and on the kernels that are running: its `csrc_hash` must equal the hash compiled
    into the LOADED library (tetris_hip_source_hash; `lib_hash`) -- a stale library next to fresh sources,
    or the reverse, yields None."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(p))
        want = lib_hash if lib_hash is not None else csrc_hash()
        if (d.get("columns", 10), d.get("rows", 20), d.get("pieces", "default"), d.get("envs")) == \
                (columns, rows, pieces, envs) and d.get("csrc_hash") == want:
            return d.get("step_kernel_hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def cpu_baseline(C, R, pieces, seconds=12.0):
    """The oracle (CPU restatement, OpenMP over envs) on a bounded sample: one thread, then all cores."""
    from oracle import oracle as orc
    import numpy as np
    visible = os.cpu_count() or 1
    try:
        visible = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(visible, int(os.environ.get("TETRIS_BENCH_CPU_THREADS", "16")))  # a 1-GPU box's CPU share is 16

    def timed(nthreads, B, budget):
        env = orc.OracleVecEnv(C, R, B, pieces=pieces, auto_reset=True, seed=0, nthreads=nthreads)
        rng = np.random.default_rng(0)

        def one():
            a = (rng.random(B) * env.n_valid).astype(np.int32)
            env.step(a)

        for _ in range(3):
            one()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < budget:
            one()
            n += 1
        return B * n / (time.perf_counter() - t0), n

    v1, n1 = timed(1, 2048, seconds / 3.0)
    vn, nn = timed(cores, 2048 * cores, seconds * 2.0 / 3.0)
    return dict(value=vn, unit="env-steps/s", cores=cores, kind="port", single_thread=v1, host_cores_visible=visible,
                sample="all-core: %d envs x %d steps on %d threads; single-thread: 2048 envs x %d steps "
                       "(oracle/tetris_oracle.c, OpenMP over envs, auto-reset, random valid actions)"
                       % (2048 * cores, nn, cores, n1),
                python_reference_quoted="~181 env-steps/s on 1 core (BASELINE.md section 2; cannot travel)")


def parse_args(argv=None):
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
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra roofline keys (4 Mi-env run, 10x40 run, copy peak)")
    ap.add_argument("--no-obs", action="store_true", help="skip the observation output (32 B per env-step less)")
    ap.add_argument("--fuse", type=int, default=1,
                    help="env-steps per kernel launch (tetris_hip_step_many: boards stay in registers between the "
                         "fused steps, every step's outputs are still written); 1 = one launch per step (headline)")
    ap.add_argument("--streams", type=int, default=1,
                    help="hold the batch of one GPU as this many env shards, each stepped on its own HIP stream "
                         "(as the shards of several GPUs are): the tail of one shard's launch overlaps the ramp of "
                         "the next.  1 = one launch over the whole batch per step (headline)")
    ap.add_argument("--graph", type=int, default=1,
                    help="capture this many steps in one HIP graph (VecTetris.capture_steps) and replay it: still "
                         "one kernel launch per step on the device, one graph launch per G steps on the host; for "
                         "small batches where the host enqueue is as long as the kernel (not combined with --fuse)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args(argv)


def _load_build_module():
    """tetris_amd/build.py by path: importing the package would pull in torch, which the parent never does."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_tetris_build", os.path.join(ROOT, "tetris_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ---- parent: start N rank processes (never touches the GPU, never imports torch) ----------------
def launch_ranks(args, entry=None, prebuild=True, deadline_s=None):
    """`entry`: the script every rank runs (default: this file; the CPU test wrapper passes itself).
    The library is built HERE, once, before any rank exists -- N ranks finding no .so would each start
    their own 7-process hipcc build of the same file."""
    n = args.gpus
    entry = entry or os.path.abspath(__file__)
    if prebuild:
        b = _load_build_module()
        if not os.path.exists(b.SO_PATH):
            b.build_hip(force=True)
    deadline_s = deadline_s or float(os.environ.get("TETRIS_BENCH_DEADLINE_S", "1500"))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # RCCL shares device buffers between the ranks of a node through dmabuf IPC handles; this pool's
        # driver supports no legacy IPC, and without the variable hipIpcGetMemHandle fails.  The image
        # exports it already (a torchrun launch inherits it the same way); the default only covers a
        # caller that scrubbed its environment.
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, entry] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # collect rank 0's stdout while watching every rank: if one dies, the others would sit in the
    # rendezvous or a collective until its timeout, so they are ended at once
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rcs = [None] * n
    failed = False
    t_start = time.monotonic()
    while any(rc is None for rc in rcs):
        for r, pr in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = pr.poll()
                if rcs[r] not in (None, 0):
                    failed = True
        if time.monotonic() - t_start > deadline_s:  # a rank hung in a collective: end the job, do not wait for RCCL's timeout
            sys.stderr.write("bench.py: ranks still running after %.0f s, ending them\n" % deadline_s)
            failed = True
        if failed:
            for r, pr in enumerate(procs):
                if rcs[r] is None:
                    pr.kill()
                    rcs[r] = pr.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write("bench.py: rank(s) failed: %s\n" % ", ".join("rank %d rc %s" % b for b in bad))
        return 1
    return 0


# ---- one rank ------------------------------------------------------------------------------------
def run_rank(args, device=None):
    """`device`: None = cuda:LOCAL_RANK (the product).  tests/bench_cpu_entry.py passes "cpu" after binding
    the package to the g++ harness build of the lane logic, to exercise this script's host code and the
    launcher without a GPU; nothing in this file selects a CPU path by itself."""
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" in os.environ and args.gpus != world:
        raise SystemExit("bench.py: --gpus %d disagrees with WORLD_SIZE=%d" % (args.gpus, world))
    backend = os.environ.get("TETRIS_BENCH_BACKEND", "nccl")  # "gloo": rehearse N ranks on fewer GPUs / on CPU
    if device is not None:
        dev = torch.device(device)
    else:
        dev_index = local_rank % max(1, torch.cuda.device_count()) if world > 1 else 0
        dev = torch.device("cuda", dev_index)
        torch.cuda.set_device(dev)
    on_gpu = dev.type == "cuda"
    # TETRIS_BENCH_FORCE_DIST=1: run the collective path with a world of one rank as well (a 1-GPU box
    # can then exercise process-group set-up and every RCCL call of the N > 1 path)
    dist_on = world > 1 or os.environ.get("TETRIS_BENCH_FORCE_DIST") == "1"
    if dist_on:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from tetris_amd import VecTetris
    from tetris_amd.distributed import DoneGather

    def sync():
        if on_gpu:
            torch.cuda.synchronize(dev)

    def cur_stream():
        return torch.cuda.current_stream(dev) if on_gpu else None

    B = args.batch
    S = max(1, args.streams) if on_gpu else 1
    if B % S:
        raise SystemExit("--batch must be a multiple of --streams")
    streams = [cur_stream()] if S == 1 else [torch.cuda.Stream(dev) for _ in range(S)]

    def on_stream(k):
        return torch.cuda.stream(streams[k]) if on_gpu else _Null()

    envs = []
    for k in range(S):  # shard k = global envs [rank*B + k*B/S, ...): the same pieces as one big batch draws
        with on_stream(k):
            envs.append(VecTetris(args.columns, args.rows, B // S, device=dev, pieces=args.pieces, auto_reset=True,
                                  seed=0, env_offset=rank * B + k * (B // S), compute_obs=not args.no_obs))
    env = envs[0]
    gather = DoneGather(B, force=dist_on)

    def barrier():
        sync()
        if dist_on:
            dist.barrier()
        sync()

    def all_totals():
        if S > 1:
            for st in streams:
                cur_stream().wait_stream(st)
        return sum(e.totals() for e in envs)

    def all_done():
        if S > 1:
            for st in streams:
                cur_stream().wait_stream(st)
        return env.done if S == 1 else torch.cat([e.done for e in envs])

    fuse = max(1, args.fuse)
    graph_steps = max(1, args.graph)
    if graph_steps > 1:
        if fuse > 1 or S > 1:
            raise SystemExit("--graph is not combined with --fuse / --streams")
        fuse = graph_steps  # (host-side bookkeeping below: `fuse` steps per call of shard_step)
    if args.steps % fuse or args.warmup % fuse:
        raise SystemExit("--steps and --warmup must be multiples of --fuse / --graph")
    traj = [None] * S
    graphs = [e.capture_steps(graph_steps) for e in envs] if graph_steps > 1 else None

    def shard_step(k):
        if graphs is not None:
            graphs[k].replay()
        elif fuse == 1:
            envs[k].step()  # action = None: uniform random valid action drawn inside the step kernel
        else:  # `fuse` steps per launch, trajectory buffers reused
            traj[k] = envs[k].step_many(fuse, out=traj[k])

    n_gathers = [0]
    n_bit_gathers = [0]
    # The gathers.  The payload -- done bitmask, per-wave counter slots -- is written by the step kernel
    # itself (tetris_hip_step_call_run_gather) into one of two buffers nothing else touches, the side
    # stream is linked behind that step with a device-scope event (tetris_hip_stream_link) and runs the
    # collective there: the stepping stream enqueues nothing for a gather and never waits for one -- the
    # exchange is off the step's critical path, as section 8(e) of the survey describes it.  (Round 2
    # packed the flags with an extra kernel on the stepping stream and recorded a default, system-scope
    # event: ~35 us of idle stepping stream per gather.)  With --fuse / --graph / --streams the payload
    # comes from the separate pack kernel and a column sum instead.
    gather_mode = os.environ.get("TETRIS_BENCH_GATHER_MODE", "side")  # experiments: "same" (collective on the stepping stream), "linkonly"
    side = torch.cuda.Stream(dev) if (dist_on and on_gpu and backend == "nccl" and gather_mode != "same") else None
    in_kernel_payload = fuse == 1 and graphs is None and S == 1
    payloads = [env.gather_payload() for _ in range(2)] if (dist_on and in_kernel_payload) else None
    payload_free = [None, None]  # event after the last collective that read buffer i
    n_payload = [0]

    def next_payload():
        i = n_payload[0] & 1
        n_payload[0] += 1
        if payload_free[i] is not None:
            payload_free[i].synchronize()  # (two gathers ago: long done)
        return i, payloads[i]

    def run_off_path(i, exchange):
        if side is None:
            exchange()
            return
        from tetris_amd import _lib
        lib = _lib.load()
        lib.check(lib.stream_link(ctypes.c_void_p(cur_stream().cuda_stream), ctypes.c_void_p(side.cuda_stream)),
                  "tetris_hip_stream_link")
        with torch.cuda.stream(side):
            if gather_mode != "linkonly":
                exchange()
            if i is not None:
                payload_free[i] = torch.cuda.Event()
                payload_free[i].record(side)

    def off_path(read, exchange):  # fallback: payload produced by extra launches on the stepping stream
        payload = read()
        if side is not None:
            payload.record_stream(side)
        run_off_path(None, lambda: exchange(payload))

    def gather_counters_now():
        off_path(all_totals, gather.gather_counters)

    def gather_bits_now():
        from tetris_amd.distributed import pack_done_bits
        off_path(lambda: pack_done_bits(all_done()), gather.gather_packed)
        n_bit_gathers[0] += 1

    def one_step(t, bits=False):
        counters = dist_on and t >= 0 and (t + 1) % max(1, args.gather_every // fuse) == 0
        if payloads is not None and (bits or counters):
            i, pl = next_payload()
            envs[0].step(gather=pl)  # the step writes the payload from its epilogue

            def exchange():
                if bits:
                    gather.gather_packed(pl["done_bits"][:(B + 7) // 8])
                if counters:
                    gather.gather_counters((pl["counters"].to(torch.int64) & 0xFFFFFFFF).sum(dim=0))
            run_off_path(i, exchange)
            n_gathers[0] += int(counters)
            n_bit_gathers[0] += int(bits)
            return
        if S == 1:
            shard_step(0)
        else:
            for k in range(S):
                with on_stream(k):
                    shard_step(k)
        if counters:
            gather_counters_now()
            n_gathers[0] += 1
        if bits:
            gather_bits_now()

    for t in range(args.warmup // fuse):
        one_step(t)
    if dist_on:  # every collective of the timed region once before it: RCCL sets up a kind of call at its first use
        one_step(max(1, args.gather_every // fuse) - 1, bits=True)
    n_gathers[0] = 0
    n_bit_gathers[0] = 0
    barrier()
    t0 = time.perf_counter()
    n_calls = args.steps // fuse
    for t in range(n_calls):
        # the done/reset gather over RCCL once, midway: it overlaps the remaining steps
        one_step(t, bits=dist_on and t == (n_calls - 1) // 2)
    sync()  # (device-wide: the side stream's collectives included; no cross-stream wait on the GPU)
    dt = time.perf_counter() - t0   # this rank's K steps + its gathers; the MAX over ranks is taken below
    barrier()
    if dist_on:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    totals = gather.gather_counters(all_totals()).cpu().tolist()

    # what the collective layer itself saw: its world size and, per rank, the env range it stepped and the
    # device it ran on (PCI bus id / uuid where torch exposes them) -- so that "N ranks on N devices" can
    # be read off the line
    me = {"rank": rank, "envs": [rank * B, (rank + 1) * B], "device": str(dev), "pid": os.getpid()}
    if on_gpu:
        pr = torch.cuda.get_device_properties(dev)
        me["gpu"] = {"name": pr.name, "pci_bus_id": getattr(pr, "pci_bus_id", None),
                     "pci_device_id": getattr(pr, "pci_device_id", None), "uuid": str(getattr(pr, "uuid", "")) or None}
    if dist_on:
        per_rank = [None] * dist.get_world_size()
        dist.all_gather_object(per_rank, me)
        ranks_info = {"world_size_observed": dist.get_world_size(), "backend_observed": dist.get_backend(),
                      "per_rank": per_rank}
    else:
        ranks_info = {"world_size_observed": 1, "backend_observed": None, "per_rank": [me]}

    # step-kernel time alone: HIP events on the launch stream (torch's current stream) around runs
    # of back-to-back launches; per-launch time = elapsed / launches (includes the ~1.5 us
    # inter-kernel gap, so it reads a few % above rocprofv3's kernel-only average).  With
    # --streams S > 1 the bracket sits on shard 0's stream: it reads the period at which that
    # stream's launches complete while the other shards' launches run beside them.
    def kernel_ms_of(step_fn, stream, n_rep=5, n_per=200):
        if not on_gpu:
            t = time.perf_counter()
            for _ in range(4):
                step_fn()
            return (time.perf_counter() - t) / 4 * 1e3
        ms = []
        for _ in range(n_rep):
            s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_ev.record(stream)
            for _ in range(n_per):
                step_fn()
            e_ev.record(stream)
            torch.cuda.synchronize(dev)
            ms.append(s_ev.elapsed_time(e_ev) / n_per)
        return sorted(ms)[len(ms) // 2]

    k_ms = kernel_ms_of(lambda: one_step(-2), streams[0])
    if graph_steps > 1:
        k_ms /= graph_steps  # one replay = graph_steps launches of the step kernel
    for e in envs:
        e.check()

    # the done/reset gather on its own (it sits outside the kernel bracket above): bitmask
    # all-gather + counter all-reduce, host-paired wall time per call
    gather_ms = None
    k_all = [k_ms]
    if dist_on:
        barrier()
        g0 = time.perf_counter()
        for _ in range(10):
            gather.gather_bits(all_done())
            gather.gather_counters(all_totals())
        sync()
        gather_ms = (time.perf_counter() - g0) / 10 * 1e3
        kt = torch.tensor([k_ms], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        parts = [torch.zeros_like(kt) for _ in range(world)]
        dist.all_gather(parts, kt)
        k_all = [float(x.item()) for x in parts]

    extras = {}
    if rank == 0 and world == 1 and on_gpu and not args.no_extras and fuse == 1 and S == 1 and not args.no_obs:
        # (a) the same kernel with a working set beyond the 256 MiB Infinity Cache (4 Mi envs, ~0.5 GB per step)
        big = VecTetris(args.columns, args.rows, 4 * B, device=dev, pieces=args.pieces, auto_reset=True, seed=0)
        for _ in range(args.warmup):
            big.step()
        extras["kernel_ms_4Mi"] = kernel_ms_of(big.step, cur_stream(), n_rep=3, n_per=50)
        del big
        # (b) config 5 shape: 10x40 boards (u64 columns), same batch
        tall = VecTetris(args.columns, 40, B, device=dev, pieces=args.pieces, auto_reset=True, seed=0)
        for _ in range(args.warmup):
            tall.step()
        tall_ms = kernel_ms_of(tall.step, cur_stream(), n_rep=3, n_per=100)
        tall_alg = algorithmic_bytes_per_env_step(tall.n_planes, tall.desc.word_bytes)
        extras["tall_10x40"] = {"kernel_ms": tall_ms, "env_steps_per_s": B / (tall_ms * 1e-3), "dtype": "u64",
                                "algorithmic_bytes_per_env_step": tall_alg,
                                "frac": tall_alg * B / (tall_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        del tall
        # (c) what this box's HBM delivers to a plain copy (1 GiB read + 1 GiB written per pass)
        src = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        dst = torch.empty_like(src)
        src.fill_(1.0)
        cp_ms = kernel_ms_of(lambda: dst.copy_(src), cur_stream(), n_rep=3, n_per=10)
        extras["peak_copy_measured"] = 2.0 * src.numel() * 4 / (cp_ms * 1e-3) / 1e9
        del src, dst
        torch.cuda.empty_cache()

    if rank == 0:
        wb = env.desc.word_bytes
        alg = algorithmic_bytes_per_env_step(env.n_planes, wb, not args.no_obs)
        alg_survey = survey_bytes_per_env_step(args.columns, wb, not args.no_obs)
        # S shard launches of B/S envs run beside each other during every period k_ms
        per_launch = 1 if graph_steps > 1 else fuse  # env-steps of every env per timed kernel launch
        achieved = alg * B * per_launch / (k_ms * 1e-3) / 1e9
        lib_hash = env._lib.source_hash()
        traffic = None if (args.no_obs or fuse > 1 or S > 1) else load_traffic(args.columns, args.rows, args.pieces, B,
                                                                             lib_hash)
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": "step_kernel" if fuse == 1 else (
                    "step_kernel (HIP graph of %d launches)" % fuse if graph_steps > 1
                    else "step_many_kernel (%d steps per launch)" % fuse),
                "kernel_ms": k_ms, "algorithmic_bytes_per_env_step": alg,
                "survey_bytes_per_env_step": alg_survey,
                "achieved_survey_bytes": alg_survey * B * per_launch / (k_ms * 1e-3) / 1e9,
                "peak_spec": HBM_PEAK_GBS,
                "note": "working set of one launch (~%d MB) fits the 256 MiB Infinity Cache, whose hits FETCH_SIZE / "
                        "WRITE_SIZE count: see frac_4Mi for the same kernel streaming from HBM" % (alg * B // 1000000)}
        if traffic is not None:
            roof["achieved_measured"] = traffic / (k_ms * 1e-3) / 1e9
            roof["traffic_source"] = "profiles/pmc_traffic.json (rocprofv3 --pmc, csrc_hash %s)" % lib_hash
        roof["library_source_hash"] = lib_hash
        if dist_on:
            roof["kernel_ms_per_rank"] = {"min": min(k_all), "max": max(k_all)}
        if "kernel_ms_4Mi" in extras:
            roof["kernel_ms_4Mi"] = extras["kernel_ms_4Mi"]
            roof["frac_4Mi"] = alg * 4 * B / (extras["kernel_ms_4Mi"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        if "peak_copy_measured" in extras:
            roof["peak_copy_measured"] = extras["peak_copy_measured"]
            roof["peak_copy_how"] = "torch copy_ of 1 GiB (read + written bytes / time), same run, same device"
            roof["frac_of_copy_peak"] = achieved / extras["peak_copy_measured"]
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
            "dtype": "u32" if wb == 4 else "u64",
            "data": "synthetic",
            "config": {"workload": "%d envs/GPU x %d GPU, %dx%d board, pieces=%s, uniform random valid actions, "
                                   "in-kernel auto-reset, device bag seed 0" % (B, world, args.columns, args.rows,
                                                                                args.pieces),
                       "envs_per_gpu": B, "observation_output": not args.no_obs,
                       "env_steps_per_launch": 1 if graph_steps > 1 else fuse, "steps_per_graph_replay": graph_steps,
                       "streams_per_gpu": S, "board": "%dx%d" % (args.columns, args.rows), "pieces": args.pieces,
                       "backend": backend if dist_on else "single", "device": dev.type,
                       "sharding": "env-index ranges, no data-path collective; RCCL gathers the done counters every "
                                   "%d steps and the done bitmask once, midway through the timed region, both on "
                                   "a side stream" % args.gather_every},
            "roofline": roof,
            "ranks": ranks_info,
            "episodes": totals[1], "lines_cleared": totals[2],
        }
        if dist_on:
            out["done_gather"] = {"ms_per_gather": gather_ms, "counter_gathers_in_timed_region": n_gathers[0],
                                  "bitmask_gathers_in_timed_region": n_bit_gathers[0],
                                  "payload": "step kernel epilogue" if payloads is not None else "pack kernel + column sum",
                                  "note": "cpu_baseline and the extra roofline keys are emitted only at world == 1"}
        if "tall_10x40" in extras:
            out["tall_10x40"] = extras["tall_10x40"]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.columns, args.rows, args.pieces, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.destroy_process_group()
    return 0


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def main(device=None, entry=None, prebuild=True):
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, entry=entry, prebuild=prebuild)
    return run_rank(args, device=device)


if __name__ == "__main__":
    sys.exit(main())

"""Multi-GPU layout: one process per GPU, contiguous env-index shards, no
data-path collective.  The only exchange is the done/reset gather used for
episode statistics and host-driven resets.

Backend "nccl" is RCCL on ROCm; "gloo" serves the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_range(total_envs, rank, world_size):
    """Contiguous slice [lo, hi) of the global env index space owned by `rank`."""
    if total_envs % world_size:
        raise ValueError("total_envs must divide evenly over the ranks")
    per = total_envs // world_size
    return rank * per, (rank + 1) * per


def pack_done_bits(done):
    """bool/uint8 [B] -> uint8 [ceil(B/8)] bitmask (bit i%8 of byte i//8).  On the device the library packs
    it with one ballot per wavefront (tetris_hip_pack_done_bits); host tensors go through torch ops."""
    from . import _lib
    lib = _lib._BINDING
    if lib is not None and done.device.type == lib.device_type and hasattr(lib, "pack_done_bits"):
        import ctypes
        d = done.view(torch.uint8) if done.dtype == torch.bool else done.to(torch.uint8)
        d = d.contiguous().flatten()
        n = d.numel()
        out = torch.empty(((n + 63) // 64) * 8, dtype=torch.uint8, device=d.device)
        stream = None
        if d.device.type == "cuda":
            stream = ctypes.c_void_p(torch.cuda.current_stream(d.device).cuda_stream)
        lib.check(lib.pack_done_bits(ctypes.c_void_p(d.data_ptr()), ctypes.c_void_p(out.data_ptr()), n, stream),
                  "tetris_hip_pack_done_bits")
        return out[:(n + 7) // 8]
    d = done.to(torch.uint8).flatten()
    pad = (-d.numel()) % 8
    if pad:
        d = torch.cat([d, d.new_zeros(pad)])
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=d.device)
    return (d.view(-1, 8) * w).sum(dim=1).to(torch.uint8)


def unpack_done_bits(bits, n):
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=bits.device)
    return ((bits.view(-1, 1) & w) != 0).flatten()[:n]


class DoneGather:
    """All-gather of per-rank done flags as bitmasks (B_local/8 bytes per rank:
    128 KiB at 1 Mi envs -- latency-bound, kept off the step's critical path by
    calling it every K steps or with async_op)."""

    def __init__(self, local_envs, group=None, force=False):
        """force: run the collectives even in a world of one rank (exercises the RCCL calls on a 1-GPU box)."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.local_envs = int(local_envs)
        self._out = None
        # gloo (CPU tests, or rehearsing ranks that share one GPU) moves device tensors through the host
        self._via_cpu = dist.is_initialized() and dist.get_backend(group) == "gloo"
        self._collective = dist.is_initialized() and (self.world > 1 or force)

    def gather_bits(self, done, async_op=False):
        return self.gather_packed(pack_done_bits(done), async_op)

    def gather_packed(self, bits, async_op=False):
        """All-gather of already packed done bitmasks (uint8 [ceil(B/8)] per rank) -> [world, ceil(B/8)]."""
        if not self._collective:
            return bits.unsqueeze(0), None
        if self._via_cpu:
            parts = [torch.empty(bits.numel(), dtype=torch.uint8) for _ in range(self.world)]
            dist.all_gather(parts, bits.cpu(), group=self.group)
            return torch.stack(parts).to(bits.device), None
        if self._out is None or self._out.device != bits.device or self._out.shape[1] != bits.numel():
            self._out = torch.empty((self.world, bits.numel()), dtype=torch.uint8, device=bits.device)
        work = dist.all_gather_into_tensor(self._out.view(-1), bits.contiguous(), group=self.group, async_op=async_op)
        return self._out, work

    def gather_indices(self, done):
        """Global env indices (int64, sorted) of every finished env on every rank."""
        out, _ = self.gather_bits(done)
        flags = torch.stack([unpack_done_bits(out[r], self.local_envs) for r in range(out.shape[0])])
        return torch.nonzero(flags.flatten(), as_tuple=False).flatten()

    def gather_counters(self, totals):
        """Sum of the env's counters (VecTetris.totals(), int64 [4]) over all ranks."""
        s = totals.clone()
        if self._collective:
            if self._via_cpu:
                c = s.cpu()
                dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group)
                return c.to(s.device)
            dist.all_reduce(s, op=dist.ReduceOp.SUM, group=self.group)
        return s

"""Sharding of the visibility pass across the GPUs of one node: one process per GPU, contiguous command
ranges per rank, no exchange of meshlet data; the only collective is one all-reduce (sum) of the visible counts per
phase — RCCL over xGMI when the process group is "nccl", gloo in the CPU tests.

The reference has no multi-GPU path (one VkPhysicalDevice, src/device.cpp:190-248); this is the new part.
"""
import numpy as np

from . import host


def command_range(command_count, rank, world):
    """contiguous [begin, end) of task commands owned by `rank` (nv_shard_range)"""
    return host.shard_range(command_count, rank, world)


def local_commands(commands, begin, end):
    """the rank's slice, padded with zeroed dummy commands to a multiple of 64 like tasksubmit does"""
    n = end - begin
    out = np.zeros((n + 63) // 64 * 64, dtype=commands.dtype)
    out[:n] = commands[begin:end]
    return out, n


def to_global_ids(local_ids, command_base):
    """clusterIndices entries are commandId (24 bits) | lane << 24 (clustercull.comp.glsl:138); rebase the rank-local
    command id by the first command the rank owns"""
    local_ids = np.asarray(local_ids, dtype=np.uint32)
    pad = local_ids == np.uint32(0xffffffff)  # clustersubmit's padding entries (clustersubmit.comp.glsl:41-44) stay ~0
    cmd = (local_ids & np.uint32(0xffffff)).astype(np.uint64) + np.uint64(command_base)
    if cmd.size and int(cmd[~pad].max(initial=0)) >= 1 << 24:
        raise ValueError("global command id does not fit the 24-bit field of a cluster index")
    out = cmd.astype(np.uint32) | (local_ids & np.uint32(0xff000000))
    out[pad] = np.uint32(0xffffffff)
    return out


class CountsReducer:
    """The passes' counts {0, task commands, visible meshlets} summed over the ranks — the one collective of the sharded
    path.  Nothing on the data path waits for it, so it is kept off the critical path twice over:
    the scatter launch writes the payload itself (nv_set_counts_sink: no extra launch per pass) into row i % B of a
    [B, 3] int64 block, and the block is reduced B passes at a time with ONE asynchronous all-reduce on the collective's
    own stream, waited for only when the block comes round again (two blocks alternate).  B = 1: one collective per pass.

        red = CountsReducer(ctx, device, batch)
        for i in range(steps): red.before_pass(i); ctx.clustercull(...); red.after_pass(i)
        red.drain(steps); total = red.last(steps)          # int64[3], summed over the ranks (world size 1: the pass's own counts)

    After a block's in-place all-reduce its rows hold SUMS; a pass overwrites its own row before the block is reduced again, and
    drain() zeroes the rows a partial last batch did not write before it reduces that block.  So: call drain(n) before last(n),
    and do not read a block between before_pass and drain — rows not yet rewritten still hold the previous use's sums.
    """

    def __init__(self, ctx, device, batch=8, stream=None, force_collective=False):
        import contextlib
        import torch
        import torch.distributed as dist
        self.ctx, self.dist, self.B = ctx, dist, max(1, int(batch))
        # the stream the context's passes are launched on (None: the current one): the collective is ordered behind it and
        # waits are issued on it
        self._on = (lambda: torch.cuda.stream(stream)) if stream is not None else contextlib.nullcontext
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # force_collective (bench.py --force-sharded): issue the all-reduces also in a process group of ONE rank — the sharded path
        # (asynchronous batched all-reduce on device tensors, its stream ordering, the waits) executed over RCCL on a one-GPU box
        self.collective = self.world > 1 or (bool(force_collective) and dist.is_available() and dist.is_initialized())
        self.blocks = [torch.zeros((self.B, 3), dtype=torch.int64, device=device) for _ in range(2)]
        self.pending = [None, None]
        # the rows' addresses, marshalled once: before_pass runs once per pass, and a pass is ~30 us
        import ctypes
        from ._lib import lib
        self._sink = lib.nv_set_counts_sink
        self._rows = [[ctypes.c_void_p(self.blocks[k][r].data_ptr()) for r in range(self.B)] for k in range(2)]

    def before_pass(self, i):
        blk, row = (i // self.B) % 2, i % self.B
        if not self.collective:  # nothing to sum: the scatter launch still leaves the pass's counts in its row, so last() holds for any world size
            self._sink(self.ctx.h, self._rows[blk][row])
            return
        if row == 0:
            with self._on():
                if self.pending[blk] is not None:
                    self.pending[blk].wait()  # the block's previous reduction (issued 2 B passes ago)
                    self.pending[blk] = None
        self._sink(self.ctx.h, self._rows[blk][row])

    def after_pass(self, i):
        if self.collective and i % self.B == self.B - 1:
            blk = (i // self.B) % 2
            with self._on():
                self.pending[blk] = self.dist.all_reduce(self.blocks[blk], async_op=True)

    def drain(self, n_steps):
        """reduces the rows of a batch the loop left unfinished, then waits for everything in flight"""
        if not self.collective:
            return
        with self._on():
            if n_steps % self.B:
                blk = (n_steps // self.B) % 2
                self.blocks[blk][n_steps % self.B:].zero_()  # rows this partial batch did not write still hold an earlier use's sums
                self.pending[blk] = self.dist.all_reduce(self.blocks[blk], async_op=True)
            for k in range(2):
                if self.pending[k] is not None:
                    self.pending[k].wait()
                    self.pending[k] = None

    def last(self, n_steps):
        """the summed counts of pass n_steps - 1 (after drain)"""
        return self.blocks[((n_steps - 1) // self.B) % 2][(n_steps - 1) % self.B].clone()


def allreduce_counts(counts):
    """counts: int64 tensor {commands, task groups, visible meshlets}; summed in place over all ranks"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts

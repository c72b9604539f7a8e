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
    cmd = (local_ids & np.uint32(0xffffff)) + np.uint32(command_base)
    if cmd.size and int(cmd.max()) >= 1 << 24:
        raise ValueError("global command id does not fit the 24-bit field of a cluster index")
    return cmd | (local_ids & np.uint32(0xff000000))


def allreduce_counts(counts):
    """counts: int64 tensor {commands, task groups, visible meshlets}; summed in place over all ranks"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts

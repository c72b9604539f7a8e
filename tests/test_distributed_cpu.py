"""N > 1 path on CPU: world_size 2, gloo.  Each rank culls its contiguous command range (the CPU oracle stands in for
the device, which is absent here), rebases its IDs, and the one all-reduce of the counts runs through
torch.distributed exactly as bench.py does with RCCL.  Concatenating the ranks' lists must give the unsharded list."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from niagara_amd import host, shard, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    draws, meshlets, commands, n = synth.cluster_scene(600, 7, seed=5)
    cd = host.build_cull_data(draw_count=len(draws), cullingEnabled=1, clusterBackfaceEnabled=1)
    b, e = shard.command_range(n, rank, world)
    local, ln = shard.local_commands(commands, b, e)
    cib, cc4 = np.zeros(len(local) * 64 + 256, np.uint32), np.zeros(4, np.uint32)
    oracle.clustercull(cd, 0, local, synth.count4_for(ln), draws, meshlets, None, None, cib, cc4)
    ids = shard.to_global_ids(cib[:cc4[0]], b)
    counts = torch.tensor([ln, (ln + 63) // 64, int(cc4[0])], dtype=torch.int64)
    shard.allreduce_counts(counts)
    np.save(os.path.join(out_dir, "ids_%d.npy" % rank), ids)
    np.save(os.path.join(out_dir, "counts_%d.npy" % rank), counts.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_concatenate_to_the_unsharded_list(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    draws, meshlets, commands, n = synth.cluster_scene(600, 7, seed=5)
    cd = host.build_cull_data(draw_count=len(draws), cullingEnabled=1, clusterBackfaceEnabled=1)
    cib, cc4 = np.zeros(len(commands) * 64 + 256, np.uint32), np.zeros(4, np.uint32)
    oracle.clustercull(cd, 0, commands, synth.count4_for(n), draws, meshlets, None, None, cib, cc4)
    ids = np.concatenate([np.load(tmp_path / ("ids_%d.npy" % r)) for r in range(world)])
    assert cc4[0] > 0 and (ids == cib[:cc4[0]]).all()
    for r in range(world):
        c = np.load(tmp_path / ("counts_%d.npy" % r))
        assert c[0] == n and c[2] == cc4[0]  # every rank holds the global sums after the all-reduce


def test_shard_ranges_cover_without_overlap():
    for total in (0, 1, 7, 64, 156250, 1562500):
        for world in (1, 2, 3, 4, 8):
            prev = 0
            for r in range(world):
                b, e = shard.command_range(total, r, world)
                assert b == prev and e >= b
                prev = e
            assert prev == total
            sizes = [shard.command_range(total, r, world)[1] - shard.command_range(total, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_global_id_field_overflow_is_refused():
    with pytest.raises(ValueError):
        shard.to_global_ids(np.array([5], np.uint32), (1 << 24) - 2)


def test_padding_entries_are_not_rebased():
    """clustersubmit pads the list with ~0 up to a multiple of 256 (clustersubmit.comp.glsl:41-44); rebasing must
    leave those entries alone instead of overflowing the 24-bit command field"""
    ids = np.array([3 | (5 << 24), 0xffffffff, 70 | (63 << 24), 0xffffffff], np.uint32)
    out = shard.to_global_ids(ids, 1000)
    assert out.tolist() == [1003 | (5 << 24), 0xffffffff, 1070 | (63 << 24), 0xffffffff]
    assert shard.to_global_ids(np.zeros(0, np.uint32), 5).size == 0


def test_bench_self_launch_command_is_the_contracts_launcher_line():
    """`python3 bench.py --gpus N ...` with no launcher in front starts its own ranks — the task contract's launcher line
    (torch.distributed.run, one node, N ranks, rendezvous on 127.0.0.1) with the caller's arguments unchanged"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    argv = ["--gpus", "4", "--steps", "20", "--warmup", "5"]
    cmd = bench.self_launch_command(4, argv, port=29517)
    assert cmd == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1", "--master-port", "29517",
                   os.path.join(root, "bench.py")] + argv
    assert bench.self_launch_command(2, [])[-2] != str(29517)  # (a free port when none is given)


@pytest.mark.skipif(torch.cuda.is_available(), reason="the failure path: needs a host WITHOUT a GPU")
def test_bench_self_launch_runs_ranks_and_propagates_their_exit_status():
    """without a launcher and with --gpus 2 the process re-executes itself under torch.distributed.run; here (no GPU) both ranks fail at
    cuda.set_device — the parent must report THEIR failure (non-zero exit, the ranks' traceback on stderr), not the old argument error"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--shared-device", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert out.returncode != 0
    assert "but WORLD_SIZE=1" not in out.stderr and "--gpus 2 but" not in out.stderr
    assert "torch.distributed" in out.stderr or "ChildFailedError" in out.stderr or "local_rank" in out.stderr  # the launcher's report of its failed ranks

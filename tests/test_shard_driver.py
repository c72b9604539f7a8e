"""examples/shard_driver.cpp — the C++ / RCCL sharded host (`ncclCommInitAll`, one stream + nv_context per device, one
`ncclAllReduce(ncclSum)` of 3 x u64 per phase) — against the oracle: the devices' rebased ID lists concatenate to the unsharded
oracle list of the pool, every rank holds the global sums.  The GPU box has one device, so the communicator has one rank there
(the RCCL calls are the same ones an 8-GPU node makes); the sharding arithmetic for N > 1 is covered on the CPU below and by
tests/test_distributed_*.py."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from niagara_amd import layouts as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "examples", "shard_driver")


def test_driver_is_built_and_prints_usage():
    assert os.path.exists(DRIVER), "examples/shard_driver is missing: run __graft_entry__.build() (make -C examples)"
    p = subprocess.run([DRIVER, "--help"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "usage" in p.stderr and "--total-meshlets" in p.stderr


def read_dump(prefix):
    blob = open(prefix + ".scene", "rb").read()
    magic, n_dev, total_draws, cpd = struct.unpack_from("<4I", blob, 0)
    assert magic == 0x4853564E
    pos = 16
    draws = np.frombuffer(blob, dtype=L.MESHDRAW, count=total_draws, offset=pos).copy()
    pos += total_draws * L.MESHDRAW.itemsize
    devs = []
    for _ in range(n_dev):
        b, e = struct.unpack_from("<2Q", blob, pos)
        d0, dn = struct.unpack_from("<2I", blob, pos + 16)
        pos += 24
        cd = np.frombuffer(blob, dtype=L.CULLDATA, count=1, offset=pos).copy()
        pos += L.CULLDATA.itemsize
        nc, = struct.unpack_from("<I", blob, pos)
        pos += 4
        commands = np.frombuffer(blob, dtype=L.TASKCMD, count=nc, offset=pos).copy()
        pos += nc * L.TASKCMD.itemsize
        nm, = struct.unpack_from("<Q", blob, pos)
        pos += 8
        meshlets = np.frombuffer(blob, dtype=L.MESHLET, count=nm, offset=pos).copy()
        pos += nm * L.MESHLET.itemsize
        devs.append(dict(b=b, e=e, d0=d0, dn=dn, cd=cd, commands=commands, meshlets=meshlets))
    assert pos == len(blob)
    out = open(prefix + ".out", "rb").read()
    pos = 0
    for d in devs:
        d["local"] = struct.unpack_from("<3Q", out, pos)
        d["reduced"] = struct.unpack_from("<3Q", out, pos + 24)
        nv, = struct.unpack_from("<I", out, pos + 48)
        d["ids"] = np.frombuffer(out, dtype=np.uint32, count=nv, offset=pos + 52).copy()
        pos += 52 + 4 * nv
    assert pos == len(out)
    return draws, cpd, devs


@pytest.mark.gpu
@pytest.mark.parametrize("total, shards", [(0, 0), (1_000_000 + 64 * 3, 0), (3_000_000 + 64 * 5, 8)])
def test_cpp_rccl_host_matches_the_oracle(tmp_path, total, shards):
    """shards = 8: eight ranks on the box's one device (--shards; RCCL refuses a device twice in one communicator, so the ranks sharing a
    device are summed on the host into the device's communicator rank and the SAME grouped ncclAllReduce runs as with one rank per device)
    — config 5's sharding, passes and ID rebasing in the C++ host, with shard boundaries inside draws"""
    import oracle
    from niagara_amd import synth
    prefix = str(tmp_path / "dump")
    cmd = ([DRIVER, "--steps", "6", "--warmup", "2", "--dump", prefix] + (["--total-meshlets", str(total)] if total else ["--draws", "1700", "--commands-per-draw", "7"]) +
           (["--shards", str(shards)] if shards else []))
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert line["counts_agree_on_all_ranks"] is True and line["devices"] >= 1 and line["scaling"] == ("strong" if total else "weak")
    draws, cpd, devs = read_dump(prefix)
    assert len(devs) == line["shards"] == (max(shards, line["devices"]) if shards else line["devices"])
    # the grouped ncclAllReduce runs in both modes (one communicator rank per device); ranks sharing a device are pre-summed on the host
    assert "ncclAllReduce" in line["allreduce"] and ("summed on the host first" in line["allreduce"]) == (line["shards"] > line["devices"])
    assert line["timing"].startswith("host-synchronised" if line["shards"] > line["devices"] else "asynchronous")
    got_ids, want_ids, visible = [], [], 0
    for d in devs:
        n = d["e"] - d["b"]
        local_draws = draws[d["d0"]:d["d0"] + d["dn"]]
        cib, cc4 = np.zeros(max(1, n * 64), np.uint32), np.zeros(4, np.uint32)
        oracle.clustercull(d["cd"], 0, d["commands"], synth.count4_for(n), local_draws, d["meshlets"], None, None, cib, cc4, threads=oracle.max_threads())
        k = int(cc4[0])
        assert d["local"] == (0, n, k) and len(d["ids"]) == k
        want = cib[:k]
        want_ids.append(((want & 0xffffff) + np.uint32(d["b"])) | (want & 0xff000000))
        got_ids.append(d["ids"])
        visible += k
    assert visible > 1000 and (np.concatenate(got_ids) == np.concatenate(want_ids)).all()
    total_cmd = sum(d["e"] - d["b"] for d in devs)
    assert line["commands_total"] == total_cmd and line["visible_total"] == visible
    for d in devs:
        assert d["reduced"] == (0, total_cmd, visible)


@pytest.mark.gpu
def test_config5_eight_shards_from_the_cpp_host():
    """BASELINE config 5's shape through the C++ host: 100 M meshlets in eight shards of 12.5 M (on the box's one device; an 8-GPU node
    runs the same command without --shards and reduces over RCCL).  The pool is too large to dump: the counts must agree on all ranks,
    cover every command, and see the share of visible meshlets the 3 M-meshlet run of the test above holds against the oracle."""
    p = subprocess.run([DRIVER, "--shards", "8", "--total-meshlets", "100000000", "--steps", "3", "--warmup", "1"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert line["shards"] == 8 and line["meshlets_total"] == 100_000_000 and line["commands_total"] == 1_562_500 and line["counts_agree_on_all_ranks"] is True
    assert 0.01 < line["visible_total"] / line["meshlets_total"] < 0.06

"""Multi-GPU path on CPU: world_size-2 `gloo` run of the capture sharding + result gather (sora_amd/shard.py).
Each rank decodes ITS block of captures (with the oracle standing in for the GPU path -- the point here is the
partition / all-gather logic, which is identical under RCCL) and every rank must end up with the full, ordered result set."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_is_a_cover():
    from sora_amd.shard import partition
    for n in (0, 1, 7, 8, 256, 257):
        for w in (1, 2, 3, 8):
            blocks = [partition(n, w, r) for r in range(w)]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == n
            for (a, ca), (b, _) in zip(blocks, blocks[1:]):
                assert a + ca == b
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1


def test_row_codec_roundtrip():
    from sora_amd.shard import results_from_rows, rows_from_results
    rs = [{"capture_id": 5, "start_sample": 176, "end_sample": 4880, "error_code": 1, "rate_kbps": 54000, "length": 1500,
           "nsym": 56, "crc32": 0xDEADBEEF, "cfo_est": -37, "flags": 0, "mpdu_offset": 1234},
          {"capture_id": 6, "start_sample": 0, "end_sample": 9, "error_code": 0x80000005, "rate_kbps": 0, "length": 0,
           "nsym": 0, "crc32": 0, "cfo_est": 12, "flags": 0, "mpdu_offset": 0}]
    assert results_from_rows(rows_from_results(rs)) == rs


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from gpu_util import make_capture
    from oracle.pyoracle import Oracle, RATES
    from sora_amd.shard import gather_rows, partition, reduce_counters, results_from_rows, rows_from_results
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = Oracle()
    ncaps = 11
    caps = [make_capture(o, RATES[i % 8], 60 + 9 * i, seed=i, rate_mhz=20, sigma=100)[0] for i in range(ncaps)]
    first, count = partition(ncaps, world, rank)
    local = []
    for i in range(first, first + count):
        for r in o.rx_capture(caps[i], 20):
            r = dict(r); r["capture_id"] = i; local.append(r)
    rows = torch.from_numpy(rows_from_results(local).copy())
    allrows, counts = gather_rows(rows, len(local), max_rows_per_rank=16)
    tot = reduce_counters([len(local), sum(r["error_code"] == 1 for r in local)])
    got = results_from_rows(allrows.numpy())
    q.put((rank, counts, tot, [(g["capture_id"], g["rate_kbps"], g["length"], g["crc32"], g["error_code"]) for g in got]))
    dist.barrier(); dist.destroy_process_group()


def test_two_rank_gloo_gather():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    outs = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    outs.sort()
    # single-process truth
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpu_util import make_capture
    from oracle.pyoracle import Oracle, RATES
    o = Oracle()
    want = []
    for i in range(11):
        c = make_capture(o, RATES[i % 8], 60 + 9 * i, seed=i, rate_mhz=20, sigma=100)[0]
        for r in o.rx_capture(c, 20):
            want.append((i, r["rate_kbps"], r["length"], r["crc32"], r["error_code"]))
    assert len(want) == 11
    for rank, counts, tot, got in outs:
        assert sum(counts) == 11 and counts == [6, 5]
        assert tot == [11, sum(1 for w in want if w[4] == 1)]
        assert got == want            # rank order == capture order: the gather needs no re-sort


def _worker_11n(rank, world, port, q):
    """the same sharding for the 802.11n graph: two-chain captures are independent units too (rows carry the MCS index as rate_kbps)"""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from oracle.pyoracle import Oracle
    from sora_amd.shard import gather_rows, partition, reduce_counters, results_from_rows, rows_from_results
    from test_oracle_11n_graph import golden_captures
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = Oracle()
    caps = [(a, b) for a, b, _, _, _ in golden_captures()]
    first, count = partition(len(caps), world, rank)
    local = []
    for i in range(first, first + count):
        for r in o.rx11n_capture(*caps[i]):
            r = {k: v for k, v in r.items() if k != "mpdu"}; r["capture_id"] = i; local.append(r)
    rows = torch.from_numpy(rows_from_results(local).copy())
    allrows, counts = gather_rows(rows, len(local), max_rows_per_rank=32)
    tot = reduce_counters([len(local), sum(r["error_code"] == 1 for r in local)])
    got = results_from_rows(allrows.numpy())
    q.put((rank, counts, tot, [(g["capture_id"], g["end_sample"], g["rate_kbps"], g["length"], g["crc32"], g["error_code"]) for g in got]))
    dist.barrier(); dist.destroy_process_group()


def test_two_rank_gloo_gather_11n():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker_11n, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    outs = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle.pyoracle import Oracle
    from test_oracle_11n_graph import golden_captures
    o = Oracle(); want = []
    for i, (a, b, _, _, _) in enumerate(golden_captures()):
        for r in o.rx11n_capture(a, b):
            want.append((i, r["end_sample"], r["rate_kbps"], r["length"], r["crc32"], r["error_code"]))
    assert len(want) >= 12
    for rank, counts, tot, got in outs:
        assert sum(counts) == len(want) and len(counts) == 2 and min(counts) > 0
        assert tot == [len(want), sum(1 for w in want if w[5] == 1)]
        assert got == want


def _worker_mpdu(rank, world, port, q):
    """rows + MPDUs: every rank's MPDU bytes reach every rank (sora_amd.shard.gather_mpdus: dense pack, three all-gathers), with the rows'
    mpdu_offset pointing into the gathered buffer -- the logic sora_shard_gather_results_mpdu runs over RCCL."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from gpu_util import make_capture
    from oracle.pyoracle import Oracle, RATES
    from sora_amd.shard import gather_mpdus, partition, results_from_rows, rows_from_results
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = Oracle()
    ncaps = 9
    caps = [make_capture(o, RATES[i % 8], 40 + 31 * i, seed=40 + i, rate_mhz=20, sigma=100 if i != 4 else 6000)[0] for i in range(ncaps)]
    first, count = partition(ncaps, world, rank)
    local = []; sparse = np.zeros(1 << 16, np.uint8); pos = 7                  # a sparse MPDU array as the library's (slot-addressed) one
    for i in range(first, first + count):
        for r in o.rx_capture(caps[i], 20):
            r = dict(r); r["capture_id"] = i
            if r["error_code"] in (1, 0x80000006):
                sparse[pos:pos + r["length"]] = np.frombuffer(r["mpdu"], np.uint8); r["mpdu_offset"] = pos; pos += r["length"] + 13
            local.append(r)
    rows = torch.from_numpy(rows_from_results(local).copy())
    allrows, allmp, counts = gather_mpdus(rows, len(local), torch.from_numpy(sparse), max_rows_per_rank=16, max_bytes_per_rank=8192)
    got = results_from_rows(allrows.numpy()); mp = allmp.numpy()
    q.put((rank, counts, [(g["capture_id"], g["error_code"], g["length"], bytes(mp[g["mpdu_offset"]:g["mpdu_offset"] + g["length"]]) if g["error_code"] in (1, 0x80000006) else b"") for g in got]))
    dist.barrier(); dist.destroy_process_group()


def test_two_rank_gloo_gather_mpdus():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker_mpdu, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    outs = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpu_util import make_capture
    from oracle.pyoracle import Oracle, RATES
    o = Oracle(); want = []
    for i in range(9):
        c = make_capture(o, RATES[i % 8], 40 + 31 * i, seed=40 + i, rate_mhz=20, sigma=100 if i != 4 else 6000)[0]
        for r in o.rx_capture(c, 20):
            want.append((i, r["error_code"], r["length"], r["mpdu"] if r["error_code"] in (1, 0x80000006) else b""))
    assert len(want) >= 8 and sum(1 for w in want if w[1] == 1) >= 6
    for rank, counts, got in outs:
        assert sum(counts) == len(want)
        assert got == want


def _worker_overflow(rank, world, port, q):
    """a rank whose MPDUs do not fit its block must still enter all three collectives (the others would wait for ever) and EVERY rank raises."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from sora_amd.shard import gather_mpdus
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = torch.zeros((2, 9), dtype=torch.int32)
    rows[:, 3] = 1; rows[0, 5] = 100; rows[1, 5] = 300 if rank == 1 else 50; rows[1, 8] = 100      # rank 1 holds 400 bytes of MPDUs, rank 0 150
    mpdu = torch.arange(1024, dtype=torch.int32).to(torch.uint8)
    try:
        gather_mpdus(rows, 2, mpdu, max_rows_per_rank=4, max_bytes_per_rank=256)
        q.put((rank, "no error"))
    except ValueError as e:
        q.put((rank, "local: %s" % e))
    except RuntimeError as e:
        q.put((rank, "remote: %s" % e))
    dist.barrier(); dist.destroy_process_group()


def test_a_rank_that_overflows_takes_every_rank_down_instead_of_hanging_them():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker_overflow, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    outs = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert outs[1].startswith("local:") and "exceed" in outs[1], outs
    assert outs[0].startswith("remote:") and "[1]" in outs[0], outs

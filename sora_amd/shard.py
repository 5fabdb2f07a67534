"""Multi-GPU sharding of the receive path: captures are the natural shard.

Independent 20 MHz captures share no state (the reference resets its context per frame,
kernel/bb/demod11/fb11ademod_config.hpp:68-95), so rank r of W runs the whole path on a contiguous block of
captures in its own HBM -- no collective on the data path.  The only exchange is the result gather:
one all-gather (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests) of fixed-size
sora_frame_result rows (9 x int32 = 36 bytes) plus their per-rank counts.
"""
import numpy as np

ROW_WORDS = 9          # sizeof(sora_frame_result) / 4


def partition(n_items, world_size, rank):
    """Static block partition: -> (first, count).  Ranks differ by at most one item."""
    base, extra = divmod(n_items, world_size)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def rows_from_results(results):
    """list of result dicts (capi.Rx.results / oracle) -> int32 array [n, 9] in sora_frame_result layout."""
    out = np.zeros((len(results), ROW_WORDS), np.uint32)
    for i, r in enumerate(results):
        out[i] = (r["capture_id"], r["start_sample"], r["end_sample"], r["error_code"], r["rate_kbps"],
                  (r["length"] & 0xFFFF) | ((r["nsym"] & 0xFFFF) << 16), r["crc32"],
                  (r["cfo_est"] & 0xFFFF) | ((r.get("flags", 0) & 0xFFFF) << 16), r.get("mpdu_offset", 0))
    return out.view(np.int32)


def results_from_rows(rows):
    rows = np.asarray(rows).view(np.uint32).reshape(-1, ROW_WORDS)
    out = []
    for w in rows:
        cfo = int(w[7] & 0xFFFF)
        out.append({"capture_id": int(w[0]), "start_sample": int(w[1]), "end_sample": int(w[2]), "error_code": int(w[3]),
                    "rate_kbps": int(w[4]), "length": int(w[5] & 0xFFFF), "nsym": int(w[5] >> 16), "crc32": int(w[6]),
                    "cfo_est": cfo - 65536 if cfo >= 32768 else cfo, "flags": int(w[7] >> 16), "mpdu_offset": int(w[8])})
    return out


def gather_rows(rows, nrows, max_rows_per_rank, group=None):
    """All-gather variable-length row blocks.  rows: int32 tensor [>=nrows, 9] on the backend's device,
    nrows: python int.  Returns (int32 tensor [total, 9] in rank order, list of per-rank counts)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = rows.device
    pad = torch.zeros((max_rows_per_rank, ROW_WORDS), dtype=torch.int32, device=dev)
    pad[:nrows] = rows[:nrows]
    counts = torch.zeros(world, dtype=torch.int32, device=dev)
    mine = torch.tensor([nrows], dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(counts, mine, group=group)
    allrows = torch.zeros((world * max_rows_per_rank, ROW_WORDS), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(allrows, pad, group=group)
    cl = [int(c) for c in counts.tolist()]
    parts = [allrows[r * max_rows_per_rank: r * max_rows_per_rank + cl[r]] for r in range(world)]
    return torch.cat(parts) if parts else allrows[:0], cl


def pack_mpdus(rows, nrows, mpdu, max_bytes):
    """This rank's MPDUs, densely packed in row order (the exchange unit of gather_mpdus).  rows: int32 tensor [>=nrows, 9] whose
    mpdu_offset (word 8) indexes the uint8 tensor `mpdu`; rows that carry an MPDU are FRAME_OK / CRC32_FAIL.  Returns (uint8 tensor
    [max_bytes] on rows' device, bytes used, int32 tensor [nrows] of the rows' dense offsets).  Runs on the rows' device (torch ops)."""
    import torch
    dev = rows.device
    r = rows[:nrows].to(torch.int64)
    err = r[:, 3] & 0xFFFFFFFF
    has = (err == 0x00000001) | (err == 0x80000006)
    ln = torch.where(has, r[:, 5] & 0xFFFF, torch.zeros_like(r[:, 5]))
    dense = torch.cumsum(ln, 0) - ln
    used = int(ln.sum().item()) if nrows else 0
    if used > max_bytes:
        raise ValueError("pack_mpdus: %d MPDU bytes exceed max_bytes %d" % (used, max_bytes))
    out = torch.zeros(max_bytes, dtype=torch.uint8, device=dev)
    if used:
        row_of = torch.repeat_interleave(torch.arange(nrows, device=dev), ln)                  # for every dense byte: its row ...
        within = torch.arange(used, device=dev) - dense[row_of]                                 # ... and its index inside the MPDU
        out[:used] = mpdu[(r[:, 8] & 0xFFFFFFFF)[row_of] + within]
    return out, used, dense.to(torch.int32)


def gather_mpdus(rows, nrows, mpdu, max_rows_per_rank, max_bytes_per_rank, group=None):
    """Rows AND MPDUs of every rank on every rank (what fb11a_demod.cpp:64-70 hands to the MAC, for a sharded batch): three all-gathers
    -- {rows, bytes} per rank, the padded row blocks, the dense MPDU blocks.  Returns (int32 tensor [total, 9] whose mpdu_offset indexes
    the second result, uint8 tensor [total bytes], per-rank row counts)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = rows.device
    # A rank whose local tables do not fit still enters all three collectives (contributing the count -1 and no bytes) and every rank
    # raises afterwards -- the others would otherwise wait forever in a collective the failed rank never enters (the C path,
    # sora_shard_gather_results_mpdu, does the same with the sentinel 0xFFFFFFFF).
    failure = None
    pad = torch.zeros((max_rows_per_rank, ROW_WORDS), dtype=torch.int32, device=dev)
    try:
        if nrows > max_rows_per_rank:
            raise ValueError("gather_mpdus: %d rows exceed max_rows_per_rank %d" % (nrows, max_rows_per_rank))
        block, used, dense = pack_mpdus(rows, nrows, mpdu, max_bytes_per_rank)
        pad[:nrows] = rows[:nrows]
        pad[:nrows, 8] = dense
    except ValueError as e:
        failure = e
        block, used, nrows = torch.zeros(max_bytes_per_rank, dtype=torch.uint8, device=dev), 0, -1
    pairs = torch.zeros((world, 2), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(pairs, torch.tensor([[nrows, used]], dtype=torch.int32, device=dev), group=group)
    allrows = torch.zeros((world * max_rows_per_rank, ROW_WORDS), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(allrows, pad, group=group)
    allmp = torch.zeros(world * max_bytes_per_rank, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(allmp, block, group=group)
    pl = pairs.tolist()
    if failure is not None:
        raise failure
    bad = [r for r in range(world) if int(pl[r][0]) < 0]
    if bad:
        raise RuntimeError("gather_mpdus: rank(s) %s reported that their tables do not fit (nothing was gathered)" % bad)
    rparts, mparts, moff = [], [], 0
    for r in range(world):
        c, b = int(pl[r][0]), int(pl[r][1])
        blk = allrows[r * max_rows_per_rank: r * max_rows_per_rank + c].clone()
        blk[:, 8] += moff
        rparts.append(blk); mparts.append(allmp[r * max_bytes_per_rank: r * max_bytes_per_rank + b]); moff += b
    return (torch.cat(rparts) if rparts else allrows[:0]), (torch.cat(mparts) if mparts else allmp[:0]), [int(p[0]) for p in pl]


def reduce_counters(values, group=None, device=None):
    """Sum small integer counters (frames, CRC-ok, samples, bits) over the ranks."""
    import torch
    import torch.distributed as dist
    t = torch.tensor(list(values), dtype=torch.int64, device=device)
    dist.all_reduce(t, group=group)
    return [int(v) for v in t.tolist()]

#!/usr/bin/env python3
"""summarize_pmc.py -- per-kernel HBM traffic of one receive call from two rocprofv3 --pmc passes.

    python tools/summarize_pmc.py <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv> <frames per launch> [<SQ_INSTS counter_collection.csv>] > profiles/rNN_traffic.json

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch (they come from the L2's memory-side request counters).
Corrections as /opt/skills/guides/MI355X_MICROARCH.md (section HBM) prescribes for gfx950 -- FETCH_SIZE reports half the bytes
of a coalesced vector read; "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your
own access pattern" -- with the calibration done: tools/calib/fetch_calib moves 1 GiB in each access shape the kernels use
(profiles/r02_k_calibration.json): vector loads, 16 B or 4 B per lane: reported / moved = 0.5; scalar loads of 32 bytes
(s_load_dwordx8, the way round 2's k_viterbi read its soft values): 1.0; stores of 16 B, 4 B and 1 B per lane: 1.0.  Every kernel
of round 3 is fed by vector loads (the trellis kernels fetch the packed soft stream with 16-bit vector loads), so FETCH_SIZE is
doubled for all of them; WRITE_SIZE is taken as reported.  The two counters cannot share a pass, so each file comes from its own
run of the same command.
"""
import csv
import json
import sys
from collections import defaultdict


SCALAR_FED = ()          # (round 2: k_viterbi, fed by 32-byte scalar loads that FETCH_SIZE counted in full)


def per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            name = name.split("(")[0].replace("sora::", "").replace("void ", "").strip()
            acc[name].append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    out = {"unit": "bytes per launch", "frames_per_launch": frames,
           "corrections": "FETCH_SIZE KiB x1024 x2 (the guide's gfx950 correction; every kernel is fed by vector loads); WRITE_SIZE KiB x1024 as reported; factors measured by tools/calib/fetch_calib (profiles/r02_k_calibration.json)",
           "kernels": {}}
    total = 0.0
    trellis = sys.argv[5] if len(sys.argv) > 5 else "k_viterbi16"             # the trellis kernel of the call being summed (the run also holds a few launches of the other one)
    rx_path = ("k_scan", "k_frame", trellis, "k_decode", "k_finish") + (("k_win_redo", "k_win_redo_finish") if trellis == "k_viterbi16w" else ())           # one receive call (split or fused chain); other kernels (k_pack, ingest, tx) are listed only
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        fb = fetch.get(k, (0.0, 0))[0] * 1024 * (1 if k in SCALAR_FED else 2)
        wb = write.get(k, (0.0, 0))[0] * 1024
        out["kernels"][k] = {"fetch_bytes": round(fb), "write_bytes": round(wb), "hbm_bytes": round(fb + wb),
                             "launches_sampled": fetch.get(k, (0.0, 0))[1]}
        if k in rx_path:
            total += fb + wb
    if len(sys.argv) > 4 and sys.argv[4]:                                     # third pass: SQ_INSTS_VALU / SQ_INSTS_SALU per dispatch
        valu = per_kernel(sys.argv[4], "SQ_INSTS_VALU"); salu = per_kernel(sys.argv[4], "SQ_INSTS_SALU")
        tv = 0.0
        for k in out["kernels"]:
            out["kernels"][k]["valu_insts"] = round(valu.get(k, (0.0, 0))[0]); out["kernels"][k]["salu_insts"] = round(salu.get(k, (0.0, 0))[0])
            if k in rx_path:
                tv += valu.get(k, (0.0, 0))[0]
        out["total_valu_insts_per_call"] = round(tv)
    try:                                                                      # the tree these counters belong to: bench.py refuses to replay them beside other sources
        import os
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        from sora_amd import build as _b
        out["sources_sha256"] = _b.sources_sha256()
    except Exception as e:
        out["sources_sha256"] = None; out["sources_sha256_error"] = repr(e)
    out["receive_call_sums"] = list(rx_path)
    out["total_hbm_bytes_per_call"] = round(total)
    out["algorithmic_bytes_per_call"] = round(frames * 4880 * 4.3375)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()

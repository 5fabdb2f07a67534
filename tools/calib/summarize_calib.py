#!/usr/bin/env python3
"""summarize_calib.py <FETCH_SIZE csv> <WRITE_SIZE csv> -- reported KiB x 1024 over the bytes each calibration kernel really moved."""
import csv, json, sys

BYTES = 1 << 30
MOVED = {"calib_read16": BYTES, "calib_read4": BYTES, "calib_read_scalar": BYTES, "calib_write16": BYTES, "calib_write4": BYTES, "calib_write1": BYTES // 4}


def counters(path, counter):
    out = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                out[row["Kernel_Name"].split("(")[0].strip()] = float(row["Counter_Value"]) * 1024
    return out


fetch, write = counters(sys.argv[1], "FETCH_SIZE"), counters(sys.argv[2], "WRITE_SIZE")
res = {"bytes_moved": MOVED, "reported_over_moved": {}}
for k, n in MOVED.items():
    res["reported_over_moved"][k] = {"FETCH_SIZE": round(fetch.get(k, 0.0) / n, 4), "WRITE_SIZE": round(write.get(k, 0.0) / n, 4)}
json.dump(res, sys.stdout, indent=1); print()

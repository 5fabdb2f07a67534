"""Per-kernel summary of a rocprofv3 --kernel-trace run's rocpd database (the image's rocprofv3 writes <name>_results.db): calls, average / total duration, grid, registers, LDS;
with --last N also the start offsets and durations of the last N dispatches (one call's chain).  Usage: python tools/rocprof_kernels.py gpurun_out/x/y_results.db [--last 10]"""
import json
import sqlite3
import sys


def main():
    db = sys.argv[1]; last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, start, end, grid_x, workgroup_x, vgpr_count, lds_size from kernels order by start"))
    agg = {}
    for name, st, en, gx, wx, vg, lds in rows:
        k = name.split("(")[0]
        a = agg.setdefault(k, {"calls": 0, "total_us": 0.0, "grid": gx, "workgroup": wx, "vgprs": vg, "lds_bytes": lds})
        a["calls"] += 1; a["total_us"] += (en - st) / 1e3
    out = {"kernels": {k: dict(v, avg_us=round(v["total_us"] / v["calls"], 2), total_us=round(v["total_us"], 1)) for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["total_us"])}}
    if last:
        t0 = rows[-last][1]
        out["last_dispatches"] = [{"kernel": n.split("(")[0], "start_us": round((s - t0) / 1e3, 1), "dur_us": round((e - s) / 1e3, 1)} for n, s, e, *_ in rows[-last:]]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

#!/bin/bash
# r04_insts.sh <tag> [bench args] -- on the GPU box: wave-level instruction counts (VALU / SALU / LDS / VMEM) and durations per kernel of one receive call,
# one call in flight, for the library named by SORA_HIP_LIB (or the product library).  -> gpurun_out/<tag>_insts.json
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-extras --check 64 --steps 3 --warmup 1 --depth 1 --min-seconds 0 --no-deliver $*"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $OUT/${TAG}_i -o p -- python $R/bench.py $ARGS > /dev/null 2> $OUT/${TAG}_i.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_k -o p -- python $R/bench.py $ARGS > /dev/null 2> $OUT/${TAG}_k.err
python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/${TAG}_i/*counter_collection.csv"):
    for row in csv.DictReader(open(f, newline="")):
        k = row["Kernel_Name"].split("(")[0].replace("sora::", "").strip()
        if k.startswith("k_"): acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: round(sum(v) / len(v)) for c, v in sorted(d.items())} for k, d in acc.items()}
for f in glob.glob("$OUT/${TAG}_k/*kernel_stats.csv"):
    for row in csv.DictReader(open(f, newline="")):
        k = row["Name"].split("(")[0].replace("sora::", "").strip()
        if k in out: out[k]["avg_ns"] = float(row["AverageNs"]); out[k]["calls"] = int(row["Calls"])
rx = [k for k in out if k in ("k_scan", "k_frame", "k_sym_front", "k_track", "k_sym_back", "k_viterbi16", "k_viterbi", "k_finish")]
tot = {c: sum(out[k].get(c, 0) for k in rx if not (k == "k_viterbi" and "k_viterbi16" in out and "$*".find("--trellis 64") < 0)) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")}
out["_total_rx_call"] = tot
json.dump(out, open("$OUT/${TAG}_insts.json", "w"), indent=1)
for k in rx + ["_total_rx_call"]:
    print(k, {c.replace("SQ_INSTS_", ""): v for c, v in out[k].items()})
PY
rm -rf $OUT/${TAG}_i $OUT/${TAG}_k

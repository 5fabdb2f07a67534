#!/bin/bash
# pmc_sq.sh <tag> [bench args ...] -- on the GPU box: SQ / GRBM counters of every kernel of the receive call (rocprofv3 --pmc, its own
# passes; dispatches are serialised by counter collection, so "saturation" is a LARGE launch (--frames 16384), not several calls in flight).
# Writes gpurun_out/<tag>_sq_counters.json (mean per launch and kernel).
set -u
TAG=${1:-r04}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-extras --check 64 --steps 3 --warmup 1 --depth 1 --min-seconds 0 --no-deliver $*"
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --output-format csv -d $OUT/${TAG}_sq$i -o p -- python $R/bench.py $ARGS > /dev/null 2> $OUT/${TAG}_sq$i.err
done
python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/${TAG}_sq*/*counter_collection.csv"):
    for row in csv.DictReader(open(f, newline="")):
        k = row["Kernel_Name"].split("(")[0].replace("sora::", "").strip()
        if k.startswith("k_"): acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: round(sum(v) / len(v)) for c, v in sorted(d.items())} for k, d in acc.items()}
out["_args"] = "$ARGS"
json.dump(out, open("$OUT/${TAG}_sq_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
rm -rf $OUT/${TAG}_sq1 $OUT/${TAG}_sq2 $OUT/${TAG}_sq3 $OUT/${TAG}_sq4

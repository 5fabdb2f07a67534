#!/bin/bash
# pmc_insts.sh <tag> -- quick look (GPU box): wave-level VALU / SALU instruction counts per kernel of one receive call (one call in flight).
set -u
TAG=${1:-insts}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PMC="--no-cpu-baseline --no-extras --check 64 --steps 3 --warmup 1 --depth 1 --trellis 16 --min-seconds 0 --no-deliver"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/${TAG}_insts -o p -- python $R/bench.py $PMC > /dev/null 2> $OUT/${TAG}_insts.err
I=$(find $OUT/${TAG}_insts -name "*counter_collection.csv" | head -1)
python3 - "$I" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print("%-40s" % k[:40], {c: round(sum(v) / len(v) / 1e6, 2) for c, v in d.items()}, "M per launch,", len(list(d.values())[0]), "launches")
PY
rm -rf $OUT/${TAG}_insts

#!/bin/bash
# pmc_insts_section.sh <tag> <bench key> -- GPU box: wave-level VALU / SALU instruction counts per kernel of one bench section (bench.py --only <key>)
set -u
TAG=${1:-insts}; KEY=${2:-rx11n}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/${TAG}_insts -o p -- python $R/bench.py --no-cpu-baseline --only $KEY > /dev/null 2> $OUT/${TAG}_insts.err
I=$(find $OUT/${TAG}_insts -name "*counter_collection.csv" | head -1)
python3 - "$I" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "sora::" not in r["Kernel_Name"]: continue
    acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print("%-40s" % k[:40], {c: round(sum(v) / len(v) / 1e6, 2) for c, v in d.items()}, "M per launch,", len(list(d.values())[0]), "launches")
PY
rm -rf $OUT/${TAG}_insts

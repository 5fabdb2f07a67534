#!/bin/bash
# pmc_section.sh <bench section> <tag> -- on the GPU box: SQ_INSTS_VALU / SQ_INSTS_SALU per kernel of one bench section (one --pmc pass)
set -u
SEC=${1:-rx11n}; TAG=${2:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $OUT/${TAG}_${SEC}_pmc -o p -- python $R/bench.py --no-cpu-baseline --only $SEC > /dev/null 2> $OUT/${TAG}_${SEC}_pmc.err
python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/${TAG}_${SEC}_pmc/*counter_collection.csv"):
    for row in csv.DictReader(open(f, newline="")):
        k = row["Kernel_Name"].split("(")[0].replace("sora::", "").strip()
        if k.startswith("k_"): acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: round(sum(v) / len(v)) for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open("$OUT/${TAG}_${SEC}_insts.json", "w"), indent=1); print(json.dumps(out, indent=1))
PY
rm -rf $OUT/${TAG}_${SEC}_pmc

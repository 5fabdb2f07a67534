#!/bin/bash
# r04_rows_insts.sh -- on the GPU box: wave-level instruction counts (VALU / SALU / LDS) per kernel launch of every widened bench row, next to the row's time:
# what bounds a row is its instruction count against the chip's sustained issue rate (profiles/r04_valu_peak.json), not HBM.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/r04_k_rows_insts.txt
for KEY in tx rx11b rx11b_cck rx11n rx11n_40; do
  timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/ri_$KEY -o p -- python $R/bench.py --no-cpu-baseline --only $KEY > $OUT/ri_$KEY.json 2> $OUT/ri_$KEY.err
  I=$(find $OUT/ri_$KEY -name "*counter_collection.csv" | head -1)
  python3 - "$I" $KEY >> $OUT/r04_k_rows_insts.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "sora::" not in r["Kernel_Name"]: continue
    acc[r["Kernel_Name"].split("(")[0].replace("sora::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== row", sys.argv[2])
tot = 0.0
for k, d in sorted(acc.items()):
    per = {c.replace("SQ_INSTS_", ""): round(sum(v) / len(v) / 1e6, 2) for c, v in d.items()}
    tot += sum(per.values())
    print("  %-28s %s M per launch, %d launches" % (k[:28], per, len(list(d.values())[0])))
print("  sum over the row's kernels (one launch each): %.1f M wave-instructions" % tot)
PY
  rm -rf $OUT/ri_$KEY
done
cat $OUT/r04_k_rows_insts.txt

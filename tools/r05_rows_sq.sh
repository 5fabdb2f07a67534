#!/bin/bash
# r05_rows_sq.sh -- on the GPU box: the SQ-counter breakdown DESIGN.md section 3.6 has for the 802.11a chain, for the kernels of every widened bench row:
# shares of the waves' cycles spent executing an instruction (of which vector), waiting (s_waitcnt / barrier), stalled at issue; LDS bank-conflict share; wave-level
# instruction counts.  Two rocprofv3 --pmc passes per row (counter collection serialises dispatches: each kernel is alone on the chip).  -> gpurun_out/r05_rows_sq.json
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for KEY in ${ROWS:-tx rx11b_cck rx11n rx11n_40 shard_32x16}; do
  i=0
  for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD"; do
    i=$((i+1))
    timeout 400 rocprofv3 --pmc $SET --output-format csv -d $OUT/rs_${KEY}_$i -o p -- python $R/bench.py --no-cpu-baseline --only $KEY > $OUT/rs_${KEY}_$i.json 2> $OUT/rs_${KEY}_$i.err
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections, os
out_dir = sys.argv[1]
res = {}
for d in sorted(glob.glob(out_dir + "/rs_*_1")):
    key = os.path.basename(d)[3:-2]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in (d, d[:-1] + "2"):
        for f in glob.glob(p + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f, newline="")):
                if "sora::" not in r["Kernel_Name"]: continue
                acc[r["Kernel_Name"].split("(")[0].replace("sora::", "").replace("void ", "").strip()][r["Counter_Name"]].append(float(r["Counter_Value"]))
    row = {}
    for k, c in acc.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        wc = m.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        row[k] = {"launches_sampled": len(c.get("SQ_WAVES", [])), "waves": round(m.get("SQ_WAVES", 0)),
                  "executing": round(m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3), "of_which_vector": round(m.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3), "lds": round(m.get("SQ_ACTIVE_INST_LDS", 0) / wc, 3),
                  "waiting": round(m.get("SQ_WAIT_ANY", 0) / wc, 3), "issue_stall": round(m.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                  "lds_bank_conflict_share_of_lds_cycles": round(m.get("SQ_LDS_BANK_CONFLICT", 0) / (m.get("SQ_LDS_IDX_ACTIVE", 0) or 1.0), 3),
                  "insts_M": {n.replace("SQ_INSTS_", "").lower(): round(m[n] / 1e6, 2) for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD") if n in m}}
    res[key] = row
json.dump(res, open(out_dir + "/r05_rows_sq.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:5000])
PY
rm -rf $OUT/rs_*_1 $OUT/rs_*_2

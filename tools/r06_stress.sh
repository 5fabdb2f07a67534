#!/bin/bash
# r06_stress.sh [captures per process] -- on the GPU box: the randomised parity hunt (tools/stress_parity.py: GPU rows against the oracle's, event for event, MPDU bytes
# included) over the round-6 kernels -- the pass-structured k_scan, k_frame with the workgroup's tracker in wave 0, every trellis kernel, the three-kernel chain, k_pipe
# in both trellis forms and k_pipe with the bound of its waits at zero (every call made again by the finishing kernel: the redo path exercised and counted) -- sixteen
# processes side by side with different seeds and batch sizes.  The output is stamped with the hash of the sources the loaded library was built from.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
N=${1:-3000}
i=0
for cfg in "--batch 40" "--batch 125" "--batch 600 --trellis 1" "--batch 1200 --trellis 1 --front 1" "--batch 300 --front 3" "--batch 125 --noise-frames 0.3" \
           "--batch 2400 --front 1 --trellis 16" "--batch 250 --front 3 --trellis 64" "--batch 125 --front 1 --trellis 16" "--batch 600 --front 1 --trellis 64 --max-frames 16" \
           "--batch 1 --depth 1 --front 4" "--batch 2 --depth 1 --front 4 --noise-frames 0.2" "--batch 1 --depth 4 --max-frames 4 --front 4" \
           "--batch 1 --depth 1 --front 4 --pipe-wait-us 0" "--batch 2 --depth 1 --front 4 --pipe-wait-us 0 --noise-frames 0.1" "--batch 1 --depth 4 --max-frames 4 --front 4 --pipe-wait-us 0"; do
  i=$((i+1))
  n=$N; case "$cfg" in *"--front 4"*) n=$((N/2));; esac
  timeout 2400 python $R/tools/stress_parity.py --captures $n --seed $((600+i)) $cfg > $OUT/r06_stress_$i.txt 2>&1 &
done
wait
{ python - <<PY
import sys; sys.path.insert(0, "$R")
from sora_amd import build as b
i = b.build_info(); print("sources_sha256", i["sources_sha256_now"], "built_from_this_tree", i["built_from_this_tree"])
PY
for j in $(seq 1 $i); do tail -1 $OUT/r06_stress_$j.txt; done; } > $OUT/r06_stress.txt
cat $OUT/r06_stress.txt
# ... and the 802.11n graph with each of its trellis kernels (1 = the window-parallel form new this round), the 802.11b graph once
( for cfg in "600 71 1" "600 72 1" "2000 73 1" "150 74 1" "600 75 64" "600 76 16"; do python $R/tools/stress_parity_11n.py $cfg 2>&1 | tail -1; done; python $R/tools/stress_parity_11b.py --captures 4000 --seed 77 2>&1 | tail -1 ) > $OUT/r06_stress_11n_11b.txt 2>&1
cat $OUT/r06_stress_11n_11b.txt >> $OUT/r06_stress.txt; cat $OUT/r06_stress_11n_11b.txt

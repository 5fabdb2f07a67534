#!/bin/bash
# r05_stress.sh -- on the GPU box: the randomised parity hunt (tools/stress_parity.py) over the round-5 kernels, twelve processes side by side with different seeds,
# batch sizes (units of one, three and more windows), kernel choices and a share of noise-behind-a-good-header frames (the re-decode path).  -> gpurun_out/r05_stress.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
N=${1:-2400}
i=0
for cfg in "--batch 40" "--batch 125" "--batch 600 --trellis 1" "--batch 1200 --trellis 1 --front 1" "--batch 300 --front 3" "--batch 125 --noise-frames 0.3" \
           "--batch 600 --noise-frames 0.2 --trellis 1" "--batch 8 --front 3 --trellis 1" "--batch 2400 --trellis 1" "--batch 250 --front 3 --trellis 64" "--batch 1 " "--batch 125 --front 1 --trellis 16"; do
  i=$((i+1))
  n=$N; case "$cfg" in *"--batch 1 "*) n=150;; *"--batch 8 "*) n=600;; esac
  timeout 1500 python $R/tools/stress_parity.py --captures $n --seed $((500+i)) $cfg > $OUT/r05_stress_$i.txt 2>&1 &
done
wait
for j in $(seq 1 $i); do tail -1 $OUT/r05_stress_$j.txt; done > $OUT/r05_stress.txt
cat $OUT/r05_stress.txt

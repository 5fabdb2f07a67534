#!/bin/bash
# r05_stress_pipe.sh -- on the GPU box: the randomised parity hunt (tools/stress_parity.py) through k_pipe (sora_rx_set_front(4)): eight processes side by side with
# different seeds, one or two captures per call (up to eight frames each), one to four calls' worth of handle, a share of noise-behind-a-good-header frames (the units'
# proof fails: the serial decode inside k_win_redo_finish).  -> gpurun_out/r05_stress_pipe.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
N=${1:-6000}
i=0
for cfg in "--batch 1 --depth 1" "--batch 2 --depth 1" "--batch 1 --depth 4 --max-frames 4" "--batch 2 --depth 1 --noise-frames 0.2" "--batch 1 --depth 2 --noise-frames 0.1" "--batch 4 --depth 1 --max-frames 4" \
           "--batch 2 --depth 2 --max-frames 4" "--batch 1 --depth 1 --max-frames 16"; do
  i=$((i+1))
  timeout 1500 python $R/tools/stress_parity.py --captures $N --seed $((900+i)) --front 4 $cfg > $OUT/r05_stress_pipe_$i.txt 2>&1 &
done
wait
for j in $(seq 1 $i); do tail -1 $OUT/r05_stress_pipe_$j.txt; done > $OUT/r05_stress_pipe.txt
cat $OUT/r05_stress_pipe.txt

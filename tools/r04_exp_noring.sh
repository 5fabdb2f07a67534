#!/bin/bash
# r04_exp_noring.sh -- upper bound of a trellis kernel whose survivor ring is not in LDS: variants of the library in which k_viterbi16 has
# NO ring and no trace-back (results wrong, durations meaningful): occupancy limited by registers (3 waves per SIMD; "lb4": forced to 4).
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
for v in main noring noring_lb4; do
  if [ "$v" = main ]; then unset SORA_HIP_LIB; else export SORA_HIP_LIB=$R/sora_amd/lib/variants/$v.so; fi
  for cfg in "0 1 4096" "0 2 4096" "16 8 4096" "16 2 16384"; do
    set -- $cfg
    timeout 300 python bench.py --no-cpu-baseline --no-extras --hw-queues $1 --depth $2 --frames $3 --trellis 16 --check 64 --min-seconds 0.5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$v hwq $1 depth $2 frames $3 ms_per_step', d['ms_per_step'], 'in_flight', {k: round(v, 4) for k, v in d['kernel_ms'].items()}, 'alone', {k: round(v, 4) for k, v in d['kernel_ms_one_call_in_flight'].items()}, 'parity', d['parity']['ok'], d['frames_crc_ok'])"
  done
done 2>&1 | tee $OUT/r04_b_exp_noring.txt

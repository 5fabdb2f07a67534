#!/bin/bash
# r04_exp_occ.sh -- does the trellis kernel gain from 4 waves per SIMD?  32768 frames per launch = 4096 trellis waves = 4 per SIMD if
# evenly placed; "noring_lb4" (no ring, no trace-back, <= 128 VGPRs) can hold them all at once, the product kernel runs two rounds of 2048.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
for v in main noring noring_lb4; do
  if [ "$v" = main ]; then unset SORA_HIP_LIB; else export SORA_HIP_LIB=$R/sora_amd/lib/variants/$v.so; fi
  for fr in 8192 32768; do
    timeout 300 python bench.py --no-cpu-baseline --no-extras --hw-queues 0 --depth 1 --frames $fr --trellis 16 --check 64 --min-seconds 0.3 --steps 10 --warmup 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$v frames $fr ms_per_step', d['ms_per_step'], 'alone', {k: round(v, 4) for k, v in d['kernel_ms_one_call_in_flight'].items()})"
  done
done 2>&1 | tee $OUT/r04_c_exp_occ.txt

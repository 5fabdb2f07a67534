#!/bin/bash
# r04_ab.sh <variant>... -- on the GPU box: the headline step at 8 calls in flight / 16 queues and at 1 and 2 calls in flight, per library variant
# ("main" = the product library).  AB_ARGS adds bench flags.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
for v in "$@"; do
  if [ "$v" = main ]; then unset SORA_HIP_LIB; else export SORA_HIP_LIB=$R/sora_amd/lib/variants/$v.so; fi
  for cfg in ${AB_CFGS:-"16:8:16" "0:2:64" "0:1:64"}; do
    IFS=: read q d t <<< "$cfg"
    timeout 300 python bench.py --no-cpu-baseline --no-extras --hw-queues $q --depth $d --trellis $t --check 256 --min-seconds 1 ${AB_ARGS:-} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$v hwq $q depth $d trellis $t ms_per_step', d['ms_per_step'], 'in_flight', {k: round(v, 4) for k, v in d['kernel_ms'].items()}, 'alone', {k: round(v, 4) for k, v in d['kernel_ms_one_call_in_flight'].items()}, 'parity', d['parity']['ok'], d['frames_crc_ok'])"
  done
done

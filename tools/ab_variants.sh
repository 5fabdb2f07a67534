#!/bin/bash
# ab_variants.sh <variant>... -- on the GPU box: the headline call with experimental builds of the library (sora_amd/lib/variants/<v>.so,
# made by sora_amd.build.build_variant; "main" = the product library), one call in flight and the library's default depth.
# Prints ms per step, the kernels' durations alone on the chip, and whether the whole batch still equals the reference graph.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in "$@"; do
  if [ "$v" = main ]; then unset SORA_HIP_LIB; else export SORA_HIP_LIB=$R/sora_amd/lib/variants/$v.so; fi
  for d in ${AB_DEPTHS:-1 3}; do
    timeout 300 python bench.py --no-cpu-baseline --no-extras --depth $d --check 256 --min-seconds 0.5 ${AB_ARGS:-} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$v depth $d ms_per_step', d['ms_per_step'], 'alone', {k: round(v, 4) for k, v in d['kernel_ms_one_call_in_flight'].items()}, 'parity', d['parity']['ok'], d['frames_crc_ok'])"
  done
done

#!/bin/bash
# ab_decode.sh -- on the GPU box: the decode kernel's two sides measured alone (experimental builds of the library; results are
# NOT correct in these builds, only the durations mean something).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in "$@"; do
  SORA_HIP_LIB=$R/sora_amd/lib/variants/$v.so timeout 300 python bench.py --no-cpu-baseline --no-extras --depth 1 --check 8 --min-seconds 0.2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['kernel_ms_one_call_in_flight'])"
done

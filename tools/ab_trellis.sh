#!/bin/bash
# ab_trellis.sh -- on the GPU box: the headline call with either trellis kernel (sora_rx_set_trellis 64 / 16) at several depths.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for t in ${AB_TRELLIS:-64 16}; do
  for d in ${AB_DEPTHS:-1 2 3 4 6 8}; do
    timeout 300 python bench.py --no-cpu-baseline --no-extras --depth $d --trellis $t --check 256 --min-seconds 0.5 ${AB_ARGS:-} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('trellis $t depth $d frames', d['frames'], 'ms_per_step', d['ms_per_step'], 'host', d.get('host_ms_per_step'), 'alone', {k: round(v, 4) for k, v in d['kernel_ms_one_call_in_flight'].items()}, 'parity', d['parity']['ok'], d['frames_crc_ok'])"
  done
done

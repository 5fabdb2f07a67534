#!/bin/bash
# r04_call_size.sh -- on the GPU box: the headline protocol (process_dev -> deliver_async -> wait -> compare) with the same captures cut into calls
# of different size; "frames:depth" pairs in SIZES.  ms per 4096 captures, and the kernels' mean durations while the calls overlap.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for cfg in ${SIZES:-4096:8 8192:4 8192:8 16384:2 16384:4 32768:2 65536:2 65536:4}; do
  IFS=: read f d <<< "$cfg"
  timeout 300 python bench.py --no-cpu-baseline --no-extras --no-plain --frames $f --depth $d --trellis 16 --check 64 --min-seconds 1 ${AB_ARGS:-} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('captures_per_call $f calls_in_flight $d ms_per_call', d['ms_per_step'], 'ms_per_4096', round(d['ms_per_step']*4096/$f, 4), 'host', d.get('host_ms_per_step'), 'in_flight', {k: round(v, 4) for k, v in d['kernel_ms'].items()}, 'wrong', d.get('delivery', {}).get('calls_with_wrong_rows'))"
done

#!/bin/bash
# ab_graph.sh -- on the GPU box: the headline step with sora_rx_set_graph off / on (the binding's harness override SORA_HIP_GRAPH)
for g in 0 1; do
  SORA_HIP_GRAPH=$g timeout 200 python bench.py --no-cpu-baseline --no-extras --check 64 2>/dev/null > /tmp/g$g.json
  python3 -c "
import json,sys
d=json.loads(open('/tmp/g$g.json').read().strip().split(chr(10))[-1])
print('graph', $g, d['ms_per_step'], d['value'], d['host_ms_per_step'], d['parity']['ok'], d['delivery']['calls_with_wrong_rows'])"
done

#!/bin/bash
# r04_clocks.sh -- on the GPU box: the shader clock and power the chip holds while the headline loop runs, per call size / calls in flight
# (rocm-smi sampled every 0.2 s during the timed region).  "frames:depth" pairs in SIZES.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for cfg in ${SIZES:-4096:8 32768:2 4096:1}; do
  IFS=: read f d <<< "$cfg"
  python bench.py --no-cpu-baseline --no-extras --no-plain --headline-only --frames $f --depth $d --check 64 --min-seconds 6 2>/dev/null > /tmp/clk_$f_$d.json &
  BP=$!
  sleep 2
  S=""; P=""
  for i in $(seq 1 24); do
    S="$S $(rocm-smi --showclocks 2>/dev/null | grep -i 'sclk' | head -1 | sed 's/.*(\([0-9]*\)Mhz).*/\1/')"
    P="$P $(rocm-smi --showpower 2>/dev/null | grep -i 'power (W)' | head -1 | sed 's/.*: *\([0-9.]*\) *$/\1/')"
    kill -0 $BP 2>/dev/null || break
  done
  wait $BP
  echo "captures_per_call $f calls_in_flight $d: $(cat /tmp/clk_$f_$d.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_per_call", d["ms_per_step"])')  sclk MHz:$S  power W:$P"
done
rocm-smi --showclocks 2>/dev/null | grep -i sclk | head -2

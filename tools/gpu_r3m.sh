#!/bin/bash
# round 3: does a short timed region (--steps 20) run slower because of what precedes it?  (warm-up length sweep)
set -u
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-copy-peak --no-companion --no-verify"
for w in 5 100 1000 5000; do for k in 20 100; do
  for rep in 1 2 3; do PVE_BENCH_TIMELINE=1 $B --steps $k --warmup $w 2>&1 | grep -E "timeline|ms_per_step" | python -c "
import sys, json
tl = sys.stdin.readline().strip(); d = json.loads(sys.stdin.readline()); print('%.2f' % (d['ms_per_step']*1e3), tl[12:])"; done; echo " <- warmup $w steps $k"
done; done

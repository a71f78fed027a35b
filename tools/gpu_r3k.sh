#!/bin/bash
# round 3: the driver's short protocol (--steps 20 --warmup 5): launch shapes with their host timelines
set -u
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-copy-peak --no-companion --no-verify --steps 20 --warmup 5"
for p in 1 2 4; do for c in 5 10 20; do
  for rep in 1 2 3; do PVE_BENCH_TIMELINE=1 $B --pipeline $p --chunk $c 2>&1 | grep -E "timeline|ms_per_step" | python -c "
import sys, json
tl = sys.stdin.readline().strip(); d = json.loads(sys.stdin.readline()); print('%.2f' % (d['ms_per_step']*1e3), tl[12:])"; done; echo " <- pipeline $p chunk $c"
done; done

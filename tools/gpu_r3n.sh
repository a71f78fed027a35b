#!/bin/bash
# round 3: the driver's short protocol with launches of decreasing length (the tail of the last launch is what is exposed)
set -u
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-copy-peak --no-companion --no-verify --steps 20 --warmup 5"
for seq in "" "5,5,5,5" "7,6,4,3" "8,6,4,2" "6,5,4,3,2" "10,6,4" "6,6,4,2,2" "4,4,4,4,4" "9,7,4" "8,5,3,2,1,1"; do
  for rep in 1 2 3 4; do PVE_BENCH_CHUNKS=$seq $B 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%.2f' % (d['ms_per_step']*1e3), end=' ')"; done; echo " <- launches [$seq]"
done

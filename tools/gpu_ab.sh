#!/bin/bash
# A/B on the GPU box: rebuild libpveenv.so with each set of extra compiler flags and time the given bench shapes.
# usage: AB_VARIANTS="-DX=1|-DX=2" AB_SHAPES="|--mode step|--lane-num 8" bash tools/gpu_ab.sh     ('' = the default shape)
set -u
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-copy-peak --no-companion --no-verify"
IFS='|' read -ra VARS <<< "${AB_VARIANTS:-}"
IFS='|' read -ra SHAPES <<< "${AB_SHAPES:-}"
VARS+=("")
[ ${#SHAPES[@]} -eq 0 ] && SHAPES=("")
for v in "${VARS[@]}"; do
  touch pve-mcc_for_unsignalized_intersection_amd/csrc/pve_hip.hip
  make -s -C pve-mcc_for_unsignalized_intersection_amd/csrc EXTRA="$v" 2>&1 | grep -E "error" 
  for args in "${SHAPES[@]}"; do
    for rep in 1 2 3; do $B $args 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%.2f' % (d['ms_per_step']*1e3), end=' ')"; done; echo " <- [$v] $args"
  done
done

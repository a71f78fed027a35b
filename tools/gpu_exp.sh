#!/bin/bash
# A/B of an experimental library build (build/libpveenv_exp.so) against the product library
set -u
export TMPDIR=/tmp
run() { "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config']['mode'], 'ms/step %.4f' % d['ms_per_step'], 'kern %.4f' % d['roofline']['kernel_ms'], 'alive %.1f' % d['mean_alive_per_env'], 'ovf', d['overflow'])"; }
B="python bench.py --no-cpu-baseline --no-copy-peak"
for lib in "" build/libpveenv_exp.so; do
  echo "lib: ${lib:-product}"
  echo -n "  lanes8 p3: "; PVE_LIBRARY_PATH=$lib run $B --lane-num 8 --pipeline 3 --steps 300
  echo -n "  lanes8 p2: "; PVE_LIBRARY_PATH=$lib run $B --lane-num 8 --pipeline 2 --steps 300
  echo -n "  lanes4 cap64 p2: "; PVE_LIBRARY_PATH=$lib run $B --lane-num 4 --capacity 64 --steps 300
done

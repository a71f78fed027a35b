#!/bin/bash
# A/B of an experimental library build (build/libpveenv_exp.so) against the product library
set -u
export TMPDIR=/tmp
run() { "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config']['mode'], 'ms/step %.4f' % d['ms_per_step'], 'kern %.4f' % d['roofline']['kernel_ms'], 'alive %.1f' % d['mean_alive_per_env'], 'ovf', d['overflow'])"; }
B="python bench.py --no-cpu-baseline --no-copy-peak"
for i in 1 2 3; do
for lib in "" build/libpveenv_exp.so; do
  echo -n "${lib:-product}  rollout: "; PVE_LIBRARY_PATH=$lib run $B --mode rollout
done; done
for lib in "" build/libpveenv_exp.so; do
  echo -n "${lib:-product}  K20: "; PVE_LIBRARY_PATH=$lib run $B --steps 20 --warmup 5
  echo -n "${lib:-product}  cap64: "; PVE_LIBRARY_PATH=$lib run $B --capacity 64
done

#!/bin/bash
# A/B of an experimental library build (build/libpveenv_exp.so) against the product library
set -u
export TMPDIR=/tmp
run() { "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config']['mode'], 'ms/step %.4f' % d['ms_per_step'], 'kern %.4f' % d['roofline']['kernel_ms'], 'alive %.1f' % d['mean_alive_per_env'], 'ovf', d['overflow'])"; }
B="python bench.py --no-cpu-baseline --no-copy-peak"
echo -n "base rollout p2: "; run $B --mode rollout
echo -n "exp  rollout p2: "; PVE_LIBRARY_PATH=build/libpveenv_exp.so run $B --mode rollout
echo -n "exp  rollout p3: "; PVE_LIBRARY_PATH=build/libpveenv_exp.so run $B --mode rollout --pipeline 3

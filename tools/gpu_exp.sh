#!/bin/bash
# A/B of an experimental library build (build/libpveenv_exp.so) against the product library
set -u
export TMPDIR=/tmp
run() { "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config']['mode'], 'ms/step %.4f' % d['ms_per_step'], 'kern %.4f' % d['roofline']['kernel_ms'], 'alive %.1f' % d['mean_alive_per_env'], 'ovf', d['overflow'])"; }
B="python bench.py --no-cpu-baseline --no-copy-peak --actor --steps 300"
echo -n "product: "; run $B
for wv in 4 8 16; do echo -n "exp wv=$wv: "; PVE_ACTOR_WV=$wv PVE_LIBRARY_PATH=build/libpveenv_exp.so run $B; done
timeout 600 env PVE_ACTOR_WV=8 PVE_LIBRARY_PATH=build/libpveenv_exp.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k actor 2>&1 | tail -2

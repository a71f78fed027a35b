#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2n
mkdir -p $O
run() { "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config']['mode'], 'ms/step %.4f' % d['ms_per_step'], 'kern %.4f' % d['roofline']['kernel_ms'], 'alive %.1f' % d['mean_alive_per_env'], 'ovf', d['overflow'])"; }
G="python bench.py --no-cpu-baseline --no-copy-peak --steps 300 --lane-num 8"
for p in 1 2 3 4 6; do echo -n "lanes8 p$p: "; run $G --pipeline $p; done
echo -n "lanes8 p1 2048 envs: "; run $G --pipeline 1 --envs 2048
echo -n "lanes8 p1 2560 envs: "; run $G --pipeline 1 --envs 2560
echo -n "lanes8 p1 1024 envs: "; run $G --pipeline 1 --envs 1024

#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2k
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "geo or general or compat" > $O/pytest_sel.log 2>&1; tail -3 $O/pytest_sel.log
run() { "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config']['mode'], 'ms/step %.4f' % d['ms_per_step'], 'kern %.4f' % d['roofline']['kernel_ms'], 'alive %.1f' % d['mean_alive_per_env'], 'ovf', d['overflow'])"; }
G="python bench.py --no-cpu-baseline --no-copy-peak --steps 300"
echo -n "lanes8 p2: "; run $G --lane-num 8
echo -n "lanes8 p3: "; run $G --lane-num 8 --pipeline 3
echo -n "lanes4 cap64 p2 rate 1200: "; run $G --lane-num 4 --capacity 64 --rate 1200
python tools/phase_profile.py --ticks 100 --lane-num 8 2>&1 | grep -v amdgpu.ids | head -15

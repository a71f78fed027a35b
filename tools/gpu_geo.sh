#!/bin/bash
# geo kernels: timings, phase profile of the 4-lane layout, geo parity tests
set -u
export TMPDIR=/tmp
run() { "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config']['mode'], 'ms/step %.4f' % d['ms_per_step'], 'kern %.4f' % d['roofline']['kernel_ms'], 'alive %.1f' % d['mean_alive_per_env'], 'ovf', d['overflow'])"; }
B="python bench.py --no-cpu-baseline --no-copy-peak"
echo -n "lanes8 p3: "; run $B --lane-num 8 --pipeline 3 --steps 300
echo -n "lanes4 cap64 p2: "; run $B --lane-num 4 --capacity 64 --steps 300
echo -n "lanes4 cap128 p2: "; run $B --lane-num 4 --capacity 128 --steps 300
python tools/phase_profile.py --ticks 100 --lane-num 4 --capacity 64 2>&1 | grep -v amdgpu.ids | head -14
if [ "${1:-}" != "notest" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "geo or general or lanes" 2>&1 | tail -3; fi

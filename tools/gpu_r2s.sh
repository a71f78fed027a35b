#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2s
mkdir -p $O
run() { "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config']['mode'], 'ms/step %.4f' % d['ms_per_step'], 'kern %.4f' % d['roofline']['kernel_ms'], 'alive %.1f' % d['mean_alive_per_env'], 'ovf', d['overflow'])"; }
B="python bench.py --no-cpu-baseline --no-copy-peak"
echo -n "step p2: "; run $B --mode step
echo -n "step p3: "; run $B --mode step --pipeline 3
echo -n "rollout p2 c25: "; run $B --mode rollout
echo -n "rollout p2 c25 wpe5: "; PVE_ROLLOUT_WPE5=1 run $B --mode rollout
echo -n "step K20: "; run $B --mode step --steps 20 --warmup 5
echo -n "cap64 rollout p1: "; run $B --capacity 64 --mode rollout --pipeline 1
echo -n "lanes8 p3: "; run $B --lane-num 8 --pipeline 3 --steps 300
echo -n "actor p3: "; run $B --actor --pipeline 3 --steps 300
python tools/phase_profile.py --ticks 100 2>&1 | grep -v amdgpu.ids | head -14
python tools/phase_profile.py --ticks 100 --many 2>&1 | grep -v amdgpu.ids | head -15
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log

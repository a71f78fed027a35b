#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2g
mkdir -p $O
python tools/phase_profile.py --ticks 100 --lane-num 8 2>&1 | grep -v amdgpu.ids | head -16
python tools/phase_profile.py --ticks 100 --lane-num 8 --geo-scan 2>&1 | grep -v amdgpu.ids | head -16
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "actor" > $O/pytest_sel.log 2>&1; tail -3 $O/pytest_sel.log
run() { "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ms/step %.4f' % d['ms_per_step'], 'kern %.4f' % d['roofline']['kernel_ms'], 'alive %.1f' % d['mean_alive_per_env'], 'ovf', d['overflow'])"; }
B="python bench.py --no-cpu-baseline --no-copy-peak --actor --steps 300"
echo -n "actor step f32 p2: "; run $B --mode step --obs-f32
echo -n "actor step f32 p1: "; run $B --mode step --obs-f32 --pipeline 1
echo -n "actor step f32 p2 grid512: "; PVE_ACTOR_GRID=512 run $B --mode step --obs-f32
echo -n "actor step f32 p2 grid768: "; PVE_ACTOR_GRID=768 run $B --mode step --obs-f32

#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2e
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "actor" > $O/pytest_actor.log 2>&1; tail -5 $O/pytest_actor.log
run() { "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ms/step %.4f' % d['ms_per_step'], 'kern %.4f' % d['roofline']['kernel_ms'], 'alive %.1f' % d['mean_alive_per_env'], 'ovf', d['overflow'])"; }
B="python bench.py --no-cpu-baseline --no-copy-peak --actor --steps 300"
echo -n "actor step f64 p2: "; run $B --mode step
echo -n "actor step f32 p2: "; run $B --mode step --obs-f32
echo -n "actor step f32 p3: "; run $B --mode step --obs-f32 --pipeline 3
echo -n "actor rollout(C loop) f32 p2: "; run $B --mode rollout --obs-f32
echo -n "actor step f32 p1: "; run $B --mode step --obs-f32 --pipeline 1
rocprofv3 --kernel-trace --stats -d $O/stats_actor -o r -- python bench.py --actor --obs-f32 --no-cpu-baseline --no-copy-peak --steps 300 --mode step > /dev/null 2>&1
python tools/rocprof_summary.py $O/stats_actor/r_results.db --tail 300 2>/dev/null | head -30

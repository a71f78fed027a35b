#!/bin/bash
# round 3, first GPU pass: the new parity tests + the bench modes that decide the headline
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { "$@" 2>gpurun_out/err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(d['config']['mode'], 'tpl', d['config']['ticks_per_launch'], 'ms/step %.4f' % d['ms_per_step'], 'kern %.4f' % r['kernel_ms'], 'alive %.1f' % d['mean_alive_per_env'], 'ovf', d['overflow'], 'verified', d['verified'], 'frac %.3f nominal %.3f' % (r['frac'], r['nominal']['frac']))" || tail -5 gpurun_out/err.log; }
B="python bench.py --no-cpu-baseline --no-copy-peak"
echo -n "rollout default: "; run $B
echo -n "rollout traj: "; run $B --trajectory 1
echo -n "rollout traj chunk10: "; run $B --trajectory 1 --chunk 10
echo -n "step: "; run $B --mode step
echo -n "driver rollout: "; run $B --steps 20 --warmup 5
echo -n "driver traj: "; run $B --steps 20 --warmup 5 --trajectory 1
echo -n "driver step: "; run $B --steps 20 --warmup 5 --mode step
echo -n "actor: "; run $B --actor
timeout 1500 python -m pytest tests -m gpu -x -q -k "driver_launch_shape or step_many or native" 2>&1 | tail -4

export TMPDIR=/tmp
echo "== trace, HOME kernel, 20 ticks chunk 12"
TRACE_K=20 TRACE_CHUNK=12 PVE_LIBRARY_PATH=$PWD/build/libpveenv_trace.so python tools/persistent_trace.py 2>&1 | grep -v amdgpu.ids
echo "== launch shapes (HOME, knob build)"
PVE_LIBRARY_PATH=$PWD/build/libpveenv_knobs.so AB_SHAPES="12:6,3;10:5,3;8:5,3;7:4;14:4;12:5,2;9:6,3;12:4,4" python tools/ab_launch_shapes.py 2>&1 | grep -v amdgpu.ids
echo "== grid sweep (HOME), pool tape"
export PVE_LIBRARY_PATH=$PWD/build/libpveenv_knobs.so
for g in 2048 2304 2560; do for s in "--steps 20 --warmup 5" "--steps 1000 --warmup 300"; do
 echo "grid $g [$s] $(PVE_PERSISTENT_GRID=$g python bench.py --no-cpu-baseline --no-copy-peak --no-companion $s 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.2f" % (d["ms_per_step"]*1e3), d.get("verified"))')"
done; done

"""The drop-in `TrafficInteraction` class (kernels run by the CPU test emulator) against the LIVE reference under
pseudo-random action tapes: both objects are driven by the same caller loop (ObjRunner = main.py:398-441) and every
record -- ids, 7x28 states, rewards, neighbour ids, per-vehicle fields, counters -- is compared every tick.
Only runs where /root/reference exists (the build container)."""
import types

import numpy as np
import pytest

from oracle.record import compare_records

pytestmark = pytest.mark.reference


def tape(scale, quant):
    def policy(t, vid, ctl, obs0):
        a = scale * np.sin(1.7 * np.asarray(vid, np.float64) + 0.31 * t + 0.001 * t * t)
        if quant:
            a = np.round(a / quant) * quant
        return a * (np.asarray(ctl) != 0)
    return policy


@pytest.mark.parametrize("lane_num,mean,seed,scale,quant,ticks", [
    (12, 3.0, 71, 3.0, None, 300), (12, 2.4, 72, 2.0, 1.0, 300), (8, 1.8, 73, 3.0, 0.5, 300), (4, 1.4, 74, 2.0, None, 300),
    (4, 1.2, 75, 3.0, 1.0, 300)])
def test_drop_in_class_vs_live_reference_random_tapes(lane_num, mean, seed, scale, quant, ticks):
    from pve_mcc_amd.traffic_interaction_scene import TrafficInteraction
    from tests.golden import ref_harness as rh
    from tests.golden.gen_golden_geo import make_stream
    from tests.hip_adapter import emulator_lib
    arr, choice = make_stream(lane_num, 400, mean, seed)
    policy = tape(scale, quant)
    ref = rh.GeoRefRunner(arr, lane_num, policy, choice=choice, want_state=True)
    try:
        args = types.SimpleNamespace(collision_thr=2, o_agent_num=6, c_mode="closer")
        kw = dict(device="cpu", _lib=emulator_lib())
        if lane_num == 8:
            kw["intentions"] = choice
        env = TrafficInteraction(arr, 150, args, show_col=False, virtual_l=True, lane_num=lane_num, **kw)
        mine = rh.ObjRunner(env, policy, want_state=True)
        for t in range(ticks):
            ra = ref.tick()
            rb = mine.tick(ref.tape)
            compare_records(ra, rb, tol=1e-9, label="drop-in/%d lanes" % lane_num)
    finally:
        ref.close()

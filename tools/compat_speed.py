#!/usr/bin/env python3
"""BASELINE config 1 (plumbing): the drop-in TrafficInteraction class driven like main.py:test() for 1000 ticks of
the 1000-veh/h stream with zero actions; prints ticks/s (the reference Python does ~205 ticks/s on one core)."""
import os
import sys
import time
import types

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pve_mcc_amd.arrivals import load_arrival_mat  # noqa: E402
from pve_mcc_amd.traffic_interaction_scene import TrafficInteraction  # noqa: E402

arr = load_arrival_mat(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "streams",
                                    "arvTimeNewVeh_new_1000_12.mat"))
args = types.SimpleNamespace(collision_thr=2, o_agent_num=6, c_mode="closer")
env = TrafficInteraction(arr, 150, args, show_col=False, virtual_l=True, lane_num=12)
t0 = time.time()
alive = 0
for i in range(1000):
    for lane in range(12):
        for ind, veh in enumerate(env.veh_info[lane]):
            env.step(lane, ind, 0)
            alive += 1
    env.scene_update()
    env.delete_vehicle()
dt = time.time() - t0
print("compat class: 1000 ticks in %.2f s = %.0f ticks/s, %.0f alive-vehicle-steps/s; id_seq=%d passed=%d" % (
    dt, 1000 / dt, alive / dt, env.id_seq, env.passed_veh))

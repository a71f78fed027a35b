"""C-ABI library checks that need no GPU: libpveenv.so loads, exports every symbol that
include/pve_env.h declares, and fails loudly (no CPU fallback) without a device; argument
validation and call-sequence errors (exercised through the emulator build of the same C-ABI source)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import pve_mcc_amd
from pve_mcc_amd import _capi
from pve_mcc_amd.batched import BatchedIntersections
from tests.hip_adapter import emulator_lib

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "pve_env.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pve_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    lib = _capi.load_library()
    syms = header_symbols()
    assert len(syms) >= 17
    for s in syms:
        assert hasattr(lib, s), "libpveenv.so does not export %s" % s
    assert sorted(_capi.EXPORTS) == syms, "binding and header disagree"
    assert lib.pve_abi_version() == _capi.ABI_VERSION
    assert lib.pve_workspace_bytes(4096, 128) > 4096 * 128 * 84
    assert lib.pve_workspace_bytes(10, 100) == 0


def test_struct_sizes_match_header():
    assert C.sizeof(_capi.PveConfig) == 9 * 8 + 8
    assert C.sizeof(_capi.PveOutputs) == 10 * 8
    assert C.sizeof(_capi.PveVehicle) == 7 * 8 + 17 * 4 + 4   # padded to 8
    assert C.sizeof(_capi.PveEnvInfo) == 8 + 4 * (1 + 12 + 12 + 3 + 3 * 16 + 1 + 1)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_fails_loudly_without_gpu():
    lib = _capi.load_library()
    cfg = _capi.PveConfig()
    lib.pve_default_config(C.byref(cfg))
    h = C.c_void_p()
    rc = lib.pve_create(C.byref(cfg), 4, 64, 0, None, None, C.byref(h))
    assert rc == -2, "pve_create must fail with PVE_ERR_NO_DEVICE on a box without a GPU"
    assert b"no CPU fallback" in lib.pve_last_error()
    with pytest.raises(_capi.PveError):
        BatchedIntersections(2, 64, np.full((4, 12), np.inf), device="cpu")


def test_argument_validation_and_call_order():
    lib = emulator_lib()
    cfg = _capi.PveConfig()
    lib.pve_default_config(C.byref(cfg))
    assert (cfg.deltaT, cfg.vm, cfg.vM, cfg.am, cfg.aM, cfg.v0, cfg.lane_cw, cfg.dis_ctl, cfg.collision_thr,
            cfg.lane_num) == (0.1, 5, 13, -3, 3, 10, 2.5, 150, 2, 12)          # ref :21-23, main.py:104
    h = C.c_void_p()
    assert lib.pve_create(C.byref(cfg), 4, 100, 0, None, None, C.byref(h)) == -1
    assert b"capacity" in lib.pve_last_error()
    assert lib.pve_create(C.byref(cfg), 0, 64, 0, None, None, C.byref(h)) == -1
    bad = _capi.PveConfig.from_buffer_copy(cfg)
    bad.lane_num = 3                                       # the 3-lane branch is broken upstream
    assert lib.pve_create(C.byref(bad), 4, 64, 0, None, None, C.byref(h)) == -1
    assert b"lane_num" in lib.pve_last_error()
    assert lib.pve_create(C.byref(cfg), 2, 64, 0, None, None, C.byref(h)) == 0
    ch = np.zeros((4, 12), np.int32)
    assert lib.pve_set_intentions(h, ch.ctypes.data_as(C.c_void_p), 4, 0) == -1      # 8-lane only (ref :389-390)
    assert lib.pve_destroy(h) == 0
    assert lib.pve_create(C.byref(cfg), 2, 64, 0, None, None, C.byref(h)) == 0
    assert lib.pve_reset(h) == -3 and b"pve_set_arrivals" in lib.pve_last_error()
    assert lib.pve_step_all(h, None, None) == -3
    arr = np.full((4, 12), np.inf)
    assert lib.pve_set_arrivals(h, arr.ctypes.data_as(C.c_void_p), 4, 2) == -1
    assert lib.pve_set_arrivals(h, arr.ctypes.data_as(C.c_void_p), 4, 0) == 0
    assert lib.pve_reset(h) == 0
    assert lib.pve_step_all(h, None, None) == 0          # NULL actions = zeros, NULL outputs = none
    info = _capi.PveEnvInfo()
    assert lib.pve_read_env(h, 5, C.byref(info)) == -1
    assert lib.pve_read_env(h, 1, C.byref(info)) == 0 and abs(info.current_time - 0.1 * 200001) < 1e-3
    p, eb = C.c_void_p(), C.c_int()
    assert lib.pve_state_field(h, b"nope", C.byref(p), C.byref(eb)) == -1
    assert lib.pve_state_field(h, b"meta", C.byref(p), C.byref(eb)) == 0 and eb.value == 4
    assert lib.pve_destroy(h) == 0


def test_actor_and_rollout_argument_validation():
    """Round-3 entry points through the emulator build of the C ABI: pve_set_actor / NULL weights, and what pve_step_many
    accepts for the training outputs."""
    lib = emulator_lib()
    arr = np.full((40, 12), np.inf)
    arr[0, :] = 0.05
    b = BatchedIntersections(2, 64, arr, device="cpu", outputs=("obs_post", "reward", "flags"), _lib=lib)
    b.reset()
    obs = torch.zeros(2, 64, 28, dtype=torch.float64)
    act = torch.zeros(2, 64, dtype=torch.float64)
    # no actor installed: NULL weights are refused (PVE_ERR_STATE), explicit weights install them
    assert lib.pve_actor_forward(b._h, None, C.c_void_p(obs.data_ptr()), C.c_void_p(act.data_ptr())) == -3
    assert b"pve_set_actor" in lib.pve_last_error()
    assert lib.pve_set_actor(b._h, None) == -1
    ro = _capi.PveRollout()
    ro.n_ticks, ro.source = 3, _capi.SRC_ACTOR
    ro.actor_obs, ro.actor_actions = obs.data_ptr(), act.data_ptr()
    o = _capi.PveOutputs()
    o.obs_post = obs.data_ptr()
    assert lib.pve_step_many(b._h, C.byref(ro), C.byref(o)) == -3          # still no actor
    w = torch.zeros(_capi.PVE_ACTOR_N_WEIGHTS, dtype=torch.float32)
    assert lib.pve_actor_forward(b._h, C.c_void_p(w.data_ptr()), C.c_void_p(obs.data_ptr()), C.c_void_p(act.data_ptr())) == 0
    assert lib.pve_actor_forward(b._h, None, C.c_void_p(obs.data_ptr()), C.c_void_p(act.data_ptr())) == 0     # installed now
    assert lib.pve_step_many(b._h, C.byref(ro), C.byref(o)) == 0
    # state_pre needs a trajectory roll-out (and obs_pre + the previous rows)
    sp = torch.zeros(3, 2, 64, 7, 28, dtype=torch.float64)
    op = torch.zeros(3, 2, 64, 28, dtype=torch.float64)
    o.state_pre, o.obs_pre, o.obs_prev_post = sp.data_ptr(), op.data_ptr(), torch.zeros(2, 64, 28, dtype=torch.float64).data_ptr()
    ro.source, ro.trajectory = _capi.SRC_ZERO, 0
    assert lib.pve_step_many(b._h, C.byref(ro), C.byref(o)) == -1 and b"trajectory" in lib.pve_last_error()
    # the 4- / 8-lane layouts take obs_pre / state_pre in pve_step_many and float32 rows too (round 4: f3 x f4); the id-indexed
    # table still excludes the training outputs
    arr8 = np.full((40, 8), np.inf)
    g = BatchedIntersections(2, 64, arr8, device="cpu", outputs=("obs_post", "obs_pre", "flags"), lane_num=8, _lib=lib)
    g.reset()
    g.step_many(2, source="zero")
    g.step(None)
    g32 = BatchedIntersections(2, 64, arr8, device="cpu", outputs=("obs_post", "obs_pre"), lane_num=8, _lib=lib, obs_dtype=torch.float32)
    g32.reset()
    g32.step_many(2, source="zero")
    g.set_action_table(torch.zeros(4, 8, dtype=torch.float64))
    with pytest.raises(_capi.PveError):
        g.step_many(2, source="table")
    # |am| outside the range the reciprocal division of the brake test covers
    cfg = _capi.PveConfig()
    lib.pve_default_config(C.byref(cfg))
    cfg.am = -1e-9
    h = C.c_void_p()
    assert lib.pve_create(C.byref(cfg), 2, 64, 0, None, None, C.byref(h)) == -1 and b"|am|" in lib.pve_last_error()


def test_item_schedule_of_the_persistent_call_has_no_tiny_items():
    """pve_debug_item_schedule (the function pve_step_many uses): the items of every call add up to the call, none is longer than
    chunk_ticks, and for calls of >= 8 ticks none is shorter than 2 ticks (the K = 1000, T = 10 default of the steady
    bench shape used to end in 1, 6, 3 -- an item of one tick cannot hide its load and flush)."""
    import bench
    lib = emulator_lib()
    assert bench.item_schedule(20, 12, lib) == [9, 7, 4]
    print(bench.item_schedule(1000, 10, lib)[-5:], bench.item_schedule(100, 10, lib), bench.item_schedule(50, 12, lib))
    assert min(bench.item_schedule(1000, 10, lib)) >= 4
    for T in (4, 5, 6, 7, 10, 12, 25, 64, 255):
        for K in range(T + 1, 1400):
            items = bench.item_schedule(K, T, lib)
            assert sum(items) == K and max(items) <= T, (K, T, items)
            if K >= 8:
                assert min(items) >= 2, (K, T, items)
    out = (C.c_int32 * 12)()
    assert lib.pve_debug_item_schedule(10, 10, out) == -1 and lib.pve_debug_item_schedule(10, 0, out) == -1

"""The CPU oracles against the LIVE reference under random / quantised action tapes (ties, collisions, dead-locks):
the parity tests pin the kernels to the oracles on such tapes, this pins the oracles themselves on them.
Every tick, every field.  Only runs where /root/reference exists (the build container)."""
import numpy as np
import pytest

from oracle.record import compare_records

pytestmark = pytest.mark.reference


def random_tape(seed, scale, quant):
    rng = np.random.default_rng(seed)

    def policy(t, vid, ctl, obs0):
        a = rng.uniform(-scale, scale, size=len(vid)).astype(np.float32).astype(np.float64)
        if quant:
            a = np.round(a / quant) * quant
        return a * (np.asarray(ctl) != 0)
    return policy


@pytest.mark.parametrize("lane_num,mean,seed,scale,quant,ticks", [
    (12, 3.0, 81, 3.0, 1.0, 350), (12, 2.6, 82, 3.0, None, 350), (12, 3.2, 83, 1.0, 0.5, 350),
    (8, 1.9, 84, 3.0, 1.0, 350), (8, 2.2, 85, 2.0, None, 300), (4, 1.5, 86, 3.0, 1.0, 350), (4, 1.3, 87, 3.0, 3.0, 300)])
def test_oracle_vs_live_reference_random_tapes(lane_num, mean, seed, scale, quant, ticks):
    from oracle.oracle_geo import OracleGeoEnv
    from tests.golden import ref_harness as rh
    from tests.golden.gen_golden_geo import make_stream
    arr, choice = make_stream(lane_num, 400, mean, seed)
    ref = rh.GeoRefRunner(arr, lane_num, random_tape(seed, scale, quant), choice=choice, want_state=True)
    try:
        orc = OracleGeoEnv(arr, lane_num, choice=choice)
        coll = lock = 0
        for t in range(ticks):
            ra = ref.tick()
            rb = orc.tick(ref.tape, want_state=True)
            compare_records(ra, rb, tol=1e-12, label="fuzz/%d lanes" % lane_num)
            coll += int(ra["collisions"]); lock += int(ra["lock"])
        assert coll > 0 or lock > 0, "the tape is meant to provoke collisions or dead-locks"
    finally:
        ref.close()

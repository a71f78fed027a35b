"""Dense, every-tick, every-field comparison of the CPU oracle with the LIVE reference.
Only runs where /root/reference exists (the build container)."""
import numpy as np
import pytest

from oracle.oracle import OracleEnv
from oracle.record import compare_records, get_policy

pytestmark = pytest.mark.reference

CASES = [("1000", "zero", 250, {}), ("1000", "sin1", 250, {}), ("200", "sin1", 1300, {}),
         ("1200", "sin1", 200, {}), ("400", "sin2", 1250, {}), ("1000", "sin3", 200, {}),
         ("1000", "sin1", 200, {"vm": 6})]


@pytest.mark.parametrize("stream,pol,ticks,kw", CASES)
def test_oracle_dense_vs_live_reference(stream, pol, ticks, kw):
    from tests.golden import ref_harness as rh
    arr = rh.load_stream(stream)
    policy = get_policy(pol)
    ref = rh.RefRunner(arr, policy, want_state=True, **kw)
    orc = OracleEnv(arr, **kw)
    for t in range(ticks):
        vid, ctl, obs0 = ref.alive_view()
        vid2, ctl2, obs02 = orc.alive_view()
        assert np.array_equal(vid, vid2) and np.array_equal(ctl, ctl2)
        assert np.allclose(obs0, obs02, rtol=0, atol=1e-12)
        acts = policy(t, vid, ctl, obs0)
        ra = ref.tick(acts)
        rb = orc.tick(acts, want_state=True)
        compare_records(ra, rb, tol=1e-12, label="%s/%s" % (stream, pol))
    assert orc.ref_would_raise == 0


GEO_CASES = [(4, "zero", 2.0, 11, 500), (4, "sin3", 1.2, 12, 500), (8, "sin2", 1.5, 13, 500),
             (8, "sin3", 1.2, 14, 400), (12, "sin3", 1.5, 15, 300)]


@pytest.mark.parametrize("lane_num,pol,mean,seed,ticks", GEO_CASES)
def test_geo_oracle_dense_vs_live_reference(lane_num, pol, mean, seed, ticks):
    """oracle/pve_oracle_geo.c vs the live reference with lane_num 4 / 8 / 12 on fresh synthetic streams
    (other seeds than the committed fixtures), every tick, every field, full 7x28 state."""
    from oracle.oracle_geo import OracleGeoEnv
    from tests.golden import ref_harness as rh
    from tests.golden.gen_golden_geo import make_stream
    arr, choice = make_stream(lane_num, 400, mean, seed)
    policy = get_policy(pol)
    ref = rh.GeoRefRunner(arr, lane_num, policy, choice=choice, want_state=True)
    try:
        orc = OracleGeoEnv(arr, lane_num, choice=choice)
        for t in range(ticks):
            ra = ref.tick()
            rb = orc.tick(ref.tape, want_state=True)
            compare_records(ra, rb, tol=1e-12, label="geo%d/%s" % (lane_num, pol))
            assert np.array_equal(ra["intent"], rb["intent"])
            assert ra["intention_re"] == rb["intention_re"]
        assert orc.ref_would_raise == 0
    finally:
        ref.close()

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

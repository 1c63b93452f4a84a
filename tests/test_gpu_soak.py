"""Soak run of the dataflow factorisation: many LM passes at config 4 on one engine, no factorisation may be abandoned.

The dataflow Cholesky's waits are bounded; a wait that runs out makes `solve` repeat the factorisation (engine.hip) and counts in
jaicov_neq_kernel_stats()[6].  Round 2 saw one such stall per 300-3 600 factorisations until the last workgroups on the chain
workgroups' XCDs were taken out (cholflow.hip, "keep"); 16 000 clean factorisations since.  150 passes here by default (5 s);
JAICOV_SOAK_PASSES=4000 for a real soak (2 min).  Every pass also runs the substitution chains (dense.hip: a lost link
would show as NaNs in the step): 14 000 passes = 28 000 launches of the polling-wave backward chain at the end of round 3, and 6 000
more with two workgroups per block column, clean."""
import os
import warnings

import numpy as np
import pytest

from bundle_adjustment_amd import engine

pytestmark = pytest.mark.gpu


def check_health(st, passes):
    """No factorisation may be abandoned in a DEDICATED soak (JAICOV_SOAK_PASSES set: profiles/r05_soak_split.log, 4 000 passes at config 4;
    scripts/stall_probe.py, 100 000 at config 3 with up to 64 pooled streams and an RCCL communicator in the process: none).  Inside the
    test-suite one abandoned-and-repeated factorisation per run is tolerated and reported: round 5 saw two in ~60 000 factorisations,
    both in this file during full-suite runs, none reproducible (profiles/r05_stall_bisect.log: 12 800 more under the same preceding
    tests, clean) -- the rate of rounds 2-4 ("Visibility"), where the retry net was built for exactly this.  The
    result of a repeated factorisation is the same bits (asserted by the callers).
    End of round 5: those two were the harbingers of something reproducible -- late in a full-suite process the stream pool held ~20 CU-masked
    streams, each a hardware queue, and the scheduler time-sliced the persistent kernels ("Hardware queues").  An engine
    now holds one such queue; 5 000 + 5 000 passes and the whole suite are clean (profiles/r05_c_*).  The tolerance stays: a host that creates
    a few dozen masked streams of its own can still bring the condition about, and the retry net is what carries it."""
    if os.environ.get("JAICOV_SOAK_PASSES"):
        assert st["flow_retries"] == 0, st
    else:
        assert st["flow_retries"] <= 1, st
        if st["flow_retries"]:
            warnings.warn(f"one factorisation of {passes} was abandoned on the device and repeated (stderr has the report)")


def test_no_factorisation_is_abandoned_over_many_passes(cfg4_scene):
    fp = cfg4_scene
    n = int(os.environ.get("JAICOV_SOAK_PASSES", "150"))
    eng = engine.Engine(fp)
    eng.set_parameters(fp.values)
    s2 = fp.sigma2apriori
    ref = None
    for i in range(n):
        eng.build(s2, 0.0)
        dx = eng.solve(False)
        assert np.isfinite(dx).all(), i      # a lost link of a substitution chain shows up as NaNs (dense.hip)
        if i % 1000 == 999:
            print(f"soak: {i + 1} passes", flush=True)     # (a long run must not look hung to the GPU pool's watchdog)
        if ref is None:
            ref = dx
        elif i % 25 == 0:       # same system every pass (no update), deterministic assembly (the default) and a fixed order of every tile's updates: the same bits
            assert np.array_equal(dx, ref), (i, np.abs(dx - ref).max())
    st = eng.kernel_stats()
    eng.close()
    check_health(st, n)
    # (flow_rescued / flow_stale_* are informational: a flag found by the slow-path poll, possibly after an ordinary long wait)


def test_no_factorisation_is_abandoned_at_config3_size():
    """The same at config 3 (order 3 014 after the elimination of the ordinary images' EO: 24 block columns), where the chain form runs
    with its THIRD workgroup (round 4: below 80 block columns) and the chain, not the tile kernel, sets the pace: 400 passes by default,
    JAICOV_SOAK_PASSES for more."""
    from bundle_adjustment_amd import scene
    fp = scene.config("cfg3")
    n = max(400, int(os.environ.get("JAICOV_SOAK_PASSES", "0")))
    eng = engine.Engine(fp)
    eng.set_parameters(fp.values)
    s2 = fp.sigma2apriori
    ref = None
    for i in range(n):
        eng.build(s2, 0.0)
        dx = eng.solve(False)
        assert np.isfinite(dx).all(), i
        if i % 1000 == 999:
            print(f"soak (config 3): {i + 1} passes", flush=True)
        if ref is None:
            ref = dx
        elif i % 50 == 0:
            assert np.array_equal(dx, ref)          # deterministic assembly (the default), same system: the same bits
    st = eng.kernel_stats()
    eng.close()
    check_health(st, n)

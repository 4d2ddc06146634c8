"""The wait protocol of the comb walker's slot loop (items pipelined across a workgroup's tickets, smx_agg_v5.hip) in a CPU
model: tools/v5_protocol_sim.py.  No GPU: the shipped period rule comes out of the library (smx_debug_v5_period), the
scheduler is random and unfair.  DESIGN.md 4.1 has the argument why no cycle of waits exists; this is the check that the
argument and the rules in the code (period(), the slot of the completion flag, the slot of the ticket) say the same thing --
and that the model is sharp enough to find the two deadlocks the kernel's bounded waits caught on the GPU."""
import ctypes as C
import os
import random
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import v5_protocol_sim as sim                                    # noqa: E402
from stereo_matching_cuda_amd import _lib                        # noqa: E402


@pytest.fixture(scope="module")
def dbg():
    lib = C.CDLL(_lib.SO_PATH)
    lib.smx_debug_v5_period.restype = C.c_int
    lib.smx_debug_v5_period.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    return lib


def test_period_rule_of_the_library(dbg):
    """period(h, K): even; the whole item (bands + 2) unless it is >= 6 and >= 2 K + 2; never below the last stage-2 slot."""
    for K in range(1, 30):
        for h in list(range(1, 140)) + [375, 376, 2000, 2160]:
            NI, q_last, P = sim.slots(h, K, lib=dbg)
            s1_last = (h + 8) // 10
            assert P % 2 == 0 and P >= q_last and P >= 4
            overlap = max(s1_last + 3, q_last)
            if overlap >= 6 and overlap >= 2 * K + 2:
                assert P == overlap + (overlap & 1) and P - 6 >= 0
            else:
                assert P == NI + 2 + ((NI + 2) & 1) and q_last + 1 <= P - 2   # the completion flag falls into the item's own period
    assert sim.slots(375, 9, lib=dbg) == (42, 41, 42)                    # KITTI


def test_shipped_protocol_never_deadlocks(dbg):
    rng = random.Random(20251005)
    cases = [sim.random_case(rng) for _ in range(1200)]
    # the boundaries of the period rule: K strips around (P - 2) / 2, the shortest overlapping period (K <= 2, h = 22 .. 29),
    # single-band images, more workgroups than items
    for K in range(1, 10):
        for h in (1, 9, 10, 12, 21, 22, 29, 30, 10 * (2 * K - 1) - 8, 10 * (2 * K - 1) + 2, 10 * (2 * K + 2), 95, 130):
            if h >= 1:
                cases.append((h, K, rng.randint(1, 24), rng.randint(2, 20)))
    for h, K, nsv, n_wg in cases:
        r, d = sim.simulate(h, K, nsv, n_wg, rng.randrange(1 << 30), lib=dbg)
        assert r == "done", (h, K, nsv, n_wg, r, d)


@pytest.mark.parametrize("variant", [{"done_rule": "at_switch"}, {"period_rule": "no_bound"}])
def test_model_finds_the_deadlocks_of_round_5(variant):
    """The two rules the GPU taught: a completion flag that waits for the workgroup's next item (h = 1, five strips: 'hand-off
    wait of work item 1319 timed out'), and items that overlap although the period is shorter than 2 K + 2 slots (h = 9)."""
    rng = random.Random(5)
    found = 0
    for _ in range(400):
        h, K, nsv, n_wg = sim.random_case(rng)
        r, _d = sim.simulate(h, K, nsv, n_wg, rng.randrange(1 << 30), **variant)
        found += r == "deadlock"
        assert r in ("done", "deadlock")
    assert found >= 5

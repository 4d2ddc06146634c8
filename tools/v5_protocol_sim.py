"""CPU model of the WAIT PROTOCOL of k_v5_walk's slot loop (smx_agg_v5.hip): workgroups taking strip-major tickets, the
front item's flag waits, the flags stage 2 publishes, the completion flag, the ticket taken a few slots ahead -- nothing of
the data flow (that is tools/v5_model.cpp).  Workgroups advance one slot at a time under a random, deliberately unfair
scheduler (which is what decides who gets which ticket); a run ends in `done` or in a DEADLOCK: every unfinished workgroup
waits for a flag nobody can publish any more.  The model is monotone (flags only grow), so a blocked state is final.

Variants reproduce the two deadlocks the bounded waits of the kernel caught in round 5:
  done_rule  "after_last" (shipped): FLAG_DONE in the slot behind the item's last stage-2 slot
             "at_switch"           : FLAG_DONE only when stage 2 moves to the next item (front slot 1 of the next period)
  fetch_rule "p_minus_6"  (shipped): the next ticket is taken in front slot P - 6
             "early"               : ... in slot -2 when P < 8 (no flag wait in front of it)
  period_rule "shipped": smx_agg_v5.h period() through the library;  "no_bound": max(s1_last + 3, q_last) without the 2 K + 2 rule

usage: python tools/v5_protocol_sim.py [cases] [seed]"""
import random
import sys

DONE = 1 << 30


def slots(h, K, period_rule="shipped", lib=None):
    """(NI, q_last, P) of an image of h rows in K strips."""
    s1_last, q_last = (h + 8) // 10, (h + 37) // 10
    NI = (h + 27) // 10 + 2
    if period_rule == "shipped":
        if lib is not None:
            import ctypes as C
            b, q, p = C.c_int(), C.c_int(), C.c_int()
            assert lib.smx_debug_v5_period(h, K, C.byref(b), C.byref(q), C.byref(p)) == 0
            assert (b.value, q.value) == (NI, q_last)
            return NI, q_last, p.value
        p = max(s1_last + 3, q_last)
        if p < 6 or p < 2 * K + 2:
            p = NI + 2
    else:
        p = max(s1_last + 3, q_last, 4)
    return NI, q_last, p + (p & 1)


class WG:
    __slots__ = ("queue", "nf", "slf", "front", "f_pred", "ending", "own2", "sl2", "pend", "blocked", "finished", "speed")

    def __init__(self, first_ticket, speed):
        self.queue = {0: first_ticket}
        self.nf, self.slf = 0, -2
        self.front, self.f_pred, self.ending = None, False, False
        self.own2, self.sl2, self.pend = None, 0, None
        self.blocked = False        # at the flag wait at the end of the current slot
        self.finished = False
        self.speed = speed


def simulate(h, K, nsv, n_wg, seed, done_rule="after_last", fetch_rule="p_minus_6", period_rule="shipped", lib=None,
             max_steps=2_000_000):
    """-> ("done" | "deadlock", detail)."""
    rng = random.Random(seed)
    NI, q_last, P = slots(h, K, period_rule, lib)
    nitems = K * nsv
    n_wg = min(n_wg, nitems)
    flag = {}                       # item -> published value
    ticket = [n_wg]
    wgs = [WG(i, rng.choice((1, 1, 1, 3, 10, 40))) for i in range(n_wg)]
    fetch_slot = (lambda: P - 6) if fetch_rule == "p_minus_6" else (lambda: P - 6 if P >= 8 else -2)
    sw = 1

    def top_and_work(w):
        if w.slf == -2:
            it = w.queue[w.nf]
            w.ending = it >= nitems
            if w.ending and w.nf == 0:
                w.finished = True
                return
            w.front = None if w.ending else it
            w.f_pred = (not w.ending) and it // nsv > 0
            w.pend = w.front
        if done_rule == "after_last" and w.own2 is not None and w.sl2 == q_last + 1:
            flag[w.own2] = DONE
        if w.slf == sw:
            if done_rule == "at_switch" and w.own2 is not None:
                flag[w.own2] = DONE
            w.own2, w.sl2 = w.pend, sw
        if w.front is not None and w.slf == fetch_slot():
            w.queue[w.nf + 1] = ticket[0]
            ticket[0] += 1
        if w.own2 is not None and 1 <= w.sl2 <= q_last and flag.get(w.own2, 0) != DONE:
            flag[w.own2] = w.sl2

    def wait_ok(w):
        if not (w.f_pred and w.slf + 2 < NI):
            return True
        return flag.get(w.front - nsv, 0) >= w.slf + 3

    def end_of_slot(w):
        if w.ending and w.slf == 1:
            w.finished = True
            return
        w.sl2 += 1
        w.slf += 1
        if w.slf > P - 3:
            w.slf = -2
            w.nf += 1

    steps = 0
    while steps < max_steps:
        live = [w for w in wgs if not w.finished]
        if not live:
            missing = [i for i in range(nitems) if flag.get(i, 0) != DONE]
            return ("done", None) if not missing else ("incomplete", missing[:5])
        moved = False
        rng.shuffle(live)
        for w in live:
            if rng.randrange(w.speed) != 0 and moved:
                continue            # (an unfair scheduler: slow workgroups sit out most rounds)
            if not w.blocked:
                top_and_work(w)
                if w.finished:
                    moved = True
                    continue
                w.blocked = True
            if wait_ok(w):
                w.blocked = False
                end_of_slot(w)
                moved = True
            steps += 1
        if not moved:
            # a full round without progress may be the scheduler's doing: check every live workgroup for real
            stuck = True
            for w in live:
                if not w.blocked or wait_ok(w):
                    stuck = False
                    break
            if stuck:
                who = [(w.front, w.slf, w.own2, w.sl2) for w in live][:6]
                return "deadlock", {"P": P, "NI": NI, "q_last": q_last, "waiting (front item, slot, stage-2 item, slot)": who}
    return "timeout", None


def random_case(rng):
    K = rng.randint(1, 9)
    h = rng.choice((rng.randint(1, 30), rng.randint(1, 120), rng.randint(100, 420)))
    nsv = rng.randint(1, 40)
    n_wg = rng.randint(2, 24)
    return h, K, nsv, n_wg


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for c in range(cases):
        h, K, nsv, n_wg = random_case(rng)
        r, d = simulate(h, K, nsv, n_wg, rng.randrange(1 << 30))
        if r != "done":
            bad += 1
            print(f"h={h} K={K} nsv={nsv} workgroups={n_wg}: {r} {d}")
    print(f"{cases} cases, {bad} not done")
    sys.exit(1 if bad else 0)

"""CPU model of the group-synchronous stopping rule (csrc/ldpc_kernel.hpp, group_decide + the host-side resolution of
csrc/ldpc_hip.hip): the PROTOCOL is checked here under random interleavings -- the kernel's implementation of it is checked
bit for bit against the reference on the GPU (tests/test_ldpc_gpu.py).

Reference rule (lib/ldpc_decoder/layered_decoder.hh:153): a batch of G frames runs `while (bad(any lane) && --trials >= 0) update`,
i.e. all frames stop at T = the first update count at which EVERY frame passes its test (or at the cap).
Protocol (round 5): ONE status word per frame, written only by that frame: after the test at count `it` it stores
(it + 1) << 1 | passed. A frame that failed continues at once. A frame that passed looks at the words of the other members: it continues
as soon as some member is past `it` or failed at `it`, stops when every member passed at `it`, polls again otherwise, and may GIVE UP
waiting (stop at `it`, to be resumed by the host-side resolution: bring every frame to the largest count reached, test there, advance
the whole group by one while some frame fails). A poll sees every word at SOME moment between the previous poll and now, independently
per word (no ordering between words is assumed: single writer, monotone values)."""
import random


def good_at(pattern, it, cap):
    """pattern: set of counts at which this frame's test passes (a frame may pass, fail again, pass later)."""
    return it in pattern


def reference_T(patterns, cap):
    for t in range(cap + 1):
        if all(good_at(p, t, cap) for p in patterns):
            return t, True
    return cap, False


def simulate(patterns, cap, rng, give_up_prob):
    n = len(patterns)
    hist = [[0] for _ in range(n)]                 # every value a member's status word ever held
    seen = [[0] * n for _ in range(n)]             # seen[f][m]: index into hist[m] of the newest value f has observed
    it = [0] * n            # update count of each member
    state = ["test"] * n    # test -> (wait | run) -> ... -> stopped
    stopped_at = [None] * n
    gave_up = False
    while any(s != "stopped" for s in state):
        f = rng.choice([i for i in range(n) if state[i] != "stopped"])
        if state[f] == "test":
            if it[f] >= cap:                       # the cap: no report needed, everybody gets here at the same count
                state[f] = "stopped"; stopped_at[f] = it[f]; continue
            ok = good_at(patterns[f], it[f], cap)
            hist[f].append(((it[f] + 1) << 1) | (1 if ok else 0))
            state[f] = "run" if not ok else "wait"
        elif state[f] == "wait":
            mine = ((it[f] + 1) << 1) | 1
            goes_on = missing = False
            for m in range(n):
                if m == f:
                    continue
                seen[f][m] = rng.randint(seen[f][m], len(hist[m]) - 1)   # any value between the last one seen and the newest
                sv = hist[m][seen[f][m]]
                goes_on |= sv > mine or sv == mine - 1
                missing |= sv < mine - 1
            if goes_on:
                state[f] = "run"
            elif not missing:
                state[f] = "stopped"; stopped_at[f] = it[f]
            elif rng.random() < give_up_prob:
                state[f] = "stopped"; stopped_at[f] = it[f]; gave_up = True
        else:  # run one update
            it[f] += 1
            state[f] = "test"
    return stopped_at, gave_up


def resolve(stopped_at, patterns, cap):
    """Host-side resolution (ldpc_group_targets_kernel + resume launches): returns the group's final count."""
    cur = list(stopped_at)
    while True:
        T = max(cur)
        if any(c != T for c in cur):
            cur = [T] * len(cur)                   # resume the early stoppers to T (no tests on the way)
            continue
        if all(good_at(p, T, cap) for p in patterns) or T >= cap:
            return T
        cur = [T + 1] * len(cur)


def test_group_stop_protocol_matches_the_reference_under_any_interleaving():
    rng = random.Random(2026)
    for trial in range(1500):
        n = rng.choice([1, 2, 3, 8, 32])
        cap = rng.choice([3, 10, 25])
        patterns = []
        for _ in range(n):
            first = rng.randint(0, cap + 2)        # may never pass within the cap
            p = set(range(first, cap + 1))
            if rng.random() < 0.3 and first + 1 <= cap:   # passes once, fails again for a while, passes later
                gap = rng.randint(1, 3)
                p -= set(range(first + 1, min(cap + 1, first + 1 + gap)))
            patterns.append(p)
        T_ref, _ = reference_T(patterns, cap)
        give_up = rng.choice([0.0, 0.0, 0.05, 1.0])
        stopped_at, gave_up = simulate(patterns, cap, rng, give_up)
        # no member ever runs past the reference's count
        assert max(stopped_at) <= T_ref, (trial, stopped_at, T_ref)
        if not gave_up:
            assert stopped_at == [T_ref] * n, (trial, stopped_at, T_ref)   # everybody leaves at the reference's count: no resume pass
        assert resolve(stopped_at, patterns, cap) == T_ref, (trial, stopped_at, T_ref)

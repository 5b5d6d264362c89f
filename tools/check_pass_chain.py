"""Is the order of the tile pass' partner read-modify-writes still fixed when the per-step workgroup barrier is replaced by
"wave w waits for wave w + 1 to have finished its previous visit" (csrc/tbk_eig_band.hip, big_pass, DESIGN_LOG.md R4.16)?

Brute force over the schedules of band_reduce_kernel::big_pass: for every block of sX, the writers in step order (what the
barrier version guarantees) must also be ordered by the chain rule's happens-before relation (program order of a wave, the
neighbour edge (w + 1, v - 1) -> (w, v), the barrier at the end of every round of own blocks).

    python tools/check_pass_chain.py        # prints the (na, NW) pairs with unordered writers: all have na < 2 NW
"""
import sys


def split_shape(na, nw_all):
    """(split, l_split, g_last, r_last, q_last, n_visits) -- the constants of big_pass."""
    n_q = (na + nw_all - 1) // nw_all
    n_t = na // 2
    q_last = n_q - 1
    r_last = na - nw_all * q_last
    g_last = nw_all // r_last
    l_try = (n_t + g_last - 1) // g_last
    split = g_last >= 2 and n_t >= 1 and l_try >= r_last
    l_split = l_try if split else 0
    n_visits = q_last * (n_t + 1) + (l_split + 1 if split else n_t + 1)
    return split, l_split, g_last, r_last, q_last, n_visits


def schedule(na, nw_all, wave_all):
    """(round, slot, active, diag, own block, partner block, in_split, last, fetch, own_valid) per visit -- the arithmetic
    of `request` in big_pass, including the split last round."""
    n_t = na // 2
    split, l_split, g_last, r_last, q_last, n_visits = split_shape(na, nw_all)
    v_last0 = q_last * (n_t + 1)
    h_i, h_j = wave_all % r_last, wave_all // r_last
    out = []
    for v in range(n_visits):
        in_split = split and v >= v_last0
        if not in_split:
            q, t = divmod(v, n_t + 1)
            slot = t
            a_raw = wave_all + nw_all * q
            own_ok = a_raw < na
            tile = own_ok
            fetch = own_ok and t == 0
            last = t == n_t
        else:
            q = q_last
            tau = v - v_last0
            slot = tau
            a_raw = nw_all * q_last + h_i
            own_ok = h_j < g_last
            t = 0 if tau == 0 else h_j * l_split + tau
            tile = own_ok and (h_j == 0 if tau == 0 else t <= n_t)
            fetch = own_ok and tau == 0
            last = tau == l_split
            t = min(t, n_t)
        a = min(a_raw, na - 1)
        active = tile and not ((na & 1) == 0 and t == n_t and t > 0 and a_raw >= n_t)
        a2 = a + t
        if a2 >= na:
            a2 -= na
        out.append((q, slot, active, t == 0, a, a2, in_split, last, fetch, own_ok))
    return out


def coverage_errors(na, nw_all):
    """Every tile {I, J} of the triangle of na blocks exactly once over all waves; a wave never works on a block whose
    operands it has not fetched in this round."""
    seen = {}
    errors = []
    for w in range(nw_all):
        own = None
        for (q, slot, active, diag, a, a2, in_split, last, fetch, own_ok) in schedule(na, nw_all, w):
            if fetch:
                own = (q, a)
            if active:
                if own != (q, a):
                    errors.append(("operands not fetched", w, q, slot, a))
                key = (min(a, a2), max(a, a2))
                seen[key] = seen.get(key, 0) + 1
    for i in range(na):
        for j in range(i, na):
            if seen.get((i, j), 0) != 1:
                errors.append(("tile count", i, j, seen.get((i, j), 0)))
    return errors


def violations(na, n_waves, members):
    """Pairs of consecutive writers of one block (inside one workgroup: its own copy of sX) the chain rule leaves unordered."""
    nw_all = n_waves * members
    n_t = na // 2
    bad = []

    def happens_before(w1, v1, w2, v2):
        q1, q2 = v1 // (n_t + 1), v2 // (n_t + 1)
        if q1 != q2:
            return q1 < q2                      # the barrier at the end of a round
        if w1 == w2:
            return v1 < v2                      # program order
        return w1 > w2 and v2 - v1 >= w1 - w2   # a path of neighbour edges (each: wave - 1, visit + 1) and program order

    for member in range(members):
        sched = [schedule(na, nw_all, member * n_waves + w) for w in range(n_waves)]
        writers = {}
        for w in range(n_waves):
            for v, (q, slot, active, diag, _own, partner, in_split, _last, _fetch, _ok) in enumerate(sched[w]):
                if active and not diag:
                    writers.setdefault((q, partner), []).append((slot, w, v, in_split))
        for key, lst in writers.items():
            lst.sort()
            for (t1, w1, v1, s1), (t2, w2, v2, s2) in zip(lst, lst[1:]):
                # a split round keeps the barrier per slot: its writers only have to sit in different slots
                if t1 == t2 or (not s1 and not happens_before(w1, v1, w2, v2)):
                    bad.append((key, (t1, w1), (t2, w2)))
    return bad


def chain_allowed(na, n_waves):
    """The kernel's condition for the chain (`chain` in big_pass)."""
    return na >= 2 * n_waves


def main():
    seen = {}
    for na in range(1, 65):
        for n_waves in (2, 4, 8):
            for members in (1, 2, 3, 4, 8):
                cov = coverage_errors(na, n_waves * members)
                if cov:
                    print("COVERAGE", na, n_waves, members, cov[:3])
                    return 1
                n_bad = len(violations(na, n_waves, members))
                if n_bad:
                    seen[(na, n_waves)] = seen.get((na, n_waves), 0) + n_bad
    for (na, n_waves), count in sorted(seen.items()):
        print("na = %2d, %d waves: %d unordered pairs (chain %s)" % (na, n_waves, count,
                                                                    "ALLOWED -- BUG" if chain_allowed(na, n_waves) else "not used"))
    wrong = [key for key in seen if chain_allowed(*key)]
    print("schedules checked: na 1..64 x 2/4/8 waves x 1/2/3/4/8 member workgroups; wrongly allowed: %d" % len(wrong))
    return 1 if wrong else 0


if __name__ == "__main__":
    sys.exit(main())

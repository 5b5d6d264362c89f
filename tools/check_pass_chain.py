"""Is the order of the tile pass' partner read-modify-writes still fixed when the per-step workgroup barrier is replaced by
"wave w waits for wave w + 1 to have finished its previous visit" (csrc/tbk_eig_band.hip, big_pass, DESIGN_LOG.md R4.16)?

Brute force over the schedules of band_reduce_kernel::big_pass: for every block of sX, the writers in step order (what the
barrier version guarantees) must also be ordered by the chain rule's happens-before relation (program order of a wave, the
neighbour edge (w + 1, v - 1) -> (w, v), the barrier at the end of every round of own blocks).

    python tools/check_pass_chain.py        # prints the (na, NW) pairs with unordered writers: all have na < 2 NW
"""
import sys


def schedule(na, nw_all, wave_all):
    """(q, t, active, diag, own block, partner block) per visit -- the arithmetic of `request` in big_pass."""
    n_q = (na + nw_all - 1) // nw_all
    n_t = na // 2
    out = []
    for v in range(n_q * (n_t + 1)):
        q, t = divmod(v, n_t + 1)
        a_raw = wave_all + nw_all * q
        a = min(a_raw, na - 1)
        active = a_raw < na and not ((na & 1) == 0 and t == n_t and t > 0 and a_raw >= n_t)
        a2 = a + t
        if a2 >= na:
            a2 -= na
        out.append((q, t, active, t == 0, a, a2))
    return out


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
            for v, (q, t, active, diag, _own, partner) in enumerate(sched[w]):
                if active and not diag:
                    writers.setdefault((q, partner), []).append((t, w, v))
        for key, lst in writers.items():
            lst.sort()
            for (t1, w1, v1), (t2, w2, v2) in zip(lst, lst[1:]):
                if t1 == t2 or not happens_before(w1, v1, w2, v2):
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

"""Is the tile pass' partner read-modify-write order still fixed when the per-step barrier is replaced by 'wave w waits for
wave w + 1 to have finished its previous visit'?  Brute force over the schedules of band_reduce_kernel::big_pass."""
import itertools, sys

def schedule(na, nw_all, wave_all):
    n_q = (na + nw_all - 1) // nw_all
    n_t = na // 2
    out = []  # per visit: (q, t, active, diag, I, I2)
    for v in range(n_q * (n_t + 1)):
        q, t = divmod(v, n_t + 1)
        a_raw = wave_all + nw_all * q
        a = min(a_raw, na - 1)
        active = a_raw < na and not ((na & 1) == 0 and t == n_t and t > 0 and a_raw >= n_t)
        a2 = a + t
        if a2 >= na: a2 -= na
        out.append((q, t, active, t == 0, a, a2))
    return out, n_t

def check(na, NW, members):
    nw_all = NW * members
    bad = 0
    for member in range(members):
        sch = [schedule(na, nw_all, member * NW + w)[0] for w in range(NW)]
        n_t = na // 2
        nv = len(sch[0])
        # events: RMW of partner block by (w, v) if active and not diag.  time model: logical clocks from the chain rule.
        # order relation: (w', v') happens-before (w, v) iff reachable through: program order; neighbor edge (w+1, v-1) -> (w, v)
        # for t >= 1; round barriers.  Reachability inside one round: (w', v') -> (w, v) iff v' <= v and w' >= w and (v - v') >= (w' - w)... 
        def hb(w1, v1, w2, v2):
            q1, q2 = v1 // (n_t + 1), v2 // (n_t + 1)
            if q1 != q2: return q1 < q2
            if w1 == w2: return v1 < v2
            # path from (w1, v1) to (w2, v2): each neighbor edge goes w+1 -> w and v-1 -> v; program order raises v
            return w1 > w2 and v2 - v1 >= w1 - w2
        writers = {}
        for w in range(NW):
            for v, (q, t, active, diag, I, I2) in enumerate(sch[w]):
                if active and not diag:
                    writers.setdefault((q, I2), []).append((t, w, v))
        for key, lst in writers.items():
            lst.sort()
            for (t1, w1, v1), (t2, w2, v2) in zip(lst, lst[1:]):
                if t1 == t2:
                    bad += 1; print('same step!', na, NW, members, key, lst)
                elif not hb(w1, v1, w2, v2):
                    bad += 1; print('unordered', na, NW, members, key, (t1, w1), (t2, w2))
    return bad

total = 0
for na in range(1, 65):
    for NW in (2, 4, 8):
        for members in (1, 2, 3, 4, 8):
            total += check(na, NW, members)
print('violations', total)

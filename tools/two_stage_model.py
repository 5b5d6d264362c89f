#!/usr/bin/env python3
"""
NumPy model of the two-stage Hermitian tridiagonalisation of csrc/tbk_eig_band.hip (64 < n <= 512):

    stage 1   dense -> band of half-width B = 8: panels of 8 rows, Householder QR of the panel, two-sided
              compact-WY update of the trailing matrix as ONE pass over its 16 x 16 tiles per panel
              (update with the previous panel's (V, W) + product with the next panel's V in the same pass)
    stage 2   band -> tridiagonal: Householder bulge chasing on the band (length-8 reflectors, 8 x 8 blocks)

This file is the executable statement of the data flow the kernels follow (same index conventions, same
order of phases, the same formulas for the reflector from Gram sums); `python tools/two_stage_model.py`
checks it against numpy.linalg.eigvalsh.  It is design tooling: nothing in the product imports it.
Reference step it replaces: scipy.linalg.eigvalsh per k-point, /root/reference/src/tbmodels/_tb_model.py:1147-1150.
"""

import numpy as np

B = 8      # band half-width = panel height
TS = 16    # MFMA tile edge


# ------------------------------------------------------------------------------------------------
# stage 1
# ------------------------------------------------------------------------------------------------
def panel_qr(y):
    """
    Householder QR of y (m x 8, rows = 'threads'), with every reflector built from ONE round of sums over the rows:
    g[c'] = sum_{i >= c} conj(y[i, c]) y[i, c'] plus the broadcast row c.  Returns V (m x 8, unit lower trapezoidal),
    tau (8), R rows (8 x 8 upper triangular, rows >= m are absent).
    """
    m = y.shape[0]
    y = y.copy()
    V = np.zeros((m, B), dtype=complex)
    tau = np.zeros(B, dtype=complex)
    for c in range(B):
        if c > m - 2:  # no row below the diagonal: nothing to eliminate (tau = 0)
            continue
        g = np.array([np.vdot(y[c:, c], y[c:, cp]) for cp in range(B)])  # one reduction round
        row = y[c, :].copy()  # broadcast by the thread that owns row c
        alpha = row[c]
        gcc = g[c].real
        sigma = gcc - abs(alpha) ** 2
        if gcc == 0.0 or (sigma == 0.0 and alpha.imag == 0.0):
            continue
        beta = -np.copysign(np.sqrt(gcc), alpha.real)
        tau[c] = (beta - alpha) / beta
        scale = 1.0 / (alpha - beta)
        v = y[:, c] * scale
        v[:c] = 0.0
        v[c] = 1.0
        V[:, c] = v
        # z[c'] = v^H y[:, c'] from the Gram sums
        z = np.conj(scale) * (g - np.conj(alpha) * row) + row
        for cp in range(c + 1, B):
            y[c:, cp] -= np.conj(tau[c]) * v[c:] * z[cp]
        y[c, c] = beta
        y[c + 1:, c] = 0.0
    return V, tau, y[:min(m, B), :]


# Threshold of the Gram-matrix panel QR below: a column whose remaining norm^2 g = G_cc - sum_{i<c} |R_ic|^2 has cancelled
# below GRAM_THRESH x G_cc ends the round -- the rows apply the reflectors found so far and a fresh Gram matrix is formed.
# Errors of the reflectors found from differences scale with eps x sqrt(G_cc / g): <= 8 eps at 1 / 64.
GRAM_THRESH = 1.0 / 64.0
GRAM_STATS = {"rounds": 0, "panels": 0, "columns": 0}


def panel_qr_gram(y, thresh=None):
    """
    The panel QR of csrc/tbk_eig_band.hip with TBK_PANEL_GRAM (round 5): ALL reflectors of a round from ONE Gram matrix
    G = P^H P of the panel's rows (formed on the matrix pipe, one meeting of the workgroup) and the explicitly tracked top
    rows -- the inner products every later reflector needs follow from G, which the reflectors leave invariant, minus the
    finished rows R[i]:  g_c[t] = G[c][t] - sum_{i<c} conj(R[i][c]) R[i][t].  Every wave runs the 8-step recurrence on
    8 x 8 data for itself; the rows then apply the reflectors of the round without meeting in between.  g_c[c] is a
    difference: when it has cancelled below `thresh` x G[c][c] the round ends in front of column c, the rows bring the
    panel up to date and the next round starts from a fresh Gram matrix of the rows and columns that are left (a column
    right behind a fresh Gram matrix never triggers: there is nothing to subtract).  Same return values as panel_qr.
    """
    if thresh is None:
        thresh = GRAM_THRESH
    m = y.shape[0]
    y = y.copy()
    V = np.zeros((m, B), dtype=complex)
    tau = np.zeros(B, dtype=complex)
    last = min(B, m - 1)  # columns c <= m - 2 have a row below the diagonal
    GRAM_STATS["panels"] += 1
    GRAM_STATS["columns"] += max(last, 0)
    c0 = 0
    while c0 < last:
        GRAM_STATS["rounds"] += 1
        G = y[c0:, :].conj().T @ y[c0:, :]           # ONE reduction round (rows above c0 are finished)
        top = y[c0:min(m, B), :].copy()              # rows c0 .. 7 of the panel, tracked by every wave for itself
        R = {}
        steps = []                                   # (c, scale, f, tau_c, beta) handed to the rows
        c = c0
        while c < last:
            g = G[c, :].copy()
            for i in range(c0, c):
                g -= np.conj(R[i][c]) * R[i]
            gcc = g[c].real
            if c > c0 and not gcc >= thresh * G[c, c].real:
                break                                # cancelled: this round ends in front of column c
            row = top[c - c0].copy()
            alpha = row[c]
            sigma = gcc - abs(alpha) ** 2
            if gcc == 0.0 or (sigma == 0.0 and alpha.imag == 0.0):
                R[c] = row                           # H_c = I: the row stays, tau = 0, v = 0
                steps.append((c, 0.0, np.zeros(B, dtype=complex), 0.0, None))
                c += 1
                continue
            beta = -np.copysign(np.sqrt(gcc), alpha.real)
            tau_c = (beta - alpha) / beta
            scale = 1.0 / (alpha - beta)
            z = np.conj(scale) * (g - np.conj(alpha) * row) + row
            f = np.conj(tau_c) * z
            f[:c + 1] = 0.0
            Rc = row - f
            Rc[c] = beta
            Rc[:c] = 0.0
            R[c] = Rc
            for i in range(c + 1, min(m, B)):        # the tracked rows below
                vtop = top[i - c0][c] * scale
                top[i - c0] = top[i - c0] - vtop * f
                top[i - c0][c] = 0.0
            steps.append((c, scale, f, tau_c, beta))
            c += 1
        # the rows' side, every row for itself
        for (cc, scale, f, tau_c, beta) in steps:
            if beta is None:
                continue
            v = y[:, cc] * scale
            v[:cc] = 0.0
            v[cc] = 1.0
            V[:, cc] = v
            tau[cc] = tau_c
            y[cc:, :] -= np.outer(v[cc:], f)
            y[cc, cc] = beta
            y[cc + 1:, cc] = 0.0
        c0 = c
    return V, tau, y[:min(m, B), :]


def t_factor(V, tau):
    """Upper triangular T of the compact WY form Q = I - V T V^H from the Gram matrix G = V^H V (one reduction round)."""
    G = V.conj().T @ V
    T = np.zeros((B, B), dtype=complex)
    for c in range(B):
        T[c, c] = tau[c]
        if c:
            T[:c, c] = -tau[c] * (T[:c, :c] @ G[:c, c])
    return T


def big_pass(A, n, s_zero_rows, VW, Vn):
    """
    One pass over the 16 x 16 tiles (I <= J) of the stored upper triangle that touch rows / columns >= 16*floor(s/16):
    tile -= VW_I . rot(VW_J)^H   (VW = [V | W] per row; rows < s are zero, so finished band rows are rewritten unchanged)
    X_I += tile Vn_J,  X_J += tile^H Vn_I  (J != I).  Tiles are walked in the kernel's cyclic order.
    """
    nbk = (n + TS - 1) // TS
    I0 = s_zero_rows // TS
    na = nbk - I0
    X = np.zeros((nbk * TS, B), dtype=complex)
    VWp = np.zeros((nbk * TS, 2 * B), dtype=complex)
    VWp[:n] = VW
    Vnp = np.zeros((nbk * TS, B), dtype=complex)
    Vnp[:n] = Vn
    WVp = np.concatenate([VWp[:, B:], VWp[:, :B]], axis=1)  # [W | V]: the k-rotated partner operand
    Ap = np.full((nbk * TS, nbk * TS), np.nan + 0j)
    Ap[:n, :n] = A
    Ap[:n, n:] = 0.0
    Ap[n:, :] = 0.0

    def visit(I, J):  # I <= J
        ri, rj = slice(TS * I, TS * I + TS), slice(TS * J, TS * J + TS)
        tile = Ap[ri, rj].copy()
        tile -= VWp[ri] @ WVp[rj].conj().T
        if I == J:
            iu = np.triu_indices(TS)
            full = np.zeros_like(tile)
            full[iu] = tile[iu]
            full = full + np.triu(full, 1).conj().T  # mirror of the stored upper part
            Ap[ri, rj][iu] = tile[iu]
            up = Ap[ri, rj]
            up[iu] = tile[iu]
            X[ri] += full @ Vnp[rj]
        else:
            Ap[ri, rj] = tile
            X[ri] += tile @ Vnp[rj]
            X[rj] += tile.conj().T @ Vnp[ri]

    for t in range(na // 2 + 1):
        for a in range(na):
            if na % 2 == 0 and t == na // 2 and a >= na // 2:
                continue
            a2 = (a + t) % na
            if t == 0:
                visit(I0 + a, I0 + a)
            else:
                visit(I0 + min(a, a2), I0 + max(a, a2))
    A[:, :] = Ap[:n, :n]
    return X[:n]


def stage1_band(H):
    """Upper-triangle-only reduction of Hermitian H to a band of half-width B; returns the band as an (n, B+1) array
    band[i, d] = H'[i, i+d]."""
    n = H.shape[0]
    A = np.triu(H).astype(complex)
    A[np.tril_indices(n, -1)] = np.nan  # nothing may read the lower triangle
    VW = np.zeros((n, 2 * B), dtype=complex)  # previous panel's [V | W], global row index
    n_panels = 0
    p = 0
    while True:
        s = B * p + B  # first row of the panel's QR = start of the trailing matrix after it
        m = n - s
        if m < 2:
            break
        g_rows = np.arange(B * p, B * p + B)
        # --- look-ahead: block row p (8 rows x columns >= 8p), brought up to date with the pending (V, W)
        cols = np.arange(B * p, n)
        x = np.zeros((len(cols), B), dtype=complex)  # thread i holds column i of the block row
        for t, i in enumerate(cols):
            for r, g in enumerate(g_rows):
                x[t, r] = A[g, i] if i >= g else np.conj(A[i, g])
        Vg, Wg = VW[g_rows, :B], VW[g_rows, B:]  # the 8 pending rows (broadcast)
        for t, i in enumerate(cols):
            x[t, :] -= Vg @ VW[i, B:].conj() + Wg @ VW[i, :B].conj()
        # diagonal block: final
        for t, i in enumerate(cols[:B]):
            for r, g in enumerate(g_rows):
                if g <= i:
                    A[g, i] = x[t, r]
        # --- panel QR on threads i >= s: y = conj(x)
        y = x[B:, :].conj()
        V, tau, R = panel_qr(y)
        for c in range(R.shape[0]):  # thread s + c writes its column of the block row: conj(R[c, r]) for r >= c
            for r in range(B):
                A[g_rows[r], s + c] = np.conj(R[c, r]) if r >= c else 0.0
        A[g_rows[0]:g_rows[0] + B, s + R.shape[0]:] = 0.0  # (the kernel leaves these stale: nobody reads them)
        T = t_factor(V, tau)
        Vn = np.zeros((n, B), dtype=complex)
        Vn[s:] = V
        # --- the pending rows were consumed by the look-ahead: zero them, then the big pass
        VW[:s, :] = 0.0
        Xr = big_pass(A, n, s, VW, Vn)
        # --- W of this panel
        X = Xr @ T
        X[:s] = 0.0
        S = T.conj().T @ (Vn.conj().T @ X)
        W = X - 0.5 * Vn @ S
        VW = np.concatenate([Vn, W], axis=1)
        n_panels += 1
        p += 1
    # final pass: apply the last pending update (no look-ahead took any of its rows)
    big_pass(A, n, B * p, VW, np.zeros((n, B), dtype=complex))
    band = np.zeros((n, B + 1), dtype=complex)
    for i in range(n):
        for d in range(min(B, n - 1 - i) + 1):
            band[i, d] = A[i, i + d]
    # everything outside the band must be negligible where it was defined
    return band, A


# ------------------------------------------------------------------------------------------------
# stage 2
# ------------------------------------------------------------------------------------------------
def larfg(x):
    """LAPACK zlarfg: (beta, v, tau) with (I - tau v v^H)^H x = beta e_1, v[0] = 1."""
    alpha = x[0]
    sigma = np.vdot(x[1:], x[1:]).real
    v = np.zeros_like(x)
    v[0] = 1.0
    if sigma == 0.0 and alpha.imag == 0.0:
        return alpha.real, v, 0.0
    beta = -np.copysign(np.sqrt(abs(alpha) ** 2 + sigma), alpha.real)
    tau = (beta - alpha) / beta
    v[1:] = x[1:] / (alpha - beta)
    return beta, v, tau


def stage2_tridiag(band):
    """
    Band (upper storage band[i, d] = H[i, i+d]) -> real tridiagonal (d, e) by bulge chasing.  Works on the LOWER band
    L[j, dd] = H[j+dd, j] = conj(band[j, dd]) with room for the bulge: dd in [0, 2B).
    """
    n = band.shape[0]
    L = np.zeros((n + 2 * B, 2 * B), dtype=complex)  # padded columns: the kernel's LDS array
    L[:n, :B + 1] = band.conj()

    def get(i, j):  # lower element (i >= j)
        return L[j, i - j]

    def put(i, j, v):
        L[j, i - j] = v

    for j in range(n - 2):
        # ---- step 0: eliminate column j below the sub-diagonal
        r0, r1 = j + 1, min(j + B, n - 1)
        x = np.array([get(i, j) for i in range(r0, r1 + 1)])
        if len(x) >= 2:
            beta, v, tau = larfg(x)
            put(r0, j, beta)
            for i in range(r0 + 1, r1 + 1):
                put(i, j, 0.0)
        else:
            v, tau = np.ones(1, dtype=complex), 0.0
        while True:
            nb = r1 - r0 + 1
            # two-sided on the diagonal block D = A[r0:r1+1, r0:r1+1]:  D <- H^H D H, H = I - tau v v^H
            D = np.zeros((nb, nb), dtype=complex)
            for a in range(nb):
                for b_ in range(a + 1):
                    D[a, b_] = get(r0 + a, r0 + b_)
                    D[b_, a] = np.conj(D[a, b_])
            for a in range(nb):
                D[a, a] = D[a, a].real
            if tau != 0.0:
                # H^H D H with H^H = I - conj(tau) v v^H
                y = D @ v
                rho = np.vdot(v, y).real
                w = tau * y - 0.5 * abs(tau) ** 2 * rho * v  # D <- D - v w^H - w v^H ... derive below
                # H^H D H = D - conj(tau) v (v^H D) - tau (D v) v^H + |tau|^2 rho v v^H
                D = D - np.conj(tau) * np.outer(v, y.conj()) - tau * np.outer(y, v.conj()) + abs(tau) ** 2 * rho * np.outer(v, v.conj())
            for a in range(nb):
                for b_ in range(a + 1):
                    put(r0 + a, r0 + b_, D[a, b_])
            # block below: rows q0..q1, columns r0..r1:  Bk <- Bk H  (right-apply), creates the bulge
            q0, q1 = r1 + 1, min(r1 + B, n - 1)
            if q0 > n - 1:
                break
            nq = q1 - q0 + 1
            Bk = np.zeros((nq, nb), dtype=complex)
            for a in range(nq):
                for b_ in range(nb):
                    if (q0 + a) - (r0 + b_) < 2 * B:
                        Bk[a, b_] = get(q0 + a, r0 + b_)
            if tau != 0.0:
                Bk = Bk - tau * np.outer(Bk @ v, v.conj())
            # next reflector from the first column of the bulge block
            x = Bk[:, 0].copy()
            if nq >= 2:
                beta, v2, tau2 = larfg(x)
                Bk[0, 0] = beta
                Bk[1:, 0] = 0.0
                if tau2 != 0.0:  # left-apply H2^H to the remaining columns
                    z = v2.conj() @ Bk[:, 1:]
                    Bk[:, 1:] -= np.conj(tau2) * np.outer(v2, z)
            else:
                v2, tau2 = np.ones(1, dtype=complex), 0.0
            for a in range(nq):
                for b_ in range(nb):
                    if (q0 + a) - (r0 + b_) < 2 * B:
                        put(q0 + a, r0 + b_, Bk[a, b_])
            r0, r1, v, tau = q0, q1, v2, tau2
    d = L[:n, 0].real.copy()
    e_c = L[:n - 1, 1].copy()
    return d, np.abs(e_c), L


def tridiag_eigvals(d, e):
    import scipy.linalg as la

    return la.eigvalsh_tridiagonal(d, e)


def main():
    rng = np.random.default_rng(0)
    for n in (65, 72, 80, 96, 100, 129, 200, 256):
        M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        H = (M + M.conj().T) / 2
        band, A = stage1_band(H)
        off = 0.0
        for i in range(n):
            for jj in range(i + B + 1, n):
                if not np.isnan(A[i, jj]):
                    off = max(off, abs(A[i, jj]))
        Hb = np.zeros((n, n), dtype=complex)
        for i in range(n):
            for dd in range(min(B, n - 1 - i) + 1):
                Hb[i, i + dd] = band[i, dd]
                Hb[i + dd, i] = np.conj(band[i, dd])
        ref = np.linalg.eigvalsh(H)
        e1 = np.abs(np.linalg.eigvalsh(Hb) - ref).max()
        d, e, _ = stage2_tridiag(band)
        e2 = np.abs(tridiag_eigvals(d, e) - ref).max()
        print("n=%3d  off-band residue %.1e  band eig err %.2e  tridiag eig err %.2e" % (n, off, e1, e2))
        assert e1 < 1e-12 * n and e2 < 1e-12 * n


if __name__ == "__main__":
    main()


# ------------------------------------------------------------------------------------------------
# stage 2 as the kernel schedules it: sweeps pipelined `stagger` ticks apart, the steps of one tick concurrent
# ------------------------------------------------------------------------------------------------
def stage2_pipelined(band, stagger, n_waves=10 ** 6):
    """Runs the chase in lockstep ticks; the steps of a tick read the state the tick started from.  Asserts that no
    step of a tick touches (reads or writes) a cell another step of the same tick writes."""
    n = band.shape[0]
    NPAD = n + 2 * B
    L = np.zeros((NPAD, 2 * B), dtype=complex)
    L[:n, :B + 1] = band.conj()

    class View:  # cell access with read / write sets, on a snapshot
        def __init__(self, snap):
            self.snap, self.reads, self.writes = snap, set(), {}

        def get(self, i, j):
            if i >= n or j >= n:
                return 0.0
            self.reads.add((i, j))
            return self.writes.get((i, j), self.snap[j, i - j])

        def put(self, i, j, v):
            if i < n and j < n:
                self.writes[(i, j)] = v

    def chase_tick(view, j, k, state):
        """Tick k of sweep j (band_chase_kernel): returns the reflector for the next tick."""
        if k == 0:
            x = np.array([view.get(j + 1 + a, j) for a in range(B)])
            beta, v, tau = larfg(x)
            view.put(j + 1, j, beta)
            for a in range(1, B):
                view.put(j + 1 + a, j, 0.0)
        else:
            v, tau = state
        r0 = j + 1 + B * k
        q0 = r0 + B
        D = np.zeros((B, B), dtype=complex)
        for a in range(B):
            for b_ in range(a + 1):
                D[a, b_] = view.get(r0 + a, r0 + b_)
                D[b_, a] = np.conj(D[a, b_])
        for a in range(B):
            D[a, a] = D[a, a].real
        y = D @ v
        rho = np.vdot(v, y).real
        D = D - np.conj(tau) * np.outer(v, y.conj()) - tau * np.outer(y, v.conj()) + abs(tau) ** 2 * rho * np.outer(v, v.conj())
        for a in range(B):
            for b_ in range(a + 1):
                view.put(r0 + a, r0 + b_, D[a, b_])
        Bk = np.array([[view.get(q0 + a, r0 + b_) for b_ in range(B)] for a in range(B)])
        Bk = Bk - tau * np.outer(Bk @ v, v.conj())
        beta, v2, tau2 = larfg(Bk[:, 0].copy())
        z = v2.conj() @ Bk
        Bk = Bk - np.conj(tau2) * np.outer(v2, z)
        Bk[0, 0] = beta
        Bk[1:, 0] = 0.0
        for a in range(B):
            for b_ in range(B):
                view.put(q0 + a, r0 + b_, Bk[a, b_])
        return v2, tau2

    n_sweeps = n - 2
    length = [(n - 1 - j + B - 1) // B for j in range(n_sweeps)]
    start = []
    for s in range(n_sweeps):
        t0 = 0 if s == 0 else start[s - 1] + stagger
        if s >= n_waves:
            t0 = max(t0, start[s - n_waves] + length[s - n_waves])
        start.append(t0)
    total = start[-1] + length[-1] if n_sweeps > 0 else 0
    states = {}
    for tick in range(total):
        active = [s for s in range(n_sweeps) if start[s] <= tick < start[s] + length[s]]
        snap = L.copy()
        views = []
        for s in active:
            view = View(snap)
            states[s] = chase_tick(view, s, tick - start[s], states.get(s))
            views.append(view)
        for i_, va in enumerate(views):
            for j_, vb in enumerate(views):
                if i_ != j_:
                    clash = (va.reads | set(va.writes)) & set(vb.writes)
                    assert not clash, "tick %d: sweeps %d and %d share cells %s" % (tick, active[i_], active[j_], sorted(clash)[:4])
        for view in views:
            for (i, j), val in view.writes.items():
                L[j, i - j] = val
    return L[:n, 0].real.copy(), np.abs(L[:n - 1, 1]), total


def stage2_window(band, n_slots, window, stagger=2, plan_only=False):
    """The pipelined chase of stage2_pipelined with the working diagonals in a CYCLIC WINDOW of `window` columns (the kernel: LDS)
    in front of a backing store (the kernel: global memory) -- the layout of band_chase4w_kernel for more orbitals than the
    LDS holds columns (csrc/tbk_eig_band.hip).  Sweep s runs in slot s % n_slots; the sweeps of generation g = s // n_slots
    see column j at window column (j + off[g]) % window with off[g + 1] = off[g] + (NE - n_slots (g + 1)), NE = n + B: the
    top of generation g + 1 follows the bottom of generation g in the window, as it does in time.  A column enters the window
    one tick before the generation's first sweep needs it and leaves the tick after its last sweep touched it; from the first
    generation whose columns all fit the window on nothing leaves any more.  Asserts: every access finds ITS column in the
    window, a column only enters a free cell, and the result equals stage2_pipelined's bit for bit.  Returns (d, e, ticks)."""
    n = band.shape[0]
    NE = n + B
    G = np.zeros((NE, 2 * B), dtype=complex)  # backing store, padded columns are zero
    G[:n, :B + 1] = band.conj()
    win = np.zeros((window, 2 * B), dtype=complex)
    tag = [None] * window  # (column, offset) a cell of the window holds

    n_sweeps = n - 2
    length = [(n - 1 - j + B - 1) // B for j in range(n_sweeps)]
    n_gen = (n_sweeps + n_slots - 1) // n_slots
    g_res = next((g for g in range(n_gen) if NE - n_slots * g <= window), n_gen)  # first generation that stays
    # a slot whose generation LEAVES the window is taken again no sooner than gap_min ticks after it started: the columns that
    # generation leaves must be back in the backing store before the next one fetches them
    gap_min = 2 * n_slots + 4
    start = []
    for s in range(n_sweeps):
        t0 = 0 if s == 0 else start[s - 1] + stagger
        if s >= n_slots:
            prev = s - n_slots
            t0 = max(t0, start[prev] + (max(length[prev], gap_min) if prev // n_slots < g_res else length[prev]))
        start.append(t0)
    total = start[-1] + length[-1] if n_sweeps > 0 else 0
    off = [0] * (n_gen + 1)
    for g in range(1, n_gen + 1):
        off[g] = off[g - 1] + ((NE - n_slots * g) if g <= g_res else 0)

    def cell(g, j):
        return (j + off[g]) % window

    def needed(g, j):  # first tick at which generation g touches column j (its sweep 0)
        base = n_slots * g
        return start[base] + max(0, (j - base - 1) // B)

    def last_touch(g, j):
        base = n_slots * g
        s_l = min(base + n_slots, n_sweeps) - 1
        if j <= s_l:
            return start[j]
        return start[s_l] + (j - s_l - 1) // B

    fetch_at, evict_at = {}, {}
    for g in range(min(g_res + 1, n_gen)):
        for j in range(n_slots * g, NE):
            fetch_at.setdefault(needed(g, j) - 1, []).append((g, j))
            if g < g_res:
                evict_at.setdefault(last_touch(g, j) + 1, []).append((g, j))

    def fetch(tick):
        for g, j in fetch_at.get(tick, ()):
            c = cell(g, j)
            assert tag[c] is None, "tick %d: column %d of generation %d enters a cell that still holds %r" % (tick, j, g, tag[c])
            win[c] = G[j]
            tag[c] = (j, off[g])

    class View:
        def __init__(self, g):
            self.g, self.reads, self.writes = g, set(), {}

        def get(self, i, j):
            assert j < NE and 0 <= i - j < 2 * B
            c = cell(self.g, j)
            assert tag[c] == (j, off[self.g]), "column %d of generation %d is not in the window (%r)" % (j, self.g, tag[c])
            self.reads.add((i, j))
            return self.writes.get((i, j), win[c, i - j])

        def put(self, i, j, v):
            if i < n and j < n:
                c = cell(self.g, j)
                assert tag[c] == (j, off[self.g])
                self.writes[(i, j)] = v

    def chase_tick(view, j, k, state):  # stage2_pipelined's, reading the padded cells too (the kernel does)
        if k == 0:
            x = np.array([view.get(j + 1 + a, j) for a in range(B)])
            beta, v, tau = larfg(x)
            view.put(j + 1, j, beta)
            for a in range(1, B):
                view.put(j + 1 + a, j, 0.0)
        else:
            v, tau = state
        r0 = j + 1 + B * k
        q0 = r0 + B
        D = np.zeros((B, B), dtype=complex)
        for a in range(B):
            for b_ in range(a + 1):
                D[a, b_] = view.get(r0 + a, r0 + b_)
                D[b_, a] = np.conj(D[a, b_])
        for a in range(B):
            D[a, a] = D[a, a].real
        y = D @ v
        rho = np.vdot(v, y).real
        D = D - np.conj(tau) * np.outer(v, y.conj()) - tau * np.outer(y, v.conj()) + abs(tau) ** 2 * rho * np.outer(v, v.conj())
        for a in range(B):
            for b_ in range(a + 1):
                view.put(r0 + a, r0 + b_, D[a, b_])
        Bk = np.array([[view.get(q0 + a, r0 + b_) for b_ in range(B)] for a in range(B)])
        Bk = Bk - tau * np.outer(Bk @ v, v.conj())
        beta, v2, tau2 = larfg(Bk[:, 0].copy())
        z = v2.conj() @ Bk
        Bk = Bk - np.conj(tau2) * np.outer(v2, z)
        Bk[0, 0] = beta
        Bk[1:, 0] = 0.0
        for a in range(B):
            for b_ in range(B):
                view.put(q0 + a, r0 + b_, Bk[a, b_])
        return v2, tau2

    stage2_window.plan = ({t: sorted(v) for t, v in fetch_at.items()}, {t: sorted(v) for t, v in evict_at.items()})
    if plan_only:
        return None
    fetch(-1)
    states = {}
    peak = 0
    for tick in range(total):
        active = [s for s in range(n_sweeps) if start[s] <= tick < start[s] + length[s]]
        views = []
        for s in active:
            view = View(s // n_slots)
            states[s] = chase_tick(view, s, tick - start[s], states.get(s))
            views.append(view)
        for view in views:
            for (i, j), val in view.writes.items():
                win[cell(view.g, j), i - j] = val
        fetch(tick)
        for g, j in evict_at.get(tick, ()):
            c = cell(g, j)
            assert tag[c] == (j, off[g])
            if j < n:
                G[j] = win[c]
            tag[c] = None
        peak = max(peak, sum(t is not None for t in tag))
    for c in range(window):  # what stayed
        if tag[c] is not None and tag[c][0] < n:
            G[tag[c][0]] = win[c]
    stage2_window.peak_columns = peak
    return G[:n, 0].real.copy(), np.abs(G[:n - 1, 1]), total


def window_plan_by_trackers(n, n_slots, window):
    """Which columns enter and leave the cyclic window at which tick, found the way band_chase4w_kernel finds them: three running
    trackers (the generation whose first sweep leads, the next sweep whose own column leaves, the generation whose last sweep
    trails) instead of a table over all columns.  Returns (fetch_at, evict_at): tick -> sorted list of (generation, column);
    tests/test_two_stage_model.py holds it equal to the table stage2_window builds, for every orbital count the kernels see."""
    NE = n + B
    n_sweeps = n - 2
    length = [(n - 1 - j + B - 1) // B for j in range(n_sweeps)]
    n_gen = (n_sweeps + n_slots - 1) // n_slots
    g_res = (NE - window + n_slots - 1) // n_slots if NE > window else 0
    gap_min = 2 * n_slots + 4
    start = []
    for s in range(n_sweeps):
        t0 = 0 if s == 0 else start[s - 1] + 2
        if s >= n_slots:
            prev = s - n_slots
            t0 = max(t0, start[prev] + (max(length[prev], gap_min) if prev // n_slots < g_res else length[prev]))
        start.append(t0)
    total = start[-1] + length[-1]
    last_fetch_gen = min(g_res, n_gen - 1)
    fetch_at, evict_at = {}, {}
    fetch_at[-1] = [(0, j) for j in range(min(9, NE))]
    g_in, t0_in, s_ev, g_out = 0, 0, 0, 0
    for tick in range(total):
        nt = tick + 1
        got = []
        if g_in < last_fetch_gen and nt >= start[n_slots * (g_in + 1)]:
            j_old = n_slots * g_in + 1 + B * (nt - t0_in)
            got += [(g_in, j) for j in range(j_old, min(j_old + 8, NE))]
            g_in += 1
            t0_in = start[n_slots * g_in]
        kk = nt - t0_in
        if kk >= 0 and g_in <= last_fetch_gen:
            base = n_slots * g_in
            j_lo = base if kk == 0 else base + 1 + B * kk
            got += [(g_in, j) for j in range(j_lo, min(base + 9 + B * kk, NE))]
        if got:
            fetch_at[tick] = sorted(got)
        out = []
        if s_ev < n_slots * g_res and s_ev < n_sweeps and start[s_ev] + 1 == tick:
            out.append((s_ev // n_slots, s_ev))
            s_ev += 1
        while g_out < g_res:
            s_l = n_slots * g_out + n_slots - 1
            ks = tick - 1 - start[s_l]
            if ks < 0:
                break
            j_lo = s_l + 1 + B * ks
            if j_lo >= NE:
                g_out += 1
                continue
            out += [(g_out, j) for j in range(j_lo, min(j_lo + 8, NE))]
            if j_lo + B < NE:
                break
            g_out += 1
        if out:
            evict_at[tick] = sorted(out)
    return fetch_at, evict_at


def check_pipeline():
    rng = np.random.default_rng(3)
    for n in (40, 67, 100):
        M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        H = (M + M.conj().T) / 2
        band, _ = stage1_band(H)
        ref = np.linalg.eigvalsh(H)
        for stagger, waves in ((3, 10 ** 6), (2, 10 ** 6), (2, 4), (3, 2)):
            d, e, ticks = stage2_pipelined(band, stagger, waves)
            err = np.abs(tridiag_eigvals(d, e) - ref).max()
            print("n=%3d stagger %d waves %7d: %4d ticks, eig err %.2e" % (n, stagger, waves, ticks, err))
            assert err < 1e-12 * n
        try:
            stage2_pipelined(band, 1)
            print("n=%3d stagger 1: no clash (unexpected)" % n)
        except AssertionError as exc:
            print("n=%3d stagger 1 clashes as expected: %s" % (n, str(exc)[:60]))

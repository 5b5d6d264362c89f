"""Model of the NEXT step for the two-stage reduction's panel QR (not in the kernels yet): all eight reflectors of
a panel from ONE Gram matrix G = P^H P plus the explicitly tracked top 8 x 8 block, with an exact round of sums only for a
column whose remaining norm^2  g = G_cc - sum_{i<c} |R_ic|^2  has cancelled below `thresh` x G_cc.

    python tools/gram_panel_model.py [thresh]      (from the repository root; thresh 1e-4 by default, 0 = never an exact round)

Runs tools/two_stage_model.py's first stage with both panel factorisations on the structured matrices of
tests/test_gpu_parity.py::test_eigensolver_structured_matrices (+ two nearly rank-deficient ones) and prints the scaled
eigenvalue error of the resulting band and how many columns needed the exact round.  Round 4: worst error 4.1e-15 with the
threshold (the one-round-per-reflector form: 4.9e-15), 8.7e-2 without it ("two equal blocks"), random matrices never need
an exact round (smallest g / G_cc 0.1 - 0.4)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import two_stage_model as M

B = M.B
stats = {"fallback": 0, "steps": 0, "worst_ratio": 1.0}


def panel_qr_gram(y, thresh=1e-4):
    m = y.shape[0]
    y = y.copy()
    V = np.zeros((m, B), dtype=complex)
    tau = np.zeros(B, dtype=complex)
    G = y.conj().T @ y                      # ONE reduction round (matrix pipe)
    top = y[:min(m, B), :].copy()            # rows 0..7, tracked explicitly by the leader
    coeff = []                               # per step: (c, tau_c, scale_c, z_c) broadcast to the rows
    for c in range(B):
        if c > m - 2:
            continue
        stats["steps"] += 1
        g = G[c, :].copy()
        for i in range(c):
            g -= np.conj(top[i, c]) * top[i, :]
        row = top[c, :].copy()
        alpha = row[c]
        gcc = g[c].real
        ratio = gcc / G[c, c].real if G[c, c].real > 0 else 1.0
        if G[c, c].real > 0 and ratio < thresh:
            # the exact round for this column: the rows bring column c.. up to date and sum again
            stats["fallback"] += 1
            ycur = y.copy()
            for (cc, t_, s_, z_, v_) in coeff:
                for cp in range(cc + 1, B):
                    ycur[cc:, cp] -= np.conj(t_) * v_[cc:] * z_[cp]
                ycur[cc + 1:, cc] = 0.0
            g = np.array([np.vdot(ycur[c:, c], ycur[c:, cp]) for cp in range(B)])
            gcc = g[c].real
        stats["worst_ratio"] = min(stats["worst_ratio"], ratio)
        sigma = gcc - abs(alpha) ** 2
        if gcc <= 0.0 or (sigma <= 0.0 and alpha.imag == 0.0):
            coeff.append((c, 0.0, 0.0, np.zeros(B, dtype=complex), np.zeros(m, dtype=complex)))
            continue
        beta = -np.copysign(np.sqrt(gcc), alpha.real)
        tau[c] = (beta - alpha) / beta
        scale = 1.0 / (alpha - beta)
        z = np.conj(scale) * (g - np.conj(alpha) * row) + row
        # the rows' side (every row on its own, no meeting): bring column c up to date, form v, apply
        ycol = y[:, c].copy()
        for (cc, t_, s_, z_, v_) in coeff:
            ycol[cc:] -= np.conj(t_) * v_[cc:] * z_[c]
        v = ycol * scale
        v[:c] = 0.0
        v[c] = 1.0
        V[:, c] = v
        coeff.append((c, tau[c], scale, z, v))
        # leader: update the tracked top rows
        for i in range(c, top.shape[0]):
            for cp in range(c + 1, B):
                top[i, cp] -= np.conj(tau[c]) * v[i] * z[cp]
        top[c, c] = beta
        top[c + 1:, c] = 0.0
    return V, tau, top


def run(n, name, mat, thresh):
    ref = np.linalg.eigvalsh(mat)
    scale = max(1e-300, np.abs(ref).max())
    out = {}
    for label, fn in (("seq", M.panel_qr), ("gram", lambda y: panel_qr_gram(y, thresh))):
        M_panel = M.panel_qr
        M.panel_qr = fn
        try:
            band, _ = M.stage1_band(mat)
            Hb = np.zeros((n, n), dtype=complex)
            for i in range(n):
                for dd in range(min(B, n - 1 - i) + 1):
                    Hb[i, i + dd] = band[i, dd]
                    Hb[i + dd, i] = np.conj(band[i, dd])
            ev = np.linalg.eigvalsh(Hb)
            out[label] = np.abs(np.sort(ev) - ref).max() / scale
        finally:
            M.panel_qr = M_panel
    return out


def main():
    thresh = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-4
    worst = 0.0
    for n in (40, 64, 97):
        rng = np.random.default_rng(100 + n)
        rand = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        rand = (rand + rand.conj().T) / 2
        cases = {
            "diagonal": np.diag(rng.standard_normal(n)).astype(complex),
            "identity": np.eye(n, dtype=complex) * 0.75,
            "random": rand,
            "graded": rand * np.outer(10.0 ** -np.arange(n) / max(1, n // 8), np.ones(n)),
            "imag_offdiag": np.diag(np.arange(n, dtype=float)) + 1j * (np.eye(n, k=1) - np.eye(n, k=-1)),
            "tiny": rand * 1e-30, "huge": rand * 1e30,
        }
        cases["graded"] = (cases["graded"] + cases["graded"].conj().T) / 2
        blk = np.zeros((n, n), dtype=complex); h = n // 2
        blk[:h, :h] = rand[:h, :h]; blk[h:, h:] = rand[:n - h, :n - h]
        cases["two_equal_blocks"] = blk
        cases["rank_one"] = np.outer(rand[:, 0], rand[:, 0].conj())
        # nearly dependent columns: a low-rank matrix plus noise
        lowrank = rand[:, :3] @ rand[:, :3].conj().T
        cases["rank3_plus_1e-9"] = lowrank + 1e-9 * rand
        cases["rank3_plus_1e-5"] = lowrank + 1e-5 * rand
        for name, mat in cases.items():
            stats.update(fallback=0, steps=0, worst_ratio=1.0)
            res = run(n, name, mat, thresh)
            worst = max(worst, res["gram"])
            print("n=%3d %-18s seq %.2e  gram %.2e  (exact rounds %d of %d steps, smallest g/G %.1e)" % (
                n, name, res["seq"], res["gram"], stats["fallback"], stats["steps"], stats["worst_ratio"]))
    print("worst gram error", worst)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""
Random shapes against the oracle (GPU box): orbital counts around every kernel boundary, odd batch sizes, 1-4 lattice
dimensions, dense / sparse storage, both conventions, mesh and random k-points, host and list / array returns.

    python tools/fuzz_parity.py [seconds] [seed] [big]

Prints one line per case that exceeds 1e-10 and a summary; exit status 1 if any case failed.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tbmodels_amd  # noqa: E402
from tbmodels_amd import synthetic as syn  # noqa: E402
from oracle import tbk_oracle as oracle  # noqa: E402  (checker)

budget = 120.0
big = False  # orbital counts up to and past the 512 limit of the wave solvers
max_cases = None
rng = np.random.default_rng(1)


def configure(seconds=120.0, seed=1, big_sizes=False, cases=None):
    """Set the run's budget, seed, size class and (optionally) a case limit -- also the entry point for tests."""
    global budget, big, max_cases, rng  # pylint: disable=global-statement
    budget, big, max_cases = float(seconds), bool(big_sizes), cases
    rng = np.random.default_rng(seed)

N_EDGES = [1, 2, 3, 4, 7, 8, 9, 12, 13, 15, 16, 17, 24, 31, 32, 33, 40, 48, 63, 64, 65, 66, 80, 96, 127, 128, 129, 150,
           191, 192, 193, 200, 255, 256, 257, 300, 383, 384, 385]
NK_EDGES = [1, 2, 3, 7, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 257, 500, 1000, 4095, 4096, 4097, 5000, 8193, 12289]


def pick_case():
    # (1025 - 1600: the launch chain of band_xl_*, round 5)
    n = int(rng.choice([300, 383, 384, 385, 448, 449, 511, 512, 513, 520, 600, 700, 1000, 1024, 1025, 1040, 1200, 1600] if big else N_EDGES))
    dim = int(rng.choice([1, 2, 3, 3, 3, 4]))
    box = {1: 12, 2: 12, 3: 12, 4: 3}[dim]
    max_r = {1: 13, 2: 313, 3: 3000, 4: 1000}[dim]
    n_r = int(min(max_r, rng.choice([0, 1, 2, 5, 7, 8, 9, 16, 33, 64, 100, 257, 600])))
    n_r = int(min(n_r, max(1, 3e7 // (n * n))))  # <= ~0.5 GB of hoppings
    # keep the oracle's scratch (NK x n x n complex, twice) under ~1.5 GB and its time in seconds
    nk_cap = max(1, int(1.5e7 / (n * n)))
    nk = int(min(nk_cap, rng.choice(NK_EDGES)))
    if n * n * max(n_r, 1) * nk > 1e9:
        nk = max(1, int(1e9 / (n * n * max(n_r, 1))))
    sparse = bool(rng.integers(0, 2))
    mesh = bool(rng.integers(0, 4) == 0) and dim >= 2
    # exact zeros, exact degeneracies and exact cancellations: small-integer / real / diagonal hoppings, and
    # k-points on multiples of 1/4 (phases exactly 0, +-1, +-i)
    kind = str(rng.choice(["random", "random", "integers", "real", "diagonal", "permutation"]))
    special_k = bool(rng.integers(0, 3) == 0)
    return dict(n=n, dim=dim, box=box, n_r=n_r, nk=nk, sparse=sparse, mesh=mesh, kind=kind, special_k=special_k)


def build(case):
    n, dim, n_r = case["n"], case["dim"], case["n_r"]
    r_vec = syn.half_space_vectors(n_r, dim=dim, box=case["box"]) if n_r else np.zeros((0, dim), dtype=np.int32)
    scale = float(rng.choice([1e-6, 1.0, 1.0, 1.0, 1e3]))
    kind = case["kind"]
    if kind == "integers":
        hop = scale * (rng.integers(-2, 3, (n_r, n, n)) + 1j * rng.integers(-2, 3, (n_r, n, n))).astype(complex)
        hop *= rng.random((n_r, n, n)) < rng.choice([0.02, 0.1, 0.5])
    elif kind == "real":
        hop = scale * rng.standard_normal((n_r, n, n)).astype(complex) / np.sqrt(max(n_r, 1) * n)
    elif kind == "diagonal":
        hop = np.zeros((n_r, n, n), dtype=complex)
        idx = np.arange(n)
        hop[:, idx, idx] = scale * rng.integers(-3, 4, (n_r, n))
    elif kind == "permutation":
        hop = np.zeros((n_r, n, n), dtype=complex)
        for r in range(n_r):
            hop[r, np.arange(n), rng.permutation(n)] = scale * rng.choice([1.0, -1.0, 1j, 0.5])
    else:
        hop = scale * (rng.standard_normal((n_r, n, n)) + 1j * rng.standard_normal((n_r, n, n))) / np.sqrt(max(n_r, 1) * n)
    if case["sparse"] and kind in ("random", "real"):
        hop *= rng.random((n_r, n, n)) < rng.choice([0.02, 0.2, 0.6])
    if n_r:
        hop[0] = (hop[0] + hop[0].conj().T) / 4.0
    pos = rng.random((n, dim))
    if case["mesh"]:
        per = max(2, int(round(case["nk"] ** (1.0 / dim))))
        axes = [np.linspace(0, 1, per, endpoint=False) + rng.random() for _ in range(dim)]
        k = np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1).reshape(-1, dim)
    else:
        k = (rng.random((case["nk"], dim)) - 0.5) * float(rng.choice([1.0, 4.0]))
    if case["special_k"]:
        k = np.round(k * 4) / 4
    return r_vec, hop, pos, np.ascontiguousarray(k), scale


def kdotp_case():
    """A random k.p model (and the k.p expansion of a random tight-binding model) against the oracle."""
    import itertools  # pylint: disable=import-outside-toplevel

    from tbmodels_amd.kdotp import KdotpModel  # pylint: disable=import-outside-toplevel

    n = int(rng.choice([1, 2, 4, 8, 17, 33, 64, 70]))
    dim = int(rng.choice([1, 2, 3]))
    order = int(rng.choice([0, 1, 2, 3]))
    powers = [p for p in itertools.product(range(order + 1), repeat=dim) if sum(p) <= order]
    if len(powers) > 3 and rng.integers(0, 2):
        powers = [powers[i] for i in sorted(rng.choice(len(powers), size=len(powers) // 2, replace=False))]
    coeffs = rng.standard_normal((len(powers), n, n)) + 1j * rng.standard_normal((len(powers), n, n))
    coeffs = coeffs + coeffs.conj().transpose(0, 2, 1)
    nk = int(rng.choice([1, 2, 33, 500, 5000]))
    k = (rng.random((nk, dim)) - 0.5) * float(rng.choice([0.1, 1.0, 3.0]))
    model = KdotpModel({p: c for p, c in zip(powers, coeffs)})
    pw = np.array(powers, dtype=np.int64).reshape(len(powers), dim)
    h_ref = oracle.kdotp_hamilton(pw, coeffs, k)
    e_ref = np.array(oracle.kdotp_eigenval(pw, coeffs, k))
    size = max(1.0, float(np.abs(h_ref).sum(axis=-1).max()))
    err = max(float(np.abs(model.hamilton(k) - h_ref).max()), float(np.abs(np.array(model.eigenval(k)) - e_ref).max()),
              float(np.abs(model.eigenval(k[0]) - e_ref[0]).max()))
    # construct_kdotp of a tight-binding model, dense and sparse
    n_r = int(rng.choice([1, 3, 20])) if dim > 1 else int(rng.choice([1, 3, 4]))
    r_vec = syn.half_space_vectors(n_r, dim=dim, box=3)
    hop = rng.standard_normal((n_r, n, n)) + 1j * rng.standard_normal((n_r, n, n))
    hop[0] = (hop[0] + hop[0].conj().T) / 4
    k0 = rng.random(dim)
    tb = tbmodels_amd.Model.from_packed(r_vec, hop, sparse=bool(rng.integers(0, 2)), size=n, dim=dim)
    got = tb.construct_kdotp(k0, order)
    pw2, cf2 = oracle.construct_kdotp(r_vec, hop, k0, order)
    scale2 = max(1.0, float(np.abs(cf2).max()))
    err2 = max(float(np.abs(got.taylor_coefficients[tuple(int(x) for x in p)] - c).max()) for p, c in zip(pw2, cf2))
    return dict(kdotp=True, n=n, dim=dim, order=order, nk=nk, n_r=n_r), err / size, err2 / scale2


def main():
    t_end = time.time() + budget
    n_cases = 0
    worst = 0.0
    failures = []
    while time.time() < t_end and (max_cases is None or n_cases < max_cases):
        if rng.integers(0, 6) == 0:
            case, err_a, err_b = kdotp_case()
            n_cases += 1
            worst = max(worst, err_a, err_b)
            if max(err_a, err_b) > 1e-10:
                failures.append((case, err_a, err_b))
                print("FAIL", case, err_a, err_b, flush=True)
            continue
        case = pick_case()
        r_vec, hop, pos, k, scale = build(case)
        model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos, sparse=case["sparse"], size=case["n"], dim=case["dim"])
        norm = float(np.abs(hop).sum(axis=(0, 2)).max()) if hop.size else 0.0  # bound on ||H||_inf / 2
        tol = 1e-10 * max(1.0, scale, norm)
        try:
            e_gpu = np.array(model.eigenval(k)).reshape(len(k), case["n"])
            e_ref = np.array(oracle.eigenval(r_vec, hop, k, n_orb=case["n"])).reshape(len(k), case["n"])
            err_e = float(np.abs(e_gpu - e_ref).max())
            n_h = min(len(k), max(1, int(2e6 / (case["n"] ** 2))))
            conv = int(rng.choice([1, 2]))
            h_gpu = model.hamilton(k[:n_h], convention=conv)
            h_ref = oracle.hamilton(r_vec, hop, k[:n_h], conv, pos=pos, n_orb=case["n"])
            err_h = float(np.abs(h_gpu - h_ref).max())
            single = np.abs(model.eigenval(k[0]) - e_ref[0]).max()
        except Exception as exc:  # pylint: disable=broad-except
            failures.append((case, repr(exc)))
            print("EXCEPTION", case, repr(exc), flush=True)
            continue
        n_cases += 1
        rel = max(err_e, err_h, single) / max(1.0, scale, norm)
        worst = max(worst, rel)
        if max(err_e, err_h, single) > tol:
            failures.append((case, err_e, err_h, single))
            print("FAIL", case, "scale", scale, "eig", err_e, "ham", err_h, "single", single, flush=True)
    print("cases %d  worst scaled error %.3g  failures %d" % (n_cases, worst, len(failures)))
    return 1 if failures else 0


if __name__ == "__main__":
    configure(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
              len(sys.argv) > 3 and sys.argv[3] == "big")
    sys.exit(main())

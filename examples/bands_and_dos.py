#!/usr/bin/env python3
"""
Band structure along a k path, a k.p expansion beside it, and a density of states from a uniform mesh -- the
pattern of the reference's `examples/kdotp/run.py` (single-k calls in a loop) plus the batched calls this package
is built for.  Runs on the silicon model of the reference's test-suite (tests/golden/cli_eigenvals); needs a GPU.

    python examples/bands_and_dos.py [model.hdf5]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tbmodels_amd  # noqa: E402


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "cli_eigenvals", "silicon_model.hdf5")
    model = tbmodels_amd.io.load(path)
    print(model)

    # 1. the reference's example: one k-point per call along a line, tight-binding against its k.p expansion
    k_star, k_dir = np.array([0.1, 0.2, 0.3]), np.array([0.3, -0.1, 0.1])
    xs = np.linspace(-0.05, 0.05, 21)
    model_kp = model.construct_kdotp(k_star, order=2)
    t0 = time.perf_counter()
    bands_tb = np.array([model.eigenval(k_star + x * k_dir) for x in xs])
    per_call = (time.perf_counter() - t0) / len(xs)
    bands_kp = np.array([model_kp.eigenval(x * k_dir) for x in xs])
    print("line of %d single-k calls: %.0f us per call; max |E_tb - E_kp| on the line: %.2e"
          % (len(xs), per_call * 1e6, np.abs(bands_tb - bands_kp).max()))

    # 2. the same line as ONE batched call
    t0 = time.perf_counter()
    batched = model.eigenval_array(k_star + xs[:, None] * k_dir)
    print("the same line as one call: %.0f us; identical to the loop within %.1e"
          % ((time.perf_counter() - t0) * 1e6, np.abs(batched - bands_tb).max()))

    # 3. density of states from a 60 x 60 x 60 mesh (folded evaluation: every mesh plane is a 2-D model)
    n = 60
    axis = np.linspace(0, 1, n, endpoint=False)
    mesh = np.stack([m.reshape(-1) for m in np.meshgrid(axis, axis, axis, indexing="ij")], axis=1)
    model.eigenval_array(mesh[:4096])  # warm up
    t0 = time.perf_counter()
    eig = model.eigenval_array(mesh)
    dt = time.perf_counter() - t0
    hist, edges = np.histogram(eig, bins=40)
    print("%d mesh points in %.1f ms (%.1f M k-points/s); bands span [%.3f, %.3f]"
          % (len(mesh), dt * 1e3, len(mesh) / dt / 1e6, eig.min(), eig.max()))
    peak = np.argmax(hist)
    print("DOS peak between %.3f and %.3f" % (edges[peak], edges[peak + 1]))


if __name__ == "__main__":
    main()

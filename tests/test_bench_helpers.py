"""Host-side helpers of bench.py that run without a GPU: usable core count and the forked all-cores CPU baseline."""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_usable_cores_is_within_the_visible_cpus():
    import bench

    cores = bench.usable_cores()
    assert 1 <= cores <= (os.cpu_count() or 1)


def test_all_cores_baseline_runs_the_oracle_in_forked_workers():
    import bench
    from tbmodels_amd import synthetic as syn

    r_vec, hop, pos = syn.dense_model_arrays(6, 9, syn.MODEL_SEED + 5)
    arrays = dict(kind="dense", n_orb=6, R=r_vec, hop=hop, pos=pos)
    k = syn.random_kpoints(500)
    result = bench.cpu_baseline_all_cores(arrays, k, per_proc=20)
    assert result["kind"] == "port" and result["cores"] == bench.usable_cores()
    assert result["value"] > 0 and result["unit"] == "k-points/s"
    assert ("%d k-points" % min(len(k), 20 * result["cores"])) in result["sample"]
    assert not bench._CPU_SHARED  # pylint: disable=protected-access
    assert np.isfinite(result["value"])


def test_hk_roofline_entries_price_the_documented_work():
    """``roofline`` of a config: executed flops of the symmetrised contraction against 78.6 TFLOP/s for dense
    hoppings (half of the algorithmic 8 N^2 N_R), compulsory output bytes against 8 TB/s for sparse ones."""
    import bench

    dense = dict(kind="dense")
    stage_ms = {"hk": 93.2, "phase": 0.0, "eig": 0.0, "ql": 0.0}
    stage_n = {"hk": 4, "phase": 0, "eig": 0, "ql": 0}
    entry = bench.hk_roofline_entry(dense, 64, 4096, 3, stage_ms, stage_n, 100_000, "cfg2", peak_measured=77.4)
    assert entry["bound"] == "mfma" and entry["flops_per_kpoint_executed"] == 8.0 * (64 * 65 / 2) * 4096
    assert entry["flops_per_kpoint_algorithmic"] == 8.0 * 64 * 64 * 4096 + 2.0 * 64 * 64
    assert abs(entry["achieved"] - 68.157e6 * 1e5 / 93.2e-3 / 1e12) < 0.01 and abs(entry["frac"] - entry["achieved"] / 78.6) < 1e-3
    assert abs(entry["algorithmic_speedup"] - 1.969) < 1e-2 and entry["kpoints_per_launch"] == 25_000
    mesh = bench.hk_roofline_entry(dense, 64, 4096, 3, stage_ms, stage_n, 1_000_000, "cfg4")
    assert mesh["frac"] is None and "folded" in mesh["note"]  # no fraction of peak for the folded evaluation
    sparse = bench.hk_roofline_entry(dict(kind="csr"), 256, 512, 3, {"hk": 30.4}, {"hk": 6}, 100_000, "cfg3")
    b_k = 16.0 * 256 * 257 / 2 + 24
    assert sparse["bound"] == "hbm" and sparse["algorithmic_bytes_per_kpoint"] == b_k
    assert abs(sparse["achieved"] - b_k * 1e5 / 30.4e-3 / 1e9) < 0.5 and abs(sparse["frac"] - sparse["achieved"] / 8000.0) < 1e-3

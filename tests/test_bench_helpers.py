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

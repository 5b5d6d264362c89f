"""
The ``eigenvals`` command end to end on the reference's own sample files
(`tests/test_cli_eigenvals.py:22-53` of the reference: both k-point inputs, atol 1e-10).
"""

import os
import subprocess
import sys

import numpy as np
import pytest

from tbmodels_amd import io
from tbmodels_amd._cli import main
from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

SAMPLES = os.path.join(GOLDEN, "cli_eigenvals")


@pytest.mark.parametrize("kpoints_file_name", ["kpoints.hdf5", "silicon_eigenvals.hdf5"])
@pytest.mark.parametrize("verbose", [[], ["-v"]])
def test_cli_eigenvals(tmp_path, capsys, kpoints_file_name, verbose):
    out = tmp_path / "out.hdf5"
    code = main(
        ["eigenvals", "-o", str(out), "-k", os.path.join(SAMPLES, kpoints_file_name), "-i", os.path.join(SAMPLES, "silicon_model.hdf5")]
        + verbose
    )
    assert code == 0
    printed = capsys.readouterr().out
    assert ("Done!" in printed) == bool(verbose)
    res = io.load(out)
    reference = io.load(os.path.join(SAMPLES, "silicon_eigenvals.hdf5"))
    assert isinstance(res, io.EigenvalsData)
    np.testing.assert_array_equal(res.kpoints.kpoints, reference.kpoints.kpoints)
    np.testing.assert_allclose(res.eigenvals, reference.eigenvals, rtol=0, atol=1e-10)


def test_cli_as_module(tmp_path):
    out = tmp_path / "out.hdf5"
    run = subprocess.run(
        [sys.executable, "-m", "tbmodels_amd", "eigenvals", "-v", "-o", str(out), "-k", os.path.join(SAMPLES, "kpoints.hdf5"),
         "-i", os.path.join(SAMPLES, "silicon_model.hdf5")],
        cwd=ROOT, capture_output=True, text=True, check=False,
    )
    assert run.returncode == 0, run.stderr[-2000:]
    assert "Calculating energy eigenvalues" in run.stdout
    reference = io.load(os.path.join(SAMPLES, "silicon_eigenvals.hdf5"))
    np.testing.assert_allclose(io.load(out).eigenvals, reference.eigenvals, rtol=0, atol=1e-10)


def test_model_file_roundtrip_on_device(tmp_path, silicon):
    """A model saved and re-loaded evaluates to the same eigenvalues (sparse flag travels too)."""
    model = io.load(os.path.join(SAMPLES, "silicon_model.hdf5"))
    want = np.array(model.eigenval(silicon["known_kpoints"]))
    for sparse in (False, True):
        model.set_sparse(sparse)
        path = tmp_path / ("m%d.hdf5" % sparse)
        model.to_hdf5_file(path)
        again = io.load(path)
        assert again._sparse == sparse
        np.testing.assert_allclose(np.array(again.eigenval(silicon["known_kpoints"])), want, rtol=0, atol=1e-12)


def test_example_script_runs():
    run = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "bands_and_dos.py")], cwd=ROOT, capture_output=True,
                         text=True, check=False)
    assert run.returncode == 0, run.stderr[-2000:]
    assert "mesh points in" in run.stdout and "single-k calls" in run.stdout

#!/opt/conda/bin/python3.9
"""
Cross-checks of tbmodels_amd/hdf5_lite.py against the real thing.  Runs under an interpreter that HAS
h5py (in the build container: /opt/conda/bin/python3.9); `tests/test_hdf5_lite.py` calls it as a child
process and skips when that interpreter is absent.

    hdf5_crosscheck.py h5py-read  FILE OUT.npz     flatten FILE with h5py -> npz of "path" -> array
    hdf5_crosscheck.py h5py-write FILE             assorted dtypes / shapes / nesting written by h5py
    hdf5_crosscheck.py h5py-chunked FILE           a chunked + gzip dataset (outside hdf5_lite's subset)
    hdf5_crosscheck.py ref-read   FILE OUT.npz     the REFERENCE's Model.from_hdf5_file on FILE (needs /root/reference)
    hdf5_crosscheck.py ref-write  IN.npz FILE      the REFERENCE's Model(...).to_hdf5 into FILE (+ type_tag)
"""

import os
import sys

import numpy as np


def _flatten(handle):
    import h5py

    out = {}

    def visit(name, obj):
        if isinstance(obj, h5py.Dataset):
            value = obj[()]
            if isinstance(value, bytes):
                value = np.array(value.decode("utf-8"))
            out[name] = np.asarray(value)

    handle.visititems(visit)
    return out


def _reference():
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import gen_golden  # the stubbed import recipe lives there

    return gen_golden._import_reference()  # pylint: disable=protected-access


def _model_arrays(model):
    keys = list(model.hop.keys())
    out = {
        "R": np.array(keys, dtype=np.int64).reshape(len(keys), model.dim),
        "hop": np.array([np.array(model.hop[k]) for k in keys]).reshape(len(keys), model.size, model.size),
        "pos": np.array(model.pos),
        "size": np.array(model.size),
        "dim": np.array(model.dim),
        "sparse": np.array(bool(model._sparse)),  # pylint: disable=protected-access
    }
    if model.uc is not None:
        out["uc"] = np.array(model.uc)
    if model.occ is not None:
        out["occ"] = np.array(model.occ)
    return out


def main():
    mode = sys.argv[1]
    if mode == "h5py-read":
        import h5py

        with h5py.File(sys.argv[2], "r") as handle:
            np.savez(sys.argv[3], **{k.replace("/", "|"): v for k, v in _flatten(handle).items()})
    elif mode == "h5py-write":
        import h5py

        rng = np.random.default_rng(7)
        with h5py.File(sys.argv[2], "w") as handle:
            handle["type_tag"] = "some.tag"
            handle["f64"] = rng.random((5, 3))
            handle["f32"] = rng.random(7).astype(np.float32)
            handle["i64"] = np.arange(-3, 9)
            handle["i32"] = np.arange(6, dtype=np.int32).reshape(2, 3)
            handle["u8"] = np.arange(5, dtype=np.uint8)
            handle["c128"] = rng.random((4, 4)) + 1j * rng.random((4, 4))
            handle["c64"] = (rng.random(3) + 1j * rng.random(3)).astype(np.complex64)
            handle["flag_true"] = True
            handle["flag_false"] = False
            handle["flags"] = np.array([True, False, True])
            handle["scalar_int"] = 42
            handle["scalar_float"] = 2.5
            handle["empty"] = np.zeros((0, 3))
            handle["fixed_str"] = np.bytes_("fixed")
            handle["unicode"] = "grüße"
            group = handle.create_group("nested")
            group["type_tag"] = "inner.tag"
            for i in range(40):  # more than one symbol-table node at h5py's default K
                sub = group.create_group(str(i))
                sub["R"] = (i, -i, 2 * i)
                sub["mat"] = rng.random((2, 2)) + 1j * rng.random((2, 2))
            handle.create_group("empty_group")
    elif mode == "h5py-chunked":
        import h5py

        with h5py.File(sys.argv[2], "w") as handle:
            handle.create_dataset("x", data=np.arange(1000.0), chunks=(100,), compression="gzip")
    elif mode == "ref-read":
        tbmodels = _reference()
        model = tbmodels.Model.from_hdf5_file(sys.argv[2])
        np.savez(sys.argv[3], **_model_arrays(model))
    elif mode == "ref-write":
        import h5py

        tbmodels = _reference()
        data = np.load(sys.argv[2])
        hop = {tuple(int(x) for x in r): mat for r, mat in zip(data["R"], data["hop"])}
        kwargs = {"hop": hop, "pos": data["pos"], "contains_cc": False, "sparse": bool(data["sparse"])}
        if "uc" in data:
            kwargs["uc"] = data["uc"]
        if "occ" in data:
            kwargs["occ"] = int(data["occ"])
        model = tbmodels.Model(**kwargs)
        with h5py.File(sys.argv[3], "w") as handle:
            handle["type_tag"] = "tbmodels.model"  # what fsc.hdf5_io adds around Model.to_hdf5
            model.to_hdf5(handle)
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()

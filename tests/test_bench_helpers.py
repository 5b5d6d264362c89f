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
    from oracle import tbk_oracle as oracle

    result, rows = bench.cpu_baseline_all_cores(arrays, k, per_proc=20)
    # the rows the workers computed come back: they are the oracle sample the GPU result is compared with, row for row
    total = min(len(k), 20 * result["cores"])
    assert rows.shape == (total, 6) and np.array_equal(rows, np.array(oracle.eigenval(r_vec, hop, k[:total])))
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


# ------------------------------------------------------------------------------------------------
# bench.py's strong-scaling leg (cfg4 at N ranks) with N = 2 / 8 on the CPU: tools/standin_lib.py stands in for libtbk ("device"
# buffers in NumPy arrays, slabs evaluated by the oracle, the all-gather through the process group) -- what is checked is the
# leg's own logic (slabs, result layout [rank][per][n], per-rank timings, parity rows of the first and the last slab, the
# trace identity over the whole mesh), which no one-GPU box can run at N > 1.
# ------------------------------------------------------------------------------------------------
sys.path.insert(0, os.path.join(ROOT, "tools"))
from standin_lib import StandInLib as _StandInLib  # noqa: E402  pylint: disable=wrong-import-position


def _strong_leg_worker(rank, world, out_dir, fail_rank=-1):
    import ctypes
    import json

    sys.path.insert(0, ROOT)
    import bench
    from tbmodels_amd import synthetic as syn
    from tbmodels_amd.rendezvous import FileGroup

    group = FileGroup(rank, world, os.path.join(out_dir, "rdzv"), token="leg")
    r_vec, hop, pos = syn.dense_model_arrays(5, 12, syn.MODEL_SEED + 77)
    arrays = dict(kind="dense", n_orb=5, R=r_vec, hop=hop, pos=pos)
    lib = _StandInLib(group, world, rank, arrays)
    lib.fail_rank = fail_rank
    try:
        entry = bench.strong_scaling_leg(lib, 0, ctypes.c_void_p(1), ctypes.c_void_p(2), group, world, rank, 5, arrays, steps=1,
                                         warmup=1, mesh=7)  # 343 points: slabs of 172 and 171 (world 2), 43 x 7 + 42 (world 8)
    except RuntimeError as exc:
        entry = {"raised": str(exc)}
    with open(os.path.join(out_dir, "leg%d.json" % rank), "w") as handle:
        json.dump(entry, handle)
    group.close()


def test_strong_scaling_leg_logic_with_two_ranks(tmp_path):
    import json

    _run_leg_workers(tmp_path, 2)
    entry = json.load(open(tmp_path / "leg0.json"))
    assert json.load(open(tmp_path / "leg1.json")) is None  # only rank 0 reports
    assert entry["scaling"] == "strong" and entry["n_gpus"] == 2 and entry["rccl_ranks"] == 2 and entry["kpoints_per_rank"] == 172
    assert entry["max_abs_err_vs_oracle"] <= 1e-12 and entry["max_trace_identity_err_4096_rows"] <= 1e-10
    assert len(entry["per_rank"]["compute_ms"]) == 2 and len(entry["per_rank"]["exposed_gather_ms"]) == 2
    assert entry["value"] > 0 and "7^3 uniform mesh in 2 contiguous slabs" in entry["workload"]


def _run_leg_workers(tmp_path, world, fail_rank=-1):
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_strong_leg_worker, args=(rank, world, str(tmp_path), fail_rank)) for rank in range(world)]
    for proc in procs:
        proc.start()
    for proc in procs:
        proc.join(timeout=300)
        assert proc.exitcode == 0, "worker exited with %r" % proc.exitcode


def test_strong_scaling_leg_with_eight_ranks_and_ragged_slabs(tmp_path):
    """World 8 on the 7^3 mesh: 343 / 8 is no whole number and no multiple of a mesh plane (49) -- slabs of 43 rows cut
    planes and lines, the last slab is short (42 rows + one padding row in the [rank][per][n] result).  Rows of every
    slab reach rank 0's trace identity; the oracle rows are those of the first and the last slab."""
    import json

    _run_leg_workers(tmp_path, 8)
    entry = json.load(open(tmp_path / "leg0.json"))
    for rank in range(1, 8):
        assert json.load(open(tmp_path / ("leg%d.json" % rank))) is None
    assert entry["n_gpus"] == 8 and entry["rccl_ranks"] == 8 and entry["kpoints_per_rank"] == 43
    assert entry["max_abs_err_vs_oracle"] <= 1e-12 and entry["max_trace_identity_err_4096_rows"] <= 1e-10
    assert len(entry["per_rank"]["compute_ms"]) == 8 and "in 8 contiguous slabs" in entry["workload"]


def test_strong_scaling_leg_failure_on_one_rank_reaches_every_rank(tmp_path):
    """Rank 5's solver fails INSIDE the pipelined gather (its status word is non-zero; like libtbk it still walks every
    collective): every one of the eight ranks raises, nobody hangs, nobody reports a number."""
    import json

    _run_leg_workers(tmp_path, 8, fail_rank=5)
    for rank in range(8):
        entry = json.load(open(tmp_path / ("leg%d.json" % rank)))
        assert entry is not None and "status words" in entry["raised"], (rank, entry)


def _guarded_worker(rank, world, out_dir, mode):
    """One rank of bench.guarded_strong_leg: `mode` "hang" -- rank 1 never arrives at the leg's first collective, rank 0
    waits in it for ever; "raise" -- the leg raises on rank 0."""
    import io
    import json
    import time

    sys.path.insert(0, ROOT)
    import bench

    result = {"metric": "m", "value": 1.0, "configs": None, "strong_scaling": None} if rank == 0 else None

    def leg():
        if mode == "raise":
            raise RuntimeError("parity failure in the cfg4 strong-scaling leg")
        if rank == 1:
            return None
        time.sleep(3600)  # the collective a lost peer never joins
        return None

    sys.stdout = open(os.path.join(out_dir, "stdout%d.txt" % rank), "w")
    failed = bench.guarded_strong_leg(result, world, 1.0, leg)
    if result is not None:
        print(json.dumps(result), flush=True)
    sys.exit(1 if failed else 0)


def test_strong_scaling_watchdog_prints_the_line_and_exits_non_zero(tmp_path):
    """A rank lost in a collective: the watchdog prints rank 0's line (top-level "strong_scaling": "timeout", the error
    under configs) and ends the process with a NON-ZERO status; a leg that raises gives "failed" and status 1."""
    import json
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    for mode, want_rc, verdict in (("hang", 3, "timeout"), ("raise", 1, "failed")):
        out = tmp_path / mode
        out.mkdir()
        proc = ctx.Process(target=_guarded_worker, args=(0, 2, str(out), mode))
        proc.start()
        proc.join(timeout=120)
        assert proc.exitcode == want_rc, (mode, proc.exitcode)
        lines = [line for line in open(out / "stdout0.txt").read().splitlines() if line.strip()]
        assert len(lines) == 1
        record = json.loads(lines[0])
        assert record["strong_scaling"] == verdict and record["value"] == 1.0
        assert "error" in record["configs"]["cfg4"] and record["configs"]["cfg4"]["n_gpus"] == 2


def test_summary_is_compact_and_carries_every_config():
    """``summary`` is the LAST key of the JSON line and at most 1.5 KB: the driver's record keeps the final 2000 characters of
    the output, and every config's value / roofline fractions / one-k latencies must survive there."""
    import json

    import bench

    def config(value, with_standalone=True):
        eig = {"frac": 0.2072}
        if with_standalone:
            eig["standalone"] = {"frac": 0.21841234}
        return {"value": value, "ms_per_step": 292.6123, "roofline": {"frac": 0.19134}, "eig_roofline": eig,
                "single_k_us": {"hamilton": 135.2, "eigenval": 1848.0, "roofline": {"frac": 0.671234}},
                "cpu_baseline": {"value": 16.6123}, "max_abs_err_vs_oracle": 3.9968028886505635e-14,
                "max_second_moment_err": 1.2345678e-15}

    result = dict(config(955117.3), config={"workload": "cfg2: dense N_orb=64 N_R=4096, 100000 random k-points per GPU"},
                  configs={name: config(v) for name, v in (("cfg1", 1.39e7), ("cfg3", 170899.0), ("cfg4", 9253450.0), ("cfg5", 16665.7))},
                  host_api={"value": 922642.1, "single_k_us": {"hamilton": 66.0, "eigenval": 180.0}, "hamilton": {"GB/s": 49.06}},
                  construct_only={"value": 1047310.0}, cpu_baseline_all_cores={"value": 452.1})
    del result["single_k_us"]  # the main line carries its one-k figures under host_api
    result["configs"]["cfg4"] = {"error": "RuntimeError: " + "x" * 300, "n_gpus": 8}  # a failed leg must not blow the budget
    summary = bench.make_summary(result)
    text = json.dumps(summary)
    assert len(text) <= 1536, len(text)
    assert list(summary)[:5] == ["cfg2", "cfg1", "cfg3", "cfg4", "cfg5"]
    assert summary["cfg2"]["v"] == 955117.0 and summary["cfg2"]["k1"] == [66.0, 180.0, None]
    assert summary["cfg3"] == {"v": 170899.0, "ms": 292.6, "hk": 0.1913, "eig": 0.2072, "eig_sa": 0.2184, "k1": [135.2, 1848.0, 0.6712],
                               "cpu": 16.61, "err": 4e-14, "m2": 1.2e-15}
    assert "error" in summary["cfg4"] and len(summary["cfg4"]["error"]) <= 60
    assert summary["host"] == 922642.0 and summary["host_h_GBs"] == 49.06 and summary["construct"] == 1047310.0 and summary["cpu_all"] == 452.1
    assert "keys" in summary
    # a multi-rank line has no host_api / construct_only / other configs: still a valid summary
    bare = bench.make_summary({"value": 7.5e6, "ms_per_step": 105.0, "config": {"workload": "cfg2: ..."}, "configs": None})
    assert bare["cfg2"]["v"] == 7.5e6 and bare["host"] is None and len(json.dumps(bare)) < 700


def test_the_scaling_drivers_command_runs_dry_on_eight_cpu_ranks():
    """The exact command the SCALE driver runs on an 8-GPU node -- ``python -m torch.distributed.run --nnodes=1 --nproc-per-node 8
    --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 --steps K --warmup W`` -- with ``--dry-ranks`` appended: eight rank
    processes walk main()'s whole control flow on the CPU (rendezvous from the launcher's environment, the communicator, the
    step loop with the overlapped gather, the per-rank split, the cfg4 strong-scaling leg, ONE JSON line from rank 0) with
    tools/standin_lib.py in place of libtbk.  No scaling curve has ever been measured (no multi-GPU node in six rounds): this
    keeps the path that will produce it runnable.  Independence / order of k-points: `_tb_model.py:1111-1123`, `:1148-1150`."""
    import json
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--dry-ranks"]
    run = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, check=False)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [line for line in run.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]
    record = json.loads(lines[0])
    assert record["n_gpus"] == 8 and record["steps"] == 3 and record["warmup"] == 1 and record["scaling"] == "weak"
    assert record["rccl_ranks"] == 8 and record["config"]["rccl_ranks"] == 8 and "DRY RUN" in record["data"]
    assert record["strong_scaling"] == "ok"
    assert len(record["per_rank"]["compute_ms"]) == 8 and len(record["per_rank"]["allgather_ms"]) == 8
    leg = record["configs"]["cfg4"]
    assert leg["scaling"] == "strong" and leg["n_gpus"] == 8 and leg["rccl_ranks"] == 8 and len(leg["per_rank"]["exposed_gather_ms"]) == 8
    assert leg["max_abs_err_vs_oracle"] <= 1e-12 and record["max_abs_err_vs_oracle"] <= 1e-12
    assert list(record)[-1] == "summary" and abs(record["summary"]["cfg4"]["v"] - leg["value"]) <= 1e-5 * leg["value"]
    assert record["value"] > 0 and record["unit"] == "k-points/s" and record["value_host_buffers"] is None

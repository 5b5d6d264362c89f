#!/usr/bin/env python3
"""
bench.py -- k-points/s of the TBmodels hot path (H(k) + eigenvalues) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2]

Metric (BASELINE.json): k-points/sec (H(k)+eig) at N_orb=64, N_R=4096.  A "step" is one pass of
``Model.eigenval`` over one batch of k-points per GPU: at N=1 the batch is BASELINE config 2 (dense
N_orb=64, N_R=4096, 100k random k-points, SURVEY.md section 8d generator).  With N>1 every rank
holds the same staged model and evaluates its own contiguous slab of an N x 100k k list (weak
scaling, per-GPU work fixed), followed by the RCCL all-gather of eigenvalue slabs inside the timed
region.  k-points and eigenvalues are resident in HBM when the clock starts.

Host side: Python + ctypes -> libtbk.so (include/tbk.h), no torch in the process: the torch wheel
bundles its own ROCm runtime, and a second runtime build next to the system one that libtbk links
crashes at import.  ``python -m torch.distributed.run`` is only the launcher (it sets RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_*); barrier, max-over-ranks and handing the RCCL unique id to the ranks go through
``tbmodels_amd.rendezvous.FileGroup`` (single node).  The eigenvalue all-gather is RCCL.

Rank 0 prints ONE JSON line with the contract fields plus ``roofline`` (the H(k) MFMA kernel, timed
with HIP events on the library's stream) and ``cpu_baseline`` (the oracle = port of the reference's
NumPy/SciPy algorithm, timed on this box's host cores on a bounded sample).
"""

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# load libtbk (system ROCm runtime) before anything else can pull in another copy of the runtime
from tbmodels_amd import _lib, synthetic  # noqa: E402  pylint: disable=wrong-import-position
from tbmodels_amd.rendezvous import group_from_env  # noqa: E402  pylint: disable=wrong-import-position

FP64_MFMA_PEAK_TFLOPS = 78.6  # AMD public spec for MI355X (vector = matrix FP64); see DESIGN.md

CONFIGS = {
    # name: (kind, n_orb, n_r, n_k per GPU, config index for the seed)
    "cfg1": ("silicon", 8, 95, 1000, 1),
    "cfg2": ("dense", 64, 4096, 100_000, 2),
    "cfg3": ("csr", 256, 512, 50_000, 3),
    # the cfg2 model on the 100 x 100 x 100 uniform mesh (meshgrid "ij" order); with N ranks every rank takes a
    # contiguous slab of 1e6 / N mesh points (the one config whose total work is fixed: not the bench line)
    "cfg4": ("dense", 64, 4096, 1_000_000, 2),
    "cfg5": ("dense", 512, 2048, 10_000, 5),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--nk", type=int, default=0, help="override k-points per GPU")
    ap.add_argument("--nr", type=int, default=0, help="override N_R (exploration only)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="k-points for the CPU baseline (0 = skip)")
    ap.add_argument("--eigensolver", default="auto", choices=["auto", "wave", "rocsolver"])
    ap.add_argument("--k-chunk", type=int, default=0)
    ap.add_argument("--construct-only", action="store_true", help="time H(k) construction alone (not the metric)")
    ap.add_argument("--dry-ranks", action="store_true",
                    help="no GPU: every rank walks the control flow of a --gpus N run with tools/standin_lib.py in place of libtbk "
                         "and a tiny model (CPU test of the exact launch command; the numbers mean nothing)")
    return ap.parse_args()


def build_model_arrays(cfg_name, n_r_override=0):
    kind, n_orb, n_r, _, idx = CONFIGS[cfg_name]
    if n_r_override:
        n_r = n_r_override
    seed = synthetic.MODEL_SEED + idx
    if kind == "silicon":
        data = np.load(os.path.join(ROOT, "tests", "golden", "silicon.npz"))
        return dict(kind="dense", n_orb=8, R=data["R"].astype(np.int32), hop=data["hop"], pos=data["pos"])
    if kind == "dense":
        r_vec, hop, pos = synthetic.dense_model_arrays(n_orb, n_r, seed)
        return dict(kind="dense", n_orb=n_orb, R=r_vec, hop=hop, pos=pos)
    r_vec, r_ptr, row, col, val, pos = synthetic.csr_model_arrays(n_orb, n_r, seed)
    return dict(kind="csr", n_orb=n_orb, R=r_vec, r_ptr=r_ptr, row=row, col=col, val=val, pos=pos)


def stage(lib, device, arrays):
    handle = ctypes.c_void_p()
    r_vec = np.ascontiguousarray(arrays["R"], dtype=np.int32)
    if arrays["kind"] == "dense":
        hop = np.ascontiguousarray(arrays["hop"], dtype=np.complex128)
        _lib.check(lib.tbk_model_create_dense(device, r_vec.shape[1], arrays["n_orb"], len(r_vec), _lib.ptr(r_vec),
                                              _lib.ptr(hop), ctypes.byref(handle)))
    else:
        _lib.check(lib.tbk_model_create_csr(device, r_vec.shape[1], arrays["n_orb"], len(r_vec), _lib.ptr(r_vec),
                                            _lib.ptr(arrays["r_ptr"]), _lib.ptr(arrays["row"]), _lib.ptr(arrays["col"]),
                                            _lib.ptr(np.ascontiguousarray(arrays["val"])), ctypes.byref(handle)))
    return handle


def measured_traffic(kernel, k_per_launch):
    """
    HBM bytes per launch of `kernel` from the newest profiles/*_traffic.json (rocprofv3 PMC passes,
    tools/summarize_profiles.py): (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per full k chunk, scaled linearly to
    this run's k-points per launch.  bench.py cannot read PMC counters itself; None when no profile exists.
    """
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as handle:
        data = json.load(handle)
    entry = data.get(kernel)
    if not entry or "hbm_bytes_per_launch" not in entry:
        return None, None
    scaled = entry["hbm_bytes_per_launch"] * k_per_launch / entry["kpoints_per_launch"]
    return scaled, os.path.relpath(files[-1], ROOT)


def cpu_baseline(arrays, kpts, sample):
    """The oracle (NumPy/SciPy port of the reference loop), single process, on `sample` k-points."""
    from oracle import tbk_oracle as oracle  # checker / baseline only

    if arrays["kind"] == "dense":
        hop = arrays["hop"]
    else:
        hop = synthetic.csr_to_dense(arrays["n_orb"], arrays["r_ptr"], arrays["row"], arrays["col"], arrays["val"])
    k = kpts[:sample]
    t0 = time.perf_counter()
    eig = oracle.eigenval(arrays["R"], hop, k)
    dt = time.perf_counter() - t0
    return len(k) / dt, dt, np.array(eig)


_CPU_SHARED = {}


def _cpu_chunk(bounds):
    from oracle import tbk_oracle as oracle  # checker / baseline only

    lo, hi = bounds
    if hi > lo:  # the rows go back to the parent: they are the parity check of the GPU result on the same k-points
        return np.array(oracle.eigenval(_CPU_SHARED["R"], _CPU_SHARED["hop"], _CPU_SHARED["k"][lo:hi]))
    return np.empty((0, _CPU_SHARED["hop"].shape[-1]))


def _cpu_worker_init():
    try:  # the oracle's ufunc passes are single-threaded; keep LAPACK from spawning a pool per process
        import threadpoolctl  # pylint: disable=import-outside-toplevel

        _CPU_SHARED["limit"] = threadpoolctl.threadpool_limits(1)
    except ImportError:
        pass


def usable_cores():
    """Host cores this process may actually use: affinity mask, capped by the cgroup CPU quota (the GPU boxes show
    256 CPUs under a 16-CPU quota; 256 busy processes there ran slower than 16)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0:
                cores = min(cores, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return cores


def cpu_baseline_all_cores(arrays, kpts, per_proc):
    """The same oracle in one process per host core, each on its own contiguous k chunk (what a
    user of the reference does with its pickle support).  Forks, so it runs BEFORE this process touches the GPU.
    Returns (entry, eigenvalues of rows [0, total) of `kpts`): the rows are the oracle sample the GPU result is held to."""
    import multiprocessing as mp  # pylint: disable=import-outside-toplevel

    procs = usable_cores()
    if arrays["kind"] == "dense":
        hop = arrays["hop"]
    else:
        hop = synthetic.csr_to_dense(arrays["n_orb"], arrays["r_ptr"], arrays["row"], arrays["col"], arrays["val"])
    total = min(len(kpts), procs * per_proc)
    edges = [total * i // procs for i in range(procs + 1)]
    _CPU_SHARED.update(R=arrays["R"], hop=hop, k=kpts)
    with mp.get_context("fork").Pool(procs, initializer=_cpu_worker_init) as pool:
        pool.map(_cpu_chunk, [(0, 0)] * procs, chunksize=1)  # every worker up and imported before the clock starts
        t0 = time.perf_counter()
        rows = np.concatenate(pool.map(_cpu_chunk, list(zip(edges[:-1], edges[1:])), chunksize=1))
        dt = time.perf_counter() - t0
    done = len(rows)
    _CPU_SHARED.clear()
    return {
        "value": round(done / dt, 2), "unit": "k-points/s", "cores": procs, "kind": "port", "host_cpus": os.cpu_count(),
        "sample": "%d k-points of this workload in %d processes (one contiguous chunk each), same oracle, %.1f s"
                  % (done, procs, dt),
    }, rows


def launch_ranks(n_ranks):
    """
    ``bench.py --gpus N`` without a launcher: start N fresh rank processes (one per GPU) and relay rank 0's JSON line.

    This process has made no GPU call yet and never makes one (importing tbmodels_amd._lib does not load libtbk): the
    ranks are new interpreters, never a re-exec of a process that touched the GPU.  They get what a launcher would
    set (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*) plus the rendezvous directory and run token of
    ``tbmodels_amd.rendezvous``.  The first rank that fails ends the run: the others are stopped (by pid) and the exit
    status is non-zero.  Returns the exit status.
    """
    import shutil  # pylint: disable=import-outside-toplevel
    import socket  # pylint: disable=import-outside-toplevel
    import subprocess  # pylint: disable=import-outside-toplevel
    import tempfile  # pylint: disable=import-outside-toplevel
    import uuid  # pylint: disable=import-outside-toplevel

    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    rdzv = tempfile.mkdtemp(prefix="tbk_rdzv_", dir=base)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    token = uuid.uuid4().hex[:16]
    procs = []
    status = 0
    try:
        for rank in range(n_ranks):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n_ranks),
                       LOCAL_WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       TBK_RDZV_DIR=rdzv, TBK_RDZV_TOKEN=token, HSA_ENABLE_IPC_MODE_LEGACY="0")
            # rank 0 prints the one JSON line on this process' stdout; whatever else a rank prints goes to stderr
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=None if rank == 0 else sys.stderr))
        alive = set(range(n_ranks))
        while alive and status == 0:
            for rank in sorted(alive):
                code = procs[rank].poll()
                if code is None:
                    continue
                alive.discard(rank)
                if code != 0:
                    status = code if code > 0 else 1
                    sys.stderr.write("[bench] rank %d exited with status %d: stopping the other ranks\n" % (rank, code))
                    break
            time.sleep(0.02)
    finally:
        for proc in procs:  # exact pids of the children started above, nothing else
            if proc.poll() is None:
                proc.terminate()
        deadline = time.monotonic() + 10.0
        for proc in procs:
            try:
                proc.wait(timeout=max(0.1, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                proc.kill()
                proc.wait()
        shutil.rmtree(rdzv, ignore_errors=True)
    return status


def trace_identity_error(arrays, k_slab, eig_all, n_rows=4096):
    """max | sum_i E_i(k) - tr H(k) | over rows drawn from the whole slab: tr H(k) = sum_R 2 Re(e^{2 pi i k.R} tr hop[R])
    is host work independent of the GPU path, and rows from everywhere cover every k chunk of the pipeline."""
    nk = len(k_slab)
    rows = np.sort(np.random.default_rng(1).choice(nk, min(nk, n_rows), replace=False))
    if arrays["kind"] == "dense":
        traces = np.einsum("rii->r", arrays["hop"])
    else:
        traces = np.zeros(len(arrays["R"]), dtype=complex)
        diag = arrays["row"] == arrays["col"]
        np.add.at(traces, np.searchsorted(arrays["r_ptr"], np.flatnonzero(diag), side="right") - 1, arrays["val"][diag])
    phase = np.exp(2j * np.pi * (k_slab[rows] @ arrays["R"].T.astype(float)))
    return float(np.abs(eig_all[rows].sum(axis=1) - 2.0 * (phase @ traces).real).max())


def second_moment_error(lib, model, n_orb, k_slab, eig_all, n_rows, chunk_bytes=256 << 20):
    """The second moment of every sampled spectrum against H(k) itself: sum_i E_i(k)^2 = ||H(k)||_F^2, with H(k) from the
    H(k) kernels (tbk_hamilton on host buffers; their parity with the oracle is pinned separately), on `n_rows` rows drawn
    from the whole slab.  Where the trace identity holds the eigensolver to the first moment only, this holds it to H on
    every row.  Returned as an equivalent eigenvalue error: max_k |sum E^2 - ||H||_F^2| / (2 sum_i |E_i|) -- a uniform
    error of that size on every eigenvalue would produce the difference (compare with the 1e-10 bar)."""
    nk = len(k_slab)
    rows = np.sort(np.random.default_rng(2).choice(nk, min(nk, n_rows), replace=False))
    k_sel = np.ascontiguousarray(k_slab[rows])
    per = max(1, int(chunk_bytes // (16 * n_orb * n_orb)))
    frob = np.empty(len(rows))
    ham = np.empty((min(per, len(rows)), n_orb, n_orb), dtype=np.complex128)
    for lo in range(0, len(rows), per):
        n = min(per, len(rows) - lo)
        _lib.check(lib.tbk_hamilton(model, _lib.ptr(k_sel[lo:lo + n]), n, 2, None, _lib.ptr(ham)))
        flat = ham[:n].view(np.float64).reshape(n, -1)
        frob[lo:lo + n] = np.einsum("ij,ij->i", flat, flat)
    e_sel = eig_all[rows]
    return float((np.abs((e_sel * e_sel).sum(axis=1) - frob) / (2.0 * np.abs(e_sel).sum(axis=1))).max())


def eig_roofline_entry(n_orb, matrices, eig_ms, steps):
    """The reduction to tridiagonal form: (16/3) n^3 flops per matrix (eigensolve, values only) over the
    HIP-event time of the reduction stage on its own stream."""
    eig_flops = 16.0 / 3.0 * n_orb ** 3
    eig_tf = eig_flops * matrices / (eig_ms * 1e-3) / 1e12
    hbm = None
    if 128 < n_orb <= 1024:
        # The two-stage reduction streams the stored triangle of the trailing matrix once per panel of 8 columns (read,
        # rank-16 update, store, product with the next panel's V in the same visit): by construction
        # 2 x 16 B x sum_p T(n - 8 (p + 1)) bytes per matrix, T(m) = m (m + 1) / 2 -- against 16 n^2 compulsory.  The [V | W] /
        # V operand blocks every visit re-reads come on top (L2 / MALL resident for the most part: DESIGN.md 5.5).
        tiles = sum((n_orb - 8 * (p + 1)) * (n_orb - 8 * (p + 1) + 1) // 2 for p in range(n_orb // 8) if n_orb - 8 * (p + 1) > 0)
        stream_bytes = 2.0 * 16.0 * tiles
        rate = stream_bytes * matrices / (eig_ms * 1e-3) / 1e9
        hbm = {"bound": "hbm", "tile_stream_bytes_per_matrix": stream_bytes, "compulsory_bytes_per_matrix": 16.0 * n_orb * n_orb,
               "achieved": round(rate, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(rate / 8000.0, 4),
               "note": "modelled tile traffic of stage one over the whole reduction stage time (both stages); "
                       "PMC totals per launch: profiles/*_pmc_cfg3.txt / _cfg5.txt"}
    return {
        "kernel": ("herm_tridiag_packed_kernel" if n_orb <= 32 else
                   "herm_tridiag4_kernel (first n - 32 steps) + herm_tridiag_packed_kernel (trailing 32 x 32)" if n_orb <= 64 else
                   "herm_tridiag_stream_kernel" if n_orb <= 128 else
                   "band_reduce_kernel (+ chase)" if n_orb <= 1024 else
                   "band_xl_* launch chain (+ chase)" if n_orb <= 4096 else "rocsolver zheevd"),
        "bound": "valu-f64" if n_orb <= 128 else "mfma",
        "flops_per_matrix": eig_flops, "achieved": round(eig_tf, 3), "peak": FP64_MFMA_PEAK_TFLOPS,
        "unit": "TFLOP/s", "frac": round(eig_tf / FP64_MFMA_PEAK_TFLOPS, 4),
        "stage_ms_per_step": round(eig_ms / steps, 3),
        "note": "stage time on the reduction's stream; other stages run beside it on other streams",
        "hbm": hbm,
    }


def hk_roofline_entry(arrays, n_orb, n_r, dim, stage_ms, stage_n, k_total, config, construct_only=False, peak_measured=None,
                      with_traffic=False):
    """The H(k) kernel of a config against the roofline that bounds it: executed FP64 flops of the MFMA
    contraction for dense hoppings, compulsory output bytes against 8 TB/s for sparse ones.  stage_ms / stage_n: HIP-event
    time and launches of the "hk" stage over the timed steps; k_total: k-points those steps evaluated."""
    hk_launches = max(1, stage_n["hk"])
    hk_ms_avg = stage_ms["hk"] / hk_launches
    k_per_launch = k_total / hk_launches
    secs = hk_ms_avg * 1e-3
    if arrays["kind"] == "dense":
        f_k = 8.0 * n_orb * n_orb * n_r + 2.0 * n_orb * n_orb  # algorithmic flops per k-point
        f_exec = 8.0 * (n_orb * (n_orb + 1) / 2) * n_r          # what the symmetrised contraction executes
        algorithmic = f_k * k_per_launch / secs / 1e12 if hk_ms_avg > 0 else 0.0
        executed = f_exec * k_per_launch / secs / 1e12 if hk_ms_avg > 0 else 0.0
        traffic, traffic_src = (None, None)
        if with_traffic:
            traffic, traffic_src = measured_traffic("hk_dense", k_per_launch)
        roofline = {
            "kernel": "hk_dense_kernel", "bound": "mfma",
            # the flops the matrix pipe EXECUTES: the staged operand is real and symmetrised, only the packed
            # upper triangle is contracted -- half of SURVEY 8d's 8 N^2 N_R per k-point
            "achieved": round(executed, 3), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(executed / FP64_MFMA_PEAK_TFLOPS, 4),
            "peak_measured": peak_measured,
            "frac_of_peak_measured": round(executed / peak_measured, 4) if peak_measured else None,
            "achieved_algorithmic": round(algorithmic, 3),
            "algorithmic_speedup": round(f_k / f_exec, 4),
            "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC: 2*FETCH_SIZE+WRITE_SIZE)",
            "traffic_source": traffic_src,
            "traffic_note": "builder-run PMC constant from profiles/ rescaled to this run's k-points per launch -- "
                            "bench.py cannot read PMC counters; not a counter of this run" if traffic else None,
            "hbm_frac": round(traffic / secs / 8.0e12, 4) if traffic and hk_ms_avg > 0 else None,
            "algorithmic_bytes_per_launch": (16.0 * n_orb * (n_orb + 1) / 2 + 8 * dim) * k_per_launch
                                            + 16.0 * n_orb * n_orb * n_r,
            "flops_per_kpoint_algorithmic": f_k, "flops_per_kpoint_executed": f_exec,
            "avg_launch_ms": round(hk_ms_avg, 4), "launches": stage_n["hk"], "kpoints_per_launch": k_per_launch,
        }
        if config in ("cfg1", "cfg4"):
            # mesh planes are evaluated on the folded model (csrc/tbk_fold.hip): the contraction executes ~13x
            # fewer flops than the direct sum these figures price (cfg1: a handful of latency-bound launches)
            roofline["note"] = ("folded / latency-bound evaluation: the flop figures would price the UNFOLDED sum, which the "
                                "matrix pipe does not execute -- no fraction of peak is reported for this config; its "
                                "dominant kernel is the n <= 64 reduction (DESIGN.md section 5.4 / 5.6)")
            for key in ("achieved", "frac", "frac_of_peak_measured", "hbm_frac"):
                roofline[key] = None
        return roofline
    # sparse: every packed element of H(k) is written once, 16 bytes (TRI mode: the upper triangle the eigensolver
    # reads; FULL for hamilton()), plus the k-point itself -- SURVEY 8d's B_k without the solver's re-read
    out_bytes = 16.0 * (n_orb * (n_orb + 1) / 2 if not construct_only else n_orb * n_orb)
    b_k = out_bytes + 8 * dim
    achieved = b_k * k_per_launch / secs / 1e9 if hk_ms_avg > 0 else 0.0
    entry = {
        "kernel": "hk_csr_lds_kernel", "bound": "hbm", "achieved": round(achieved, 2), "peak": 8000.0,
        "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": None,
        "algorithmic_bytes_per_kpoint": b_k,
        "avg_launch_ms": round(hk_ms_avg, 4), "launches": stage_n["hk"], "kpoints_per_launch": k_per_launch,
    }
    if "val" in arrays:
        # the other side of the same kernel: one complex multiply-add (8 flops) per stored hopping entry and k-point on the
        # FP64 vector pipe -- at the bench model's fill the record walk, not the 16 B per element it writes, is what binds it
        records = float(len(arrays["val"]))
        valu = 8.0 * records * k_per_launch / secs / 1e12 if hk_ms_avg > 0 else 0.0
        entry["valu"] = {"records_per_kpoint": records, "flops_per_kpoint": 8.0 * records, "achieved": round(valu, 3),
                         "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(valu / FP64_MFMA_PEAK_TFLOPS, 4)}
    return entry


def staged_operand_bytes(arrays):
    """Bytes of the staged hopping operand one H(k) evaluation has to read at least once: the dense
    Bt[K][ncol_pad] of (re, im) pairs, K = 2 N_R rounded up to 16 rows, ncol = N (N + 1) / 2 rounded up to 64; for sparse
    hoppings the 16-byte records (value + packed offsets) of the element-major CSR."""
    n_orb = arrays["n_orb"]
    if arrays["kind"] == "dense":
        k_rows = (2 * len(arrays["R"]) + 15) // 16 * 16
        ncol_pad = (n_orb * (n_orb + 1) // 2 + 63) // 64 * 64
        return 16.0 * k_rows * ncol_pad
    return 16.0 * len(arrays["val"])


def single_k_latency(lib, model, k_slab, n_orb, calls=128, warm=8, arrays=None):
    """Wall-clock of ONE-k-point calls on host buffers -- what Z2Pack-style callers do (_tb_model.py:1103-1108) -- and, with
    `arrays`, the roofline of the H(k) kernels of such a call: HIP-event time of the "hk" stage of 8 further calls (for a
    dense model hk_gemv_kernel + its finish kernel: one read of the staged operand + one write of H) against 8 TB/s."""
    one_h = np.empty((1, n_orb, n_orb), dtype=np.complex128)
    one_e = np.empty((1, n_orb))
    single = {}
    nk = len(k_slab)
    # (stage timing off for the wall-clock: with it every call records and reads back HIP events, ~10 us per call)
    _lib.check(lib.tbk_model_set_option(model, _lib.TBK_OPT_TIMING, 0))
    for name, call in (("hamilton", lambda q: lib.tbk_hamilton(model, _lib.ptr(k_slab[q:q + 1]), 1, 2, None, _lib.ptr(one_h))),
                       ("eigenval", lambda q: lib.tbk_eigenval(model, _lib.ptr(k_slab[q:q + 1]), 1, _lib.ptr(one_e)))):
        for q in range(warm):
            _lib.check(call(q % nk))
        t1 = time.perf_counter()
        for q in range(warm, warm + calls):
            _lib.check(call(q % nk))
        single[name] = round((time.perf_counter() - t1) / calls * 1e6, 1)
    if arrays is not None:
        ms = (ctypes.c_double * _lib.TBK_T_COUNT)()
        launches = (ctypes.c_int64 * _lib.TBK_T_COUNT)()
        _lib.check(lib.tbk_model_set_option(model, _lib.TBK_OPT_TIMING, 1))
        _lib.check(lib.tbk_get_timing(model, None, None, 1))
        timed = 8
        for q in range(timed):
            _lib.check(lib.tbk_hamilton(model, _lib.ptr(k_slab[q % nk:q % nk + 1]), 1, 2, None, _lib.ptr(one_h)))
        _lib.check(lib.tbk_get_timing(model, ms, launches, 1))
        hk_us = ms[_lib.STAGE_NAMES.index("hk")] / timed * 1e3
        nbytes = staged_operand_bytes(arrays) + 16.0 * n_orb * n_orb
        rate = nbytes / (hk_us * 1e-6) / 1e12 if hk_us > 0 else 0.0
        single["roofline"] = {
            "kernel": "hk_gemv_kernel + hk_finish_wide_kernel" if arrays["kind"] == "dense" else "hk_csr_lds_kernel",
            "bound": "hbm", "kernel_us": round(hk_us, 2), "bytes": nbytes, "TB/s": round(rate, 3), "peak": 8.0,
            "frac": round(rate / 8.0, 4),
            "note": "HIP events around the H(k) stage of a one-k hamilton call (all its kernels and the gaps between them); "
                    "bytes = one read of the staged operand + one write of the full H; kernel traces: profiles/*_single_k_*.csv; "
                    "PMC (profiles/r06_single_k.txt): hk_gemv_kernel fetches 1.006 x / 1.003 x the staged bytes at cfg2 / cfg5",
        }
    _lib.check(lib.tbk_model_set_option(model, _lib.TBK_OPT_TIMING, 1))  # (the callers measure with stage timing on)
    _lib.check(lib.tbk_get_timing(model, None, None, 1))
    return single


def config_kpoints(name, nk, dim):
    if name == "cfg1":
        return np.ascontiguousarray(synthetic.uniform_grid(10)[:nk])
    if name == "cfg4":
        return np.ascontiguousarray(synthetic.grid_slab(100, 0, nk))
    return np.ascontiguousarray(np.random.default_rng(synthetic.K_SEED).random((nk, dim)))


def standalone_reduction(lib, device, n_orb, matrices, reps=3):
    """The reduction stage alone on the chip (tbk_reduce_standalone: device-made random Hermitian matrices, HIP events around
    the reduction only): us per matrix and fraction of the FP64 peak for the whole reduction and, at the two-stage sizes, for
    each stage -- beside the in-pipeline stage time, during which the next chunk's H(k) shares the FP64 pipe."""
    us = (ctypes.c_double * 3)()
    _lib.check(lib.tbk_reduce_standalone(device, n_orb, matrices, reps, us))
    flops = 16.0 / 3.0 * n_orb ** 3
    entry = {"matrices_per_launch": matrices, "reps": reps, "us_per_matrix": round(us[0], 4),
             "achieved": round(flops / (us[0] * 1e-6) / 1e12, 3), "unit": "TFLOP/s",
             "frac": round(flops / (us[0] * 1e-6) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
             "note": "tbk_reduce_standalone: nothing else on the chip; (16/3) n^3 flops per matrix over the HIP-event time"}
    if us[1] > 0.0:
        entry["stage1_us_per_matrix"] = round(us[1], 4)
        entry["stage2_us_per_matrix"] = round(us[2], 4)
        entry["stage1_frac"] = round(flops / (us[1] * 1e-6) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4)
        entry["stages_note"] = ("each stage in a launch of its own (up to 256 orbitals the product fuses them into one kernel: "
                                "us_per_matrix is that kernel)")
    return entry


def run_other_config(lib, device, name, model=None, arrays=None, steps=2, warmup=1, cpu_all=None, cpu_all_rows=None):
    """
    One of the other BASELINE configs at its FULL size on this GPU, after the main clock and outside ``ms_per_step``:
    `steps` timed passes of tbk_eigenval_device_hint over the config's k list (resident in HBM), the stage times, the
    reduction's roofline entry, and the same two correctness checks as the main line (oracle sample, trace identity).
    """
    _, n_orb, _, nk, _ = CONFIGS[name]
    t_build = time.perf_counter()
    own = model is None
    if arrays is None:
        arrays = build_model_arrays(name)
    if own:
        model = stage(lib, device, arrays)
    dim = arrays["R"].shape[1]
    n_r = len(arrays["R"])
    k = config_kpoints(name, nk, dim)
    build_s = time.perf_counter() - t_build
    d_k, d_e = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.check(lib.tbk_device_malloc(device, k.nbytes, ctypes.byref(d_k)))
    _lib.check(lib.tbk_device_malloc(device, nk * n_orb * 8, ctypes.byref(d_e)))
    try:
        _lib.check(lib.tbk_memcpy_h2d(device, d_k, _lib.ptr(k), k.nbytes))
        hint = _lib.ptr(k) if name in ("cfg1", "cfg4") else None  # meshes: the host list is the structure hint
        _lib.check(lib.tbk_model_set_option(model, _lib.TBK_OPT_TIMING, 1))
        for _ in range(warmup):
            _lib.check(lib.tbk_eigenval_device_hint(model, d_k, hint, nk, d_e))
        _lib.check(lib.tbk_eigenval_check(model))
        _lib.check(lib.tbk_get_timing(model, None, None, 1))
        t0 = time.perf_counter()
        for _ in range(steps):
            _lib.check(lib.tbk_eigenval_device_hint(model, d_k, hint, nk, d_e))
        _lib.check(lib.tbk_synchronize(model))
        elapsed = time.perf_counter() - t0
        _lib.check(lib.tbk_eigenval_check(model))
        ms = (ctypes.c_double * _lib.TBK_T_COUNT)()
        launches = (ctypes.c_int64 * _lib.TBK_T_COUNT)()
        _lib.check(lib.tbk_get_timing(model, ms, launches, 1))
        stage_ms = {stage_name: ms[i] for i, stage_name in enumerate(_lib.STAGE_NAMES)}
        stage_n = {stage_name: launches[i] for i, stage_name in enumerate(_lib.STAGE_NAMES)}
        eig = np.empty((nk, n_orb))
        _lib.check(lib.tbk_memcpy_d2h(device, _lib.ptr(eig), d_e, eig.nbytes))
        # sum E^2 = ||H(k)||_F^2 on 4096 rows (512 at 512 orbitals: 4 MiB of H per row) -- the eigensolver held to H itself
        moment_rows = 4096 if n_orb <= 256 else 512
        moment_err = second_moment_error(lib, model, n_orb, k, eig, moment_rows)
        _lib.check(lib.tbk_get_timing(model, None, None, 1))
        single = None
        if os.environ.get("TBK_BENCH_SKIP_HOSTAPI") != "1":
            # (a one-k call at 512 orbitals takes ~10 ms: fewer of them)
            single = single_k_latency(lib, model, k, n_orb, calls=32 if n_orb > 128 else 128, warm=4 if n_orb > 128 else 8, arrays=arrays)
            _lib.check(lib.tbk_get_timing(model, None, None, 1))
    finally:
        lib.tbk_device_free(device, d_k)
        lib.tbk_device_free(device, d_e)
        if own:
            lib.tbk_model_destroy(model)
    trace_err = trace_identity_error(arrays, k, eig)
    # the oracle as checker AND as this config's CPU baseline (cfg1 in full, a bounded sample of the
    # others, linear in NK; one process = the reference's own usage).  TBK_BENCH_CPU_LIGHT=1: the checker's few rows only.
    light = os.environ.get("TBK_BENCH_CPU_LIGHT") == "1"
    sample = ({"cfg1": 64, "cfg3": 4, "cfg4": 16, "cfg5": 1} if light else
              {"cfg1": 1000, "cfg3": 48, "cfg4": 256, "cfg5": 4}).get(name, 8)
    cpu_rate, cpu_dt, cpu_eig = cpu_baseline(arrays, k, sample)
    parity = float(np.abs(cpu_eig - eig[:len(cpu_eig)]).max())
    oracle_rows = sample
    if cpu_all_rows is not None and len(cpu_all_rows):
        # the rows the all-core baseline computed anyway (before the GPU was touched): the same k-points, every one compared
        parity = max(parity, float(np.abs(cpu_all_rows - eig[:len(cpu_all_rows)]).max()))
        oracle_rows = max(sample, len(cpu_all_rows))
    eig_roofline = eig_roofline_entry(n_orb, nk * steps, stage_ms["eig"], steps) if stage_ms["eig"] > 0 else None
    if eig_roofline is not None and n_orb > 64 and os.environ.get("TBK_BENCH_SKIP_STANDALONE") != "1":
        eig_roofline["standalone"] = standalone_reduction(lib, device, n_orb, 4096 if n_orb <= 256 else 2048)
    entry = {
        "workload": "%s: %s N_orb=%d N_R=%d, %d %s k-points, eigenval (H(k)+eig), 1 GPU"
                    % (name, arrays["kind"], n_orb, n_r, nk, "grid" if name in ("cfg1", "cfg4") else "random"),
        "value": round(nk * steps / elapsed, 1), "unit": "k-points/s", "steps": steps, "warmup": warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 3),
        "stage_ms_per_step": {key: round(v / steps, 3) for key, v in stage_ms.items()},
        "eig_roofline": eig_roofline,
        "roofline": hk_roofline_entry(arrays, n_orb, n_r, dim, stage_ms, stage_n, nk * steps, name),
        "cpu_baseline": {
            "value": round(cpu_rate, 3), "unit": "k-points/s", "cores": 1, "kind": "port",
            "sample": "%d of the %d k-points of this workload, oracle/tbk_oracle.py (NumPy loop over R + scipy eigvalsh), "
                      "one process, %.1f s" % (sample, nk, cpu_dt),
            "host_cpus": os.cpu_count(),
        },
        "cpu_baseline_all_cores": cpu_all,
        "single_k_us": single,
        "max_abs_err_vs_oracle": parity,
        "oracle_sample": "%d rows (the first %d k-points of the workload%s)"
                         % (oracle_rows, oracle_rows, ": the rows of cpu_baseline_all_cores" if oracle_rows > sample else ""),
        "max_trace_identity_err_4096_rows": trace_err,
        "max_second_moment_err": moment_err,
        "second_moment_note": None if moment_err is None else
                              "max |sum E^2 - ||H(k)||_F^2| / (2 sum |E|) over %d rows of the workload, H(k) from tbk_hamilton"
                              % min(nk, 4096 if n_orb <= 256 else 512),
        "model_build_and_staging_s": round(build_s, 2),
    }
    if not (parity <= 1e-10 and trace_err <= 1e-10 and (moment_err is None or moment_err <= 1e-10)):
        raise SystemExit("parity failure in %s: %r" % (name, entry))
    return entry


def _sig(x, digits=4):
    """`x` rounded to `digits` significant digits (None stays None): the summary has to stay short."""
    if x is None:
        return None
    return float("%.*g" % (digits, x))


def make_summary(result):
    """The figures of every config in <= 1.5 KB, to be the LAST key of the JSON line: the driver's record keeps the final
    2000 characters of the output, and the full line is ~15 KB.  Per config: value, ms_per_step,
    roofline.frac of its H(k) kernel, eig_roofline.frac in the pipeline and standalone, one-k latencies with the H(k)
    fraction of 8 TB/s of such a call, cpu_baseline.value, max_abs_err_vs_oracle (+ the second-moment check)."""
    def entry(src):
        if not isinstance(src, dict) or "value" not in src:
            return {"error": str((src or {}).get("error", "missing"))[:60]} if isinstance(src, dict) else None
        eig = src.get("eig_roofline") or {}
        single = src.get("single_k_us") or (src.get("host_api") or {}).get("single_k_us") or {}
        return {
            "v": _sig(src.get("value"), 6), "ms": _sig(src.get("ms_per_step")),
            "hk": _sig((src.get("roofline") or {}).get("frac")),
            "eig": _sig(eig.get("frac")), "eig_sa": _sig((eig.get("standalone") or {}).get("frac")),
            "k1": [single.get("hamilton"), single.get("eigenval"), _sig((single.get("roofline") or {}).get("frac"))],
            "cpu": _sig((src.get("cpu_baseline") or {}).get("value")),
            "err": _sig(src.get("max_abs_err_vs_oracle"), 2), "m2": _sig(src.get("max_second_moment_err"), 2),
        }
    main_name = (result.get("config") or {}).get("workload", "cfg2")[:4]
    out = {main_name: entry(result)}
    for name, src in sorted((result.get("configs") or {}).items()):
        out[name] = entry(src)
    host = result.get("host_api") or {}
    out["host"] = _sig(host.get("value"), 6)
    out["host_h_GBs"] = _sig((host.get("hamilton") or {}).get("GB/s"))
    out["construct"] = _sig((result.get("construct_only") or {}).get("value"), 6)
    out["cpu_all"] = _sig((result.get("cpu_baseline_all_cores") or {}).get("value"))
    out["keys"] = ("v=k-points/s ms=ms_per_step hk=roofline.frac eig=eig_roofline.frac eig_sa=.standalone.frac "
                   "k1=[one-k hamilton us,eigenval us,H(k) frac of 8TB/s] cpu=cpu_baseline k/s err=max_abs_err_vs_oracle "
                   "m2=second moment; host=host_api.value")
    return out


def strong_scaling_leg(lib, device, model, comm, group, world, rank, n_orb, arrays, steps=3, warmup=1, mesh=100):
    """cfg4 at N ranks (every rank calls this): rank r evaluates rows slab_bounds(10^6, N, r) of the 100^3 mesh.  One step =
    tbk_eigenval_device_gather (slab eigenvalues + pipelined all-gather into the [N][per][n] result on every rank) +
    a wait for the communicator's stream; barrier on both sides of the timed steps, MAX over ranks.  Returns rank 0's
    entry for ``configs.cfg4`` (None elsewhere)."""
    from tbmodels_amd.sharding import slab_bounds  # pylint: disable=import-outside-toplevel

    total = mesh ** 3  # (mesh: the CPU test-suite walks this function with a small mesh and a stand-in library)
    lo, hi = slab_bounds(total, world, rank)
    per = -(-total // world)
    k = np.ascontiguousarray(synthetic.grid_slab(mesh, lo, hi))
    pointers = []

    def dmalloc(nbytes):
        p = ctypes.c_void_p()
        _lib.check(lib.tbk_device_malloc(device, max(int(nbytes), 8), ctypes.byref(p)))
        pointers.append(p)
        return p

    def sync():
        _lib.check(lib.tbk_comm_synchronize(comm))
        _lib.check(lib.tbk_synchronize(model))

    class _Alone:  # N = 1 (TBK_BENCH_FORCE_COMM=1): no process group
        @staticmethod
        def barrier():
            return None

        @staticmethod
        def allreduce_max(x):
            return x

        @staticmethod
        def all_gather_array(a):
            return [a]

    if group is None:
        group = _Alone()
    try:
        d_k = dmalloc(k.nbytes)
        d_all = dmalloc(world * per * n_orb * 8)
        d_st = dmalloc(world * 8)
        _lib.check(lib.tbk_memcpy_h2d(device, d_k, _lib.ptr(k), k.nbytes))
        _lib.check(lib.tbk_model_set_option(model, _lib.TBK_OPT_TIMING, 0))

        def step():
            _lib.check(lib.tbk_eigenval_device_gather(comm, model, d_k, _lib.ptr(k), hi - lo, per, 0, d_all, d_st))

        for _ in range(warmup):
            step()
        sync()
        group.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync()
        group.barrier()
        elapsed = group.allreduce_max(time.perf_counter() - t0)
        # one more step, split: this rank's kernels alone, then what the pipelined call adds on top (the exposed gather)
        group.barrier()
        t1 = time.perf_counter()
        _lib.check(lib.tbk_eigenval_device_hint(model, d_k, _lib.ptr(k), hi - lo, d_all))
        _lib.check(lib.tbk_synchronize(model))
        compute_ms = (time.perf_counter() - t1) * 1e3
        group.barrier()
        t1 = time.perf_counter()
        step()
        sync()
        call_ms = (time.perf_counter() - t1) * 1e3
        parts = group.all_gather_array(np.array([compute_ms, call_ms]))
        status = np.empty(world)
        _lib.check(lib.tbk_memcpy_d2h(device, _lib.ptr(status), d_st, status.nbytes))
        if status.max() != 0:
            raise RuntimeError("cfg4 strong-scaling leg: status words %r" % (status,))
        if rank != 0:
            return None
        # checks on rank 0, over rows of EVERY rank's slab: trace identity on 4096 rows of the whole mesh, oracle on the
        # first rows of the first and the last slab
        eig = np.empty((world * per, n_orb))
        _lib.check(lib.tbk_memcpy_d2h(device, _lib.ptr(eig), d_all, eig.nbytes))
        k_all = synthetic.grid_slab(mesh, 0, total)
        trace_err = trace_identity_error(arrays, k_all, eig[:total])
        last_lo = slab_bounds(total, world, world - 1)[0]
        rows = np.r_[0:8, last_lo:last_lo + 8]
        _, _, cpu_eig = cpu_baseline(arrays, k_all[rows], len(rows))
        parity = float(np.abs(cpu_eig - eig[rows]).max())
        n_ranks = ctypes.c_int(0)
        _lib.check(lib.tbk_comm_ranks(comm, ctypes.byref(n_ranks), None))
        entry = {
            "workload": "cfg4: dense N_orb=%d N_R=%d, %d^3 uniform mesh in %d contiguous slabs (one per GPU), eigenval + RCCL "
                        "all-gather of the eigenvalues to every rank (pipelined behind the k chunks)"
                        % (n_orb, len(arrays["R"]), mesh, world),
            "value": round(total * steps / elapsed, 1), "unit": "k-points/s", "scaling": "strong", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3),
            "rccl_ranks": n_ranks.value, "kpoints_per_rank": per,
            "per_rank": {"compute_ms": [round(float(p[0]), 3) for p in parts],
                         "call_with_gather_ms": [round(float(p[1]), 3) for p in parts],
                         "exposed_gather_ms": [round(float(p[1] - p[0]), 3) for p in parts],
                         "note": "one extra step per rank: the slab's kernels alone (tbk_eigenval_device_hint), then the "
                                 "pipelined call (tbk_eigenval_device_gather + wait): the difference is what the gather adds"},
            "max_abs_err_vs_oracle": parity, "oracle_sample": len(rows),
            "max_trace_identity_err_4096_rows": trace_err,
        }
        if not (parity <= 1e-10 and trace_err <= 1e-10):
            raise RuntimeError("parity failure in the cfg4 strong-scaling leg: %r" % (entry,))
        return entry
    finally:
        for p in pointers:
            lib.tbk_device_free(device, p)


def guarded_strong_leg(result, world, limit, leg):
    """Runs `leg()` (the cfg4 strong-scaling leg; every rank calls this) under a watchdog and files its outcome in `result`
    (rank 0's main record, None elsewhere): the entry under configs and a TOP-LEVEL "strong_scaling": "ok" | "failed" |
    "timeout".  A leg that raises (a failed collective, its own parity check) returns True -- the caller prints the line and
    exits non-zero.  A leg that does not come back within `limit` seconds (a rank lost in a collective) ends THIS process
    from the watchdog thread: the line is printed with "timeout", then os._exit(3) -- the process exits, it is never
    re-executed or restarted (it has touched the GPU)."""
    import threading  # pylint: disable=import-outside-toplevel

    key = "cfg4" if world > 1 else "cfg4_one_rank_communicator"

    def attach(entry, verdict):
        if result is not None:
            if result.get("configs") is None:
                result["configs"] = {}
            result["configs"][key] = entry
            result["strong_scaling"] = verdict

    def give_up():
        attach({"error": "the strong-scaling leg did not finish within %s s" % limit, "scaling": "strong", "n_gpus": world}, "timeout")
        if result is not None:
            print(json.dumps(result), flush=True)
        sys.stderr.write("[bench] the cfg4 strong-scaling leg timed out after %s s\n" % limit)
        sys.stderr.flush()
        os._exit(3)  # pylint: disable=protected-access

    # (rank 0 -- the one with the record -- gives up first: a launcher that sees another rank fail may stop rank 0 before it
    # has printed the line)
    watchdog = threading.Timer(limit if result is not None else limit + 30.0, give_up)
    watchdog.daemon = True
    watchdog.start()
    failed = False
    try:
        attach(leg(), "ok")
    except Exception as exc:  # pylint: disable=broad-except
        failed = True
        attach({"error": "%s: %s" % (type(exc).__name__, exc), "scaling": "strong", "n_gpus": world}, "failed")
    watchdog.cancel()
    return failed


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))  # no launcher: this process becomes one (it never touches the GPU)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))

    kind, n_orb, n_r, nk_gpu, cfg_idx = CONFIGS[args.config]
    if args.nk:
        nk_gpu = args.nk
    if args.nr:
        n_r = args.nr
    arrays = build_model_arrays(args.config, args.nr) if not args.dry_ranks else None
    if args.dry_ranks:
        # the launch mechanics of an N-rank run on the CPU: 5 orbitals, 12 lattice vectors, 96 k-points per rank, a 7^3 mesh for
        # the strong-scaling leg -- the stand-in evaluates slabs with the oracle, "eigenvalues per second" are meaningless
        r_dry, hop_dry, pos_dry = synthetic.dense_model_arrays(5, 12, synthetic.MODEL_SEED + 77)
        arrays = dict(kind="dense", n_orb=5, R=r_dry, hop=hop_dry, pos=pos_dry)
        n_orb, nk_gpu = 5, (args.nk or 96)
        os.environ.setdefault("TBK_BENCH_STRONG_MESH", "7")
    n_r = len(arrays["R"])
    dim = arrays["R"].shape[1]

    # the global k list (N x nk_gpu rows); every rank materialises only its slab
    if args.config == "cfg1":
        k_all = synthetic.uniform_grid(10)
        k_slab = k_all[:nk_gpu]
    elif args.config == "cfg4":
        from tbmodels_amd.sharding import slab_bounds  # pylint: disable=import-outside-toplevel

        if not args.nk:
            lo, hi = slab_bounds(100 ** 3, world, rank)
            nk_gpu = slab_bounds(100 ** 3, world, 0)[1]  # equal slabs for the all-gather (1e6 divides by 1, 2, 4, 8)
            hi = lo + nk_gpu
        else:
            lo, hi = rank * nk_gpu, (rank + 1) * nk_gpu
        k_slab = synthetic.grid_slab(100, lo, hi)
    else:
        rng = np.random.default_rng(synthetic.K_SEED)
        k_slab = None
        for r in range(rank + 1):  # same stream as random_kpoints(world * nk_gpu), slab by slab
            k_slab = rng.random((nk_gpu, dim))
    k_slab = np.ascontiguousarray(k_slab)

    cpu_all = None
    cpu_all_rows = None
    cpu_all_other = {}
    per_proc_all = {"cfg1": 64, "cfg2": 256, "cfg3": 24, "cfg4": 256, "cfg5": 2}  # ~5-10 s each
    run_others = (world == 1 and args.config == "cfg2" and not args.construct_only and not args.nk and not args.nr
                  and not args.dry_ranks and os.environ.get("TBK_BENCH_SKIP_CONFIGS") != "1")
    if rank == 0 and world == 1 and args.cpu_sample != 0 and not args.construct_only and not args.dry_ranks:
        cpu_all, cpu_all_rows = cpu_baseline_all_cores(arrays, k_slab, per_proc_all[args.config])
        if run_others and os.environ.get("TBK_BENCH_CPU_LIGHT") != "1":
            # all-core row of the sparse config too; it forks, so it runs here, before the GPU is touched
            arrays3 = build_model_arrays("cfg3")
            cpu_all_other["cfg3"] = cpu_baseline_all_cores(arrays3, config_kpoints("cfg3", CONFIGS["cfg3"][3], 3), per_proc_all["cfg3"])
            del arrays3

    group = group_from_env() if world > 1 else None
    if args.dry_ranks:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from standin_lib import StandInLib  # pylint: disable=import-outside-toplevel,import-error

        lib = StandInLib(group, world, rank, arrays)
        device = 0
    else:
        lib = _lib.lib()
        if _lib.device_count() < 1:
            raise SystemExit("bench.py needs a GPU: libtbk has no CPU path")
        device = local_rank % _lib.device_count()
    model = stage(lib, device, arrays)
    solver = {"auto": _lib.TBK_EIG_AUTO, "wave": _lib.TBK_EIG_WAVE, "rocsolver": _lib.TBK_EIG_ROCSOLVER}[args.eigensolver]
    _lib.check(lib.tbk_model_set_option(model, _lib.TBK_OPT_EIGENSOLVER, solver))
    if args.k_chunk:
        _lib.check(lib.tbk_model_set_option(model, _lib.TBK_OPT_K_CHUNK, args.k_chunk))

    def dmalloc(nbytes):
        p = ctypes.c_void_p()
        _lib.check(lib.tbk_device_malloc(device, nbytes, ctypes.byref(p)))
        return p

    d_k = dmalloc(k_slab.nbytes)
    _lib.check(lib.tbk_memcpy_h2d(device, d_k, _lib.ptr(k_slab), k_slab.nbytes))
    e_count = nk_gpu * n_orb
    d_e_pair = [dmalloc(e_count * 8), dmalloc(e_count * 8)]  # alternate per step: a gather may still read one
    d_e = d_e_pair[0]
    d_h = None
    if args.construct_only:
        d_h = dmalloc(nk_gpu * n_orb * n_orb * 16)
    d_gather = None
    comm = None
    collective = "none"
    rccl_ranks = 0
    host_gather = None
    force_comm = os.environ.get("TBK_BENCH_FORCE_COMM") == "1"  # exercise RCCL with a 1-rank communicator
    if world > 1 or force_comm:
        d_gather_pair = [dmalloc(world * e_count * 8), dmalloc(world * e_count * 8)]
        d_gather = d_gather_pair[0]
        uid = np.zeros(128, dtype=np.uint8)
        if rank == 0:
            _lib.check(lib.tbk_comm_unique_id(_lib.ptr(uid)))
        if group is not None:
            uid = np.frombuffer(group.broadcast_bytes(uid.tobytes(), src=0), dtype=np.uint8).copy()
        comm = ctypes.c_void_p()
        # RCCL prints a version banner when a communicator is created; keep stdout for the one JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            status = lib.tbk_comm_create(device, world, rank, _lib.ptr(uid), ctypes.byref(comm))
        finally:
            # the banner sits in the C library's stdout buffer: flush it while fd 1 still points at stderr, or it comes
            # out at exit, behind the JSON line
            try:
                ctypes.CDLL(None).fflush(None)
            except (OSError, AttributeError):
                pass
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
        all_ok = (status == 0) if group is None else group.allreduce_min(1.0 if status == 0 else 0.0) == 1.0
        if all_ok:
            collective = "rccl all-gather (xGMI) of device buffers on its own stream, overlapping the next step"
            n_ranks, my_rank = ctypes.c_int(0), ctypes.c_int(-1)
            _lib.check(lib.tbk_comm_ranks(comm, ctypes.byref(n_ranks), ctypes.byref(my_rank)))
            rccl_ranks = n_ranks.value  # ncclCommCount: what RCCL itself says joined
            if rccl_ranks != world or my_rank.value != rank:
                raise SystemExit("RCCL communicator has %d ranks (this one is %d), expected %d (rank %d)"
                                 % (rccl_ranks, my_rank.value, world, rank))
        else:
            # e.g. several ranks sharing one GPU: RCCL refuses.  A scaling run must not print a number RCCL never
            # produced: fail, unless the host-gather fallback was asked for explicitly (launch-mechanics tests)
            message = "[bench] RCCL communicator unavailable (%s)" % _lib.last_error()
            if os.environ.get("TBK_BENCH_ALLOW_HOST_GATHER") != "1":
                raise SystemExit(message + "; set TBK_BENCH_ALLOW_HOST_GATHER=1 to gather through the host instead")
            sys.stderr.write(message + "; TBK_BENCH_ALLOW_HOST_GATHER=1: falling back to a host all-gather\n")
            if status == 0:
                lib.tbk_comm_destroy(comm)
            comm = None
            collective = "host all-gather (RCCL unavailable, TBK_BENCH_ALLOW_HOST_GATHER=1)"
            h_slab = np.empty((nk_gpu, n_orb))

            def host_gather():
                _lib.check(lib.tbk_memcpy_d2h(device, _lib.ptr(h_slab), d_e_pair[(step_no[0] - 1) & 1], h_slab.nbytes))
                group.all_gather_array(h_slab)

    step_no = [0]
    k_hint = _lib.ptr(k_slab) if args.config == "cfg4" else None

    def step():
        if args.construct_only:
            _lib.check(lib.tbk_hamilton_device(model, d_k, nk_gpu, 2, None, d_h))
            return
        # the all-gather of step s runs on the communicator's stream under the kernels of step s+1; the
        # eigenvalue / gather buffers alternate, and before a pair is reused its gather (two steps back) is done
        pair = step_no[0] & 1
        step_no[0] += 1
        d_out = d_e_pair[pair]
        if comm is not None:
            _lib.check(lib.tbk_comm_wait_slot(comm, model, pair))
        # (the mesh config hands over the host array it uploaded: device lists are never read back, so runs of a
        # shared k component are only recognised through it -- include/tbk.h)
        _lib.check(lib.tbk_eigenval_device_hint(model, d_k, k_hint, nk_gpu, d_out))
        if comm is not None:
            _lib.check(lib.tbk_comm_allgather_f64_overlapped(comm, model, d_out, d_gather_pair[pair], e_count, pair))
        elif host_gather is not None:
            _lib.check(lib.tbk_synchronize(model))
            host_gather()

    def barrier():
        _lib.check(lib.tbk_synchronize(model))
        if comm is not None:
            _lib.check(lib.tbk_comm_synchronize(comm))
        if group is not None:
            group.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    if not args.construct_only:
        _lib.check(lib.tbk_eigenval_check(model))
    _lib.check(lib.tbk_model_set_option(model, _lib.TBK_OPT_TIMING, 1))
    ms = (ctypes.c_double * _lib.TBK_T_COUNT)()
    launches = (ctypes.c_int64 * _lib.TBK_T_COUNT)()
    _lib.check(lib.tbk_get_timing(model, ms, launches, 1))

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if group is not None:
        elapsed = group.allreduce_max(elapsed)
    if not args.construct_only:
        _lib.check(lib.tbk_eigenval_check(model))
    _lib.check(lib.tbk_get_timing(model, ms, launches, 1))
    stage_ms = {name: ms[i] for i, name in enumerate(_lib.STAGE_NAMES)}
    stage_n = {name: launches[i] for i, name in enumerate(_lib.STAGE_NAMES)}

    # --- after the clock: per-rank split of one step (N > 1), and the host-buffer API leg (N = 1) ---------------
    per_rank = None
    if world > 1 and comm is not None and not args.construct_only:
        split = np.zeros(2)
        barrier()
        t1 = time.perf_counter()
        _lib.check(lib.tbk_eigenval_device_hint(model, d_k, k_hint, nk_gpu, d_e_pair[0]))
        _lib.check(lib.tbk_synchronize(model))
        split[0] = (time.perf_counter() - t1) * 1e3
        barrier()
        t1 = time.perf_counter()
        _lib.check(lib.tbk_comm_allgather_f64(comm, model, d_e_pair[0], d_gather_pair[0], e_count))
        _lib.check(lib.tbk_synchronize(model))
        split[1] = (time.perf_counter() - t1) * 1e3
        parts = group.all_gather_array(split)
        per_rank = {"compute_ms": [round(float(p[0]), 3) for p in parts],
                    "allgather_ms": [round(float(p[1]), 3) for p in parts],
                    "note": "one step, not overlapped: eigenval on the rank's slab, then the RCCL all-gather alone"}
    host_api = None
    if world == 1 and rank == 0 and not args.construct_only and not args.dry_ranks and os.environ.get("TBK_BENCH_SKIP_HOSTAPI") != "1":
        # SURVEY 8(d) "Evidence": wall-clock through the drop-in surface -- host k in, host eigenvalues out (H2D of k,
        # all kernels, the non-finite check, D2H), i.e. tbk_eigenval, what Model.eigenval_array calls; plus the
        # reference's return type (a Python list of row arrays, _tb_model.py:1148-1150)
        e_host = np.empty((nk_gpu, n_orb))
        _lib.check(lib.tbk_eigenval(model, _lib.ptr(k_slab), nk_gpu, _lib.ptr(e_host)))  # sizes the staging buffers
        t1 = time.perf_counter()
        _lib.check(lib.tbk_eigenval(model, _lib.ptr(k_slab), nk_gpu, _lib.ptr(e_host)))
        dt_call = time.perf_counter() - t1
        t1 = time.perf_counter()
        as_list = list(e_host)
        dt_list = time.perf_counter() - t1
        del as_list
        # Z2Pack-style callers evaluate ONE k-point per call (_tb_model.py:1103-1108): wall-clock of such calls
        single = single_k_latency(lib, model, k_slab, n_orb, calls=32 if n_orb > 128 else 128, warm=4 if n_orb > 128 else 8, arrays=arrays)
        host_api = {
            "value": round(nk_gpu / dt_call, 1), "unit": "k-points/s", "ms_per_call": round(dt_call * 1e3, 3),
            "single_k_us": single,
            "includes": "H2D of k, all kernels, non-finite check, D2H of eigenvalues (tbk_eigenval on host buffers)",
            "list_return_ms": round(dt_list * 1e3, 3),
            "value_with_list_return": round(nk_gpu / (dt_call + dt_list), 1),
        }
        _lib.check(lib.tbk_get_timing(model, None, None, 1))  # drop the stage events of these two extra calls
    construct = None
    if world == 1 and rank == 0 and not args.construct_only and not args.dry_ranks and os.environ.get("TBK_BENCH_SKIP_HOSTAPI") != "1":
        # SURVEY 8(d) "Metric": k-points/s for H(k) construction alone, the result resident in HBM (tbk_hamilton_device, the
        # FULL matrix of Model.hamilton); and hard part 5: the batch hamilton() wall-clock through HOST buffers, PCIe-bound
        nk_c = min(nk_gpu, max(1, int(12e9 // (n_orb * n_orb * 16))))  # at most 12 GB of H
        d_hc = dmalloc(nk_c * n_orb * n_orb * 16)
        try:
            _lib.check(lib.tbk_hamilton_device(model, d_k, nk_c, 2, None, d_hc))
            _lib.check(lib.tbk_synchronize(model))
            c_steps = 3
            t1 = time.perf_counter()
            for _ in range(c_steps):
                _lib.check(lib.tbk_hamilton_device(model, d_k, nk_c, 2, None, d_hc))
            _lib.check(lib.tbk_synchronize(model))
            dt_c = (time.perf_counter() - t1) / c_steps
        finally:
            lib.tbk_device_free(device, d_hc)
        construct = {"value": round(nk_c / dt_c, 1), "unit": "k-points/s", "ms_per_step": round(dt_c * 1e3, 3), "steps": c_steps,
                     "kpoints": nk_c, "includes": "tbk_hamilton_device: phase rows + contraction + scatter to the full "
                     "H[k][i][j] (both triangles), k and H resident in HBM"}
        nk_h = min(nk_gpu, max(1, int(1.4e9 // (n_orb * n_orb * 16))))  # ~1.3 GB of H: 20 000 k-points at 64 orbitals
        h_host = np.zeros((nk_h, n_orb, n_orb), dtype=np.complex128)   # written once: a warm buffer (pages exist)
        _lib.check(lib.tbk_hamilton(model, _lib.ptr(k_slab), nk_h, 2, None, _lib.ptr(h_host)))
        t1 = time.perf_counter()
        _lib.check(lib.tbk_hamilton(model, _lib.ptr(k_slab), nk_h, 2, None, _lib.ptr(h_host)))
        dt_h = time.perf_counter() - t1
        h_trace = float(np.abs(np.trace(h_host[:64], axis1=1, axis2=2).imag).max())
        if host_api is not None:
            host_api["hamilton"] = {"value": round(nk_h / dt_h, 1), "unit": "k-points/s", "kpoints": nk_h,
                                    "GB/s": round(h_host.nbytes / dt_h / 1e9, 2), "ms_per_call": round(dt_h * 1e3, 3),
                                    "includes": "tbk_hamilton on host buffers: H2D of k, kernels, chunked D2H of H into a "
                                                "warm (already written) host array",
                                    "max_abs_imag_trace_64_rows": h_trace}
        del h_host
        _lib.check(lib.tbk_get_timing(model, None, None, 1))
    peak_measured = None
    if rank == 0 and arrays["kind"] == "dense" and os.environ.get("TBK_BENCH_SKIP_PEAK") != "1":
        tf = ctypes.c_double(0.0)
        _lib.check(lib.tbk_mfma_f64_peak(device, ctypes.byref(tf)))  # ~1 s: a bare v_mfma_f64_16x16x4_f64 loop
        peak_measured = round(tf.value, 2)

    # --- correctness checks on rank 0 (after the clock stopped) ----------------------------------
    # (i) the first k-points of the slab against the oracle (with the CPU baseline below);
    # (ii) sum_i E_i(k) = tr H(k) = sum_R 2 Re(e^{2 pi i k.R} tr hop[R]) on rows drawn from the whole
    #      slab, so every k chunk of the pipeline is covered (host work independent of the GPU path)
    eig_head = np.empty((min(nk_gpu, 64), n_orb))
    trace_err = None
    moment_err = None
    parity_all = None
    if not args.construct_only and rank == 0:
        d_e = d_e_pair[(step_no[0] - 1) & 1]  # the buffer of the last step
        _lib.check(lib.tbk_memcpy_d2h(device, _lib.ptr(eig_head), d_e, eig_head.nbytes))
        eig_all = np.empty((nk_gpu, n_orb))
        _lib.check(lib.tbk_memcpy_d2h(device, _lib.ptr(eig_all), d_e, eig_all.nbytes))
        trace_err = trace_identity_error(arrays, k_slab, eig_all)
        if not args.dry_ranks:
            moment_err = second_moment_error(lib, model, n_orb, k_slab, eig_all, 4096 if n_orb <= 256 else 512)
            _lib.check(lib.tbk_get_timing(model, None, None, 1))
        if cpu_all_rows is not None and len(cpu_all_rows):
            # every row the all-core CPU baseline computed (the first rows of this slab) against the GPU's eigenvalues
            parity_all = float(np.abs(cpu_all_rows - eig_all[:len(cpu_all_rows)]).max())
        del eig_all

    result = None
    if rank == 0:
        total_k = world * nk_gpu * args.steps
        value = total_k / elapsed
        roofline = hk_roofline_entry(arrays, n_orb, n_r, dim, stage_ms, stage_n, nk_gpu * args.steps, args.config,
                                     construct_only=args.construct_only, peak_measured=peak_measured,
                                     with_traffic=(args.config == "cfg2" and not args.nr))

        # the reduction to tridiagonal form (the dominant kernel of the configs above 64 orbitals): (16/3) n^3 flops per
        # matrix (eigensolve, values only) over the HIP-event time of the reduction stage on its own stream
        eig_roofline = None
        if not args.construct_only and stage_ms.get("eig", 0.0) > 0.0:
            eig_roofline = eig_roofline_entry(n_orb, nk_gpu * args.steps, stage_ms["eig"], args.steps)
            if world == 1 and not args.dry_ranks and os.environ.get("TBK_BENCH_SKIP_STANDALONE") != "1" and os.environ.get("TBK_BENCH_SKIP_HOSTAPI") != "1":
                eig_roofline["standalone"] = standalone_reduction(
                    lib, device, n_orb, 32768 if n_orb <= 64 else 8192 if n_orb <= 128 else 4096 if n_orb <= 256 else 2048)

        sample = args.cpu_sample
        if sample < 0:
            # ~10-15 s of single-core work each (17-19 ms per k-point at the headline shape)
            sample = {"cfg1": 1000, "cfg2": 640, "cfg3": 48, "cfg4": 640, "cfg5": 4}[args.config]
        cpu = None
        parity = None
        oracle_rows = 0
        if world > 1:
            sample = min(sample, 64)  # N > 1: the oracle only as the checker; the baseline is reported at N = 1
        if sample > 0 and not args.construct_only:
            cpu_rate, cpu_dt, cpu_eig = cpu_baseline(arrays, k_slab, sample)
            n_cmp = min(len(cpu_eig), len(eig_head))
            parity = float(np.abs(cpu_eig[:n_cmp] - eig_head[:n_cmp]).max())
            oracle_rows = n_cmp
            if parity_all is not None:
                parity = max(parity, parity_all)
                oracle_rows = max(n_cmp, len(cpu_all_rows))
        if sample > 0 and not args.construct_only and world == 1:
            cpu = {
                "value": round(cpu_rate, 2), "unit": "k-points/s", "cores": 1, "kind": "port",
                "sample": "%d of the %d k-points of this workload, oracle/tbk_oracle.py (NumPy loop over R + "
                          "scipy eigvalsh), one process, %.1f s" % (sample, nk_gpu, cpu_dt),
                "host_cpus": os.cpu_count(),
            }
        # the other BASELINE configs at full size, after the clock (not part of value / ms_per_step): the cfg2 model on
        # the 100^3 mesh (cfg4, this GPU's share at N = 1 = the whole mesh) reuses the staged handle
        other = None
        if run_others:
            other = {}
            _lib.check(lib.tbk_model_set_option(model, _lib.TBK_OPT_EIGENSOLVER, _lib.TBK_EIG_AUTO))
            other["cfg4"] = run_other_config(lib, device, "cfg4", model=model, arrays=arrays)
            for name in ("cfg1", "cfg3", "cfg5"):
                entry_all, rows_all = cpu_all_other.get(name, (None, None))
                other[name] = run_other_config(lib, device, name, cpu_all=entry_all, cpu_all_rows=rows_all)
        result = {
            "metric": "k-points/sec (H(k)+eig) at N_orb=%d, N_R=%d" % (n_orb, n_r) if not args.construct_only
                      else "k-points/sec (H(k) construction only) at N_orb=%d, N_R=%d" % (n_orb, n_r),
            "value": round(value, 1),
            # the same call on HOST buffers (H2D of k and D2H of the eigenvalues inside: SURVEY 8(d) "Evidence"); N = 1 only
            "value_host_buffers": host_api["value"] if host_api else None,
            "unit": "k-points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong" if args.config == "cfg4" and not args.nk else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" if not args.dry_ranks else "synthetic; DRY RUN on the CPU (tools/standin_lib.py in place of libtbk, a 5-orbital model): launch mechanics only, the numbers mean nothing",
            "config": {
                "workload": "%s: %s N_orb=%d N_R=%d, %d %s k-points per GPU, eigenval (H(k)+eig)"
                            % (args.config, arrays["kind"], n_orb, n_r, nk_gpu,
                               "grid" if args.config in ("cfg1", "cfg4") else "random"),
                "kpoints_per_gpu": nk_gpu,
                "sharding": "contiguous k slabs, hoppings replicated, all-gather of eigenvalue slabs" if world > 1
                            else "single GPU",
                "collective": collective,
                "rccl_ranks": rccl_ranks,
                "eigensolver": args.eigensolver,
            },
            "roofline": roofline,
            "eig_roofline": eig_roofline,
            "cpu_baseline": cpu,
            "cpu_baseline_all_cores": cpu_all,
            "host_api": host_api,
            "rccl_ranks": rccl_ranks,
            "per_rank": per_rank,
            "configs": other,
            "stage_ms_per_step": {k: round(v / args.steps, 3) for k, v in stage_ms.items()},
            "max_abs_err_vs_oracle": parity,
            "oracle_sample": "%d rows (the first %d k-points of the slab%s)"
                             % (oracle_rows, oracle_rows, ": the rows of cpu_baseline_all_cores" if parity_all is not None else ""),
            "max_trace_identity_err_4096_rows": trace_err,
            "max_second_moment_err": moment_err,
            "construct_only": construct,
            "strong_scaling": None,
        }
    # --- after everything else: BASELINE config 4, the one STRONG-scaling config -- the 100^3 mesh of the staged model in
    # `world` contiguous slabs, every rank's slab evaluated into its rows of the full result with the RCCL all-gather of
    # finished row blocks pipelined behind the k chunks (tbk_eigenval_device_gather), the gather INSIDE the timing.  Every
    # rank takes part (N = 1 only with TBK_BENCH_FORCE_COMM=1: a one-rank communicator).  The main line above is complete
    # at this point: if the leg fails or does not come back (a rank lost in a collective), the line is printed WITHOUT it
    # -- with the error under configs.cfg4 -- instead of being lost with the process.
    strong_failed = False
    if comm is not None and args.config == "cfg2" and not args.construct_only and not args.nr \
            and ((os.environ.get("TBK_BENCH_SKIP_CONFIGS") != "1" and not args.nk) or os.environ.get("TBK_BENCH_STRONG") == "1"):
        strong_failed = guarded_strong_leg(
            result, world, float(os.environ.get("TBK_BENCH_STRONG_TIMEOUT", "240")),
            lambda: strong_scaling_leg(lib, device, model, comm, group, world, rank, n_orb, arrays,
                                       mesh=int(os.environ.get("TBK_BENCH_STRONG_MESH", "100"))))
    if rank == 0 and result is not None:
        result["summary"] = make_summary(result)  # LAST key: what survives in a record that keeps the end of the line
        print(json.dumps(result), flush=True)
    if strong_failed:  # the main line stands and says "strong_scaling": "failed"; the exit status below reports it too
        sys.stderr.write("[bench] the cfg4 strong-scaling leg failed on rank %d: see configs in the JSON line\n" % rank)

    if comm is not None:
        lib.tbk_comm_destroy(comm)
    lib.tbk_model_destroy(model)
    if group is not None:
        group.close()
    if rank == 0 and result:
        for key in ("max_abs_err_vs_oracle", "max_trace_identity_err_4096_rows", "max_second_moment_err"):
            if result.get(key) is not None and not result[key] <= 1e-10:
                raise SystemExit("parity failure: %s = %g" % (key, result[key]))
    if strong_failed:
        sys.exit(1)


if __name__ == "__main__":
    main()

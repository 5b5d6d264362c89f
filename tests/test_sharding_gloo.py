"""
The N > 1 path on CPU: world_size-2 ``gloo`` process group, contiguous k slabs, all-gather of
eigenvalue slabs, result in caller order on every rank.  The per-rank evaluator is the oracle here
(no GPU in this suite); on a GPU box the same class evaluates slabs through libtbk and gathers over RCCL.
"""

import os
import socket
import sys

import numpy as np
import pytest

from tbmodels_amd.sharding import slab_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_slab_bounds_cover_and_order():
    for n_k in (0, 1, 2, 7, 8, 9, 1000, 100_000):
        for world in (1, 2, 3, 4, 8):
            edges = [slab_bounds(n_k, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n_k
            for (a0, a1), (b0, b1) in zip(edges, edges[1:]):
                assert a1 == b0 and a0 <= a1 and b0 <= b1
            per = -(-n_k // world)
            assert all(hi - lo <= per for lo, hi in edges)


def _free_port():
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def _worker(rank, world, port, n_k, out_dir, backend="gloo"):
    sys.path.insert(0, ROOT)
    import tbmodels_amd
    from tbmodels_amd import synthetic as syn
    from tbmodels_amd.sharding import ShardedEigenval
    from oracle import tbk_oracle as oracle

    if backend == "gloo":
        import torch.distributed as dist

        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
        group = dist
    else:
        from tbmodels_amd.rendezvous import FileGroup

        group = FileGroup(rank, world, os.path.join(out_dir, "rdzv"))
    r_vec, hop, pos = syn.dense_model_arrays(6, 10, syn.MODEL_SEED + 42)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    calls = []

    def evaluate(k_slab):
        calls.append(len(k_slab))
        return np.array(oracle.eigenval(r_vec, hop, k_slab))

    sharded = ShardedEigenval(model, group, device=None, evaluate=evaluate)
    k = syn.random_kpoints(n_k)
    result = sharded(k)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), result)
    np.save(os.path.join(out_dir, "calls%d.npy" % rank), np.array(calls))
    if backend == "gloo":
        group.barrier()
        group.destroy_process_group()
    else:
        assert sharded.group.allreduce_max(float(rank)) == world - 1
        assert sharded.group.broadcast_bytes(b"id-%d" % rank, src=1) == b"id-1"
        sharded.group.close()


@pytest.mark.parametrize(
    "backend,n_k,world",
    [("gloo", 11, 2), ("gloo", 64, 2), ("gloo", 1, 2), ("file", 11, 2), ("file", 2, 2), ("file", 37, 8), ("file", 5, 8)],
)
def test_world_size_n_allgather(tmp_path, backend, n_k, world):
    import multiprocessing as mp

    from tbmodels_amd import synthetic as syn
    from oracle import tbk_oracle as oracle

    port = _free_port()
    # fresh interpreters: torch (gloo) is imported only inside the workers -- never in this process,
    # which may already hold libtbk and its system ROCm runtime (see tbmodels_amd/rendezvous.py)
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(rank, world, port, n_k, str(tmp_path), backend)) for rank in range(world)]
    for proc in procs:
        proc.start()
    for proc in procs:
        proc.join(timeout=300)
        assert proc.exitcode == 0, "worker exited with %r" % proc.exitcode
    r_vec, hop, _ = syn.dense_model_arrays(6, 10, syn.MODEL_SEED + 42)
    expected = np.array(oracle.eigenval(r_vec, hop, syn.random_kpoints(n_k)))
    total_calls = 0
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % rank))
        assert got.shape == expected.shape
        assert np.abs(got - expected).max() < 1e-13  # every rank holds the full, ordered result
        calls = np.load(os.path.join(str(tmp_path), "calls%d.npy" % rank))
        lo, hi = slab_bounds(n_k, world, rank)
        assert calls.sum() == hi - lo  # each rank evaluated exactly its slab
        total_calls += calls.sum()
    assert total_calls == n_k


def _failing_worker(rank, world, out_dir, token):
    sys.path.insert(0, ROOT)
    import tbmodels_amd
    from tbmodels_amd import synthetic as syn
    from tbmodels_amd.rendezvous import FileGroup
    from tbmodels_amd.sharding import ShardedEigenval
    from oracle import tbk_oracle as oracle

    group = FileGroup(rank, world, os.path.join(out_dir, "rdzv"), token=token)
    r_vec, hop, pos = syn.dense_model_arrays(6, 10, syn.MODEL_SEED + 42)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)

    def evaluate(k_slab):
        if not np.isfinite(k_slab).all():  # scipy's check_finite on this rank's slab only
            raise ValueError("array must not contain infs or NaNs")
        return np.array(oracle.eigenval(r_vec, hop, k_slab))

    sharded = ShardedEigenval(model, group, device=None, evaluate=evaluate)
    k = syn.random_kpoints(12)
    k[1, 0] = np.nan  # lands in rank 0's slab
    try:
        sharded(k)
        outcome = "returned"
    except ValueError:
        outcome = "ValueError"
    # both ranks are still in step: the next collective and a clean second call go through
    good = sharded(syn.random_kpoints(12))
    assert good.shape == (12, 6)
    with open(os.path.join(out_dir, "outcome%d.txt" % rank), "w") as handle:
        handle.write(outcome)
    group.close()


def test_failure_on_one_rank_is_raised_on_every_rank(tmp_path):
    """A NaN in ONE rank's slab: every rank raises (the others used to return the gathered NaN rows and run on into
    the next collective alone), and the group stays usable.  The rendezvous directory still holds files of an
    earlier run with the same names: the run token keeps them out."""
    import multiprocessing as mp

    rdzv = tmp_path / "rdzv"
    rdzv.mkdir()
    for name in ("ag.000001.r0", "ag.000001.r1", "bc.000001", "stale-run.ag.000001.r1"):
        (rdzv / name).write_bytes(b"\xff" * 8)  # leftovers of a crashed run (old naming and another token)
    # ... and of a crashed run with THIS token (a re-run with the same --rdzv-id and port): its hand-shake, its
    # broadcast (an old RCCL id) and its last exchange -- the fresh name space of the hand-shake keeps them out, and rank
    # 0 removes them
    (rdzv / "this-run.hello").write_text("0123456789abcdef fedcba9876543210")
    (rdzv / "this-run.hi.r1").write_text("fedcba9876543210")
    for name in ("this-run-0123456789abcdef.go", "this-run-0123456789abcdef.bc.000001", "this-run-0123456789abcdef.ag.000002.r1",
                 "this-run.bc.000001", "this-run.ag.000001.r0"):
        (rdzv / name).write_bytes(b"\xff" * 8)
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_failing_worker, args=(rank, 2, str(tmp_path), "this-run")) for rank in range(2)]
    for proc in procs:
        proc.start()
    for proc in procs:
        proc.join(timeout=300)
        assert proc.exitcode == 0, "worker exited with %r" % proc.exitcode
    for rank in range(2):
        assert (tmp_path / ("outcome%d.txt" % rank)).read_text() == "ValueError"
    assert sorted(p.name for p in rdzv.iterdir()) == ["ag.000001.r0", "ag.000001.r1", "bc.000001", "stale-run.ag.000001.r1"]

"""
Process groups for the sharded path: the few host-side collectives it needs (rank / world, barrier,
broadcast of the 128-byte RCCL id, max-reduce of a timing, host all-gather for GPU-less runs).

Two implementations of one small interface (``rank``, ``world``, ``barrier()``, ``broadcast_bytes()``,
``allreduce_max()``, ``all_gather_array()``):

``FileGroup``   torch-free, single node: a directory of atomically renamed files.  This is what
                ``bench.py`` uses under ``python -m torch.distributed.run`` (which only has to set
                RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT): importing torch next to libtbk would put a
                second ROCm runtime build into the process (the wheel bundles its own), which crashes
                when the system runtime is already loaded.
``TorchGroup``  adapter over an initialised ``torch.distributed`` (gloo or nccl) for callers that live
                in a torch process anyway; import torch BEFORE the first ``tbmodels_amd`` GPU call there.

The data path (eigenvalue slabs) never goes through these on a GPU box: it is the RCCL all-gather of
``tbk_comm_allgather_f64``.
"""

import os
import time

import numpy as np

__all__ = ("FileGroup", "TorchGroup", "group_from_env")


def _run_token():
    """
    A string that is the same in every rank of ONE launch and differs between launches: the launcher's pid plus its
    start time (``/proc/<ppid>/stat`` field 22, so a recycled pid does not collide) -- or, when the launcher names its
    run (``TORCHELASTIC_RUN_ID`` other than torchrun's static default "none"), that id plus ``MASTER_PORT``.
    ``TBK_RDZV_TOKEN`` overrides both for ranks that share neither (``bench.py --gpus N`` sets it for its ranks).
    """
    token = os.environ.get("TBK_RDZV_TOKEN")
    if token:
        return token
    run_id = os.environ.get("TORCHELASTIC_RUN_ID", "")
    if run_id and run_id != "none" and os.environ.get("MASTER_PORT"):
        # a launcher-wide id: also right when the ranks are started through per-rank wrapper shells (no shared parent).
        # The restart count tells the worker generations of ONE elastic run apart (``--max-restarts``: same id, same
        # port, same directory -- and the files of the generation that crashed are still there).  A re-run that reuses
        # --rdzv-id and the port after a crash is told apart by FileGroup's first exchange (a nonce, below).
        return "%s-%s-r%s" % (run_id, os.environ["MASTER_PORT"], os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"))
    ppid = os.getppid()
    start = "0"
    try:
        with open("/proc/%d/stat" % ppid) as handle:
            start = handle.read().rsplit(")", 1)[1].split()[19]  # field 22 (starttime), counted after "(comm)"
    except (OSError, IndexError):
        pass
    return "%d-%s" % (ppid, start)


class FileGroup:
    """
    Single-node process group backed by a shared directory (``/dev/shm`` when available).

    Every file name carries the launch's run token (:func:`_run_token`): a directory that still holds the files of an
    earlier, crashed run -- a reused ``TBK_RDZV_DIR`` -- cannot feed this run a stale RCCL id or stale slabs.
    """

    def __init__(self, rank, world, path, poll_s=2e-4, timeout_s=600.0, token=None, first_timeout_s=None):
        self.rank = int(rank)
        self.world = int(world)
        self.path = path
        self.poll_s = poll_s
        self.timeout_s = timeout_s
        # the FIRST exchange is where ranks with different tokens / directories never meet: fail that one fast
        if first_timeout_s is None:
            first_timeout_s = float(os.environ.get("TBK_RDZV_FIRST_TIMEOUT", "180"))
        self.first_timeout_s = min(float(first_timeout_s), timeout_s)
        self.token = token or _run_token()
        self._seq = 0
        self._met = False     # one exchange with every peer has completed
        self._mine = []       # (sequence number, path) of the files this rank wrote and has not removed yet
        os.makedirs(path, exist_ok=True)
        self._fresh = self.world == 1  # the hand-shake below has given this launch its own name space

    # -- a name space no earlier launch can have written into ------------------------------------
    def _handshake(self):
        """
        The run token alone does not tell a launch from an earlier one that used the same token -- a re-run with the same
        ``--rdzv-id`` and port after a crash, whose last exchange and broadcast files (an old RCCL id!) are still in the
        directory.  So the first exchange is a two-way hand-shake on fresh random numbers: every peer r writes
        ``<token>.hi.r<r>`` with a nonce P_r of its own; rank 0 keeps publishing ``<token>.hello`` = (its nonce N, the
        P_r it currently sees); a peer accepts a hello only if it carries ITS nonce -- such a hello was written after this
        peer started, by this launch's rank 0 -- and acknowledges inside the name space ``<token>-<N>``, which rank 0
        waits for.  From then on every file name carries ``<token>-<N>``: stale files cannot match.
        """
        base = self.token
        deadline = time.monotonic() + self.first_timeout_s

        def write(name, text):
            tmp = os.path.join(self.path, ".%s.%d.tmp" % (name, self.rank))
            with open(tmp, "w") as handle:
                handle.write(text)
            os.replace(tmp, os.path.join(self.path, name))

        def read(name):
            try:
                with open(os.path.join(self.path, name)) as handle:
                    return handle.read()
            except OSError:
                return None

        def expired(what):
            if time.monotonic() > deadline:
                raise TimeoutError(
                    "rendezvous: rank %d of %d waited %.0f s for %s (run token %r, directory %r: every rank of a launch "
                    "must see the same two -- set TBK_RDZV_TOKEN / TBK_RDZV_DIR when the ranks do not share a parent "
                    "process)" % (self.rank, self.world, self.first_timeout_s, what, base, self.path))
            time.sleep(self.poll_s)

        nonce = os.urandom(8).hex()
        if self.rank == 0:
            space = "%s-%s" % (base, nonce)
            seen = None
            while True:
                peers = [read("%s.hi.r%d" % (base, r)) or "" for r in range(1, self.world)]
                if peers != seen:
                    write("%s.hello" % base, " ".join([nonce] + [p or "-" for p in peers]))
                    seen = peers
                if all(read("%s.ack.r%d" % (space, r)) is not None for r in range(1, self.world)):
                    break
                expired("the peers' acknowledgements")
            for name in os.listdir(self.path):  # the hand-shake files of this launch, and whatever older launches left
                stale = (name.startswith(base + ".") or name.startswith(base + "-")) and not name.startswith(space + ".")
                if stale or name.startswith(space + ".ack."):
                    try:
                        os.unlink(os.path.join(self.path, name))
                    except OSError:
                        pass
            write("%s.go" % space, "1")
            self._mine.append((0, os.path.join(self.path, "%s.go" % space)))
        else:
            write("%s.hi.r%d" % (base, self.rank), nonce)
            while True:
                hello = (read("%s.hello" % base) or "").split()
                if len(hello) == self.world and hello[self.rank] == nonce:
                    break
                expired("rank 0's hello")
            space = "%s-%s" % (base, hello[0])
            write("%s.ack.r%d" % (space, self.rank), "1")
            while read("%s.go" % space) is None:
                expired("rank 0's go")
        self.token = space
        self._fresh = True

    # -- primitives ---------------------------------------------------------------------------
    def _put(self, name, data):
        name = "%s.%s" % (self.token, name)
        tmp = os.path.join(self.path, ".%s.%d.tmp" % (name, self.rank))
        with open(tmp, "wb") as handle:
            handle.write(data)
        final = os.path.join(self.path, name)
        os.replace(tmp, final)  # atomic: readers never see a partial file
        self._mine.append((self._seq, final))

    def _get(self, name):
        target = os.path.join(self.path, "%s.%s" % (self.token, name))
        limit = self.timeout_s if self._met else self.first_timeout_s
        deadline = time.monotonic() + limit
        while True:
            try:
                with open(target, "rb") as handle:
                    return handle.read()
            except FileNotFoundError:
                if time.monotonic() > deadline:
                    raise TimeoutError(
                        "rendezvous: rank %d of %d waited %.0f s for %s (run token %r, directory %r: every rank of a "
                        "launch must see the same two -- set TBK_RDZV_TOKEN / TBK_RDZV_DIR when the ranks do not share "
                        "a parent process)" % (self.rank, self.world, limit, target, self.token, self.path))
                time.sleep(self.poll_s)

    def _retire(self, before):
        """Remove this rank's files of exchanges older than sequence number `before`.  Called after an all-gather with
        that number completed: every rank has then written its part of it, i.e. finished reading everything earlier,
        so the number of files (each a tmpfs page and an inode) stays bounded however long the run."""
        keep = []
        for seq, path in self._mine:
            if seq < before:
                try:
                    os.unlink(path)
                except OSError:
                    pass
            else:
                keep.append((seq, path))
        self._mine = keep

    def _next(self, tag):
        if not self._fresh:
            self._handshake()
        self._seq += 1
        return "%s.%06d" % (tag, self._seq)

    # -- collectives --------------------------------------------------------------------------
    def all_gather_bytes(self, data):
        """Every rank contributes ``data``; returns the list of all contributions in rank order."""
        key = self._next("ag")
        self._put("%s.r%d" % (key, self.rank), bytes(data))
        parts = [self._get("%s.r%d" % (key, r)) for r in range(self.world)]
        self._met = True
        self._retire(self._seq)
        return parts

    def barrier(self):
        self.all_gather_bytes(b"\x01")

    def broadcast_bytes(self, data, src=0):
        key = self._next("bc")
        if self.rank == src:
            self._put(key, bytes(data))
            return bytes(data)
        return self._get(key)

    def allreduce_max(self, value):
        parts = self.all_gather_bytes(np.float64(value).tobytes())
        return float(max(np.frombuffer(p, dtype=np.float64)[0] for p in parts))

    def allreduce_min(self, value):
        parts = self.all_gather_bytes(np.float64(value).tobytes())
        return float(min(np.frombuffer(p, dtype=np.float64)[0] for p in parts))

    def all_gather_array(self, array):
        """All-gather of equally shaped float64 arrays; returns a list of arrays in rank order."""
        array = np.ascontiguousarray(array, dtype=np.float64)
        parts = self.all_gather_bytes(array.tobytes())
        return [np.frombuffer(p, dtype=np.float64).reshape(array.shape) for p in parts]

    def close(self):
        """Final barrier; rank 0 removes the directory once every other rank has said it is done reading."""
        self.barrier()
        if self.rank != 0:
            self._put("bye.r%d" % self.rank, b"\x01")
            return
        for r in range(1, self.world):
            self._get("bye.r%d" % r)
        # this run's files only: the directory may be a user-supplied one that other runs share
        for name in os.listdir(self.path):
            if name.startswith(self.token + ".") or name.startswith("." + self.token + "."):
                try:
                    os.unlink(os.path.join(self.path, name))
                except OSError:
                    pass
        try:
            os.rmdir(self.path)
        except OSError:
            pass


class TorchGroup:
    """The same interface over ``torch.distributed`` (CPU tensors; works with gloo and nccl groups)."""

    def __init__(self, dist):
        import torch  # pylint: disable=import-outside-toplevel

        self._torch = torch
        self._dist = dist
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()

    def barrier(self):
        self._dist.barrier()

    def broadcast_bytes(self, data, src=0):
        buf = np.frombuffer(bytes(data), dtype=np.uint8).copy()
        tensor = self._torch.from_numpy(buf)
        self._dist.broadcast(tensor, src=src)
        return buf.tobytes()

    def _reduce(self, value, op):
        tensor = self._torch.tensor([float(value)], dtype=self._torch.float64)
        self._dist.all_reduce(tensor, op=op)
        return float(tensor.item())

    def allreduce_max(self, value):
        return self._reduce(value, self._dist.ReduceOp.MAX)

    def allreduce_min(self, value):
        return self._reduce(value, self._dist.ReduceOp.MIN)

    def all_gather_array(self, array):
        array = np.ascontiguousarray(array, dtype=np.float64)
        pieces = [self._torch.zeros(array.shape, dtype=self._torch.float64) for _ in range(self.world)]
        self._dist.all_gather(pieces, self._torch.from_numpy(array))
        return [p.numpy() for p in pieces]

    def close(self):
        self.barrier()


def group_from_env():
    """
    The :class:`FileGroup` of a ``torch.distributed.run`` / ``torchrun`` / ``bench.py --gpus N`` launch on one node:
    ranks from RANK / WORLD_SIZE, directory named after MASTER_PORT and the run token (:func:`_run_token`: the same in
    every rank of one launch).  ``TBK_RDZV_DIR`` overrides the directory.
    """
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    token = _run_token()
    path = os.environ.get("TBK_RDZV_DIR")
    if not path:
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
        safe = "".join(c if c.isalnum() or c in "-_" else "_" for c in token)
        path = os.path.join(base, "tbk_rdzv_%s_%s" % (os.environ.get("MASTER_PORT", "0"), safe))
    return FileGroup(rank, world, path, token=token)

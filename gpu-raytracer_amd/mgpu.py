"""Host side of the N-GPU image split above the C ABI (include/okrt.h, "multi-GPU"): who drives which rank, how the
ranks of a multi-process job find each other, and the gather / reduce calls — no torch, no import-order rules.

Three ways a job runs (`launch_plan`):

  single            one process, one context (N = 1)
  one_process       `python bench.py --gpus N` started plainly: ONE process drives N contexts, rank r on device r
                    (okrt_comm_create_all + okrt_mgpu_gather_accum_all: one RCCL group around all sends / receives)
  process_per_gpu   started by a launcher that exports RANK / WORLD_SIZE / LOCAL_RANK (torch.distributed.run, mpirun,
                    a shell loop): one process per rank; rank 0 makes the RCCL id (okrt_comm_unique_id), the others read
                    it from a FileStore, everybody calls okrt_comm_create + okrt_mgpu_gather_accum

With fewer GPUs than ranks the same code paths run as a REHEARSAL (ranks share GPUs): one_process uses the library's
copy transport (RCCL refuses two ranks on one device), process_per_gpu gathers through host memory. The reference is
single-GPU (one immediate context, DX11.cpp:12-19); pixels are independent and seeds use global coordinates
(RaytracerCS.hlsl:430), which is what makes the split exact.
"""
import ctypes as C
import json
import os
import tempfile
import time

import numpy as np

from . import _abi, tiling

P = C.c_void_p


def launch_plan(gpus_arg, env, visible_gpus):
    """Decides the mode from the command line and the environment alone (no GPU call).

    Returns dict(mode, world, rank, local_ranks=[ranks this process drives], devices=[device of each of them],
    rehearsal=bool). `visible_gpus` may be None when not yet known (then devices / rehearsal are filled for a machine
    with enough GPUs)."""
    world_env = int(env.get("WORLD_SIZE", "0") or 0)
    if world_env > 1:  # a launcher started one process per rank; it decides the world size, not --gpus
        rank = int(env.get("RANK", "0"))
        local_rank = int(env.get("LOCAL_RANK", str(rank)))
        if not 0 <= rank < world_env:
            raise ValueError(f"RANK {rank} outside WORLD_SIZE {world_env}")
        vis = visible_gpus if visible_gpus else world_env
        # a launcher that narrows every rank's view to its own GPU (HIP_ / ROCR_ / CUDA_VISIBLE_DEVICES naming one device)
        # leaves one visible GPU per process, and they are DIFFERENT GPUs: device 0 everywhere, not a rehearsal
        masks = [env.get(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES") if env.get(k)]
        own_gpu = visible_gpus == 1 and any(len(m.split(",")) == 1 for m in masks)
        return dict(mode="process_per_gpu", world=world_env, rank=rank, local_ranks=[rank], devices=[0 if own_gpu else local_rank % vis],
                    rehearsal=visible_gpus is not None and visible_gpus < world_env and not own_gpu)
    n = max(int(gpus_arg), 1)
    if n == 1:
        return dict(mode="single", world=1, rank=0, local_ranks=[0], devices=[0], rehearsal=False)
    vis = visible_gpus if visible_gpus else n
    return dict(mode="one_process", world=n, rank=0, local_ranks=list(range(n)), devices=[r % vis for r in range(n)],
                rehearsal=visible_gpus is not None and visible_gpus < n)


def settle_shared_devices(plan, store, env):
    """Under a launcher no rank can tell alone whether its peers sit on the same GPU: a box whose one GPU is exposed through
    a visibility mask looks, from inside a rank, exactly like a launcher that masks a GPU per rank. So the ranks compare notes
    over the store: two ranks of one node with the same masks and the same device index share a GPU, and the job is a
    REHEARSAL (host-memory exchange; RCCL cannot put two ranks on one device). Collective; returns the settled plan."""
    if plan["mode"] != "process_per_gpu" or store is None:
        return plan
    import json
    import socket
    mine = json.dumps([socket.gethostname(), [env.get(k, "") for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")],
                       plan["devices"][0]]).encode()
    seats = store.allgather(mine)
    return dict(plan, rehearsal=len(set(seats)) < len(seats))


def visible_gpus(fns):
    n = C.c_int(0)
    st = fns.device_count(C.byref(n))
    if st != 0 or n.value < 1:
        raise _abi.OkrtError("no GPU visible to libokrt.so (okrt_device_count): the product has no CPU path")
    return n.value


class FileStore:
    """Rendezvous of the ranks of ONE node through a directory (all ranks share a parent process, the launcher): atomic
    small files for set / get, counted barriers, all-gather of small payloads. Enough to hand the 128-byte RCCL id from
    rank 0 to the others and to bracket a timed region; it carries no pixel data except in rehearsal mode."""

    def __init__(self, rank, world, key=None, root=None, timeout=600.0):
        self.rank, self.world, self.timeout = rank, world, timeout
        if key is None:
            key = os.environ.get("OKRT_RDZV_KEY") or default_rendezvous_key(os.environ)
        base = root or ("/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir())
        self.dir = os.path.join(base, "okrt_rdzv_" + "".join(ch if ch.isalnum() or ch in "-_." else "_" for ch in key))
        os.makedirs(self.dir, exist_ok=True)
        self._seq = 0

    def set(self, name, payload):
        tmp = os.path.join(self.dir, f".{name}.{self.rank}.tmp")
        with open(tmp, "wb") as f:
            f.write(payload)
        os.replace(tmp, os.path.join(self.dir, name))  # atomic: a reader sees nothing or everything

    def get(self, name):
        path = os.path.join(self.dir, name)
        deadline = time.monotonic() + self.timeout
        delay = 1e-4
        while True:
            try:
                with open(path, "rb") as f:
                    return f.read()
            except FileNotFoundError:
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rank {self.rank}: {path} did not appear within {self.timeout} s (did a peer die?)")
                time.sleep(delay)
                delay = min(delay * 1.5, 2e-3)

    def allgather(self, payload):
        """Every rank's bytes, in rank order. Collective: all ranks must call it the same number of times."""
        self._seq += 1
        self.set(f"ag{self._seq}.{self.rank}", payload)
        return [self.get(f"ag{self._seq}.{r}") for r in range(self.world)]

    def barrier(self):
        self.allgather(b"")

    def tight_barrier(self, gather_first=False):
        """A barrier the ranks LEAVE within microseconds of each other and that costs microseconds once the last rank has arrived:
        what brackets a timed region (the file-polling barrier lets its early arrivals sleep up to 2 ms between looks, which is
        the length of an 8-GPU, 20-frame region). Every rank bumps its own 8-byte counter in a shared memory-mapped file (one
        64-byte line per rank) and SPINS until every counter has reached the round. gather_first = True puts the ordinary
        barrier in front, so that ranks which may be seconds apart (scene set-up) do their waiting asleep."""
        import mmap
        import struct
        if getattr(self, "_bar", None) is None:
            path = os.path.join(self.dir, "tight_barrier")
            # a directory reused under the same OKRT_RDZV_KEY may hold an old file with large counters, which would let the first
            # rounds pass without waiting: rank 0 writes a zeroed file first, the others map it only after an ordinary barrier
            if self.rank == 0:
                with open(path, "wb") as f:
                    f.write(b"\0" * (64 * self.world))
            self.barrier()
            fd = os.open(path, os.O_RDWR)
            try:
                self._bar = mmap.mmap(fd, 64 * self.world)
            finally:
                os.close(fd)
            self._bar_round = 0
        if gather_first:
            self.barrier()
        self._bar_round += 1
        struct.pack_into("<Q", self._bar, 64 * self.rank, self._bar_round)
        deadline = time.monotonic() + self.timeout
        for r in range(self.world):
            spins = 0
            while struct.unpack_from("<Q", self._bar, 64 * r)[0] < self._bar_round:
                spins += 1
                if spins & 0xFFF == 0:
                    now = time.monotonic()
                    if now > deadline:
                        raise TimeoutError(f"rank {self.rank}: rank {r} never reached tight barrier {self._bar_round}")
                    if now - (deadline - self.timeout) > 0.005:
                        time.sleep(0.0005)  # a peer that is milliseconds late is not about to arrive within microseconds: stop burning the core

    def close(self):
        """Collective. After a last barrier (every rank has finished every earlier read) each rank signs off; rank 0
        waits for all sign-offs — nobody reads after writing theirs — and removes the directory."""
        import shutil
        self.barrier()
        if getattr(self, "_bar", None) is not None:
            self._bar.close()
            self._bar = None
        self.set(f"bye.{self.rank}", b"")
        if self.rank == 0:
            for r in range(self.world):
                self.get(f"bye.{r}")
            shutil.rmtree(self.dir, ignore_errors=True)


def default_rendezvous_key(env):
    """Unique per job on this node: the launcher's run id and port plus the identity (pid + start time) of the parent
    process all ranks share. Override with OKRT_RDZV_KEY when ranks are started from different parents."""
    ppid = os.getppid()
    start = "0"
    try:
        with open(f"/proc/{ppid}/stat") as f:
            start = f.read().rsplit(")", 1)[1].split()[19]  # field 22: start time in clock ticks since boot
    except (OSError, IndexError):
        pass
    return f"{env.get('TORCHELASTIC_RUN_ID', 'run')}_{env.get('MASTER_PORT', '0')}_{ppid}_{start}"


class TileGroup:
    """The ranks this process drives (contexts with their tile set) plus the communicator(s) that join them with the rest
    of the job. `gather()` / `reduce()` are collective over the whole job and asynchronous on the contexts' streams
    (the host-memory rehearsal excepted: it reads the tiles back)."""

    def __init__(self, fns, plan, ctx_handles, store=None, frame=None, read_tile=None, root=None):
        """frame = (width, height, stripe_rows); read_tile(k) -> packed float accumulation of local context k (only the
        host-memory rehearsal and the one-rank case use it). root: the rank the exchanges deliver to; default = the rank that
        owns the fewest rows of the striped split (tiling.lightest_rank), rank 0 when the frame is not given."""
        self.fns, self.plan, self.store = fns, plan, store
        if root is None:
            root = tiling.lightest_rank(frame[1], frame[2], plan["world"]) if frame and plan["world"] > 1 else 0
        self.root = int(root)
        self.h = list(ctx_handles)
        self.n = len(self.h)
        self.world = plan["world"]
        self.frame, self.read_tile = frame, read_tile
        self.comms = []
        self.transport = "none"
        self._host_full = None
        if self.world == 1:
            return
        if plan["mode"] == "one_process":
            ctxs = (P * self.n)(*self.h)
            comms = (P * self.n)()
            st = fns.comm_create_all(ctxs, self.n, comms)
            if st != 0:
                raise _abi.OkrtError(f"okrt_comm_create_all: status {st}: {fns.comm_last_error(None).decode()}")
            self.comms = [P(comms[k]) for k in range(self.n)]
            self._ctx_arr, self._comm_arr = ctxs, comms
            self.transport = fns.comm_transport(self.comms[0]).decode()
        elif plan["rehearsal"]:
            self.transport = "host memory (REHEARSAL: ranks share a GPU, RCCL cannot join them)"
        else:
            # One RCCL communicator over all ranks. A rank that cannot join (no unique id, ncclCommInitRank refused) must not leave the
            # others waiting: every rank reports, and if any failed ALL fall back to the host-memory exchange, which says so in
            # `transport` (the bench line then carries a flagged gather instead of no line at all).
            problem = ""
            ident = (C.c_uint8 * 128)()
            if plan["rank"] == 0:
                st = fns.comm_unique_id(ident)
                if st != 0:
                    problem = f"okrt_comm_unique_id: status {st}: {fns.comm_last_error(None).decode()}"
                store.set("rccl_id", b"" if problem else bytes(ident))
            raw = store.get("rccl_id")
            comm = P()
            if len(raw) != 128:
                problem = problem or "rank 0 has no RCCL unique id"
            else:
                ident = (C.c_uint8 * 128).from_buffer_copy(raw)
                st = fns.comm_create(self.h[0], ident, plan["rank"], self.world, C.byref(comm))
                if st != 0:
                    problem = f"okrt_comm_create: status {st}: {fns.comm_last_error(None).decode()}"
                    comm = P()
            problems = [b.decode() for b in store.allgather(problem.encode())]
            if any(problems):
                if comm:
                    fns.comm_destroy(comm)
                first = next(k for k, m in enumerate(problems) if m)
                self.transport = f"host memory (FALLBACK, not RCCL: rank {first}: {problems[first]})"
            else:
                self.comms = [comm]
                self.transport = fns.comm_transport(comm).decode()

    @property
    def is_root(self):
        return self.root in self.plan["local_ranks"]

    @property
    def root_local(self):
        """Index of the root's context among the ones this process drives (is_root must hold)."""
        return list(self.plan["local_ranks"]).index(self.root)

    def exchange_ms(self):
        """Root only: device time of the last gather / reduce on the root's stream (okrt_comm_exchange_ms); None elsewhere and
        for the host-memory rehearsal."""
        if not self.is_root or not self.comms:
            return None
        ms = C.c_float(0)
        st = self.fns.comm_exchange_ms(self.comms[self.root_local], C.byref(ms))
        return float(ms.value) if st == 0 else None

    @property
    def host_only(self):
        return self.world > 1 and not self.comms

    def _ck(self, st, what):
        if st != 0:
            msgs = "; ".join(self.fns.last_error(h).decode() for h in self.h)
            raise _abi.OkrtError(f"{what}: status {st}: {msgs}")

    def _through_host(self, samples_split):
        width, height, stripe_rows = self.frame
        mine = np.ascontiguousarray(self.read_tile(0), dtype=np.float32)
        parts = self.store.allgather(mine.tobytes())
        if not self.is_root:
            return
        tiles = [np.frombuffer(b, dtype=np.float32).reshape(-1, width, 4) for b in parts]
        self._host_full = np.sum(np.stack(tiles), axis=0, dtype=np.float32) if samples_split else tiling.deinterleave(tiles, height, stripe_rows)

    def gather(self):
        """Tile split: every rank's packed float4 tile -> the full frame on the root (read_full())."""
        if self.world == 1:
            return
        if self.host_only:
            self._through_host(False)
        elif self.plan["mode"] == "one_process":
            self._ck(self.fns.mgpu_gather_accum_all(self._ctx_arr, self._comm_arr, self.n, self.root, None), "okrt_mgpu_gather_accum_all")
        else:
            self._ck(self.fns.mgpu_gather_accum(self.h[0], self.comms[0], self.root, None), "okrt_mgpu_gather_accum")

    def reduce(self):
        """Samples split: sum of the whole-frame accumulations on the root (read_full())."""
        if self.world == 1:
            return
        if self.host_only:
            self._through_host(True)
        elif self.plan["mode"] == "one_process":
            self._ck(self.fns.mgpu_reduce_accum_all(self._ctx_arr, self._comm_arr, self.n, self.root, None), "okrt_mgpu_reduce_accum_all")
        else:
            self._ck(self.fns.mgpu_reduce_accum(self.h[0], self.comms[0], self.root, None), "okrt_mgpu_reduce_accum")

    def read_full(self):
        """On the root's process: the combined H x W x 4 frame of the last gather / reduce (synchronises). Elsewhere: None."""
        if not self.is_root:
            return None
        width, height, _ = self.frame
        if self.world == 1:
            return np.ascontiguousarray(self.read_tile(0), dtype=np.float32).reshape(height, width, 4)
        if self.host_only:
            return self._host_full
        out = np.empty((height, width, 4), np.float32)
        k = self.root_local
        self._ck(self.fns.mgpu_read_gathered(self.h[k], self.comms[k], out.ctypes.data_as(P)), "okrt_mgpu_read_gathered")
        return out

    def close(self):
        for c in self.comms:
            self.fns.comm_destroy(c)
        self.comms = []


def allreduce_json(store, value, combine):
    """Small host-side reductions of a multi-process job (ray counts, elapsed time): every rank contributes a JSON value,
    every rank gets combine(list of values in rank order)."""
    if store is None:
        return combine([value])
    return combine([json.loads(b.decode()) for b in store.allgather(json.dumps(value).encode())])

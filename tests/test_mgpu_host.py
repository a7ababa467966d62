"""The launcher logic of the N-GPU path on CPU (gpu-raytracer_amd/mgpu.py): which process drives which rank, the file
rendezvous that replaces torch.distributed for a one-process-per-GPU job, and the N-rank tile / samples split through
the same TileGroup calls bench.py makes — with the oracle standing in for the device and the host-memory transport of
the rehearsal mode (RCCL needs GPUs; the RCCL leg itself is tests/test_gpu_mgpu.py and the driver's multi-GPU run)."""
import importlib
import multiprocessing as mp
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import REPO

mgpu = importlib.import_module("gpu-raytracer_amd.mgpu")


def test_launch_plan_plain_and_under_a_launcher():
    lp = mgpu.launch_plan
    assert lp(1, {}, 1) == dict(mode="single", world=1, rank=0, local_ranks=[0], devices=[0], rehearsal=False)
    # plain `python bench.py --gpus 8` on an 8-GPU node: one process, rank r on device r
    p = lp(8, {}, 8)
    assert p["mode"] == "one_process" and p["world"] == 8 and p["local_ranks"] == list(range(8)) and p["devices"] == list(range(8)) and not p["rehearsal"]
    # the same command on a 1-GPU box is a rehearsal: every rank on device 0
    p = lp(2, {}, 1)
    assert p["mode"] == "one_process" and p["devices"] == [0, 0] and p["rehearsal"]
    p = lp(4, {}, 2)
    assert p["devices"] == [0, 1, 0, 1] and p["rehearsal"]
    # under torch.distributed.run the launcher's world size wins over --gpus, one rank per process
    env = dict(WORLD_SIZE="4", RANK="2", LOCAL_RANK="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29500")
    p = lp(1, env, 8)
    assert p == dict(mode="process_per_gpu", world=4, rank=2, local_ranks=[2], devices=[2], rehearsal=False)
    p = lp(4, env, 1)
    assert p["devices"] == [0] and p["rehearsal"]
    # a launcher that gives every rank its own GPU through a visibility mask: one visible device per process, NOT a rehearsal
    p = lp(4, dict(env, HIP_VISIBLE_DEVICES="2"), 1)
    assert p["devices"] == [0] and not p["rehearsal"]
    assert lp(4, dict(env, HIP_VISIBLE_DEVICES="0,1,2,3"), 1)["rehearsal"]  # a mask of several devices but one visible: still shared
    assert lp(4, dict(WORLD_SIZE="1", RANK="0"), 8)["mode"] == "one_process"  # WORLD_SIZE=1 is no launcher at all
    with pytest.raises(ValueError):
        lp(1, dict(WORLD_SIZE="2", RANK="5"), 2)
    # before the GPU count is known the plan already names mode and world (nothing touches a GPU to decide them)
    assert lp(8, {}, None)["mode"] == "one_process" and lp(8, {}, None)["world"] == 8


def _store_rank(rank, world, key, root, out):
    sys.path.insert(0, REPO)
    m = importlib.import_module("gpu-raytracer_amd.mgpu")
    st = m.FileStore(rank, world, key=key, root=root, timeout=60)
    got = st.allgather(bytes([rank]) * (rank + 1))
    if rank == 0:
        st.set("blob", b"x" * 128)
    blob = st.get("blob")
    st.barrier()
    total = m.allreduce_json(st, {"n": rank + 1}, lambda ds: sum(d["n"] for d in ds))
    # who shares a GPU is settled between the ranks: the same one-device mask everywhere (a 1-GPU box) is a rehearsal even
    # though each rank alone would read it as "my own GPU"; a different mask per rank is not; nor are distinct devices
    env = dict(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    box_mask = m.settle_shared_devices(m.launch_plan(1, dict(env, HIP_VISIBLE_DEVICES="3"), 1), st, dict(env, HIP_VISIBLE_DEVICES="3"))
    own_mask = m.settle_shared_devices(m.launch_plan(1, dict(env, HIP_VISIBLE_DEVICES=str(rank)), 1), st, dict(env, HIP_VISIBLE_DEVICES=str(rank)))
    no_mask = m.settle_shared_devices(m.launch_plan(1, env, 8), st, env)
    crowded = m.settle_shared_devices(m.launch_plan(1, env, 2), st, env)  # 3 ranks, 2 GPUs: ranks 0 and 2 on device 0
    # the barrier that brackets a timed region: nobody leaves before the last rank arrived, and all leave together
    stamps = []
    for k in range(5):
        time.sleep(0.003 * ((rank + k) % world))  # staggered arrivals
        arrive = time.monotonic()  # (CLOCK_MONOTONIC: one clock for every process of the machine)
        st.tight_barrier(gather_first=(k % 2 == 0))  # with and without the sleeping barrier in front
        stamps.append((arrive, time.monotonic()))
    st.close()
    out.put((rank, [len(g) for g in got], [g[:1] for g in got], len(blob), total,
             (box_mask["rehearsal"], own_mask["rehearsal"], no_mask["rehearsal"], crowded["rehearsal"]), stamps))


def test_file_store_allgather_barrier_close(tmp_path):
    world = 3
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_store_rank, args=(r, world, "t_%d" % os.getpid(), str(tmp_path), out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, lens, firsts, blob, total, shared, stamps in res:
        assert lens == [1, 2, 3] and firsts == [b"\x00", b"\x01", b"\x02"] and blob == 128 and total == 6
        assert shared == (True, False, False, True)
    for k in range(5):
        arrive = [r[6][k][0] for r in res]
        leave = [r[6][k][1] for r in res]
        assert min(leave) >= max(arrive)  # nobody left before the last rank arrived
        # all leave together: microseconds on an idle machine; the bound allows for a loaded 8-core CI container (the file-polling
        # barrier it replaces in front of bench.py's timed region let early arrivals sleep up to 2 ms between looks)
        assert max(leave) - min(leave) < 0.02, (k, leave)
    skews = sorted(max(r[6][k][1] for r in res) - min(r[6][k][1] for r in res) for k in range(5))
    print("tight barrier exit skew (s), 5 rounds:", ["%.6f" % s for s in skews])
    assert not [d for d in os.listdir(tmp_path) if d.startswith("okrt_rdzv_")]  # rank 0 removed the directory


def _tile_rank(rank, world, key, root, out_path):
    """What bench.py does per rank in process_per_gpu mode, with the oracle as the device and the host-memory transport."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import conftest
    m = importlib.import_module("gpu-raytracer_amd.mgpu")
    env = dict(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    plan = m.launch_plan(1, env, 1)  # one "GPU" for `world` ranks: the rehearsal plan
    assert plan["rehearsal"] and plan["mode"] == "process_per_gpu"
    store = m.FileStore(rank, world, key=key, root=root, timeout=120)
    oracle = conftest.Oracle()
    oracle.lib.okrt_oracle_set_threads.argtypes = [conftest.C.c_void_p, conftest.C.c_int]
    desc = conftest.scenes.small_mesh_scene(seed=2, n_objects=4, detail=5, with_lights=True)
    W, H, SR, FRAMES = 48, 27, 4, 3
    rt, sc, res = oracle.instantiate(desc, W, H)
    oracle.lib.okrt_oracle_set_threads(rt.ctx.h, 1)
    rt.ctx.set_tile(SR, rank, world)
    group = m.TileGroup(oracle.fns, plan, [rt.ctx.h], store, frame=(W, H, SR), read_tile=lambda k: rt.ctx.read_accum())
    assert group.host_only and "REHEARSAL" in group.transport
    # the root is the rank with the fewest rows, the last one among equals: 27 rows in 4-row stripes -> 15 / 12 rows (2 ranks), 11 / 8 / 8 (3)
    assert group.root == {2: 1, 3: 2}[world] and group.is_root == (rank == group.root)
    rt.renderFrames(FRAMES)
    group.gather()
    full = group.read_full()
    # samples split: whole frame, rank r renders frame indices 2r+1, 2r+2, one sum at the end
    rt.ctx.set_tile(H, 0, 1)
    rt.resetAccumulation()
    rt.render_data.numAccumulationFrames = rank * 2
    rt.renderFrames(2)
    group.reduce()
    summed = group.read_full()
    if group.is_root:
        ref_rt, _, _ = oracle.instantiate(desc, W, H)
        ref_rt.renderFrames(FRAMES)
        ref = ref_rt.ctx.read_accum()
        ref_rt.resetAccumulation()
        ref_rt.renderFrames(2 * world)
        ref_sum = ref_rt.ctx.read_accum()
        ok_tiles = np.array_equal(full, ref, equal_nan=True) and bool((full[..., 3] == FRAMES).all())
        ok_samples = np.allclose(summed, ref_sum, rtol=1e-5, atol=1e-6, equal_nan=True) and bool((summed[..., 3] == 2 * world).all())
        np.save(out_path, np.array([float(ok_tiles), float(ok_samples)]))
    else:
        assert full is None and summed is None
    group.close()
    store.close()


def test_the_gather_root_is_the_lightest_rank():
    """bench.py, mgpu.TileGroup and examples/headless_host_mgpu.cpp deliver the tiles to the rank that owns the fewest rows (it is
    done with its own frames first): rank N - 1 for 1080 rows in 8-row stripes over 8 ranks (128 rows against 136)."""
    t = importlib.import_module("gpu-raytracer_amd.tiling")
    assert t.lightest_rank(1080, 8, 8) == 7 and [len(t.owned_rows(1080, 8, r, 8)) for r in (0, 6, 7)] == [136, 136, 128]
    assert t.lightest_rank(1080, 8, 4) == 3 and t.lightest_rank(1080, 8, 2) == 1
    assert t.lightest_rank(2160, 8, 8) == 7  # 270 stripes: ranks 0-5 own 34, ranks 6 and 7 own 33: the last of the equals
    assert t.lightest_rank(64, 8, 8) == 7 and t.lightest_rank(64, 8, 1) == 0
    assert t.lightest_rank(20, 8, 8) == 7  # ranks that own nothing at all
    for world in (2, 3, 5, 8):
        rows = [len(t.owned_rows(1080, 8, r, world)) for r in range(world)]
        assert rows[t.lightest_rank(1080, 8, world)] == min(rows)


@pytest.mark.parametrize("world", [2, 3])
def test_tile_and_samples_split_through_the_file_store(tmp_path, world):
    """world processes, no torch: the union of the ranks' tiles is BIT-identical to the full frame; the samples split
    sums to the same accumulation (float order aside)."""
    ctx = mp.get_context("spawn")
    out = str(tmp_path / "ok.npy")
    procs = [ctx.Process(target=_tile_rank, args=(r, world, "tile_%d" % os.getpid(), str(tmp_path), out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    ok = np.load(out)
    assert ok[0] == 1.0 and ok[1] == 1.0, ok


def test_bench_without_a_gpu_fails_loudly_not_silently():
    """No GPU in this container: bench.py (any --gpus) must stop at the device count with a clear message — the product
    has no CPU path — and must not get as far as importing torch or building a scene."""
    for gpus in ("1", "2"):
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", gpus, "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env, timeout=120)
        if r.returncode == 0:
            pytest.skip("a GPU is visible here: the loud-failure path cannot be exercised")
        assert "no GPU visible" in r.stderr and "torch" not in r.stderr, r.stderr[-400:]


def test_a_rank_that_cannot_join_rccl_sends_every_rank_to_the_flagged_host_exchange(tmp_path):
    """process_per_gpu, real devices: if ANY rank fails to create the communicator (here rank 1's okrt_comm_create), all ranks must take
    the host-memory exchange together — the communicator rank 0 did get is destroyed again — and `transport` must say it is not RCCL."""
    import ctypes as C
    import threading
    m = importlib.import_module("gpu-raytracer_amd.mgpu")

    class Fns:
        def __init__(self, rank):
            self.rank, self.destroyed = rank, []

        def comm_unique_id(self, ident):
            ident[0] = 42
            return 0

        def comm_create(self, ctx, ident, rank, world, comm_ref):
            assert ident[0] == 42 and world == 2
            if rank == 1:
                return 7
            comm_ref._obj.value = 0xBEEF
            return 0

        def comm_last_error(self, _):
            return b"ncclCommInitRank: unhandled system error"

        def comm_destroy(self, comm):
            self.destroyed.append(comm.value)
            return 0

        def comm_transport(self, comm):
            return b"rccl"

    out = {}

    def run(rank):
        plan = dict(world=2, rank=rank, local_ranks=[rank], devices=[rank], mode="process_per_gpu", rehearsal=False)
        store = m.FileStore(rank, 2, key="fallback", root=str(tmp_path), timeout=60)
        fns = Fns(rank)
        tile = np.full((2, 4, 4), float(rank + 1), np.float32)  # 2 rows of a 4-wide, 4-row frame in 2-row stripes
        g = m.TileGroup(fns, plan, [C.c_void_p(1)], store, frame=(4, 4, 2), read_tile=lambda k: tile)
        g.gather()
        out[rank] = (g.transport, g.host_only, fns.destroyed, g.read_full() if g.is_root else None, g.exchange_ms())
        store.barrier()
        store.close()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    assert set(out) == {0, 1}
    for rank in (0, 1):
        transport, host_only, destroyed, full, ms = out[rank]
        assert host_only and "FALLBACK" in transport and "rank 1" in transport and "okrt_comm_create: status 7" in transport and ms is None
        assert destroyed == ([0xBEEF] if rank == 0 else [])
    root = [r for r in (0, 1) if out[r][3] is not None]
    assert len(root) == 1
    full = out[root[0]][3]
    assert full.shape == (4, 4, 4) and (full[:2] == 1.0).all() and (full[2:] == 2.0).all()

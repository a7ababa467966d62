"""The C-ABI multi-GPU exchange (include/okrt.h "multi-GPU", csrc/okrt_mgpu.hip) as far as ONE GPU can exercise it:
the communicator / gather entry points with a one-rank communicator, and the root-side de-interleave with the packed
tiles of 2, 4 and 8 ranks rendered by one device. The N-rank send / receive leg needs N GPUs (the driver's SCALE run);
the same partition arithmetic runs on CPU with gloo ranks in tests/test_tiles_distributed.py."""
import ctypes as C
import importlib

import numpy as np
import pytest

from conftest import abi, scenes

pytestmark = pytest.mark.gpu
tiling = importlib.import_module("gpu-raytracer_amd.tiling")
P = C.c_void_p


@pytest.fixture(scope="module")
def hip():
    lib = C.CDLL("libamdhip64.so")  # the runtime instance libokrt.so already loaded
    lib.hipMalloc.argtypes = [C.POINTER(P), C.c_size_t]
    lib.hipFree.argtypes = [P]
    lib.hipMemcpy.argtypes = [P, P, C.c_size_t, C.c_int]
    return lib


def test_one_rank_communicator_gathers_the_frame():
    desc = scenes.small_mesh_scene(seed=4, n_objects=4, detail=6)
    rt, sc, res = scenes.instantiate(desc, 96, 54)
    rt.renderFrames(3)
    want = rt.ctx.read_accum()
    fns = rt.ctx.fns
    ctxs = (P * 1)(rt.ctx.h)
    comms = (P * 1)()
    assert fns.comm_create_all(ctxs, 1, comms) == 0, fns.comm_last_error(None)
    assert fns.mgpu_gather_accum(rt.ctx.h, comms[0], 0, None) == 0, fns.last_error(rt.ctx.h)
    got = np.zeros_like(want)
    assert fns.mgpu_read_gathered(rt.ctx.h, comms[0], got.ctypes.data_as(P)) == 0
    assert np.array_equal(got, want, equal_nan=True)
    # a tile that is not this rank's share of the communicator is refused, not gathered wrongly
    rt.ctx.set_tile(8, 1, 2)
    assert fns.mgpu_gather_accum(rt.ctx.h, comms[0], 0, None) == -4 and b"okrt_set_tile" in fns.last_error(rt.ctx.h)
    assert fns.mgpu_gather_accum(rt.ctx.h, comms[0], 3, None) == -1  # root out of range
    assert fns.comm_destroy(comms[0]) == 0
    rt.close()


@pytest.mark.parametrize("world,stripe", [(2, 8), (4, 8), (8, 8), (8, 270), (3, 5)])
def test_deinterleave_rebuilds_the_frame_from_rank_tiles(hip, world, stripe):
    """The root-side kernel on the tiles the ranks of a `world`-way split own (config 5's 8-row stripes and 270-row bands
    among them; 2160 rows like the 4K frame, narrow so that eight renders stay cheap)."""
    desc = scenes.small_mesh_scene(seed=7, n_objects=3, detail=5, with_textures=False)
    W, H = 64, 2160 if stripe in (8, 270) else 37
    rt, sc, res = scenes.instantiate(desc, W, H)
    rt.render()
    full = rt.ctx.read_accum().copy()
    rd = rt.render_data
    pad = tiling.max_owned_rows(H, stripe, world)
    tiles = np.zeros((world, pad, W, 4), np.float32)
    for r in range(world):
        rt.ctx.set_tile(stripe, r, world)
        rt.ctx.render(rd)  # same frame index
        t = rt.ctx.read_accum()
        tiles[r, : t.shape[0]] = t
    rt.ctx.set_tile(H, 0, 1)
    d_tiles, d_full = P(), P()
    assert hip.hipMalloc(C.byref(d_tiles), tiles.nbytes) == 0 and hip.hipMalloc(C.byref(d_full), full.nbytes) == 0
    assert hip.hipMemcpy(d_tiles, tiles.ctypes.data_as(P), tiles.nbytes, 1) == 0
    fns = rt.ctx.fns
    assert fns.mgpu_deinterleave(rt.ctx.h, d_tiles, world, stripe, pad, d_full) == 0, fns.last_error(rt.ctx.h)
    rt.ctx.synchronize()
    got = np.zeros_like(full)
    assert hip.hipMemcpy(got.ctypes.data_as(P), d_full, got.nbytes, 2) == 0
    hip.hipFree(d_tiles); hip.hipFree(d_full)
    assert np.array_equal(got, full, equal_nan=True)
    assert np.array_equal(tiling.deinterleave(list(tiles), H, stripe), full, equal_nan=True)  # the Python twin bench.py uses
    assert fns.mgpu_deinterleave(rt.ctx.h, d_tiles, world, stripe, 1, d_full) == -1 or pad == 1  # stride too small is refused
    rt.close()


def _split_contexts(desc, W, H, world, stripe, whole_frame=False):
    rts = []
    for r in range(world):
        rt, sc, res = scenes.instantiate(desc, W, H)  # every rank on device 0: the rehearsal layout
        if not whole_frame:
            rt.ctx.set_tile(stripe, r, world)
        rts.append((rt, sc, res))
    return rts


@pytest.mark.parametrize("world,stripe,H", [(2, 8, 54), (3, 5, 37), (8, 8, 135), (3, 8, 16)])
def test_n_contexts_in_one_process_gather_through_the_copy_transport(world, stripe, H):
    """What `python bench.py --gpus N` does on a box with fewer GPUs: N contexts in one process, rank r owning stripes
    r, r+N, ... (okrt_set_tile), joined by okrt_comm_create_all — contexts that share a device get the copy transport —
    and okrt_mgpu_gather_accum_all: the gathered frame is BIT-identical to one context's full frame. (3, 8, 16): the third
    rank owns ZERO rows and must still be a valid member of the collective."""
    desc = scenes.small_mesh_scene(seed=11, n_objects=4, detail=6, with_lights=True)
    W, FRAMES = 96, 5
    ref, _, _ = scenes.instantiate(desc, W, H)
    ref.renderFrames(FRAMES)
    want = ref.ctx.read_accum().copy()
    ref.close()
    rts = _split_contexts(desc, W, H, world, stripe)
    fns = rts[0][0].ctx.fns
    ctxs = (P * world)(*[rt.ctx.h for rt, _, _ in rts])
    comms = (P * world)()
    assert fns.comm_create_all(ctxs, world, comms) == 0, fns.comm_last_error(None)
    assert fns.comm_transport(comms[0]) == b"copy"
    assert sum(rt.ctx.tile_rows() for rt, _, _ in rts) == H
    for rnd in range(2):  # twice: the peers' streams wait for the root's copy before they render on
        for rt, _, _ in rts:
            rt.renderFrames(FRAMES if rnd == 0 else 2)
        assert fns.mgpu_gather_accum_all(ctxs, comms, world, 0, None) == 0, [fns.last_error(c) for c in ctxs]
    got = np.zeros((H, W, 4), np.float32)
    assert fns.mgpu_read_gathered(ctxs[0], comms[0], got.ctypes.data_as(P)) == 0
    ref, _, _ = scenes.instantiate(desc, W, H)
    ref.renderFrames(FRAMES + 2)
    want2 = ref.ctx.read_accum()
    assert np.array_equal(got, want2, equal_nan=True) and (got[..., 3] == FRAMES + 2).all()
    assert not np.array_equal(want, want2)
    ref.close()
    # a subset of the ranks is refused (its sends would have no receive), and so is a rank passed twice
    if world > 2:
        assert fns.mgpu_gather_accum_all(ctxs, comms, world - 1, 0, None) == -1
    dup_c = (P * world)(*([ctxs[0]] * world))
    dup_m = (P * world)(*([comms[0]] * world))
    assert fns.mgpu_gather_accum_all(dup_c, dup_m, world, 0, None) == -1
    for k in range(world):
        assert fns.comm_destroy(comms[k]) == 0
    for rt, _, _ in rts:
        rt.close()


def test_samples_split_reduce_through_the_copy_transport():
    """bench.py --split samples: every rank renders the WHOLE frame for its own block of frame indices; one reduce sums
    the accumulations on rank 0 (float order aside it is the accumulation of all the frames)."""
    desc = scenes.small_mesh_scene(seed=12, n_objects=4, detail=6)
    W, H, world, per = 80, 45, 3, 4
    rts = _split_contexts(desc, W, H, world, 8, whole_frame=True)
    fns = rts[0][0].ctx.fns
    ctxs = (P * world)(*[rt.ctx.h for rt, _, _ in rts])
    comms = (P * world)()
    assert fns.comm_create_all(ctxs, world, comms) == 0, fns.comm_last_error(None)
    for r, (rt, _, _) in enumerate(rts):
        rt.render_data.numAccumulationFrames = r * per
        rt.renderFrames(per)
    assert fns.mgpu_reduce_accum_all(ctxs, comms, world, 0, None) == 0, [fns.last_error(c) for c in ctxs]
    got = np.zeros((H, W, 4), np.float32)
    assert fns.mgpu_read_gathered(ctxs[0], comms[0], got.ctypes.data_as(P)) == 0
    ref, _, _ = scenes.instantiate(desc, W, H)
    ref.renderFrames(world * per)
    want = ref.ctx.read_accum()
    assert (got[..., 3] == world * per).all() and np.allclose(got, want, rtol=1e-5, atol=1e-6, equal_nan=True)
    # a tile-split context is refused by the reduce (it needs whole frames)
    rts[1][0].ctx.set_tile(8, 1, world)
    assert fns.mgpu_reduce_accum_all(ctxs, comms, world, 0, None) == -4
    ref.close()
    for k in range(world):
        fns.comm_destroy(comms[k])
    for rt, _, _ in rts:
        rt.close()


def test_device_count_and_frame_size():
    import ctypes
    fns = abi.load_device_library()
    n = ctypes.c_int(0)
    assert fns.device_count(ctypes.byref(n)) == 0 and n.value >= 1
    ctx = importlib.import_module("gpu-raytracer_amd.host").Context()
    assert ctx.frame_size() == (0, 0)
    ctx.resize(40, 30)
    ctx.set_tile(8, 1, 2)
    assert ctx.frame_size() == (40, 30) and ctx.tile_rows() == 14
    ctx.close()


def test_rccl_is_bound_at_run_time_with_every_symbol_the_exchange_uses():
    """What a 1-GPU box can say about the RCCL leg: librccl.so.1 loads on first use, every entry point the gather / reduce
    call is found (okrt_comm_unique_id fails with the missing symbol's name otherwise), and ncclGetUniqueId hands out a
    128-byte id. The sends / receives themselves need N GPUs."""
    import ctypes
    fns = abi.load_device_library()
    a, b = (ctypes.c_uint8 * 128)(), (ctypes.c_uint8 * 128)()
    assert fns.comm_unique_id(a) == 0, fns.comm_last_error(None)
    assert fns.comm_unique_id(b) == 0
    assert any(a) and bytes(a) != bytes(b)
    assert fns.comm_unique_id(None) == -1

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

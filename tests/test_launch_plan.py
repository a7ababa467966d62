"""How okrt_render_frames / the deferred okrt_render cut a progressive accumulation into launch sets (csrc/okrt_launch_plan.h,
answered without a GPU by the test hook okrt_debug_launch_plan in libokrt.so): big tiles keep one set per 64 frames, a rank's
share of an N-way split is issued as two sub-sets that ride the pipeline slots, small sets ride them as they are."""
import ctypes as C

import numpy as np

from conftest import abi


def plan(rows, width, frames, batch_limit=64, pipeline=4, split_rays=0):
    lib = C.CDLL(abi.device_library_path())
    fn = lib.okrt_debug_launch_plan
    fn.restype = C.c_uint32
    fn.argtypes = [C.c_uint32] * 5 + [C.c_uint64, C.c_void_p, C.c_uint32]
    out = np.zeros((256, 2), np.uint32)
    n = fn(rows, width, frames, batch_limit, pipeline, split_rays, out.ctypes.data_as(C.c_void_p), 256)
    assert n <= 256
    return [(int(f), bool(r)) for f, r in out[:n]]


def test_full_frames_keep_one_set_per_batch():
    assert plan(1080, 1920, 20) == [(20, False)]                     # the driver's --steps 20 on one GPU: 41 M rays, one set
    assert plan(1080, 1920, 128) == [(64, False), (64, False)]
    assert plan(2160, 3840, 64) == [(64, False)]
    assert plan(1080, 1920, 3) == [(3, True)]                        # up to OKRT_OPT_PIPELINE frames: rides the slots as it is
    assert plan(1080, 1920, 72, batch_limit=16) == [(16, False)] * 4 + [(8, False)]  # an allocator-limited host: sets of 16


def test_a_ranks_share_of_a_split_is_issued_as_two_sub_sets():
    assert plan(136, 1920, 20) == [(10, True), (10, True)]           # 1/8 of 1080p x 20 frames = 5.2 M rays
    assert plan(136, 1920, 21) == [(11, True), (10, True)]
    assert plan(270, 1920, 20) == [(10, True), (10, True)]           # 1/4: 10.4 M
    assert plan(540, 1920, 20) == [(20, False)]                      # 1/2: 20.7 M rays, above the threshold
    assert plan(136, 1920, 128) == [(64, False), (64, False)]        # 64-frame sets of a 1/8 tile: 16.7 M, in one piece
    assert plan(136, 1920, 128, split_rays=1 << 30) == [(32, True)] * 4
    assert plan(136, 1920, 5) == [(3, True), (2, True)]
    assert plan(136, 1920, 4) == [(4, True)]
    # every frame is issued exactly once, whatever the shape
    rng = np.random.default_rng(3)
    for _ in range(300):
        rows, frames, limit = int(rng.integers(1, 2200)), int(rng.integers(1, 300)), int(rng.integers(1, 65))
        sets = plan(rows, 1920, frames, batch_limit=limit, pipeline=int(rng.integers(0, 33)))
        assert sum(f for f, _ in sets) == frames and all(1 <= f <= 64 for f, _ in sets)


def test_pipelining_off_means_no_sub_sets():
    assert plan(136, 1920, 20, pipeline=0) == [(20, False)]
    assert plan(136, 1920, 3, pipeline=0) == [(3, False)]

"""GPU tests at BASELINE.json's full sizes through size-independent properties (the oracle would take minutes
there), plus the boundary's error behaviour and edge cases."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest

from conftest import abi, host, scenes

pytestmark = pytest.mark.gpu
tiling = importlib.import_module("gpu-raytracer_amd.tiling")


@pytest.fixture(scope="module")
def sponza():
    desc = scenes.sponza_proxy()
    assert desc.num_triangles == scenes.SPONZA_TRIANGLES and len(desc.meshes) == scenes.SPONZA_MESHES and len(desc.textures) == scenes.SPONZA_TEXTURES
    rt, sc, res = scenes.instantiate(desc, 1920, 1080)
    yield rt
    rt.close()


def test_sponza_1080p_fast_equals_reference_order(sponza):
    rt = sponza
    accs = []
    for trav in (abi.TRAVERSAL_FAST, abi.TRAVERSAL_REFERENCE):
        rt.ctx.set_option(abi.OPT_TRAVERSAL, trav)
        rt.resetAccumulation()
        rt.render()
        accs.append(rt.ctx.read_accum())
    rt.ctx.set_option(abi.OPT_TRAVERSAL, abi.TRAVERSAL_FAST)
    same = (accs[0] == accs[1]).all(-1) | (np.isnan(accs[0]).any(-1) & np.isnan(accs[1]).any(-1))
    assert same.mean() >= 0.9995, same.mean()  # ties only: 390 meshes share many coincident edges
    l2 = np.sqrt(((accs[0][..., :3] - accs[1][..., :3]) ** 2).sum(-1))
    assert (l2[np.isfinite(l2)] <= 1e-3).mean() >= 0.9999


def test_sponza_1080p_accumulation_is_a_running_sum(sponza):
    """acc after frames 1..3 == sum of the three frames rendered alone (same frame indices); alpha counts frames."""
    rt = sponza
    rt.resetAccumulation()
    for _ in range(3):
        rt.render()
    acc3 = rt.ctx.read_accum()
    assert (acc3[..., 3] == 3).all()
    total = np.zeros_like(acc3)
    rd = rt.render_data
    for n in (1, 2, 3):
        rt.ctx.reset_accumulation()
        rd.numAccumulationFrames = n
        rt.ctx.render(rd)
        frame = rt.ctx.read_accum()
        total = (total + frame).astype(np.float32)  # (a+b)+c in float32, the order the kernel adds in
    fin = np.isfinite(acc3).all(-1)
    assert np.array_equal(acc3[fin], total[fin])
    res = rt.ctx.read_result()
    assert res.shape == (1080, 1920, 4) and (res[..., 3] > 0).all()


def test_sponza_1080p_tile_split_is_bit_identical(sponza):
    """Rows rendered by a context that owns every other 8-row stripe equal the same rows of the full frame."""
    rt = sponza
    rt.resetAccumulation()
    rt.render()
    full = rt.ctx.read_accum()
    rd = rt.render_data
    tiles = []
    for r in range(2):
        rt.ctx.set_tile(8, r, 2)
        rt.ctx.render(rd)  # same frame index
        tiles.append(rt.ctx.read_accum())
    rt.ctx.set_tile(1080, 0, 1)
    merged = tiling.deinterleave(tiles, 1080, 8)
    assert np.array_equal(merged, full, equal_nan=True)


def test_sponza_1080p_rays_per_frame_and_timing(sponza):
    rt = sponza
    rt.resetAccumulation()
    rt.ctx.reset_counters()
    rt.ctx.set_option(abi.OPT_TIMING, 4)
    for _ in range(4):
        rt.render()
    trace, shade, other, frames = rt.ctx.stage_ms()
    rt.ctx.set_option(abi.OPT_TIMING, 0)
    c = rt.ctx.counters()
    assert frames == 4 and c["pixels"] == 4 * 1920 * 1080
    assert 1920 * 1080 * 4 < c["segments"] <= 2 * 1920 * 1080 * 4  # NUM_BOUNCES = 1: at most 2 segments per sample
    assert 0 < trace < 100 and 0 < shade < 100 and abs(other) < 5


def test_dragon_871k_fast_equals_reference_order():
    desc = scenes.dragon_proxy(871414)
    assert desc.num_triangles == 871416  # + the 2-triangle ground quad
    rt, sc, res = scenes.instantiate(desc, 1280, 720)
    accs = []
    for trav in (abi.TRAVERSAL_FAST, abi.TRAVERSAL_REFERENCE):
        rt.ctx.set_option(abi.OPT_TRAVERSAL, trav)
        rt.resetAccumulation()
        rt.render()
        accs.append(rt.ctx.read_accum())
    rt.close()
    same = (accs[0] == accs[1]).all(-1) | (np.isnan(accs[0]).any(-1) & np.isnan(accs[1]).any(-1))
    assert same.mean() >= 0.999, same.mean()


def test_glass_1080p_fast_equals_reference_order_at_8_bounces():
    """Config 3 at its full size: glass + revolver-proxy + IBL, 1920x1080, with NUM_BOUNCES = 8 so that the refraction
    chains (Q23: at HEAD's single bounce glass is only entered, never left) run through every queue segment."""
    desc = scenes.glass_revolver_proxy()
    rt, sc, res = scenes.instantiate(desc, 1920, 1080)
    rt.ctx.set_option(abi.OPT_MAX_BOUNCES, 8)
    accs = []
    for trav in (abi.TRAVERSAL_FAST, abi.TRAVERSAL_REFERENCE):
        rt.ctx.set_option(abi.OPT_TRAVERSAL, trav)
        rt.resetAccumulation()
        rt.render()
        accs.append(rt.ctx.read_accum())
    rt.close()
    same = (accs[0] == accs[1]).all(-1) | (np.isnan(accs[0]).any(-1) & np.isnan(accs[1]).any(-1))
    l2 = np.sqrt(((accs[0][..., :3] - accs[1][..., :3]) ** 2).sum(-1))
    assert same.mean() >= 0.998 and (l2[np.isfinite(l2)] <= 1e-3).mean() >= 0.999, same.mean()


def test_sponza_4k_eight_way_tile_split_is_bit_identical_and_builders_agree():
    """Config 5 on one GPU: the 3840x2160 frame rendered as the eight 8-row-stripe tiles the eight ranks would own
    equals the frame rendered whole, bit for bit (batched frames included); and the device-built structure gives the
    same picture as the host-built one up to equal-distance ties."""
    desc = scenes.sponza_proxy()
    W, H = 3840, 2160
    rt, sc, res = scenes.instantiate(desc, W, H)
    rt.renderFrames(3)
    full = rt.ctx.read_accum().copy()
    tiles = []
    for r in range(8):
        rt.ctx.set_tile(8, r, 8)
        rt.resetAccumulation()
        rt.renderFrames(3)
        tiles.append(rt.ctx.read_accum().copy())
    rt.ctx.set_tile(H, 0, 1)
    merged = tiling.deinterleave(tiles, H, 8)
    assert np.array_equal(merged, full, equal_nan=True)
    rt.ctx.set_option(abi.OPT_BUILDER, abi.BUILDER_DEVICE)
    rt.resetAccumulation()
    rt.renderFrames(3)
    dev = rt.ctx.read_accum()
    rt.close()
    same = (dev == full).all(-1) | (np.isnan(dev).any(-1) & np.isnan(full).any(-1))
    assert same.mean() >= 0.999, same.mean()


# ---- boundary behaviour ---------------------------------------------------------------------------------------
def test_error_codes_and_messages():
    ctx = host.Context()
    rd = abi.RenderData()
    rd.textureDims[0], rd.textureDims[1] = 8, 8
    assert ctx.fns.render(ctx.h, C.byref(rd)) == -4  # OKRT_ERR_STATE: render before resize
    assert b"resize" in ctx.fns.last_error(ctx.h)
    ctx.resize(8, 8)
    tri = np.zeros((2, 9), np.float32)
    assert ctx.set_table_raw(abi.TABLE_TRI_POS, tri, 40, 2) == -1  # wrong element stride
    assert b"stride" in ctx.fns.last_error(ctx.h)
    assert ctx.fns.set_table(ctx.h, 3, None, 4, 0) == -1  # slot 3 is an image, not a table
    rd.numSpheres = 2
    assert ctx.fns.render(ctx.h, C.byref(rd)) == -5  # OKRT_ERR_RANGE: count exceeds the uploaded table
    rd.numSpheres = 0
    rd.textureDims[0] = 9
    assert ctx.fns.render(ctx.h, C.byref(rd)) == -1  # textureDims must match the framebuffer
    rd.textureDims[0] = 8
    # a BVH whose child index points backwards (a cycle) must be refused, not traversed
    ctx.set_table(abi.TABLE_TRI_POS, tri)
    ctx.set_table(abi.TABLE_TRI_INFO, np.zeros((2, 33), np.float32))
    bad = np.zeros((3, 9), np.uint32); bad[0, 8] = 0  # node 0 -> children 0,1
    bad[1:, 8] = 0xFFFFFFFF
    ctx.set_table(abi.TABLE_BVH, bad)
    assert ctx.fns.render(ctx.h, C.byref(rd)) == -5
    assert b"BVH node 0" in ctx.fns.last_error(ctx.h)
    good = bad.copy(); good[0, 8] = 1; good[1, 6:8] = [0, 1]; good[2, 6:8] = [1, 5]  # leaf range beyond t0
    ctx.set_table(abi.TABLE_BVH, good)
    assert ctx.fns.render(ctx.h, C.byref(rd)) == -5
    good[2, 6:8] = [1, 2]
    ctx.set_table(abi.TABLE_BVH, good)
    assert ctx.fns.render(ctx.h, C.byref(rd)) == 0  # valid again (no octree -> nothing to hit)
    assert ctx.fns.set_option(ctx.h, 99, 0) == -1
    assert ctx.fns.set_tile(ctx.h, 0, 0, 1) == -1 and ctx.fns.set_tile(ctx.h, 4, 2, 2) == -1
    ctx.close()
    h = C.c_void_p()
    assert ctx.fns.create(9999, C.byref(h)) == -2 and not h.value  # no such device


def test_empty_scene_renders_the_environment_only():
    ctx = host.Context()
    ctx.resize(33, 17)  # not a multiple of any tile size: every pixel is still covered (Q15)
    faces = np.zeros((6, 4, 4, 4), np.uint8); faces[..., 0] = 255
    ctx.set_envmap(faces)
    rd = abi.RenderData()
    rd.accumulationEnabled = 1; rd.numAccumulationFrames = 1
    rd.textureDims[0], rd.textureDims[1] = 33, 17
    for i in range(16):
        rd.cameraInverseProjectionMatrix[i] = rd.cameraInverseViewMatrix[i] = float(i % 5 == 0)
    rd.cameraNearZ = 0.1
    for trav in (abi.TRAVERSAL_FAST, abi.TRAVERSAL_REFERENCE):
        ctx.set_option(abi.OPT_TRAVERSAL, trav)
        ctx.reset_accumulation()
        ctx.render(rd)
        acc = ctx.read_accum()
        assert acc.shape == (17, 33, 4) and np.allclose(acc[..., 0], 1.0) and (acc[..., 1:3] == 0).all() and (acc[..., 3] == 1).all()
    ctx.set_envmap(None)  # unbound cube: black
    ctx.reset_accumulation(); ctx.render(rd)
    assert (ctx.read_accum()[..., :3] == 0).all()
    ctx.close()


def test_table_update_with_a_different_count_reallocates():
    """GPUStorage::update grows/shrinks with the count (GPUStorage.h:36-37)."""
    desc = scenes.small_mesh_scene(seed=3, n_objects=3, detail=5, with_textures=False)
    rt, sc, res = scenes.instantiate(desc, 64, 36)
    rt.render()
    a = rt.ctx.read_accum()
    rt.loadMeshAndBvhData(8, 2)   # "Rebuild BVH tree" (Application.cpp:408-411): different node count, same image
    rt.createOctTree(sc, 1, 1)
    rt.resetAccumulation()
    rt.render()
    b = rt.ctx.read_accum()
    same = (a == b).all(-1)
    assert same.mean() > 0.995
    rt.close()


def test_render_frames_is_bit_identical_to_single_frames(sponza):
    """okrt_render_frames(F) must leave exactly what F okrt_render calls leave (accumulation AND resolve)."""
    rt = sponza
    rt.resetAccumulation()
    for _ in range(5):
        rt.render()
    seq_acc, seq_res = rt.ctx.read_accum(), rt.ctx.read_result()
    rt.resetAccumulation()
    rt.renderFrames(3)
    rt.renderFrames(2)
    assert rt.getNumAccumulationFrames() == 5
    bat_acc, bat_res = rt.ctx.read_accum(), rt.ctx.read_result()
    assert np.array_equal(seq_acc, bat_acc, equal_nan=True) and np.array_equal(seq_res, bat_res)
    # also through a tile, with a batch longer than the internal cap (16) and with max_bounces = 0
    rt.ctx.set_option(abi.OPT_MAX_BOUNCES, 0)
    rt.ctx.set_tile(8, 1, 4)
    rt.resetAccumulation()
    for _ in range(19):
        rt.render()
    a = rt.ctx.read_accum()
    rt.resetAccumulation()
    rt.renderFrames(19)
    b = rt.ctx.read_accum()
    rt.ctx.set_tile(1080, 0, 1)
    rt.ctx.set_option(abi.OPT_MAX_BOUNCES, 1)
    assert np.array_equal(a, b, equal_nan=True) and (a[..., 3] == 19).all()


def test_checkpoint_resume_is_bit_identical():
    """Save the float accumulation + the host's frame counter after 3 frames, restore into a fresh context, render 2
    more: identical to 5 uninterrupted frames."""
    desc = scenes.small_mesh_scene(seed=6, n_objects=4, detail=6)
    rt, sc, res = scenes.instantiate(desc, 96, 54)
    rt.renderFrames(5)
    full = rt.ctx.read_accum()
    rt.resetAccumulation()
    rt.renderFrames(3)
    saved, n = rt.ctx.read_accum(), rt.getNumAccumulationFrames()
    rt.close()
    rt2, sc2, res2 = scenes.instantiate(desc, 96, 54)
    rt2.ctx.write_accum(saved)
    rt2.render_data.numAccumulationFrames = n
    rt2.renderFrames(2)
    assert np.array_equal(rt2.ctx.read_accum(), full, equal_nan=True)
    rt2.close()


def test_deferred_frames_are_bit_identical_and_unobservable():
    """OKRT_OPT_DEFER: okrt_render calls of the accumulation loop wait and are traced together. Same framebuffers as
    immediate execution, also when the loop is interrupted by what a host does between frames: re-uploading unchanged
    light tables (RayTracer::updateBuffers does, every frame), a real table change, a read-back, a DOF edit."""
    desc = scenes.small_mesh_scene(seed=9, n_objects=6, detail=10, with_lights=True)

    def run(defer):
        rt, sc, res = scenes.instantiate(desc, 256, 144)
        rt.ctx.set_option(abi.OPT_DEFER, defer)
        snaps = []
        for i in range(23):
            rt.render()  # host layer: updateBuffers (light/sphere tables re-uploaded, identical) + okrt_render
            if i == 4:
                snaps.append(rt.ctx.read_accum().copy())  # a read-back in the middle of a deferred run
            if i == 9:
                rt.setDOF(0.5, 12.0)  # RenderData changes: the waiting frames must be traced with the old one
            if i == 15:
                snaps.append(rt.ctx.counters()["pixels"])
        snaps.append(rt.ctx.read_accum().copy())
        snaps.append(rt.ctx.read_result().copy())
        rt.close()
        return snaps

    a, b = run(16), run(0)
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True) if isinstance(x, np.ndarray) and x.dtype.kind == "f" else np.array_equal(x, y)
    assert a[2][..., 3].max() == 23  # alpha counts frames: nothing was lost at the end of the loop


def test_launch_sets_shrink_to_the_queue_memory_budget():
    """A launch set whose queues do not fit (here: an artificial 24 MiB budget; in the field: hipMalloc saying no) is
    split in halves until it fits; the framebuffers do not change."""
    desc = scenes.small_mesh_scene(seed=4, n_objects=5, detail=10, with_lights=True)
    rt, sc, res = scenes.instantiate(desc, 320, 180)
    rt.renderFrames(48)
    want = rt.ctx.read_accum().copy()
    rt.resetAccumulation()
    rt.ctx.set_option(abi.OPT_QUEUE_MEMORY_MB, 24)  # 320x180 needs 7.8 MiB per frame in flight: sets of 3 at most
    rt.renderFrames(48)
    got = rt.ctx.read_accum()
    assert np.array_equal(got, want, equal_nan=True)
    rt.ctx.set_option(abi.OPT_QUEUE_MEMORY_MB, 1)  # not even one frame fits: a single frame is still attempted
    rt.resetAccumulation()
    rt.renderFrames(5)
    assert np.array_equal(rt.ctx.read_accum()[..., 3], np.full((180, 320), 5.0, np.float32))
    rt.close()


def test_the_default_queue_budget_is_a_quarter_of_the_free_memory(monkeypatch):
    """okrt.h, OKRT_OPT_QUEUE_MEMORY_MB: a host that never sets the bound gets a quarter of the device memory free at okrt_resize
    (OKRT_QUEUE_MEMORY_ASSUME_FREE_MB stands in for hipMemGetInfo here: 96 MiB "free" -> 24 MiB, the case of the test above);
    an explicit 0 lifts it again. Framebuffers do not change either way."""
    desc = scenes.small_mesh_scene(seed=4, n_objects=5, detail=10, with_lights=True)
    rt, sc, res = scenes.instantiate(desc, 320, 180)
    rt.renderFrames(48)
    want = rt.ctx.read_accum().copy()
    rt.close()
    monkeypatch.setenv("OKRT_QUEUE_MEMORY_ASSUME_FREE_MB", "96")
    rt, sc, res = scenes.instantiate(desc, 320, 180)
    rt.ctx.set_option(abi.OPT_TIMING, 64)
    rt.renderFrames(48)
    assert np.array_equal(rt.ctx.read_accum(), want, equal_nan=True)
    _, launches, frames = rt.ctx.interval_ms()
    assert frames // launches <= 3, (launches, frames)  # 7.8 MiB per frame in flight under a 24 MiB bound
    rt.ctx.set_option(abi.OPT_QUEUE_MEMORY_MB, 0)
    rt.ctx.set_option(abi.OPT_TIMING, 64)
    rt.resetAccumulation()
    rt.renderFrames(48)
    assert np.array_equal(rt.ctx.read_accum(), want, equal_nan=True)
    _, launches, frames = rt.ctx.interval_ms()
    assert launches <= 2 and frames == 48, (launches, frames)  # one set (issued as two sub-sets on the pipeline slots: a small tile)
    rt.close()


def test_launch_sets_shrink_when_the_allocator_says_no():
    """The same split on a REAL allocator failure (okrt.h, OKRT_OPT_QUEUE_MEMORY_MB: "by this bound or by the allocator"):
    OKRT_OPT_DEBUG_FAIL_ALLOC makes the n-th queue allocation report hipErrorOutOfMemory. The failed launch set is retried
    at half the size and must come back OKRT_OK — a stale out-of-memory left behind as HIP's "last error" used to
    turn the successful retry into OKRT_ERR_DEVICE — with the accumulation bit-identical and every frame counted."""
    desc = scenes.small_mesh_scene(seed=4, n_objects=5, detail=10, with_lights=True)
    rt, sc, res = scenes.instantiate(desc, 320, 180)
    rt.renderFrames(8)
    want = rt.ctx.read_accum().copy()
    rt.close()
    for nth in (1, 2, 5, 7):  # first queue array, a later one, the hit records, the radiance slots
        rt, sc, res = scenes.instantiate(desc, 320, 180)
        rt.ctx.set_option(abi.OPT_TIMING, 8)
        rt.ctx.set_option(abi.OPT_DEBUG_FAIL_ALLOC, nth)
        rt.renderFrames(8)  # raises on any status but OKRT_OK
        got = rt.ctx.read_accum()
        assert np.array_equal(got, want, equal_nan=True), nth
        assert rt.getNumAccumulationFrames() == 8 and (got[..., 3] == 8).all()
        ms, launches, frames = rt.ctx.interval_ms()
        assert launches == 2 and frames == 8, (nth, launches, frames)  # two sets of four; the refused attempt left no entry behind
        _, _, _, staged = rt.ctx.stage_ms()
        assert staged == 8
        rt.ctx.set_option(abi.OPT_DEBUG_FAIL_ALLOC, 0)
        rt.close()


def test_interval_ms_only_sums_launch_sets_of_the_newest_shape():
    """okrt_get_interval_ms: 100 frames = a set of 64 and a set of 36; the 36-frame shape is the newest, so exactly that launch is
    reported (a mixed sum would describe a fictitious 50-frame launch). With sub-sets (OKRT_PIPELINE_SPLIT_PARTS=2) a frame this
    small is issued as 32, 32, 18 and 18 and the two 18-frame launches are."""
    desc = scenes.small_mesh_scene(seed=4, n_objects=4, detail=6)
    for parts, newest, after in (("1", (1, 36), (2, 128)), ("2", (2, 36), (4, 128))):
        os.environ["OKRT_PIPELINE_SPLIT_PARTS"] = parts
        try:
            rt, sc, res = scenes.instantiate(desc, 128, 72)
        finally:
            del os.environ["OKRT_PIPELINE_SPLIT_PARTS"]
        rt.ctx.set_option(abi.OPT_TIMING, 16)
        rt.renderFrames(100)
        ms, launches, frames = rt.ctx.interval_ms()
        assert (launches, frames) == newest and len(ms) == 4 and all(m >= 0 for m in ms), (parts, launches, frames)
        rt.renderFrames(64)
        ms, launches, frames = rt.ctx.interval_ms()
        assert (launches, frames) == after, (parts, launches, frames)
        rt.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_call_sequences_deferred_equals_immediate(seed):
    """API fuzz: the same random sequence of host calls — frames, batched frames, read-backs, option changes, DOF edits,
    accumulation resets / toggles, tile changes, a resize, an entity move — drives one context with deferral (32) and
    one without; every observation and the final framebuffers must be identical."""
    desc = scenes.small_mesh_scene(seed=30 + seed, n_objects=5, detail=8, with_lights=True, octree=(0, 50))

    def run(defer):
        rng = np.random.default_rng(seed)
        rt, sc, res = scenes.instantiate(desc, 192, 108)
        rt.ctx.set_option(abi.OPT_DEFER, defer)
        seen = []
        for _ in range(60):
            op = int(rng.integers(0, 12))
            if op <= 3:
                for _ in range(int(rng.integers(1, 9))):
                    rt.render()
            elif op == 4:
                rt.renderFrames(int(rng.integers(1, 40)))
            elif op == 5:
                seen.append(rt.ctx.read_accum().copy())
            elif op == 6:
                rt.setDOF(float(rng.uniform(0, 1)), float(rng.uniform(5, 20)))
            elif op == 7:
                rt.resetAccumulation()
            elif op == 8:
                rt.ctx.set_option(abi.OPT_MAX_BOUNCES, int(rng.integers(0, 4)))
            elif op == 9:
                n = int(rng.integers(1, 4))
                rt.ctx.set_tile(8, int(rng.integers(0, n)), n)
                rt.resetAccumulation()
            elif op == 10:
                sc.set_transform(0, position=tuple(rng.uniform(-4, 4, 3)), rotation=(0.0, float(rng.uniform(0, 90)), 0.0))
                rt.createOctTree(sc, 0, 50)
                rt.resetAccumulation()
            else:
                seen.append(rt.ctx.counters()["pixels"])
        seen.append(rt.ctx.read_accum().copy())
        seen.append(rt.ctx.read_result().copy())
        rt.close()
        return seen

    a, b = run(32), run(0)
    assert len(a) == len(b)
    for x, y in zip(a, b):
        if isinstance(x, np.ndarray):
            assert x.shape == y.shape and np.array_equal(x, y, equal_nan=(x.dtype.kind == "f"))
        else:
            assert x == y



def test_peek_shows_a_resolved_picture_without_executing_deferred_frames():
    """okrt_peek_result_rgba8 (the presenting host's read-back, Application.cpp:106-128): the picture it returns is the
    exact resolve of some earlier frame count — never a torn or half-accumulated one —, asking for it neither executes
    the deferred frames nor loses them, and the frame count it reports only moves forward."""
    desc = scenes.small_mesh_scene(seed=6, n_objects=5, detail=8, with_lights=True)
    W, H = 160, 90
    # reference pictures: the resolve after exactly k frames, k = 0 .. 40
    ref, _, _ = scenes.instantiate(desc, W, H)
    ref.ctx.set_option(abi.OPT_DEFER, 0)
    pictures = {0: np.zeros((H, W, 4), np.uint8)}
    for k in range(1, 41):
        ref.render()
        pictures[k] = ref.ctx.read_result().copy()
    ref.close()
    rt, sc, res = scenes.instantiate(desc, W, H)
    rt.ctx.set_option(abi.OPT_DEFER, 8)
    seen = []
    for k in range(1, 41):
        rt.render()  # frames wait in groups of 8
        img, n = rt.ctx.peek_result()
        assert n <= k and n % 8 == 0 and (not seen or n >= seen[-1]), (k, n, seen[-3:])
        assert np.array_equal(img, pictures[n]), (k, n)
        seen.append(n)
    assert seen[-1] >= 24  # the loop really saw the accumulation progress (not just the initial zeros)
    assert rt.getNumAccumulationFrames() == 40
    acc = rt.ctx.read_accum()  # this DOES flush
    assert (acc[..., 3] == 40).all() and np.array_equal(rt.ctx.read_result(), pictures[40])
    img, n = rt.ctx.peek_result()
    assert n == 40 and np.array_equal(img, pictures[40])
    # a resize drops the stale pictures instead of handing out one of the wrong size
    rt.onResize(64, 36)
    img, n = rt.ctx.peek_result()
    assert img.shape == (36, 64, 4) and n == 0 and not img.any()
    rt.close()


def test_peek_between_pipelined_launch_sets_is_never_torn():
    """Launch sets that ride the pipeline slots only order their accumulation kernels; the copy a peek reads is made after a set's
    accumulation on that set's stream, and the NEXT set's accumulation — another stream — rewrites the resolved picture. It must wait
    for the copy as well (round 3 did not order the two): every peeked picture is the exact resolve of some earlier frame count."""
    desc = scenes.small_mesh_scene(seed=6, n_objects=5, detail=8, with_lights=True)
    W, H = 640, 360
    sizes = [1, 2, 4, 3, 1, 1, 4, 2, 5, 10, 1, 3] * 4  # (5 and 10: issued as sub-sets, which ride the slots too)
    ref, _, _ = scenes.instantiate(desc, W, H)
    ref.ctx.set_option(abi.OPT_DEFER, 0)
    ref.ctx.set_option(abi.OPT_PIPELINE, 0)
    pictures = {0: np.zeros((H, W, 4), np.uint8)}
    for k in range(1, sum(sizes) + 1):
        ref.render()
        pictures[k] = ref.ctx.read_result().copy()
    ref.close()
    rt, sc, res = scenes.instantiate(desc, W, H)
    rt.ctx.set_option(abi.OPT_DEFER, 0)
    rt.ctx.set_option(abi.OPT_PIPELINE, 4)
    rt.ctx.peek_result()  # from now on every launch set leaves a copy behind
    done, last, moved = 0, 0, 0
    for m in sizes:
        rt.renderFrames(m) if m > 1 else rt.render()
        done += m
        img, n = rt.ctx.peek_result()
        assert last <= n <= done, (done, n, last)
        assert np.array_equal(img, pictures[n]), (done, n)
        moved += n > last
        last = n
    assert moved >= 8  # the peeks really saw sets finish while others were in flight
    assert np.array_equal(rt.ctx.read_result(), pictures[done])
    rt.close()


def test_sub_sets_of_a_small_tile_are_bit_identical_to_one_set(monkeypatch):
    """okrt_render_frames on a small tile (one rank's stripes) can issue a launch set as two sub-sets on the pipeline slots
    (csrc/okrt_launch_plan.h; OKRT_PIPELINE_SPLIT_PARTS=2 — the default until round 5 ordered the camera tiles longest first);
    framebuffers are the ones a single set — and single frames — produce, bit for bit."""
    monkeypatch.setenv("OKRT_PIPELINE_SPLIT_PARTS", "2")
    desc = scenes.small_mesh_scene(seed=9, n_objects=6, detail=9, with_lights=True)

    def run(pipeline, chunks):
        rt, sc, res = scenes.instantiate(desc, 384, 216)
        rt.ctx.set_tile(8, 3, 8)  # rank 3 of an 8-way split
        rt.ctx.set_option(abi.OPT_PIPELINE, pipeline)
        rt.ctx.set_option(abi.OPT_DEFER, 0)
        for m in chunks:
            rt.renderFrames(m)
        acc, img = rt.ctx.read_accum().copy(), rt.ctx.read_result().copy()
        rt.close()
        return acc, img

    a = run(4, [20, 21, 64, 5])
    b = run(0, [20, 21, 64, 5])
    c = run(0, [1] * 110)
    assert (a[0][..., 3] == 110).all()
    assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1])
    assert np.array_equal(a[0], c[0], equal_nan=True) and np.array_equal(a[1], c[1])


def test_two_launch_sets_in_flight_are_bit_identical_to_one_at_a_time():
    """OKRT_OPT_PIPELINE: consecutive launch sets alternate between two queue slots and two streams and only their
    accumulation kernels are ordered. Whatever the host does in between — single frames, batches of every size (40 frames
    is beyond the pipelined set size), read-backs, a camera change, a reset, a table upload, a peek — the framebuffers
    are the ones one launch set at a time produces, bit for bit, and nothing is lost at the end."""
    desc = scenes.small_mesh_scene(seed=8, n_objects=5, detail=9, with_lights=True)

    def run(pipeline):
        rt, sc, res = scenes.instantiate(desc, 224, 126)
        rt.ctx.set_option(abi.OPT_PIPELINE, pipeline)
        rt.ctx.set_option(abi.OPT_DEFER, 0)
        snaps = []
        for _ in range(5):
            rt.render()
        rt.renderFrames(7)
        rt.renderFrames(40)
        for _ in range(3):
            rt.render()
        snaps.append(rt.ctx.read_accum().copy())
        rt.setDOF(0.4, 10.0)
        for _ in range(3):
            rt.renderFrames(16)
        snaps.append(rt.ctx.read_result().copy())
        rt.resetAccumulation()
        rt.renderFrames(3)
        rt.render()
        rt.ctx.set_option(abi.OPT_DEFER, 4)
        for _ in range(11):
            rt.render()
        snaps.append(rt.ctx.counters()["pixels"])
        rt.renderFrames(2)
        snaps.append(rt.ctx.read_accum().copy())
        snaps.append(rt.ctx.read_result().copy())
        assert rt.getNumAccumulationFrames() == 3 + 1 + 11 + 2
        rt.close()
        return snaps

    a, b = run(32), run(0)  # every set of up to 32 frames pipelined / none
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True) if isinstance(x, np.ndarray) and x.dtype.kind == "f" else np.array_equal(x, y)
    assert (a[3][..., 3] == 17).all()
    c = run(4)  # the default
    assert np.array_equal(c[3], b[3], equal_nan=True) and np.array_equal(c[4], b[4])


def test_a_deferring_render_reports_the_allocator_refusal_itself():
    """OKRT_OPT_DEFER: the queues of the deferred launch set are allocated as soon as a second frame waits, so running out
    of memory is reported inside the host's render loop — here the injected refusal halves the set size, the loop goes on,
    and the picture is the one immediate execution gives — instead of at whichever unrelated call flushes later."""
    desc = scenes.small_mesh_scene(seed=4, n_objects=5, detail=10, with_lights=True)
    rt, sc, res = scenes.instantiate(desc, 320, 180)
    rt.ctx.set_option(abi.OPT_DEFER, 0)
    for _ in range(20):
        rt.render()
    want = rt.ctx.read_accum().copy()
    rt.close()
    rt, sc, res = scenes.instantiate(desc, 320, 180)
    rt.ctx.set_option(abi.OPT_DEFER, 16)
    rt.render()                                        # waits alone: nothing reserved yet
    rt.ctx.set_option(abi.OPT_DEBUG_FAIL_ALLOC, 3)     # (setting an option executes the waiting frame)
    rt.render()
    rt.render()                                        # second waiting frame: reserves 16 frames -> refused -> 8 frames
    rt.ctx.set_option(abi.OPT_DEBUG_FAIL_ALLOC, 0)
    for _ in range(17):
        rt.render()
    got = rt.ctx.read_accum()
    assert np.array_equal(got, want, equal_nan=True) and (got[..., 3] == 20).all()
    rt.ctx.set_option(abi.OPT_TIMING, 8)
    for _ in range(16):
        rt.render()
    rt.ctx.synchronize()
    _, launches, frames = rt.ctx.interval_ms()
    assert (launches, frames) == (2, 16)  # the remembered limit: sets of 8 (two sub-sets of 4 each under OKRT_PIPELINE_SPLIT_PARTS=2)
    rt.close()


def test_a_small_deferral_reserves_the_slots_its_sets_will_ride():
    """OKRT_OPT_DEFER <= OKRT_OPT_PIPELINE: the deferred sets take turns on the pipeline slots with frame slots, so that is what the
    reservation allocates (round 3 reserved slot 0 in the other mode: the refusal then surfaced at whichever call flushed)."""
    desc = scenes.small_mesh_scene(seed=4, n_objects=5, detail=10, with_lights=True)
    rt, sc, res = scenes.instantiate(desc, 320, 180)
    rt.ctx.set_option(abi.OPT_DEFER, 0)
    for _ in range(14):
        rt.render()
    want = rt.ctx.read_accum().copy()
    rt.close()
    rt, sc, res = scenes.instantiate(desc, 320, 180)
    rt.ctx.set_option(abi.OPT_DEFER, 4)
    rt.render()
    rt.ctx.set_option(abi.OPT_DEBUG_FAIL_ALLOC, 2)     # (executes the waiting frame, then arms the hook)
    rt.render()
    rt.render()                                        # second waiting frame: reserves sets of 4 on every slot -> refused -> sets of 2
    rt.ctx.set_option(abi.OPT_DEBUG_FAIL_ALLOC, 0)     # (a refusal that had been left for a later flush would fire here or below)
    for _ in range(11):
        rt.render()
    got = rt.ctx.read_accum()
    assert np.array_equal(got, want, equal_nan=True) and (got[..., 3] == 14).all()
    rt.ctx.set_option(abi.OPT_TIMING, 8)
    for _ in range(8):
        rt.render()
    rt.ctx.synchronize()
    _, launches, frames = rt.ctx.interval_ms()
    assert (launches, frames) == (4, 8)  # the remembered limit: sets of 2
    rt.close()


def test_a_granted_reservation_leaves_larger_launch_sets_alone():
    """The reservation a deferring okrt_render makes (16 frames) is not a limit: when the allocator granted it, a later
    okrt_render_frames(64) is still ONE launch set of 64 frames (only a refusal lowers the remembered set size)."""
    desc = scenes.small_mesh_scene(seed=4, n_objects=4, detail=6)
    rt, sc, res = scenes.instantiate(desc, 128, 72)
    rt.ctx.set_option(abi.OPT_DEFER, 16)
    for _ in range(40):
        rt.render()
    rt.ctx.synchronize()
    rt.ctx.set_option(abi.OPT_TIMING, 8)
    rt.renderFrames(64)
    _, launches, frames = rt.ctx.interval_ms()
    assert (launches, frames) == (1, 64)  # one set of 64 frames (two sub-sets of 32 on a frame this small under OKRT_PIPELINE_SPLIT_PARTS=2)
    rt.ctx.set_option(abi.OPT_PIPELINE, 0)
    rt.ctx.set_option(abi.OPT_TIMING, 8)
    rt.renderFrames(64)
    _, launches, frames = rt.ctx.interval_ms()
    assert (launches, frames) == (1, 64)
    rt.close()


def test_longest_first_tile_order_leaves_the_picture_alone(tmp_path):
    """Round 5: the fused camera kernel issues its 8x8 tiles in descending order of the steps they took in the previous launch set
    (okrt_fast_shade.hip: orderTilesKernel). The order is read from the environment once per process (OKRT_FAST_LPT), so the two
    pictures come from separate processes: accumulation and resolve after a mix of launch shapes must be the same bytes, through a
    striped tile as well (a rank's share: the shards are short there). The same kernel cuts every full group of eight of a tile's
    packets as 4x2 pixels x 8 frames and every full group of four of the rest as 4x4 x 4 (frame-stacked packets, OKRT_FAST_STACK;
    off with the order): the launch shapes below give sets of 1, 7, 13 and 11 frames — plain 8x8 packets; 4 + 3 plain; 8 + 4 + 1;
    8 + 3 plain — and the third process keeps the order but not the stacking. Which rays share a packet must not change a byte."""
    import subprocess
    import sys
    code = (
        "import sys, os, importlib, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "scenes = importlib.import_module('gpu-raytracer_amd.scenes'); abi = importlib.import_module('gpu-raytracer_amd._abi')\n"
        "out = []\n"
        "for tile in (None, (8, 1, 3)):\n"
        "    rt, _, _ = scenes.instantiate(scenes.small_mesh_scene(seed=11, n_objects=6, detail=9), 320, 200)\n"
        "    if tile: rt.ctx.set_tile(*tile)\n"
        "    for m in (1, 7, 1, 13, 11):\n"
        "        rt.renderFrames(m)\n"
        "    for _ in range(9): rt.render()\n"
        "    out += [rt.ctx.read_accum(), rt.ctx.read_result().astype(np.float32)]\n"
        "    rt.close()\n"
        "np.savez(sys.argv[1], *out)\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = []
    for tag, extra in (("default", {}), ("frame_major", {"OKRT_FAST_LPT": "0"}), ("unstacked", {"OKRT_FAST_STACK": "0"})):
        path = str(tmp_path / f"{tag}.npz")
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        files.append(np.load(path))
    a = files[0]
    assert len(a.files) == 4
    for b in files[1:]:
        for k in a.files:
            assert np.array_equal(a[k], b[k], equal_nan=True), k
    assert (a["arr_0"][..., 3] == 42).all()


@pytest.mark.parametrize("case", range(8))
def test_tile_split_and_batching_leave_random_scenes_bit_identical(case):
    """The two invariances every multi-GPU and every batched run rests on, over scenes nobody arranged: the rows of an N-way striped split
    equal the full frame's rows, and K frames in one okrt_render_frames equal K okrt_render calls — odd frame sizes (partly filled 8x8
    tiles, a last stripe that is cut short), lit and unlit scenes (with and without the last-segment shortcut), 0-4 bounces, both builders."""
    rng = np.random.default_rng(900 + case)
    kw = dict(seed=int(rng.integers(1, 10**6)), n_objects=int(rng.integers(1, 8)), detail=int(rng.integers(3, 11)), with_textures=bool(rng.integers(0, 2)),
              with_lights=bool(rng.integers(0, 2)), transforms=bool(rng.integers(0, 2)))
    desc = scenes.small_mesh_scene(**kw)
    if rng.integers(0, 2):  # no emitter: with no lights either the last-segment shortcut is taken
        for e in desc.entities:
            e.material = {k: v for k, v in e.material.items() if k not in ("emissionColour", "emissionPower")}
    W, H = int(rng.integers(20, 300)), int(rng.integers(17, 170))
    bounces, frames, n = int(rng.integers(0, 5)), int(rng.integers(2, 7)), int(rng.integers(2, 5))
    stripe = int(rng.choice([8, 16]))
    builder = abi.BUILDER_DEVICE if rng.integers(0, 2) else abi.BUILDER_HOST
    rt, _, _ = scenes.instantiate(desc, W, H)
    rt.ctx.set_option(abi.OPT_MAX_BOUNCES, bounces)
    rt.ctx.set_option(abi.OPT_BUILDER, builder)
    rt.renderFrames(frames)
    full = rt.ctx.read_accum()
    rt.resetAccumulation()
    for _ in range(frames):
        rt.render()
    assert np.array_equal(rt.ctx.read_accum(), full, equal_nan=True), (kw, W, H, bounces, frames)
    tiles = []
    for r in range(n):
        rt.ctx.set_tile(stripe, r, n)
        rt.resetAccumulation()
        rt.renderFrames(frames)
        t = rt.ctx.read_accum()
        rows = tiling.owned_rows(H, stripe, r, n)
        assert t.shape[0] >= len(rows)
        tiles.append(t[: len(rows)])
        assert np.array_equal(t[: len(rows)], full[rows], equal_nan=True), (kw, W, H, bounces, frames, n, stripe, r)
    assert np.array_equal(tiling.deinterleave(tiles, H, stripe), full, equal_nan=True)
    rt.close()

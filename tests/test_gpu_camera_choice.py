"""The camera segment is traced by the 8x8 packet kernel or ray by ray, chosen per scene by a probe of the first launch set after
a build (okrt_fast_pipeline.hip: probeCameraSegment; the reference's loop is RaytracerCS.hlsl:278-326 either way). These tests
pin: the choice on both kinds of scene, that it is the same for every tile of a split (it probes the full frame), that it is
re-made after a rebuild, and that the picture is the reference-order picture up to equal-distance ties and the oracle's within
SURVEY 8c whichever kernel ran."""
import importlib

import numpy as np
import pytest

from conftest import abi, compare_accum, scenes

pytestmark = pytest.mark.gpu
tiling = importlib.import_module("gpu-raytracer_amd.tiling")


def per_ray(rt):
    return bool(rt.ctx.build_info()["flags"] & abi.BUILD_FLAG_CAMERA_PER_RAY)


def fast_and_reference(rt, frames=2):
    out = []
    for trav in (abi.TRAVERSAL_FAST, abi.TRAVERSAL_REFERENCE):
        rt.ctx.set_option(abi.OPT_TRAVERSAL, trav)
        rt.resetAccumulation()
        rt.renderFrames(frames)
        out.append(rt.ctx.read_accum())
    rt.ctx.set_option(abi.OPT_TRAVERSAL, abi.TRAVERSAL_FAST)
    return out


@pytest.fixture(scope="module")
def dense():
    """Sub-pixel triangles: 400 000 of them behind 256 x 144 pixels (the dragon8m workload's regime at test size)."""
    return scenes.dragon_proxy(num_triangles=400000)


def test_sub_pixel_triangles_take_the_per_ray_kernel_and_keep_the_picture(dense, oracle):
    W, H = 256, 144
    rt, _, _ = scenes.instantiate(dense, W, H)
    assert not per_ray(rt)  # nothing rendered yet: undecided reads as the packet kernel
    fast, ref = fast_and_reference(rt)
    assert per_ray(rt)
    same = (fast == ref).all(-1) | (np.isnan(fast).any(-1) & np.isnan(ref).any(-1))
    assert same.mean() >= 0.999, same.mean()
    ort, _, _ = oracle.instantiate(dense, W, H)
    ort.render()
    ort.render()
    frac, rmse, nanmm, _ = compare_accum(fast, ort.ctx.read_accum(), frames=2)
    assert frac >= 0.995 and rmse <= 0.02 and nanmm == 0, (frac, rmse)
    # every tile of a split makes the full frame's choice and renders the full frame's rows bit for bit
    rt.resetAccumulation()
    rt.renderFrames(2)
    full = rt.ctx.read_accum()
    rt.close()
    for idx in (0, 2):
        rt, _, _ = scenes.instantiate(dense, W, H)
        rt.ctx.set_tile(8, idx, 3)
        rt.renderFrames(2)
        tile = rt.ctx.read_accum()
        assert per_ray(rt)
        rows = tiling.owned_rows(H, 8, idx, 3)
        assert np.array_equal(tile[: len(rows)], full[rows], equal_nan=True)
        rt.close()


def test_large_triangles_keep_the_packet_kernel_and_a_rebuild_chooses_again(dense):
    W, H = 256, 144
    coarse = scenes.small_mesh_scene(seed=9, n_objects=5, detail=8)
    rt, _, _ = scenes.instantiate(coarse, W, H)
    fast, ref = fast_and_reference(rt)
    assert not per_ray(rt)
    same = (fast == ref).all(-1) | (np.isnan(fast).any(-1) & np.isnan(ref).any(-1))
    assert same.mean() >= 0.999
    # the device builder makes another structure of the same scene: the probe runs again (and agrees)
    rt.ctx.set_option(abi.OPT_BUILDER, abi.BUILDER_DEVICE)
    rt.resetAccumulation()
    rt.renderFrames(2)
    assert not per_ray(rt)
    rt.close()
    rt, _, _ = scenes.instantiate(dense, W, H)
    rt.ctx.set_option(abi.OPT_BUILDER, abi.BUILDER_DEVICE)
    rt.renderFrames(1)
    assert per_ray(rt)
    rt.close()

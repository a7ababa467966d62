"""OKRT_OPT_LAST_SEGMENT_SHORTCUT: on the last path segment of a scene without emitters and lights a HIT adds
`emission * contribution * (1 - transparencyFactor)` = +-0 (RaytracerCS.hlsl:509-510; lights :448-487; the environment only on
a miss, :489-493), so the fast pipeline does not fetch / sample / draw for those hits, the bounce-ray kernel stores hit-or-miss
and only the misses are shaded. These tests pin that (a) the framebuffers are the full path's BIT FOR BIT in every launch shape,
(b) anything that could make the skipped term non-zero selects the full path by itself, and (c) the pictures still equal the
oracle's — which always shades in full — in both situations."""
import numpy as np
import pytest

from conftest import abi, compare_accum, scenes

pytestmark = pytest.mark.gpu


def dark_scene(**kw):
    """small_mesh_scene without its emitter: textures, a normal map, glass, two analytic spheres, open sky (misses)."""
    desc = scenes.small_mesh_scene(**kw)
    for e in desc.entities:
        e.material = {k: v for k, v in e.material.items() if k not in ("emissionColour", "emissionPower")}
    return desc


def flags(rt):
    """okrt_build_info.flags (meaningful once the host layer has uploaded every table: after the first render)."""
    return rt.ctx.build_info()["flags"]


def render_both(rt, frames, how):
    """Accumulation + resolve with the shortcut on and off; `how`: 'frames' = one okrt_render_frames, 'calls' = one okrt_render each."""
    out = []
    for on in (1, 0):
        rt.ctx.set_option(abi.OPT_LAST_SEGMENT_SHORTCUT, on)
        rt.resetAccumulation()
        if how == "frames":
            rt.renderFrames(frames)
        else:
            for _ in range(frames):
                rt.render()
        out.append((rt.ctx.read_accum(), rt.ctx.read_result()))
    rt.ctx.set_option(abi.OPT_LAST_SEGMENT_SHORTCUT, 1)
    return out


@pytest.mark.parametrize("bounces", [0, 1, 2, 5])
def test_shortcut_equals_full_shading_bit_for_bit(bounces):
    rt, _, _ = scenes.instantiate(dark_scene(), 320, 180)
    rt.ctx.set_option(abi.OPT_MAX_BOUNCES, bounces)
    for how, frames in (("frames", 7), ("calls", 3), ("frames", 1)):
        (a_on, r_on), (a_off, r_off) = render_both(rt, frames, how)
        assert flags(rt) & abi.BUILD_FLAG_DARK_MATERIALS
        assert np.array_equal(a_on, a_off, equal_nan=True) and np.array_equal(r_on, r_off), (bounces, how, frames)
        assert (a_on[..., 3] == frames).all()
    # launch shapes without frame slots: one set at a time, single frames written straight into the accumulation
    rt.ctx.set_option(abi.OPT_PIPELINE, 0)
    rt.ctx.set_option(abi.OPT_DEFER, 0)
    (a_on, r_on), (a_off, r_off) = render_both(rt, 2, "calls")
    assert np.array_equal(a_on, a_off, equal_nan=True) and np.array_equal(r_on, r_off)
    # accumulation off (the resolve of a single sample)
    rt.toggleAccumulation(False)
    (a_on, r_on), (a_off, r_off) = render_both(rt, 1, "calls")
    assert np.array_equal(r_on, r_off)
    rt.close()


def test_shortcut_with_the_unfused_camera_segment_and_a_tile(monkeypatch):
    """The other kernels that append the last segment's rays: shadeKernel<FIRST> (camera segment as three kernels) and
    shadeKernel<NEXT> at bounce lastBounce - 1 (above); here through a striped tile as well."""
    monkeypatch.setenv("OKRT_FAST_FUSED", "0")
    rt, _, _ = scenes.instantiate(dark_scene(seed=5), 256, 144)
    rt.ctx.set_tile(8, 1, 3)
    (a_on, r_on), (a_off, r_off) = render_both(rt, 5, "frames")
    assert np.array_equal(a_on, a_off, equal_nan=True) and np.array_equal(r_on, r_off)
    rt.close()


def test_any_hit_on_the_last_segment_leaves_the_same_picture():
    rt, _, _ = scenes.instantiate(dark_scene(), 320, 180)
    for bounces in (1, 3):
        rt.ctx.set_option(abi.OPT_MAX_BOUNCES, bounces)
        pics = []
        for any_hit in (0, 1):
            rt.ctx.set_option(abi.OPT_LAST_SEGMENT_ANY_HIT, any_hit)
            rt.resetAccumulation()
            rt.renderFrames(6)
            pics.append(rt.ctx.read_accum())
        assert np.array_equal(pics[0], pics[1], equal_nan=True)
    rt.ctx.set_option(abi.OPT_LAST_SEGMENT_ANY_HIT, 0)
    rt.close()


def test_counters_see_the_full_path():
    """tex_taps / mesh_hits count the reference's shading work: with OKRT_OPT_COUNTERS the last segment is shaded in full."""
    rt, _, _ = scenes.instantiate(dark_scene(), 160, 90)
    got = []
    for on in (1, 0):
        rt.ctx.set_option(abi.OPT_LAST_SEGMENT_SHORTCUT, on)
        rt.ctx.set_option(abi.OPT_COUNTERS, 1)
        rt.ctx.reset_counters()
        rt.resetAccumulation()
        rt.renderFrames(2)
        c = rt.ctx.counters()
        got.append(tuple(c[k] for k in ("segments", "mesh_hits", "sphere_hits", "tex_taps", "env_taps")))
        rt.ctx.set_option(abi.OPT_COUNTERS, 0)
    assert got[0] == got[1] and got[0][3] > 0
    rt.close()


def with_emitter(desc):
    desc.entities[2].material.update(emissionColour=(1.0, 0.8, 0.5), emissionPower=3.0)


def with_bright_albedo(desc):
    desc.entities[0].material.update(albedo=(1.5, 0.9, 0.9))  # throughput may grow: 0 * contribution is no longer provably finite


def with_nan_emission(desc):
    desc.entities[3].material.update(emissionColour=(float("nan"), 0.0, 0.0), emissionPower=0.0)


def with_infinite_power(desc):
    desc.spheres[0][2].update(emissionColour=(0.0, 0.0, 0.0), emissionPower=float("inf"))  # 0 * inf = NaN on every hit of that sphere


@pytest.mark.parametrize("spoil", [with_emitter, with_bright_albedo, with_nan_emission, with_infinite_power])
def test_a_material_that_could_contribute_selects_the_full_path(oracle, spoil):
    desc = dark_scene()
    spoil(desc)
    w, h = 192, 108
    rt, _, _ = scenes.instantiate(desc, w, h)
    rt.renderFrames(2)
    acc = rt.ctx.read_accum()
    assert not (flags(rt) & abi.BUILD_FLAG_DARK_MATERIALS)
    rt.ctx.set_option(abi.OPT_TRAVERSAL, abi.TRAVERSAL_REFERENCE)
    rt.resetAccumulation()
    rt.renderFrames(2)
    ref = rt.ctx.read_accum()
    rt.close()
    same = (acc == ref).all(-1) | (np.isnan(acc).any(-1) & np.isnan(ref).any(-1))
    assert same.mean() >= 0.999, same.mean()  # equal-distance ties only
    ort, _, _ = oracle.instantiate(desc, w, h)
    ort.render()
    ort.render()
    frac, rmse, nanmm, _ = compare_accum(acc, ort.ctx.read_accum(), frames=2)
    assert frac >= 0.995 and rmse <= 0.02 and nanmm <= 0.002, (frac, rmse, nanmm)  # SURVEY 8c, 1-spp rule per frame
    if spoil in (with_nan_emission, with_infinite_power):
        assert np.isnan(acc).any()  # the NaN the reference would show is shown


def test_a_light_in_the_frame_selects_the_full_path_and_the_dark_scene_equals_the_oracle(oracle):
    """Same dark materials, once without and once with lights (the per-frame half of the condition), both against the oracle."""
    w, h = 192, 108
    for lights in (False, True):
        desc = dark_scene(with_lights=lights)
        rt, _, _ = scenes.instantiate(desc, w, h)
        rt.renderFrames(3)
        acc = rt.ctx.read_accum()
        assert flags(rt) & abi.BUILD_FLAG_DARK_MATERIALS
        rt.ctx.set_option(abi.OPT_LAST_SEGMENT_SHORTCUT, 0)
        rt.resetAccumulation()
        rt.renderFrames(3)
        full = rt.ctx.read_accum()
        rt.close()
        assert np.array_equal(acc, full, equal_nan=True)
        ort, _, _ = oracle.instantiate(desc, w, h)
        for _ in range(3):
            ort.render()
        frac, rmse, nanmm, _ = compare_accum(acc, ort.ctx.read_accum(), frames=3)
        assert frac >= 0.995 and rmse <= 0.02 and nanmm == 0, (lights, frac, rmse)


@pytest.mark.parametrize("case", range(10))
def test_shortcut_equals_full_shading_on_random_scenes(case):
    """The same property over scenes nobody arranged: random object counts, tessellations, textures, rotations and scales, odd frame sizes that
    leave partly filled 8x8 tiles, a striped tile now and then, 1-4 bounces, batches and single calls — shortcut on and off, bit for bit."""
    rng = np.random.default_rng(500 + case)
    kw = dict(seed=int(rng.integers(1, 10**6)), n_objects=int(rng.integers(1, 9)), detail=int(rng.integers(3, 12)), with_textures=bool(rng.integers(0, 2)),
              with_lights=False, transforms=bool(rng.integers(0, 2)))
    if kw["transforms"] and rng.integers(0, 2):
        kw["scales"] = [float(rng.uniform(1.0, 2.5)) for _ in range(3)]
    W, H = int(rng.integers(17, 260)), int(rng.integers(9, 150))
    rt, _, _ = scenes.instantiate(dark_scene(**kw), W, H)
    rt.ctx.set_option(abi.OPT_MAX_BOUNCES, int(rng.integers(1, 5)))
    if rng.integers(0, 3) == 0:
        n = int(rng.integers(2, 4))
        rt.ctx.set_tile(8, int(rng.integers(0, n)), n)
    how, frames = ("frames", int(rng.integers(1, 9))) if rng.integers(0, 2) else ("calls", int(rng.integers(1, 4)))
    (a_on, r_on), (a_off, r_off) = render_both(rt, frames, how)
    assert flags(rt) & abi.BUILD_FLAG_DARK_MATERIALS
    assert np.array_equal(a_on, a_off, equal_nan=True) and np.array_equal(r_on, r_off), (kw, W, H, how, frames)
    rt.close()

"""How much can the UNPINNED part of the sampler move the picture?  (D3D11 fixed-function sampling — fixed-point
bilinear weights, filtering across cube-face edges — is not in the reference repository; RaytracerCS.hlsl:88,135, sampler state
RayTracer.cpp:99-109.)  The oracle's default is exact fp32 weights and clamp-to-edge inside a cube face; two oracle-only models
bracket the hardware: weights snapped to 8 fractional bits (the D3D11 minimum precision) and seamless cube edges.  The small versions
of BASELINE configs 3 and 4 (tests/golden/make_golden.py: golden_scenes) are rendered with each model against the default, at 1 spp
and at 256 spp; what they change must sit inside the tolerance the GPU-vs-oracle parity tests allow (8c's N >= 256 rule), so that
tolerance really does cover the unpinned choice.  CPU only; the measured table is in DESIGN.md section 2."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import abi, scenes  # noqa: F401

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_golden import golden_scenes  # noqa: E402

MODELS = {"weights8": (8, 0), "seamless": (0, 1), "both": (8, 1)}


def render(oracle, desc, w, h, frames, bounces, model):
    rt, sc, res = oracle.instantiate(desc, w, h)
    fn = oracle.lib.okrt_oracle_set_sampler_model
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int]
    assert fn(rt.ctx.h, *model) == 0
    rt.ctx.set_option(abi.OPT_MAX_BOUNCES, bounces)
    rt.renderFrames(frames)
    acc = rt.ctx.read_accum().copy()
    rt.close()
    return acc


def deltas(a, b, frames):
    x, y = a[..., :3].astype(np.float64) / frames, b[..., :3].astype(np.float64) / frames
    fin = np.isfinite(x).all(-1) & np.isfinite(y).all(-1)
    d = np.where(fin[..., None], x - y, 0.0)
    l2 = np.sqrt((d * d).sum(-1))
    return l2, float(np.sqrt((d * d).mean())), fin


@pytest.mark.parametrize("name", ["glass_b4", "sponza_tiny"])
def test_unpinned_sampler_choices_stay_inside_the_parity_tolerance(oracle, name):
    desc, w, h, _, bounces = golden_scenes()[name]
    rows = []
    for frames in (1, 256):
        base = render(oracle, desc, w, h, frames, bounces, (0, 0))
        again = render(oracle, desc, w, h, frames, bounces, (0, 0))
        assert np.array_equal(base, again, equal_nan=True)  # the default model is deterministic: every difference below is the model's
        for model, args in MODELS.items():
            acc = render(oracle, desc, w, h, frames, bounces, args)
            l2, rmse, fin = deltas(acc, base, frames)
            rows.append((frames, model, float(l2.max()), float(np.percentile(l2[fin], 99.9)), rmse, float((l2[fin] > 1e-3).mean()), float((l2[fin] > 4 / 255).mean())))
            assert (np.isfinite(acc[..., :3]).all(-1) == np.isfinite(base[..., :3]).all(-1)).all()
            assert (l2 > 0).any(), (name, model)  # the model really changes something on this scene
            if frames >= 256:  # SURVEY 8c, N >= 256: per-pixel L2 <= 4/255 for >= 99.9 % of the finite pixels, RMSE <= 1/255
                assert (l2[fin] <= 4 / 255).mean() >= 0.999, (name, model, rows[-1])
                assert rmse <= 1 / 255, (name, model, rows[-1])
            else:
                # 1 spp: whole-image RMSE <= 0.02 holds with a wide margin. The per-pixel half of the 1-spp rule (>= 99.5 % within 1e-3) is
                # a GPU-vs-oracle rule — both sides use the SAME sampler model — and is NOT a bound against D3D hardware: a weight off by
                # 2^-9 moves a magnified high-contrast texel pair by up to 0.002 per channel (1.3 % of glass_b4's pixels exceed 1e-3, none
                # exceeds 2e-3 + the rare flipped branch). Against hardware the claim is the N >= 256 rule asserted above.
                assert rmse <= 0.02 and np.percentile(l2[fin], 99.9) <= 4 / 255, (name, model, rows[-1])
    for r in rows:
        print("%-12s spp %3d %-9s max L2 %.5f  p99.9 %.5f  rmse %.6f  frac > 1e-3 %.5f  frac > 4/255 %.5f" % ((name,) + r))

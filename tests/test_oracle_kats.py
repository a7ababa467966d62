"""Known-answer tests pinning the oracle's leaf functions to the shader text, one by one.

The reference has no tests or golden vectors, so every expected value below was derived
by hand / with independent integer or float64 arithmetic from the HLSL formulas, citing the lines."""
import ctypes as C
import math

import numpy as np
import pytest

F3 = C.c_float * 3


def f3(*v):
    return F3(*v)


@pytest.fixture(scope="module")
def k(oracle):
    lib = oracle.lib
    lib.okrt_oracle_kat_pcg_hash.restype = C.c_uint32
    lib.okrt_oracle_kat_random_float.restype = C.c_float
    for n in ("ray_sphere", "ray_triangle", "ray_aabb", "reflectance"):
        getattr(lib, "okrt_oracle_kat_" + n).restype = C.c_float
    lib.okrt_oracle_kat_ray_sphere.argtypes = [F3, F3, F3, C.c_float]
    lib.okrt_oracle_kat_ray_triangle.argtypes = [F3, F3, C.c_float * 9, C.c_float * 2]
    lib.okrt_oracle_kat_ray_aabb.argtypes = [F3, F3, F3, F3]
    lib.okrt_oracle_kat_reflectance.argtypes = [C.c_float, C.c_float]
    lib.okrt_oracle_kat_refract.argtypes = [F3, F3, C.c_float, F3]
    lib.okrt_oracle_kat_transparency_bounce.argtypes = [F3, F3, C.c_float, C.POINTER(C.c_uint32), F3]
    lib.okrt_oracle_kat_reflect_direction.argtypes = [F3, F3, C.c_float, C.POINTER(C.c_uint32), F3]
    lib.okrt_oracle_kat_random_vector.argtypes = [C.POINTER(C.c_uint32), F3]
    lib.okrt_oracle_kat_point_in_circle.argtypes = [C.POINTER(C.c_uint32), C.c_float * 2]
    return lib


def pcg_py(seed):
    """GPU-Utilities.hlsli:18-25 in Python integers. `747796405u + 2891336453u` is ONE constant (Q2)."""
    M = 0xFFFFFFFF
    seed = (seed * ((747796405 + 2891336453) & M)) & M
    seed = (((seed >> ((seed >> 28) + 4)) ^ seed) * 277803737) & M
    seed = ((seed >> 22) ^ seed) & M
    return seed


def test_pcg_multiplier_is_sum_of_both_constants():
    assert (747796405 + 2891336453) & 0xFFFFFFFF == 3639132858


@pytest.mark.parametrize("seed", [0, 1, 2, 74813, 0xDEADBEEF, 0xFFFFFFFF, 1920 * 74813 * 2])
def test_pcg_hash_chain(k, seed):
    s = C.c_uint32(seed)
    expect = seed
    for _ in range(16):
        expect = pcg_py(expect)
        got = k.okrt_oracle_kat_pcg_hash(C.byref(s))
        assert got == expect and s.value == expect  # state == output (:24)


def test_pcg_hash_zero_is_fixed_point(k):
    s = C.c_uint32(0)
    assert k.okrt_oracle_kat_pcg_hash(C.byref(s)) == 0  # -> randomFloat 0 -> log(0) = -inf downstream (Q1)


def test_random_float_is_hash_over_2_pow_32_and_can_reach_one(k):
    s = C.c_uint32(12345)
    h = pcg_py(12345)
    assert k.okrt_oracle_kat_random_float(C.byref(s)) == np.float32(np.float32(h) / np.float32(4294967296.0))
    # (float)0xFFFFFFFF rounds to 2^32 (:53), and so does any hash >= 0xFFFFFF80: the result is exactly 1.0
    assert np.float32(0xFFFFFF80) / np.float32(4294967296.0) == np.float32(1.0)
    assert np.float32(0xFFFFFF7F) / np.float32(4294967296.0) < np.float32(1.0)


def test_seed_formula_and_primary_ray_centre(k, oracle):
    """seed = x + (y + 74813) * W * (N + 1) (RaytracerCS.hlsl:430); the ray through the image centre of an
    identity camera points down +z (createRay :177-206, LH projection)."""
    lib = oracle.lib
    rd = conftest_rd(64, 48, n=3)
    lib.okrt_oracle_kat_primary_ray.argtypes = [C.POINTER(type(rd)), C.c_uint32, C.c_uint32, F3, F3, C.POINTER(C.c_uint32)]
    o, d, seed = F3(), F3(), C.c_uint32()
    lib.okrt_oracle_kat_primary_ray(C.byref(rd), 32, 24, o, d, C.byref(seed))
    s = (32 + (24 + 74813) * 64 * (3 + 1)) & 0xFFFFFFFF
    for _ in range(4):  # 2 jitter draws + 2 DOF draws (:182-183, :210)
        s = pcg_py(s)
    assert seed.value == s
    assert abs(d[0]) < 0.03 and abs(d[1]) < 0.06 and d[2] > 0.99
    assert list(o) == [0.0, 0.0, 0.0]
    # top row maps ABOVE +1 before the jitter: pos.y = H - y (Q10), so row 0 looks up
    lib.okrt_oracle_kat_primary_ray(C.byref(rd), 32, 0, o, d, C.byref(seed))
    assert d[1] > 0.5


def conftest_rd(w, h, n=1):
    from conftest import abi
    rd = abi.RenderData()
    rd.accumulationEnabled = 1
    rd.numAccumulationFrames = n
    rd.textureDims[0], rd.textureDims[1] = w, h
    # inverse of perspectiveFovLH(90 deg, w, h, 0.1, 1000) in the uploaded (row-major inverse) order
    hh = 1.0 / math.tan(math.radians(45.0))
    ww = hh * h / w
    zn, zf = 0.1, 1000.0
    A, B = (zf + zn) / (zf - zn), -(2 * zf * zn) / (zf - zn)
    inv = np.zeros((4, 4), np.float64)
    inv[0, 0], inv[1, 1] = 1 / ww, 1 / hh
    inv[2, 3] = 1.0
    inv[3, 2], inv[3, 3] = 1 / B, -A / B
    for i in range(16):
        rd.cameraInverseProjectionMatrix[i] = inv.flat[i]
        rd.cameraInverseViewMatrix[i] = np.eye(4).flat[i]
    rd.cameraNearZ = 0.1
    rd.cameraUpDir[1] = 1.0
    rd.cameraRightDir[0] = 1.0
    rd.debugMaxCount = 500
    return rd


# ---- Collision (GPU-Utilities.hlsli:95-198) -------------------------------------------------------------
def test_ray_sphere_cases(k):
    # outside, aimed at the centre: near root = dist - r
    assert k.okrt_oracle_kat_ray_sphere(f3(0, 0, -5), f3(0, 0, 1), f3(0, 0, 0), 2.0) == 3.0
    # inside: far root (:113)
    assert k.okrt_oracle_kat_ray_sphere(f3(0, 0, 0), f3(0, 0, 1), f3(0, 0, 0), 2.0) == 2.0
    # behind and outside: -1 (:105-106)
    assert k.okrt_oracle_kat_ray_sphere(f3(0, 0, 5), f3(0, 0, 1), f3(0, 0, 0), 2.0) == -1.0
    # passes beside it: -1 (:109-110)
    assert k.okrt_oracle_kat_ray_sphere(f3(3, 0, -5), f3(0, 0, 1), f3(0, 0, 0), 2.0) == -1.0
    # grazing offset 1: sideB = sqrt(4 - 1)
    got = k.okrt_oracle_kat_ray_sphere(f3(1, 0, -5), f3(0, 0, 1), f3(0, 0, 0), 2.0)
    assert got == pytest.approx(5 - math.sqrt(3), rel=1e-6)


def test_ray_triangle_barycentrics_follow_the_edge_convention(k):
    """E1 = p0 - p2, E2 = p1 - p2 (:124): u weights p0, v weights p1, 1-u-v weights p2."""
    tri = (C.c_float * 9)(0, 0, 0, 4, 0, 0, 0, 4, 0)  # p0, p1, p2
    uv = (C.c_float * 2)()
    t = k.okrt_oracle_kat_ray_triangle(f3(1, 1, -3), f3(0, 0, 1), tri, uv)
    assert t == 3.0
    # hit point (1,1,0) = u*p0 + v*p1 + w*p2 -> v = 0.25, w = 0.25, u = 0.5
    assert uv[0] == pytest.approx(0.5) and uv[1] == pytest.approx(0.25)
    # at p0 exactly
    t = k.okrt_oracle_kat_ray_triangle(f3(0, 0, -1), f3(0, 0, 1), tri, uv)
    assert t == 1.0 and uv[0] == pytest.approx(1.0) and uv[1] == pytest.approx(0.0)
    # no back-face culling: same hit from the other side
    assert k.okrt_oracle_kat_ray_triangle(f3(1, 1, 3), f3(0, 0, -1), tri, uv) == 3.0
    # outside (u + v > 1), parallel, behind
    assert k.okrt_oracle_kat_ray_triangle(f3(3, 3, -3), f3(0, 0, 1), tri, uv) == -1.0
    assert k.okrt_oracle_kat_ray_triangle(f3(1, 1, -3), f3(1, 0, 0), tri, uv) == -1.0
    assert k.okrt_oracle_kat_ray_triangle(f3(1, 1, 3), f3(0, 0, 1), tri, uv) == -1.0


def test_ray_aabb_distance(k):
    FLT_MAX = np.finfo(np.float32).max
    # from outside: the near distance
    assert k.okrt_oracle_kat_ray_aabb(f3(0, 0, -5), f3(0, 0, 1), f3(-1, -1, -1), f3(1, 1, 1)) == 4.0
    # origin inside: negative near distance is returned as is (:195-196)
    assert k.okrt_oracle_kat_ray_aabb(f3(0, 0, 0), f3(0, 0, 1), f3(-1, -1, -1), f3(1, 1, 1)) == -1.0
    # miss, behind
    assert k.okrt_oracle_kat_ray_aabb(f3(3, 0, -5), f3(0, 0, 1), f3(-1, -1, -1), f3(1, 1, 1)) == FLT_MAX
    assert k.okrt_oracle_kat_ray_aabb(f3(0, 0, 5), f3(0, 0, 1), f3(-1, -1, -1), f3(1, 1, 1)) == FLT_MAX
    # diagonal ray: 1/0 = inf semantics on no axis here; check a slanted entry distance
    d = 1 / math.sqrt(2)
    got = k.okrt_oracle_kat_ray_aabb(f3(-3, 0, -3), f3(d, 0, d), f3(-1, -1, -1), f3(1, 1, 1))
    assert got == pytest.approx(2 * math.sqrt(2), rel=1e-6)


# ---- BSDF quirks (RaytracerCS.hlsl:100-131, GPU-Utilities.hlsli:85-93) ---------------------------------------
def test_reflectance_uses_one_PLUS_r0(k):  # Q5
    for cos, idx in [(1.0, 1.5), (0.5, 1.5), (0.2, 1 / 1.5), (0.0, 1.33)]:
        r0 = ((1 - idx) / (1 + idx)) ** 2
        expect = r0 + (1 + r0) * (1 - cos) ** 5
        assert k.okrt_oracle_kat_reflectance(cos, idx) == pytest.approx(expect, rel=2e-6, abs=1e-7)


def test_reflectance_pow_of_negative_base_is_nan(k):
    """HLSL pow(x, y) = exp2(y * log2(x)): NaN for x < 0, i.e. when cosine rounds above 1."""
    assert math.isnan(k.okrt_oracle_kat_reflectance(1.0000001, 1.5))


def test_user_refract(k):  # Q8
    out = F3()
    k.okrt_oracle_kat_refract(f3(0, 0, 1), f3(0, 0, -1), 1 / 1.5, out)  # straight through
    assert list(out) == pytest.approx([0, 0, 1])
    d = np.array([math.sin(0.5), 0, math.cos(0.5)])
    n = np.array([0, 0, -1.0])
    k.okrt_oracle_kat_refract(f3(*d), f3(*n), 1 / 1.5, out)
    cos_t = min(float(-d @ n), 1.0)
    perp = (1 / 1.5) * (d + cos_t * n)
    par = -math.sqrt(abs(1 - perp @ perp)) * n
    assert list(out) == pytest.approx(list(perp + par), rel=1e-6, abs=1e-7)


def test_transparency_bounce_always_consumes_one_draw(k):  # Q6: SM5 `||` does not short-circuit
    out = F3()
    # total internal reflection from inside (ior 1.5, grazing): cannot_refract is true, the draw still happens
    d = np.array([math.sin(1.2), 0, math.cos(1.2)])
    for direction, normal, ior in [(d, (0, 0, 1), 1.5), ((0, 0, 1), (0, 0, -1), 1.5), ((0, 0, 1), (0, 0, -1), 1.0)]:
        s = C.c_uint32(777)
        k.okrt_oracle_kat_transparency_bounce(f3(*direction), f3(*normal), ior, C.byref(s), out)
        assert s.value == pcg_py(777)


def test_transparency_bounce_reflected_branch_still_refracts(k):  # Q7
    out = F3()
    d = np.array([math.sin(1.2), 0.0, math.cos(1.2)])
    n_in = np.array([0.0, 0.0, 1.0])   # dot(d, n) > 0: back face, normal flipped, ratio = ior
    s = C.c_uint32(5)
    k.okrt_oracle_kat_transparency_bounce(f3(*d), f3(*n_in), 1.5, C.byref(s), out)
    n = -n_in
    assert 1.5 * math.sqrt(1 - (float(-d @ n)) ** 2) > 1  # cannot refract
    r = d - 2 * (d @ n) * n                       # reflect(direction, normal)
    cos_t = min(float(-r @ n), 1.0)
    perp = 1.5 * (r + cos_t * n)
    expect = perp - math.sqrt(abs(1 - perp @ perp)) * n   # ... and then refract() anyway
    assert list(out) == pytest.approx(list(expect), rel=2e-5, abs=2e-6)


def test_reflect_direction_draw_count_and_specular_limit(k):
    out = F3()
    s = C.c_uint32(99)
    k.okrt_oracle_kat_reflect_direction(f3(0, -1, 0), f3(0, 1, 0), 0.0, C.byref(s), out)
    exp_seed = 99
    for _ in range(6):  # getRandomVector = 3 normals x 2 draws, consumed even at roughness 0 (:109)
        exp_seed = pcg_py(exp_seed)
    assert s.value == exp_seed
    assert list(out) == pytest.approx([0, 1, 0], abs=1e-6)  # roughness 0 -> mirror direction


def test_random_vector_box_muller_order(k):
    """randomFloatNormalDistribution: u1 (theta) is drawn BEFORE u2 (rho) (:59-60)."""
    s = C.c_uint32(4242)
    out = F3()
    k.okrt_oracle_kat_random_vector(C.byref(s), out)
    st, vals = 4242, []
    for _ in range(3):
        st = pcg_py(st); u1 = np.float32(st) / np.float32(2 ** 32)
        st = pcg_py(st); u2 = np.float32(st) / np.float32(2 ** 32)
        vals.append(math.sqrt(-2 * math.log(float(u2))) * math.cos(2 * 3.14159265 * float(u1)))
    v = np.array(vals) / np.linalg.norm(vals)
    assert list(out) == pytest.approx(list(v), rel=1e-4, abs=1e-5)
    assert s.value == st


def test_point_in_circle_order(k):
    s = C.c_uint32(31337)
    out = (C.c_float * 2)()
    k.okrt_oracle_kat_point_in_circle(C.byref(s), out)
    st = pcg_py(31337); ang = float(np.float32(st) / np.float32(2 ** 32)) * 2 * 3.14159265
    st = pcg_py(st); r = math.sqrt(float(np.float32(st) / np.float32(2 ** 32)))
    assert list(out) == pytest.approx([math.cos(ang) * r, math.sin(ang) * r], rel=1e-4, abs=1e-5)


# ---- sampling ----------------------------------------------------------------------------------------------
def test_bilinear_wrap_and_slice_rounding(oracle):
    ctx = oracle.context()
    tex = np.zeros((2, 2, 2, 4), np.uint8)
    tex[0, 0, 0] = [255, 0, 0, 255]; tex[0, 0, 1] = [0, 255, 0, 255]
    tex[0, 1, 0] = [0, 0, 255, 255]; tex[0, 1, 1] = [255, 255, 255, 255]
    tex[1] = 51  # 0.2
    ctx.set_textures(tex)
    lib = oracle.lib
    lib.okrt_oracle_kat_sample_texture.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.c_float, F3]
    out = F3()
    lib.okrt_oracle_kat_sample_texture(ctx.h, 0, 0.25, 0.25, out)   # centre of texel (0,0)
    assert list(out) == pytest.approx([1, 0, 0])
    lib.okrt_oracle_kat_sample_texture(ctx.h, 0, 0.5, 0.25, out)    # halfway between texel (0,0) and (1,0)
    assert list(out) == pytest.approx([0.5, 0.5, 0])
    lib.okrt_oracle_kat_sample_texture(ctx.h, 0, 0.0, 0.25, out)    # u = 0: WRAP blends texel 1 and texel 0 (RayTracer.cpp:102)
    assert list(out) == pytest.approx([0.5, 0.5, 0])
    lib.okrt_oracle_kat_sample_texture(ctx.h, 0, 1.25, -0.75, out)  # whole periods away from (0.25, 0.25)
    assert list(out) == pytest.approx([1, 0, 0])
    lib.okrt_oracle_kat_sample_texture(ctx.h, 0, 0.5, 0.5, out)     # centre of the 2x2: mean of all four
    assert list(out) == pytest.approx([0.5, 0.5, 0.5])
    lib.okrt_oracle_kat_sample_texture(ctx.h, 1, 0.3, 0.7, out)
    assert list(out) == pytest.approx([0.2, 0.2, 0.2])
    lib.okrt_oracle_kat_sample_texture(ctx.h, 7, 0.3, 0.7, out)     # slice index clamps to the array
    assert list(out) == pytest.approx([0.2, 0.2, 0.2])
    ctx.close()


def test_cube_faces_follow_the_d3d_table(oracle):
    """Face order +X,-X,+Y,-Y,+Z,-Z (RayTracer.cpp:377-399); (sc,tc) per the D3D/GL cube table."""
    ctx = oracle.context()
    faces = np.zeros((6, 2, 2, 4), np.uint8)
    for f in range(6):
        faces[f, :, :, 0] = 40 * (f + 1)
        faces[f, 0, 0, 1] = 255  # mark texel (x=0, y=0) of every face in green
    ctx.set_envmap(faces)
    lib = oracle.lib
    lib.okrt_oracle_kat_sample_env.argtypes = [C.c_void_p, F3, F3]
    out = F3()
    for f, d in enumerate([(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]):
        lib.okrt_oracle_kat_sample_env(ctx.h, f3(*d), out)
        assert out[0] == pytest.approx(40 * (f + 1) / 255), f
    # +X face: sc = -z, tc = -y -> texel (0,0) is at z > 0, y > 0
    lib.okrt_oracle_kat_sample_env(ctx.h, f3(1, 0.5, 0.5), out)
    assert out[1] == pytest.approx(1.0)
    lib.okrt_oracle_kat_sample_env(ctx.h, f3(1, -0.5, -0.5), out)
    assert out[1] == pytest.approx(0.0)
    # +Y face: sc = +x, tc = +z -> texel (0,0) at x < 0, z < 0
    lib.okrt_oracle_kat_sample_env(ctx.h, f3(-0.5, 1, -0.5), out)
    assert out[1] == pytest.approx(1.0)
    # -Z face: sc = -x, tc = -y -> texel (0,0) at x > 0, y > 0
    lib.okrt_oracle_kat_sample_env(ctx.h, f3(0.5, 0.5, -1), out)
    assert out[1] == pytest.approx(1.0)
    ctx.set_envmap(None)  # unbound: black (RayTracer.cpp:349-350)
    lib.okrt_oracle_kat_sample_env(ctx.h, f3(0, 1, 0), out)
    assert list(out) == [0, 0, 0]
    ctx.close()


def test_unorm8_newton_step_is_exact():
    """csrc/okrt_device.h unorm8(): c * fl(1/255) followed by one fma Newton step equals the correctly rounded c / 255.f
    for every 8-bit value (the device code decodes texels this way; the oracle divides)."""
    c = np.arange(256, dtype=np.float64)
    r = np.float64(np.float32(1.0) / np.float32(255.0))
    q = (c * r).astype(np.float32).astype(np.float64)               # fl32(c * r)
    e = (c - q * 255.0).astype(np.float32).astype(np.float64)       # fma(-q, 255, c): the product is exact in f64
    v = (e * r + q).astype(np.float32)                              # fma(e, r, q)
    assert np.array_equal(v, np.arange(256, dtype=np.float32) / np.float32(255.0))

"""The device's leaf functions against the oracle's, bit for bit, on thousands of random and adversarial inputs
(okrt_debug_kat runs csrc/okrt_device.h's functions on the GPU; okrt_oracle_kat_* are the CPU restatement that
test_oracle_kats.py pins to the shader text). Everything that contains no transcendental — RNG, the three
intersection routines, the user refract, both texture samplers — must be IDENTICAL: the render-level tolerances of
test_gpu_parity.py are then left to cover sin/cos/log/exp2/log2/atan2 only."""
import ctypes as C

import numpy as np
import pytest

from conftest import abi, host, scenes

pytestmark = pytest.mark.gpu
F3 = C.c_float * 3
N = 4000


def bits(u):
    return np.asarray(u, np.uint32).view(np.float32)


def same_bits(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return np.array_equal(a.view(np.uint32), b.view(np.uint32)) or np.array_equal(a, b, equal_nan=True) and \
        np.array_equal(np.isnan(a), np.isnan(b))


@pytest.fixture(scope="module")
def gpu():
    ctx = host.Context()
    yield ctx
    ctx.close()


@pytest.fixture(scope="module")
def lib(oracle):
    l = oracle.lib
    l.okrt_oracle_kat_pcg_hash.restype = C.c_uint32
    l.okrt_oracle_kat_random_float.restype = C.c_float
    for n in ("ray_sphere", "ray_triangle", "ray_aabb"):
        getattr(l, "okrt_oracle_kat_" + n).restype = C.c_float
    l.okrt_oracle_kat_ray_sphere.argtypes = [F3, F3, F3, C.c_float]
    l.okrt_oracle_kat_ray_triangle.argtypes = [F3, F3, C.c_float * 9, C.c_float * 2]
    l.okrt_oracle_kat_ray_aabb.argtypes = [F3, F3, F3, F3]
    l.okrt_oracle_kat_refract.argtypes = [F3, F3, C.c_float, F3]
    l.okrt_oracle_kat_sample_texture.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.c_float, F3]
    l.okrt_oracle_kat_sample_env.argtypes = [C.c_void_p, F3, F3]
    return l


def test_rng(gpu, lib):
    rng = np.random.default_rng(1)
    seeds = np.concatenate([[0, 1, 2, 0xFFFFFFFF, 74813, 1920 * 74813 * 2], rng.integers(0, 2**32, N)]).astype(np.uint32)
    got = gpu.kat(abi.KAT_PCG_HASH, bits(seeds)[:, None], 2).view(np.uint32)
    gotf = gpu.kat(abi.KAT_RANDOM_FLOAT, bits(seeds)[:, None], 2)
    for i, s in enumerate(seeds):
        c = C.c_uint32(int(s))
        assert lib.okrt_oracle_kat_pcg_hash(C.byref(c)) == got[i, 0] == got[i, 1] == c.value
        c = C.c_uint32(int(s))
        want = lib.okrt_oracle_kat_random_float(C.byref(c))
        assert np.float32(want) == gotf[i, 0] and c.value == gotf[i, 1:].view(np.uint32)[0]


def rays(rng, n, spread=10.0):
    o = rng.normal(0, spread, (n, 3)).astype(np.float32)
    d = rng.normal(0, 1, (n, 3)).astype(np.float32)
    d[: n // 8] = np.sign(d[: n // 8])                        # axis-diagonal directions
    d[n // 8: n // 4, rng.integers(0, 3)] = 0.0               # a zero component: 1/0 = inf in the slab test
    return o, d


def test_ray_sphere(gpu, lib):
    rng = np.random.default_rng(2)
    o, d = rays(rng, N)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    pos = rng.normal(0, 10, (N, 3)).astype(np.float32)
    rad = rng.uniform(0.1, 12, N).astype(np.float32)
    pos[:50] = o[:50]                                         # origin at the centre
    aim = slice(1000, 3000)                                   # half of the rays are aimed near their sphere
    t = pos[aim] + rng.normal(0, 0.6, (2000, 3)).astype(np.float32) * rad[aim, None] - o[aim]
    d[aim] = (t / np.linalg.norm(t, axis=1, keepdims=True)).astype(np.float32)
    got = gpu.kat(abi.KAT_RAY_SPHERE, np.concatenate([o, d, pos, rad[:, None]], 1), 1)[:, 0]
    want = np.array([lib.okrt_oracle_kat_ray_sphere(F3(*o[i]), F3(*d[i]), F3(*pos[i]), float(rad[i])) for i in range(N)], np.float32)
    assert same_bits(got, want)
    assert (got > 0).mean() > 0.2 and (got == -1).mean() > 0.2


def test_ray_triangle(gpu, lib):
    rng = np.random.default_rng(3)
    o, d = rays(rng, N, 4.0)
    tri = rng.normal(0, 4, (N, 9)).astype(np.float32)
    tri[:40, 3:6] = tri[:40, 0:3]                             # degenerate: two equal vertices
    d[40:80] = (tri[40:80, 3:6] - tri[40:80, 0:3])           # ray parallel to an edge
    o[80:400] = (tri[80:400, 0:3] + tri[80:400, 3:6] + tri[80:400, 6:9]) / 3 - d[80:400] * 2  # aimed at the centroid
    o[400:500] = tri[400:500, 0:3] - d[400:500]               # aimed exactly at a vertex
    got = gpu.kat(abi.KAT_RAY_TRIANGLE, np.concatenate([o, d, tri], 1), 3)
    want = np.zeros((N, 3), np.float32)
    for i in range(N):
        uv = (C.c_float * 2)()
        want[i, 0] = lib.okrt_oracle_kat_ray_triangle(F3(*o[i]), F3(*d[i]), (C.c_float * 9)(*tri[i]), uv)
        want[i, 1:] = uv[0], uv[1]
    hit = want[:, 0] > 0
    assert same_bits(got[:, 0], want[:, 0])
    assert same_bits(got[hit, 1:], want[hit, 1:])             # barycentrics are only defined on a hit
    assert hit.sum() > 300


def test_ray_aabb(gpu, lib):
    rng = np.random.default_rng(4)
    o, d = rays(rng, N)
    a, b = rng.normal(0, 8, (N, 3)).astype(np.float32), rng.normal(0, 8, (N, 3)).astype(np.float32)
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    o[:200] = ((lo[:200] + hi[:200]) / 2).astype(np.float32)  # origin inside: negative entry distance
    lo[200:260, 1] = hi[200:260, 1]                           # flat box
    aim = slice(1000, 3000)                                   # half of the rays are aimed into their box
    d[aim] = (lo[aim] + (hi[aim] - lo[aim]) * rng.uniform(0, 1, (2000, 3)).astype(np.float32)) - o[aim]
    got = gpu.kat(abi.KAT_RAY_AABB, np.concatenate([o, d, lo, hi], 1), 1)[:, 0]
    want = np.array([lib.okrt_oracle_kat_ray_aabb(F3(*o[i]), F3(*d[i]), F3(*lo[i]), F3(*hi[i])) for i in range(N)], np.float32)
    assert same_bits(got, want)
    assert (got < 3e38).mean() > 0.2


def test_user_refract(gpu, lib):
    rng = np.random.default_rng(5)
    d = rng.normal(0, 1, (N, 3)).astype(np.float32)
    n = rng.normal(0, 1, (N, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True).astype(np.float32)
    ratio = rng.choice(np.array([1 / 1.5, 1.5, 1.0, 1 / 1.33, 2.4], np.float32), N)
    got = gpu.kat(abi.KAT_REFRACT, np.concatenate([d, n, ratio[:, None]], 1), 3)
    want = np.zeros((N, 3), np.float32)
    for i in range(N):
        out = F3()
        lib.okrt_oracle_kat_refract(F3(*d[i]), F3(*n[i]), float(ratio[i]), out)
        want[i] = out[0], out[1], out[2]
    assert same_bits(got, want)


@pytest.mark.parametrize("side", [64, 48])  # power of two (exact reciprocal path) and not
def test_texture_sampler(gpu, oracle, lib, side):
    rng = np.random.default_rng(6)
    tex = rng.integers(0, 256, (3, side, side, 4), dtype=np.uint8)
    octx = oracle.context()
    octx.set_textures(tex)
    gpu.set_textures(tex)
    idx = rng.integers(0, 5, N).astype(np.uint32)             # beyond the array: clamps
    uv = rng.uniform(-3, 3, (N, 2)).astype(np.float32)
    uv[:100] = np.round(uv[:100] * side) / side               # exactly on texel edges
    uv[100:140] = [[0.0, 0.0]]
    uv[140:150] = [[np.inf, np.nan]]                          # non-finite coordinates are zeroed
    got = gpu.kat(abi.KAT_SAMPLE_TEXTURE, np.concatenate([bits(idx)[:, None], uv], 1), 3)
    want = np.zeros((N, 3), np.float32)
    for i in range(N):
        out = F3()
        lib.okrt_oracle_kat_sample_texture(octx.h, int(idx[i]), float(uv[i, 0]), float(uv[i, 1]), out)
        want[i] = out[0], out[1], out[2]
    octx.close()
    assert same_bits(got, want)


def test_environment_sampler(gpu, oracle, lib):
    rng = np.random.default_rng(7)
    faces = rng.integers(0, 256, (6, 32, 32, 4), dtype=np.uint8)
    octx = oracle.context()
    octx.set_envmap(faces)
    gpu.set_envmap(faces)
    d = rng.normal(0, 1, (N, 3)).astype(np.float32)
    d[:60] = np.sign(d[:60])                                  # cube corners / edges: face selection ties
    d[60:70] = 0.0
    got = gpu.kat(abi.KAT_SAMPLE_ENV, d, 3)
    want = np.zeros((N, 3), np.float32)
    for i in range(N):
        out = F3()
        lib.okrt_oracle_kat_sample_env(octx.h, F3(*d[i]), out)
        want[i] = out[0], out[1], out[2]
    octx.close()
    assert same_bits(got, want)


def test_reference_order_closest_hit(oracle, lib):
    """The reference-order traversal (octree -> instances -> per-mesh BVH, RaytracerCS.hlsl:218-334) ray by ray: hit
    distance, hit flag and the reference's own bbCheckCount / triCheckCount must equal the oracle's exactly."""
    desc = scenes.small_mesh_scene(seed=17, n_objects=6, detail=10, with_lights=False)
    rt, sc, res = scenes.instantiate(desc, 64, 36)
    ort, _, _ = oracle.instantiate(desc, 64, 36)
    rt.updateBuffers()
    ort.updateBuffers()
    rd = ort.render_data
    lib.okrt_oracle_kat_closest_hit.argtypes = [C.c_void_p, C.POINTER(type(rd)), F3, F3, F3, C.c_uint32 * 3]
    rng = np.random.default_rng(8)
    n = 3000
    o = rng.normal(0, 14, (n, 3)).astype(np.float32)
    o[:, 1] = np.abs(o[:, 1]) + 1.0
    target = rng.normal(0, 5, (n, 3)).astype(np.float32)
    d = target - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rec = np.concatenate([o, d, np.full((n, 1), rd.numSpheres, np.uint32).view(np.float32)], 1)
    got = rt.ctx.kat(abi.KAT_CLOSEST_HIT, rec, 6)
    gi = got.view(np.uint32)
    hits = 0
    for i in range(n):
        out, ids = F3(), (C.c_uint32 * 3)()
        lib.okrt_oracle_kat_closest_hit(ort.ctx.h, C.byref(rd), F3(*o[i]), F3(*d[i]), out, ids)
        assert np.float32(out[0]).view(np.uint32) == gi[i, 0], (i, out[0], got[i, 0])
        assert (ids[0], ids[1], ids[2]) == (gi[i, 1], gi[i, 2], gi[i, 3]), i
        hits += ids[0]
    assert hits > n // 3
    # the derived structure, host- and device-built: same rays, same closest-hit DISTANCE bit for bit (identity
    # instances; conservative culling cannot lose the nearest triangle). Which triangle carries it may differ on ties.
    for builder in (abi.BUILDER_HOST, abi.BUILDER_DEVICE):
        rt.ctx.set_option(abi.OPT_BUILDER, builder)
        fast = rt.ctx.kat(abi.KAT_CLOSEST_HIT_FAST, rec, 5)
        fi = fast.view(np.uint32)
        assert np.array_equal(fi[:, 0], gi[:, 0]) and np.array_equal(fi[:, 1], gi[:, 1]), builder
        hit = gi[:, 1] == 1
        assert (fi[hit, 3] == gi[hit, 4]).mean() > 0.999 and (fi[hit, 4] == gi[hit, 5]).mean() > 0.995
        assert fi[:, 2].mean() < gi[:, 2].mean()  # 4-wide node visits vs the reference's box checks, even unordered
    rt.close()
    ort.close()


# ---- functions with transcendentals: device libm (ocml) vs glibc agree to the last ulp or two, not to the bit ---------
def close(a, b, rel=4e-6, abs_=4e-6):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    ok = np.isclose(a, b, rtol=rel, atol=abs_) | (np.isnan(a) & np.isnan(b)) | (np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b)))
    return ok


def test_random_vector_reflectance_and_bsdf_directions(gpu, lib):
    rng = np.random.default_rng(9)
    n = 2000
    lib.okrt_oracle_kat_random_vector.argtypes = [C.POINTER(C.c_uint32), F3]
    lib.okrt_oracle_kat_reflectance.argtypes = [C.c_float, C.c_float]
    lib.okrt_oracle_kat_reflectance.restype = C.c_float
    lib.okrt_oracle_kat_reflect_direction.argtypes = [F3, F3, C.c_float, C.POINTER(C.c_uint32), F3]
    lib.okrt_oracle_kat_transparency_bounce.argtypes = [F3, F3, C.c_float, C.POINTER(C.c_uint32), F3]
    seeds = rng.integers(1, 2**32, n).astype(np.uint32)
    got = gpu.kat(abi.KAT_RANDOM_VECTOR, bits(seeds)[:, None], 4)
    for i in range(n):
        s, out = C.c_uint32(int(seeds[i])), F3()
        lib.okrt_oracle_kat_random_vector(C.byref(s), out)
        assert close(got[i, :3], [out[0], out[1], out[2]]).all(), (i, got[i], list(out))
        assert s.value == got[i, 3:].view(np.uint32)[0]  # six draws consumed on both sides
    cosv = rng.uniform(0, 1, n).astype(np.float32)
    idx = rng.choice(np.array([1 / 1.5, 1.5, 1.0, 1.33, 2.4], np.float32), n)
    got = gpu.kat(abi.KAT_REFLECTANCE, np.stack([cosv, idx], 1), 1)[:, 0]
    want = [lib.okrt_oracle_kat_reflectance(float(cosv[i]), float(idx[i])) for i in range(n)]
    assert close(got, want).all()
    d = rng.normal(0, 1, (n, 3)).astype(np.float32)
    nn = rng.normal(0, 1, (n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    nn /= np.linalg.norm(nn, axis=1, keepdims=True).astype(np.float32)
    rough = rng.uniform(0, 1, n).astype(np.float32)
    ior = rng.choice(np.array([1.5, 1.33, 1.0, 2.4], np.float32), n)
    g1 = gpu.kat(abi.KAT_REFLECT_DIRECTION, np.concatenate([d, nn, rough[:, None], bits(seeds)[:, None]], 1), 4)
    g2 = gpu.kat(abi.KAT_TRANSPARENCY_BOUNCE, np.concatenate([d, nn, ior[:, None], bits(seeds)[:, None]], 1), 4)
    flips = 0
    for i in range(n):
        s, out = C.c_uint32(int(seeds[i])), F3()
        lib.okrt_oracle_kat_reflect_direction(F3(*d[i]), F3(*nn[i]), float(rough[i]), C.byref(s), out)
        assert close(g1[i, :3], [out[0], out[1], out[2]], 2e-5, 2e-5).all() and s.value == g1[i, 3:].view(np.uint32)[0], i
        s = C.c_uint32(int(seeds[i]))
        lib.okrt_oracle_kat_transparency_bounce(F3(*d[i]), F3(*nn[i]), float(ior[i]), C.byref(s), out)
        assert s.value == g2[i, 3:].view(np.uint32)[0]      # the Fresnel draw is always consumed (Q6)
        flips += not close(g2[i, :3], [out[0], out[1], out[2]], 2e-5, 2e-5).all()  # refl > rnd can flip on a last-ulp pow
    assert flips <= 2, flips


def test_primary_ray(gpu, oracle, lib):
    desc = scenes.small_mesh_scene(seed=3)
    ort, _, _ = oracle.instantiate(desc, 640, 360)
    ort.setDOF(0.7, 9.0)
    ort.updateBuffers()
    rd = ort.render_data
    lib.okrt_oracle_kat_primary_ray.argtypes = [C.POINTER(type(rd)), C.c_uint32, C.c_uint32, F3, F3, C.POINTER(C.c_uint32)]
    rdf = np.frombuffer(bytes(rd), np.float32)
    assert rdf.size == 60
    rng = np.random.default_rng(10)
    n = 2000
    px, py = rng.integers(0, 640, n).astype(np.uint32), rng.integers(0, 360, n).astype(np.uint32)
    rec = np.concatenate([bits(px)[:, None], bits(py)[:, None], np.broadcast_to(rdf, (n, 60))], 1)
    got = gpu.kat(abi.KAT_PRIMARY_RAY, rec, 7)
    for i in range(n):
        o, d, s = F3(), F3(), C.c_uint32(0)
        lib.okrt_oracle_kat_primary_ray(C.byref(rd), int(px[i]), int(py[i]), o, d, C.byref(s))
        assert s.value == got[i, 6:].view(np.uint32)[0]
        assert close(got[i, :3], list(o), 1e-5, 1e-5).all() and close(got[i, 3:6], list(d), 1e-5, 1e-5).all(), i
    ort.close()


@pytest.mark.parametrize("case", ["plain", "plus_zero", "minus_zero", "nan_basis"])
def test_primary_ray_without_a_lens_keeps_the_oracles_bits(gpu, oracle, lib, case):
    """Depth of field off (the reference's default): the device skips the lens trigonometry, whose product with strength 0 is a
    signed zero that cannot change an origin component unless that component is -0 (okrt_device.h: primaryRay). The origin must be
    the oracle's BIT FOR BIT — the oracle always evaluates the lens — with the same RNG state after the two always-consumed draws:
    an ordinary position, a +0 component (unchanged by either zero), a -0 component (the device takes the full path: -0 + +0 = +0),
    a NaN in the camera basis (full path: the NaN reaches the origin)."""
    desc = scenes.small_mesh_scene(seed=3)
    ort, _, _ = oracle.instantiate(desc, 640, 360)
    ort.setDOF(0.0, 9.0)
    ort.updateBuffers()
    rd = type(ort.render_data).from_buffer_copy(bytes(ort.render_data))
    if case == "plus_zero":
        rd.cameraPosition[1] = 0.0
    elif case == "minus_zero":
        rd.cameraPosition[1] = -0.0
        rd.cameraPosition[2] = -0.0
    elif case == "nan_basis":
        rd.cameraUpDir[0] = float("nan")
    lib.okrt_oracle_kat_primary_ray.argtypes = [C.POINTER(type(rd)), C.c_uint32, C.c_uint32, F3, F3, C.POINTER(C.c_uint32)]
    rdf = np.frombuffer(bytes(rd), np.float32)
    rng = np.random.default_rng(11)
    n = 1500
    px, py = rng.integers(0, 640, n).astype(np.uint32), rng.integers(0, 360, n).astype(np.uint32)
    rec = np.concatenate([bits(px)[:, None], bits(py)[:, None], np.broadcast_to(rdf, (n, 60))], 1)
    got = gpu.kat(abi.KAT_PRIMARY_RAY, rec, 7)
    want_o = np.zeros((n, 3), np.float32)
    for i in range(n):
        o, d, s = F3(), F3(), C.c_uint32(0)
        lib.okrt_oracle_kat_primary_ray(C.byref(rd), int(px[i]), int(py[i]), o, d, C.byref(s))
        want_o[i] = list(o)
        assert s.value == got[i, 6:].view(np.uint32)[0], i
        assert close(got[i, 3:6], list(d), 1e-5, 1e-5).all(), i
    assert np.array_equal(got[:, :3].view(np.uint32), want_o.view(np.uint32)), case  # origins bit for bit, signs of zeros and NaNs included
    if case == "minus_zero":
        assert (want_o[:, 1].view(np.uint32) == 0).any() and (want_o[:, 1].view(np.uint32) == 0x80000000).any()  # both signs really occur
    ort.close()


def test_short_reciprocal_and_square_root_are_the_ieee_ones_for_every_float(gpu):
    """okrt_device.h computes 1/x and sqrt(x) in the traversal with the hardware estimate plus the last correction step of the
    compiler's IEEE sequence (3 and 9 instructions instead of 10 and 16) wherever the exponent is mid-range. All 2^32 bit
    patterns go through both forms on the device: not one may differ (the conventions in okrt_device.h say IEEE divide and
    sqrt; test_ray_triangle and test_primary_ray above compare the functions that use them bit for bit with the CPU
    checker's true divisions and square roots)."""
    per = 1 << 14
    first = (np.arange(0, 1 << 32, per, dtype=np.uint64)).astype(np.uint32)
    rec = np.stack([bits(first), bits(np.full(first.shape, per, np.uint32))], axis=1)
    out = gpu.kat(abi.KAT_IEEE_SHORTCUTS, rec, 4).view(np.uint32)
    bad_rcp, bad_sqrt = int(out[:, 0].sum()), int(out[:, 1].sum())
    where = out[(out[:, 0] > 0) | (out[:, 1] > 0)][:4]
    assert bad_rcp == 0 and bad_sqrt == 0, (bad_rcp, bad_sqrt, [[hex(int(v)) for v in row] for row in where])


def test_the_bounce_kernel_itself_ray_by_ray():
    """OKRT_KAT_CLOSEST_HIT_TRACE puts the probe's rays on a sharded queue and traces them with traceKernel — the persistent bounce-ray
    kernel with its lane refill, shard stealing, step blocks, LDS stack and spill slab — where CLOSEST_HIT_FAST walks the same
    structure with one thread and a private stack. Same closest-hit distance and hit flag for every ray, bit for bit (the
    triangle carrying it may differ on equal-distance ties); ray counts that are no multiple of anything; and the sphere table."""
    rng = np.random.default_rng(12)
    for desc, n in ((scenes.small_mesh_scene(seed=5, n_objects=7, detail=12, with_textures=False), 70001),
                    (scenes.dragon_proxy(80000), 200003),
                    (scenes.test_scene_spheres(), 999)):
        rt, sc, res = scenes.instantiate(desc, 64, 36)
        rt.calculateProjectionData(); rt.updateBuffers()
        o = rng.normal(0, 14, (n, 3)).astype(np.float32)
        target = rng.normal(0, 5, (n, 3)).astype(np.float32)
        d = target - o
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        rec = np.concatenate([o, d, np.full((n, 1), rt.render_data.numSpheres, np.uint32).view(np.float32)], 1)
        fast = rt.ctx.kat(abi.KAT_CLOSEST_HIT_FAST, rec, 5).view(np.uint32)
        trace = rt.ctx.kat(abi.KAT_CLOSEST_HIT_TRACE, rec, 5).view(np.uint32)
        assert np.array_equal(trace[:, 0], fast[:, 0]) and np.array_equal(trace[:, 1], fast[:, 1]), desc.name
        hit = fast[:, 1] == 1
        assert hit.sum() > n // 20
        assert (trace[hit, 3] == fast[hit, 3]).mean() > 0.999 and (trace[hit, 4] == fast[hit, 4]).mean() > 0.995
        rt.close()


def test_an_empty_child_slot_that_collapses_under_rounding_does_not_end_the_ray():
    """A Node4Q's unused child slots carry an inverted box (lower plane 255, upper plane 0) that no ray enters — except when
    255 * step / d is below half an ulp of the plane distance on all three axes AND the three distances are equal: a tiny node
    seen from far away by a ray through its grid origin. Then tnear == tfar and the slot IS entered; its reference is
    kEmptyRef, which the bounce kernel also uses for "no more work", so round 3's kernel dropped the rest of the ray's stack
    there. Built here: a cluster a few float spacings wide at (30, 30, 30), seen from 8 000 units away (255 grid steps of its node
    are far below half an ulp of that distance), behind which a large wall must still be hit."""
    cluster = (np.array([30.0, 30.0, 30.0]) + np.random.default_rng(2).uniform(0, 6e-6, (30, 3, 3))).astype(np.float32)
    wall = np.array([[[2000, -3000, -3000], [2000, 6000, -3000], [2000, -3000, 6000]], [[2000, 6000, 6000], [2000, -3000, 6000], [2000, 6000, -3000]]], np.float32)
    desc = scenes.SceneDescription("collapse")
    desc.meshes = [(np.concatenate([cluster, wall]), np.zeros((32, 33), np.float32))]
    desc.entities = [scenes.MeshEntity(0, {})]
    rt, sc, res = scenes.instantiate(desc, 16, 16)
    rt.ctx.commit()
    nodes, tris = rt.ctx.fast_structure()
    exps = nodes[:, 3]
    extent = 255.0 * np.ldexp(1.0, (exps & 255).astype(np.int64) - 127)
    small = [k for k in range(len(nodes)) if extent[k] < 1e-2 and (nodes[k, 4:8] == 0xFFFFFFFF).any() and (nodes[k, 4:8] != 0xFFFFFFFF).any()]
    assert small, "the builder made no small node with an unused slot"
    recs = []
    dn = np.float32(1.0) / np.sqrt(np.float32(3.0))  # the kernel normalises (1, 1, 1): three equal components
    iq = np.float32(1.0) / dn
    for k in small:
        O = nodes[k, 0:3].view(np.float32)
        steps = np.ldexp(np.float32(1.0), ((exps[k] >> np.array([0, 8, 16])) & 255).astype(np.int64) - 127).astype(np.float32)
        for L in (8192.0, 16384.0, 24576.0, 32768.0, 49152.0, 65536.0):
            ro = (O - np.float32(L)).astype(np.float32)
            D = (O - ro).astype(np.float32)
            B = (D * iq).astype(np.float32)
            absorbed = all(np.float32(255.0 * float(steps[a] * iq) + float(B[a])) == B[a] for a in range(3))  # fma(255, A, B) == B
            if D[0] == D[1] == D[2] and absorbed:  # every plane of every child slot, the empty ones too, is at the same distance
                recs.append(np.concatenate([ro, np.float32([1, 1, 1]), np.zeros(1, np.float32)]))
    assert recs, "no ray through a grid origin with three equal, absorbing plane distances"
    rec = np.array(recs, np.float32)
    fast = rt.ctx.kat(abi.KAT_CLOSEST_HIT_FAST, rec, 5).view(np.uint32)
    trace = rt.ctx.kat(abi.KAT_CLOSEST_HIT_TRACE, rec, 5).view(np.uint32)
    assert (fast[:, 1] == 1).all()  # every such ray hits the cluster or, past it, the wall
    assert np.array_equal(trace[:, 0], fast[:, 0]) and np.array_equal(trace[:, 1], fast[:, 1]), (trace[:, :2], fast[:, :2])
    rt.close()

"""Known-answer tests for the CONTROL FLOW of the shader's `main` (RaytracerCS.hlsl:421-537), derived independently of the
C++ oracle: a sphere-only re-derivation of one whole pixel written here in Python straight from the HLSL text — integer
arithmetic for the PCG stream (exact), float64 for the geometry — and compared with the oracle's accumulation on scenes
built to drive the branches the leaf KATs (test_oracle_kats.py) cannot reach:

  * a two-segment path that ends on an emitter (emission is added AFTER `contribution *= albedo`, quirk Q12; the second
    segment's miss adds the environment; the loop ends after NUM_BOUNCES + 1 segments without an environment term);
  * transparency with total internal reflection from INSIDE a glass sphere (back face: ratio = ior, `cannot_refract`, the
    always-consumed Fresnel draw Q6, refract-after-reflect Q7, the user `refract` with sqrt(abs(..)) Q8), several bounces;
  * the spot-light gate (`cosTheta < maxAngle -> continue`, six draws consumed either way, :456-466), point lights with
    the in-front / behind modifier (:397-418) and the directional light that contributes nothing on segment 0 (Q13).

The Python path follows the shader line by line (citations inline); the only shared inputs are the RenderData matrices,
which the compiled-glm goldens pin separately (tests/test_golden.py). Agreement is required to 2e-4 on >= 99 % of the
pixels (float64 vs float32 can flip a stochastic comparison that lands within rounding of its threshold); measured: every
pixel, to 1e-7 on the sphere scenes and 2e-4 on the textured mesh scene."""
import math

import numpy as np
import pytest

from conftest import abi, host

M32 = 0xFFFFFFFF
PI = 3.14159265359  # GPU-Utilities.hlsli: #define PI (float literal)


class Rng:
    def __init__(self, seed):
        self.s = seed & M32

    def hash(self):  # GPU-Utilities.hlsli:18-25; one multiply by the SUM of the two constants (Q2)
        s = (self.s * ((747796405 + 2891336453) & M32)) & M32
        s = (((s >> ((s >> 28) + 4)) ^ s) * 277803737) & M32
        self.s = ((s >> 22) ^ s) & M32
        return self.s

    def f(self):  # :51-54, float32 division by float(0xFFFFFFFF) = 2^32
        return float(np.float32(np.float32(self.hash()) / np.float32(4294967296.0)))

    def normal(self):  # :56-62: theta first, then rho
        theta = 2 * PI * self.f()
        u = self.f()
        rho = math.sqrt(-2 * math.log(u)) if u > 0 else math.inf
        return rho * math.cos(theta)

    def vector(self):  # :64-70
        v = np.array([self.normal(), self.normal(), self.normal()])
        return v / np.linalg.norm(v)


def normalize(v):
    return v / np.linalg.norm(v)


def reflect(d, n):
    return d - 2.0 * np.dot(d, n) * n


def ray_sphere(o, d, c, r):  # GPU-Utilities.hlsli:98-114
    to = c - o
    tca = np.dot(to, d)
    m2 = np.dot(to, to)
    r2 = r * r
    if tca < 0 and m2 > r2:
        return -1.0
    a = m2 - tca * tca
    if a > r2:
        return -1.0
    b = math.sqrt(r2 - a)
    return tca - b if m2 > r2 else tca + b


def reflectance(cosine, idx):  # RaytracerCS.hlsl:100-105, (1 + r0) (Q5)
    r0 = ((1 - idx) / (1 + idx)) ** 2
    base = 1 - cosine
    return r0 + (1 + r0) * (base ** 5 if base >= 0 else math.nan)


def refract_user(d, n, ratio):  # GPU-Utilities.hlsli:85-93 (Q8)
    cos_t = min(np.dot(-d, n), 1.0)
    perp = ratio * (d + cos_t * n)
    return perp - math.sqrt(abs(1 - np.dot(perp, perp))) * n


def transparency_bounce(d, n, ior, rng):  # RaytracerCS.hlsl:114-131
    front = np.dot(d, n) < 0
    ratio = 1.0 / ior if front else ior
    if not front:
        n = -n
    cos_t = np.dot(-d, n)
    sin_t = math.sqrt(max(1 - cos_t * cos_t, 0.0)) if 1 - cos_t * cos_t >= 0 else math.nan
    cannot = ratio * sin_t > 1
    u = rng.f()  # no short-circuit in SM5: always drawn (Q6)
    refl = reflectance(cos_t, ratio)
    if cannot or refl > u:
        d = reflect(d, n)
    return refract_user(d, n, ratio)  # also after reflecting (Q7)


def get_lighting(o, d, hit, lpos, lrad, lcol, lint):  # :397-418
    dist = ray_sphere(o, d, lpos, lrad)
    if dist < 0:
        return np.zeros(3)
    mod = 1.0
    if hit and hit["distance"] > dist:
        mod *= max(np.dot(d, -hit["normal"]), 0.0)
    elif hit and hit["distance"] < dist:
        mod *= hit["mat"]["transparency"] * max(np.dot(d, -hit["normal"]), 0.0)
    return lcol * lint * mod


def pixel(px, py, rd, spheres, lights, env_rgb, num_bounces):
    """One sample of pixel (px, py): RaytracerCS.hlsl:421-537 for a sphere-only scene. Returns `light`."""
    W, H = rd.textureDims[0], rd.textureDims[1]
    rng = Rng(px + (py + 74813) * W * (rd.numAccumulationFrames + 1))  # :430
    invP = np.array(list(rd.cameraInverseProjectionMatrix), np.float64).reshape(4, 4)
    invV = np.array(list(rd.cameraInverseViewMatrix), np.float64).reshape(4, 4)
    x = (px + rng.f() - 0.5) / W * 2 - 1  # :179-187, y uses H - py (Q10)
    y = ((H - py) + rng.f() - 0.5) / H * 2 - 1
    target = invP @ np.array([x, y, rd.cameraNearZ, 1.0])
    d = (invV @ np.append(normalize(target[:3] / target[2]), 0.0))[:3]
    o = np.array(list(rd.cameraPosition), np.float64)
    ang = rng.f() * 2 * PI  # applyDOF :208-216, always two draws
    jit = np.array([math.cos(ang), math.sin(ang)]) * math.sqrt(rng.f()) * rd.dofStrength
    off = np.array(list(rd.cameraRightDir)) * jit[0] + np.array(list(rd.cameraUpDir)) * jit[1]
    focus = o + d * (rd.dofDistance + rd.cameraNearZ)
    o = o + off
    d = normalize(focus - o)

    light, contribution = np.zeros(3), np.ones(3)
    for i in range(num_bounces + 1):
        hit = None
        best = 3.402823466e38
        for s in spheres:  # :232-242
            t = ray_sphere(o, d, s["pos"], s["radius"])
            if 0 < t < best:
                best = t
                pos = o + d * t
                hit = {"distance": t, "pos": pos, "normal": normalize(pos - s["pos"]), "mat": s}
        for (lpos, lcol, lint, lrad) in lights.get("point", []):  # :448-454
            light = light + get_lighting(o, d, hit, lpos, lrad, lcol, lint) * contribution
        for (lpos, ldir, lcol, lint, lrad, cos_max) in lights.get("spot", []):  # :456-466
            to_light = normalize(lpos + rng.vector() * 0.01 - o)
            if np.dot(to_light, -ldir) < cos_max:
                continue
            light = light + get_lighting(o, d, hit, lpos, lrad, lcol, lint) * contribution
        for (ldir, lcol, lint, cos_eff) in lights.get("dir", []):  # :468-486
            if max(np.dot(d, -ldir), 0.0) < cos_eff:
                continue
            mod = 1.0
            if hit and i:
                mod *= hit["mat"]["transparency"] * max(np.dot(d, -hit["normal"]), 0.0)
            elif not i:
                mod = 0.0
            light = light + lcol * lint * contribution * mod
        if not hit:  # :489-493
            light = light + env_rgb * contribution
            break
        m = hit["mat"]
        metallic = 1.0 if m["metallic"] * (1 - m["roughness"]) >= rng.f() else 0.0  # :499-501, in this order
        specular = 1.0 if m["specular"] * (1 - m["roughness"]) >= rng.f() else 0.0
        transp = 1.0 if m["transparency"] >= rng.f() else 0.0
        rough = m["roughness"] * (1 - specular)
        diffuse = normalize(hit["normal"] + rng.vector())  # findReflectDirection :107-112
        refl = normalize(reflect(d, hit["normal"]) * (1 - rough) + diffuse * rough)
        refr = transparency_bounce(d, hit["normal"], m["ior"], rng)
        bounce = normalize(refl * (1 - transp) + refr * transp)
        contribution = contribution * (m["albedo"] * (1 - metallic) + metallic)  # :509, then emission with the NEW contribution (Q12)
        light = light + m["emission"] * m["power"] * contribution * (1 - transp)
        o = hit["pos"] + bounce * 0.001
        d = bounce
    return light


def forward_vec(euler_deg):
    """Transform::getForwardVec (Components.h:20-23): normalised column 2 of toMat3(quat(radians(euler))), glm's
    Euler-to-quaternion (gtc/quaternion.inl) and quaternion-to-matrix (gtc/quaternion.inl mat3_cast) formulas."""
    e = np.radians(np.array(euler_deg, float)) * 0.5
    c, s_ = np.cos(e), np.sin(e)
    w = c[0] * c[1] * c[2] + s_[0] * s_[1] * s_[2]
    x = s_[0] * c[1] * c[2] - c[0] * s_[1] * s_[2]
    y = c[0] * s_[1] * c[2] + s_[0] * c[1] * s_[2]
    z = c[0] * c[1] * s_[2] - s_[0] * s_[1] * c[2]
    return normalize(np.array([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)]))


def mat(albedo=(1, 1, 1), roughness=1.0, metallic=0.0, specular=1.0, emission=(1, 1, 1), power=0.0, transparency=0.0, ior=1.0):
    return dict(albedo=np.array(albedo, float), roughness=roughness, metallic=metallic, specular=specular, emission=np.array(emission, float), power=power,
                transparency=transparency, ior=ior)


def run_case(oracle, spheres, camera, W, H, bounces, frames=1, env=None, point=(), spot=(), dirl=()):
    """Renders the scene with the oracle through the host layer and with the Python path; returns (oracle acc, python acc)."""
    sc = host.Scene()
    for s in spheres:
        sc.add_sphere(sc.create_entity(tuple(s["pos"])), s["radius"],
                      host.Material(albedo=tuple(s["albedo"]), roughness=s["roughness"], metallic=s["metallic"], specular=s["specular"],
                                    emissionColour=tuple(s["emission"]), emissionPower=s["power"], transparency=s["transparency"], indexOfRefraction=s["ior"]))
    for (pos, col, inten, rad) in point:
        sc.add_point_light(sc.create_entity(tuple(pos)), tuple(col), inten, rad)
    for (pos, rot, col, inten, rad, max_angle) in spot:
        sc.add_spot_light(sc.create_entity(tuple(pos), tuple(rot)), tuple(col), inten, rad, max_angle)
    for (rot, col, inten, eff) in dirl:
        sc.add_dir_light(sc.create_entity(rotation=tuple(rot)), tuple(col), inten, eff)
    sc.add_camera(sc.create_entity(camera[0], camera[1]), camera[2], 0.1, 1000.0)
    rt = host.RayTracer(api=oracle.api, fns=oracle.fns)
    env_cross = None
    env_rgb = np.zeros(3)
    if env is not None:  # a constant-colour cube: any sampler returns exactly c / 255
        env_cross = np.zeros((12, 16, 4), np.uint8)
        env_cross[..., :3] = env
        env_cross[..., 3] = 255
        env_rgb = np.array(env, float) / 255.0
    rt.initiate(W, H, host.ResourceManager(), env_cross)
    rt.setScene(sc)
    rt.createOctTree(sc, 0, 50)
    rt.ctx.set_option(abi.OPT_MAX_BOUNCES, bounces)
    want = np.zeros((H, W, 3))
    for _ in range(frames):
        rt.render()
        rd = rt.render_data
        # what the host uploads for the lights (RayTracer.cpp:434-480): cos of the angles, the transform's forward vector
        lights = {"point": [(np.array(pos, float), np.array(col, float), inten, rad) for (pos, col, inten, rad) in point],
                  "spot": [(np.array(pos, float), forward_vec(rot), np.array(col, float), inten, rad, math.cos(math.radians(max_angle)))
                           for (pos, rot, col, inten, rad, max_angle) in spot],
                  "dir": [(forward_vec(rot), np.array(col, float), inten, math.cos(math.radians(eff))) for (rot, col, inten, eff) in dirl]}
        # the sphere table is filed in entt view order (reverse creation): closest-hit ties aside the order is irrelevant
        for py in range(H):
            for px in range(W):
                want[py, px] += pixel(px, py, rd, [dict(s, pos=np.array(s["pos"], float)) for s in spheres], lights, env_rgb, bounces)
    got = rt.ctx.read_accum()
    rt.close()
    return got, want


def agree(got, want, frac=0.99):
    d = np.abs(got[..., :3].astype(np.float64) - want)
    fin = np.isfinite(d).all(-1)
    ok = (d[fin].max(-1) <= 2e-4 + 2e-4 * np.abs(want[fin]).max(-1))
    assert ok.mean() >= frac, (ok.mean(), float(d[fin].max()))
    return ok.mean()


def test_two_segment_path_through_emission_and_environment(oracle):
    """A diffuse grey ball over a tinted emitter floor, constant sky. Segment 0 hits ball or floor; the floor is an emitter
    with albedo (0.5, 0.25, 1): its emission is tinted by its own albedo (Q12: contribution is multiplied first). Segment
    1 either misses (sky * contribution) or hits again: then only emission is added and the loop ends WITHOUT a sky term."""
    spheres = [dict(mat(albedo=(0.8, 0.8, 0.8)), pos=(0.0, 1.0, 6.0), radius=1.0),
               dict(mat(albedo=(0.5, 0.25, 1.0), emission=(1.0, 0.5, 0.25), power=2.0), pos=(0.0, -1000.0, 6.0), radius=1000.0)]
    got, want = run_case(oracle, spheres, ((0.0, 1.5, 0.0), (5.0, 0.0, 0.0), 70.0), 24, 16, 1, frames=2, env=(51, 102, 204))
    agree(got, want)
    floor = want[-1, 12]  # bottom-centre pixel looks at the floor: emission(1,0.5,0.25) * 2 * albedo(0.5,0.25,1) per frame, plus the bounce
    assert floor[0] >= 2 * 1.0 - 1e-9 and floor[1] >= 2 * 0.25 - 1e-9 and floor[2] >= 2 * 0.5 - 1e-9
    assert (got[..., 3] == 2).all()


@pytest.mark.parametrize("bounces", [1, 4])
def test_transparency_with_total_internal_reflection(oracle, bounces):
    """The camera sits INSIDE a glass sphere (radius 10, ior 1.5), 7 units off-centre, looking along +z: the centre ray
    meets the wall from behind with sin(theta) = 0.7 > 1 / 1.5, so `cannot_refract` holds and the path reflects — through
    the user `refract` (Q7, Q8) — and keeps bouncing inside; rays towards +x leave through the wall (sin(theta) small).
    A red emitter ball outside makes leaving paths visible. transparency = 1: no emission term ever (:510)."""
    glass = dict(mat(albedo=(1, 1, 1), roughness=0.0, specular=1.0, transparency=1.0, ior=1.5), pos=(0.0, 0.0, 0.0), radius=10.0)
    lamp = dict(mat(albedo=(1, 1, 1), emission=(1.0, 0.1, 0.1), power=3.0), pos=(30.0, 0.0, 4.0), radius=8.0)
    got, want = run_case(oracle, [glass, lamp], ((7.0, 0.0, 0.0), (0.0, 40.0, 0.0), 90.0), 24, 16, bounces, env=(30, 60, 90))
    agree(got, want)
    # the geometry of the claim above, by hand: centre ray of an unrotated camera at (7,0,0) -> wall at z = sqrt(51)
    n = np.array([7.0, 0.0, math.sqrt(51.0)]) / 10.0
    assert math.sqrt(1 - n[2] ** 2) * 1.5 > 1.0


def test_spot_light_gate_point_light_sides_and_directional_rule(oracle):
    """A ball in front of the camera; a spot light whose cone (max angle 35 deg) faces the camera (gate open on segment 0:
    the camera sees its light sphere) and whose bounce rays mostly fall outside it (gate closed: `continue`, yet six draws
    are consumed, which the RNG-synchronous comparison would expose); a point light BEHIND the ball (hit in front of the
    light: `distance < dist` -> transparency * ... = 0) and one in front; a directional light that must add nothing on
    segment 0 and only through transparency on later ones (Q13)."""
    spheres = [dict(mat(albedo=(0.9, 0.6, 0.3), roughness=0.6, specular=0.8), pos=(0.0, 0.0, 8.0), radius=1.5),
               dict(mat(albedo=(0.4, 0.4, 0.4)), pos=(0.0, -1001.5, 8.0), radius=1000.0)]
    point = [((0.0, 0.0, 14.0), (1.0, 0.2, 0.2), 3.0, 1.0),   # behind the ball for the centre rays
             ((-3.0, 2.5, 5.0), (0.2, 1.0, 0.2), 2.0, 1.2)]   # in front, visible as a disc
    # forward = -z: facing the camera. Gate on segment 0: cos = 6 / |(3, 1, 6)| = 0.885 >= cos 35 = 0.819 -> open
    spot = [((3.0, 2.0, 6.0), (0.0, 180.0, 0.0), (0.3, 0.3, 1.0), 4.0, 1.2, 35.0)]
    dirl = [((20.0, 0.0, 0.0), (1.0, 1.0, 1.0), 0.7, 25.0)]
    got, want = run_case(oracle, spheres, ((0.0, 1.0, 0.0), (0.0, 0.0, 0.0), 80.0), 24, 16, 1, frames=2, point=point, spot=spot, dirl=dirl)
    agree(got, want)
    assert want[..., 2].max() > 1.0 and want[..., 1].max() > 0.5  # the spot (blue) and the front point light (green) are seen


# ---- the same independent path with MESHES: brute-force closest hit + the shading fetch of RaytracerCS.hlsl:362-392 ----
# Every triangle of every instance is tested (no octree, no BVH: the closest hit does not depend on the visit order except
# on exact ties), with the shader's Moeller-Trumbore (E1 = p0 - p2, E2 = p1 - p2, GPU-Utilities.hlsli:117-155), the
# per-instance local ray and world-space distance (:273-276,311-314), barycentric interpolation (u -> p0, v -> p1,
# w -> p2, :368-371), M * n without inverse-transpose (Q11), normal mapping (:167-175), texture-clamped material values
# (:138-165), bilinear WRAP sampling of the RGBA8 array with texel centres at +0.5, and a cube map with one constant colour
# per face (so only the D3D face-selection table matters: +X -X +Y -Y +Z -Z by the major axis). Inputs taken from the
# host layer are the tables it uploads (t0, t1, t7, the scaled texture array), which tests/test_host_layer.py and
# test_golden.py pin against the compiled reference; everything downstream is re-derived here.
FACE_CELLS = [(1, 2), (1, 0), (0, 1), (2, 1), (1, 1), (1, 3)]  # +X -X +Y -Y +Z -Z in the 4x3 cross (RayTracer.cpp:377-399)
FACE_COLOURS = [(200, 60, 60), (60, 200, 60), (60, 60, 200), (200, 200, 60), (60, 200, 200), (200, 60, 200)]


def constant_face_cross(face=4):
    img = np.zeros((3 * face, 4 * face, 4), np.uint8)
    for (row, col), c in zip(FACE_CELLS, FACE_COLOURS):
        img[row * face:(row + 1) * face, col * face:(col + 1) * face, :3] = c
    img[..., 3] = 255
    return img


def env_face_colour(d):
    ax, ay, az = abs(d[0]), abs(d[1]), abs(d[2])
    if az >= ax and az >= ay:
        f = 4 if d[2] >= 0 else 5
    elif ay >= ax:
        f = 2 if d[1] >= 0 else 3
    else:
        f = 0 if d[0] >= 0 else 1
    return np.array(FACE_COLOURS[f], float) / 255.0


def sample_texture(tex, idx, uv):
    """Texture2DArray.SampleLevel(LINEAR, WRAP) on RGBA8 slices [n, S, S, 4]: (rgb)."""
    S = tex.shape[1]
    x, y = uv[0] * S - 0.5, uv[1] * S - 0.5
    x0, y0 = math.floor(x), math.floor(y)
    fx, fy = x - x0, y - y0
    sl = tex[int(idx)]

    def t(i, j):
        return sl[j % S, i % S, :3].astype(float) / 255.0
    top = t(x0, y0) * (1 - fx) + t(x0 + 1, y0) * fx
    bot = t(x0, y0 + 1) * (1 - fx) + t(x0 + 1, y0 + 1) * fx
    return top * (1 - fy) + bot * fy


def closest_mesh_hit(o, d, inst, tri_pos, best):
    """Brute force over all instances: returns (distance, instance, triangle, u, v) or None. Vectorised per instance."""
    out = None
    for k, m in enumerate(inst):
        lo = (m["Minv"] @ np.append(o, 1.0))[:3]
        ld = normalize((m["Minv"] @ np.append(d, 0.0))[:3])
        P = tri_pos[m["tri0"]:m["tri1"]]
        p0, p1, p2 = P[:, 0], P[:, 1], P[:, 2]
        E1, E2 = p0 - p2, p1 - p2
        c1 = np.cross(ld, E2)
        det = (E1 * c1).sum(1)
        ok = ~((det < 1e-6) & (det > -1e-6))
        inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
        rm = lo - p2
        u = (rm * c1).sum(1) * inv
        ok &= ~(u < -1e-6)
        c2 = np.cross(rm, E1)
        v = (c2 @ ld) * inv
        ok &= ~((v < -1e-6) | (u + v > 1.0))
        t = (E2 * c2).sum(1) * inv
        ok &= ~(t < -1e-6) & (t > 0)  # :306-308: distanceToHit <= 0 -> continue
        if not ok.any():
            continue
        dist = np.linalg.norm((m["M"][:3, :3] @ (ld[:, None] * t[None, :])).T, axis=1)  # |M * (localDir * t)|, w = 0
        dist = np.where(ok, dist, np.inf)
        j = int(np.argmin(dist))
        if dist[j] < best:
            best = dist[j]
            out = (best, k, m["tri0"] + j, u[j], v[j])
    return out


def mesh_pixel(px, py, rd, spheres, inst, tri_pos, tri_info, tex, lights, num_bounces):
    W, H = rd.textureDims[0], rd.textureDims[1]
    rng = Rng(px + (py + 74813) * W * (rd.numAccumulationFrames + 1))
    invP = np.array(list(rd.cameraInverseProjectionMatrix), np.float64).reshape(4, 4)
    invV = np.array(list(rd.cameraInverseViewMatrix), np.float64).reshape(4, 4)
    x = (px + rng.f() - 0.5) / W * 2 - 1
    y = ((H - py) + rng.f() - 0.5) / H * 2 - 1
    target = invP @ np.array([x, y, rd.cameraNearZ, 1.0])
    d = (invV @ np.append(normalize(target[:3] / target[2]), 0.0))[:3]
    o = np.array(list(rd.cameraPosition), np.float64)
    ang = rng.f() * 2 * PI
    jit = np.array([math.cos(ang), math.sin(ang)]) * math.sqrt(rng.f()) * rd.dofStrength
    off = np.array(list(rd.cameraRightDir)) * jit[0] + np.array(list(rd.cameraUpDir)) * jit[1]
    focus = o + d * (rd.dofDistance + rd.cameraNearZ)
    o = o + off
    d = normalize(focus - o)
    light, contribution = np.zeros(3), np.ones(3)
    for i in range(num_bounces + 1):
        hit, best = None, 3.402823466e38
        for s in spheres:
            t = ray_sphere(o, d, s["pos"], s["radius"])
            if 0 < t < best:
                best = t
                pos = o + d * t
                hit = {"distance": t, "pos": pos, "normal": normalize(pos - s["pos"]), "mat": dict(s)}
        mh = closest_mesh_hit(o, d, inst, tri_pos, best)
        if mh:
            dist, k, tri, bu, bv = mh
            m = inst[k]
            w3 = np.array([bu, bv, 1.0 - (bu + bv)])
            info = tri_info[tri]  # [3, 11]: normal 0:3, uv 3:5, tangent 5:8, bitangent 8:11
            uv = w3 @ info[:, 3:5]
            A = m["M"][:3, :3]
            n = normalize(A @ (w3 @ info[:, 0:3]))
            mat_ = dict(m["mat"])
            if mat_["normal_map"] is not None:
                tg, bt = normalize(A @ (w3 @ info[:, 5:8])), normalize(A @ (w3 @ info[:, 8:11]))
                s3 = sample_texture(tex, mat_["normal_map"], uv) * 2 - 1
                n = normalize(s3[0] * tg + s3[1] * bt + s3[2] * n)
            if mat_["albedo_tex"] is not None:
                mat_["albedo"] = sample_texture(tex, mat_["albedo_tex"], uv)
            for key in ("roughness", "metallic", "specular"):
                if mat_[key + "_tex"] is not None:
                    mat_[key] = min(max(sample_texture(tex, mat_[key + "_tex"], uv)[0], mat_[key]), 1.0)
            hit = {"distance": dist, "pos": o + d * dist, "normal": n, "mat": mat_}
        for (lpos, lcol, lint, lrad) in lights.get("point", []):
            light = light + get_lighting(o, d, hit, lpos, lrad, lcol, lint) * contribution
        for (lpos, ldir, lcol, lint, lrad, cos_max) in lights.get("spot", []):
            to_light = normalize(lpos + rng.vector() * 0.01 - o)
            if np.dot(to_light, -ldir) < cos_max:
                continue
            light = light + get_lighting(o, d, hit, lpos, lrad, lcol, lint) * contribution
        for (ldir, lcol, lint, cos_eff) in lights.get("dir", []):
            if max(np.dot(d, -ldir), 0.0) < cos_eff:
                continue
            mod = 1.0
            if hit and i:
                mod *= hit["mat"]["transparency"] * max(np.dot(d, -hit["normal"]), 0.0)
            elif not i:
                mod = 0.0
            light = light + lcol * lint * contribution * mod
        if not hit:
            light = light + env_face_colour(d) * contribution
            break
        m = hit["mat"]
        metallic = 1.0 if m["metallic"] * (1 - m["roughness"]) >= rng.f() else 0.0
        specular = 1.0 if m["specular"] * (1 - m["roughness"]) >= rng.f() else 0.0
        transp = 1.0 if m["transparency"] >= rng.f() else 0.0
        rough = m["roughness"] * (1 - specular)
        diffuse = normalize(hit["normal"] + rng.vector())
        refl = normalize(reflect(d, hit["normal"]) * (1 - rough) + diffuse * rough)
        refr = transparency_bounce(d, hit["normal"], m["ior"], rng)
        bounce = normalize(refl * (1 - transp) + refr * transp)
        contribution = contribution * (m["albedo"] * (1 - metallic) + metallic)
        light = light + m["emission"] * m["power"] * contribution * (1 - transp)
        o = hit["pos"] + bounce * 0.001
        d = bounce
    return light


def _material_from_record(rec):
    """68-byte Material (GPU-Structs.hlsli:37-71) -> the dict mesh_pixel uses."""
    f = rec.view(np.float32)
    u = rec.view(np.uint32)
    tid = lambda v: None if v == 0xFFFFFFFF else int(v)  # noqa: E731
    return dict(albedo=np.array(f[0:3], float), albedo_tex=tid(u[3]), roughness=float(f[4]), roughness_tex=tid(u[5]), metallic=float(f[6]),
                metallic_tex=tid(u[7]), specular=float(f[8]), specular_tex=tid(u[9]), normal_map=tid(u[10]), emission=np.array(f[11:14], float),
                power=float(f[14]), transparency=float(f[15]), ior=float(f[16]))


def test_mesh_scene_with_textures_normal_maps_instances_and_lights(oracle):
    """A small mixed scene — rotated, translated instances; albedo / roughness / normal-map textures; a glass blob and an
    emitter; two analytic spheres; point, spot and directional light; a cube map — through the oracle and through the
    brute-force Python path above, two frames. Exercises the shading fetch end to end, which no leaf KAT covers."""
    from conftest import scenes
    desc = scenes.small_mesh_scene(seed=3, n_objects=4, detail=4, transforms=True, with_lights=True, octree=(0, 50), tex_size=8)
    desc.env_cross = constant_face_cross(4)
    W, H, frames = 32, 18, 2
    rt, sc, res = oracle.instantiate(desc, W, H)
    tri_pos = rt.table(0).view(np.float32).reshape(-1, 3, 3).astype(np.float64)
    tri_info = rt.table(1).view(np.float32).reshape(-1, 3, 11).astype(np.float64)
    tex = rt.table(3)
    inst = []
    for rec in rt.table(7):
        f = rec[:128].view(np.float32).astype(np.float64)
        u = rec[128:136].view(np.uint32)
        inst.append(dict(M=f[:16].reshape(4, 4), Minv=f[16:32].reshape(4, 4), tri0=int(u[0]), tri1=int(u[1]), mat=_material_from_record(rec[160:228])))
    spheres = [dict(mat(**{k: v for k, v in m.items() if k in ("albedo", "roughness", "specular", "transparency")}, ior=m.get("indexOfRefraction", 1.0)),
                    pos=np.array(p, float), radius=r) for (p, r, m) in desc.spheres]
    lights = {"point": [(np.array(p, float), np.array(c, float), i, r) for (p, c, i, r) in desc.point_lights],
              "spot": [(np.array(p, float), forward_vec(rot), np.array(c, float), i, r, math.cos(math.radians(a))) for (p, rot, c, i, r, a) in desc.spot_lights],
              "dir": [(forward_vec(rot), np.array(c, float), i, math.cos(math.radians(a))) for (rot, c, i, a) in desc.dir_lights]}
    assert len(inst) == 5 and any(m["mat"]["normal_map"] is not None for m in inst) and any(m["mat"]["transparency"] == 1.0 for m in inst)
    want = np.zeros((H, W, 3))
    for _ in range(frames):
        rt.render()
        rd = rt.render_data
        for py in range(H):
            for px in range(W):
                want[py, px] += mesh_pixel(px, py, rd, spheres, inst, tri_pos, tri_info, tex, lights, 1)
    got = rt.ctx.read_accum()
    rt.close()
    agree(got, want)

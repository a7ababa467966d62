"""Seeded procedural stand-ins for the benchmark scenes.

The assets the reference's configs name are not redistributable / not in the container
(reference .MISSING_LARGE_BLOBS:1-14: sponza.obj, dragon.obj, rev_*.png, ...), so every config
runs on a deterministic procedural scene of matching size and structure:

  config 1  test_scene_spheres   4 analytic spheres (values from Pics/before.PNG's ImGui panel)
  config 2  dragon_proxy         (2,3) torus-knot tube with 3-octave value-noise displacement,
                                 80 000 or 871 414 triangles, on a 2-triangle ground quad
  config 3  glass_revolver_proxy double-walled glass of revolution + textured "revolver" blob, IBL
  config 4  sponza_proxy         atrium: 262 267 triangles in 390 mesh entities, 26 textures

Each builder returns a `SceneDescription`; `instantiate()` turns one into ResourceManager / Scene /
RayTracer objects of the host layer (any device backend).
"""
from dataclasses import dataclass, field

import os

import numpy as np

F32 = np.float32


# ---- noise ------------------------------------------------------------------------------------
def _hash3(ix, iy, iz, seed):
    h = (ix.astype(np.uint32) * np.uint32(0x8DA6B343)) ^ (iy.astype(np.uint32) * np.uint32(0xD8163841)) ^ \
        (iz.astype(np.uint32) * np.uint32(0xCB1AB31F)) ^ np.uint32(seed)
    h ^= h >> np.uint32(13)
    h *= np.uint32(0x5BD1E995)
    h ^= h >> np.uint32(15)
    return (h & np.uint32(0xFFFFFF)).astype(np.float64) / float(0x1000000)


def value_noise3(p, seed):
    """Trilinear value noise in [0,1) at points p[...,3] (float64)."""
    f = np.floor(p)
    t = p - f
    t = t * t * (3.0 - 2.0 * t)
    i = f.astype(np.int64)
    out = 0.0
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                w = (t[..., 0] if dx else 1 - t[..., 0]) * (t[..., 1] if dy else 1 - t[..., 1]) * (t[..., 2] if dz else 1 - t[..., 2])
                out = out + w * _hash3(i[..., 0] + dx, i[..., 1] + dy, i[..., 2] + dz, seed)
    return out


def fbm3(p, seed, octaves=3):
    a, s, out = 0.5, 1.0, 0.0
    for o in range(octaves):
        out = out + a * value_noise3(p * s, seed + 101 * o)
        a *= 0.5
        s *= 2.0
    return out


# ---- mesh helpers -----------------------------------------------------------------------------
def _normalize(v):
    n = np.linalg.norm(v, axis=-1, keepdims=True)
    return v / np.where(n > 0, n, 1.0)


def grid_surface(pos_fn, nu, nv, uv_scale=(1.0, 1.0), flip=False):
    """Triangulate P(u,v), (u,v) in [0,1]^2, on an nu x nv cell lattice: 2*nu*nv triangles.

    Returns (positions [N,3,3] f32, info [N,33] f32) with per-vertex normal / uv / tangent /
    bitangent (TriangleInfo layout, GPU-Structs.hlsli:100-111)."""
    u = np.linspace(0.0, 1.0, nu + 1)
    v = np.linspace(0.0, 1.0, nv + 1)
    U, V = np.meshgrid(u, v, indexing="ij")
    P = pos_fn(U, V)
    e = 1e-4
    Tu = (pos_fn(np.clip(U + e, 0, 1), V) - pos_fn(np.clip(U - e, 0, 1), V))
    Tv = (pos_fn(U, np.clip(V + e, 0, 1)) - pos_fn(U, np.clip(V - e, 0, 1)))
    Tu, Tv = _normalize(Tu), _normalize(Tv)
    Nn = _normalize(np.cross(Tu, Tv))
    if flip:
        Nn = -Nn
    UV = np.stack([U * uv_scale[0], V * uv_scale[1]], axis=-1)
    vert = np.concatenate([Nn, UV, Tu, Tv], axis=-1)  # [nu+1, nv+1, 11]

    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    i, j = i.ravel(), j.ravel()
    quads = [(i, j), (i + 1, j), (i + 1, j + 1), (i, j + 1)]
    tri_idx = [(0, 1, 2), (0, 2, 3)]
    # (vertices are rounded to f32 BEFORE the gather — elementwise, so the result is the same bits as rounding the gathered f64
    # arrays — and gathered by flat index: 8 M triangles in 10 s instead of 70)
    Pf, vertf = P.astype(F32).reshape(-1, 3), vert.astype(F32).reshape(-1, 11)
    pos = np.empty((2 * nu * nv, 3, 3), dtype=F32)
    info = np.empty((2 * nu * nv, 3, 11), dtype=F32)
    for t, corners in enumerate(tri_idx):
        for k, c in enumerate(corners):
            ci, cj = quads[c]
            flat = ci * (nv + 1) + cj
            pos[t::2, k] = Pf[flat]
            info[t::2, k] = vertf[flat]
    return pos, info.reshape(-1, 33)


def box_mesh(lo, hi, div=1, uv_scale=1.0):
    """Axis-aligned box as 6 tessellated faces: 12*div*div triangles."""
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    ps, infos = [], []
    for axis in range(3):
        a, b = (axis + 1) % 3, (axis + 2) % 3
        for side in (0, 1):
            def fn(U, V, axis=axis, a=a, b=b, side=side):
                P = np.empty(U.shape + (3,))
                P[..., axis] = hi[axis] if side else lo[axis]
                P[..., a] = lo[a] + U * (hi[a] - lo[a])
                P[..., b] = lo[b] + V * (hi[b] - lo[b])
                return P
            p, i = grid_surface(fn, div, div, (uv_scale, uv_scale), flip=(side == 0))
            ps.append(p)
            infos.append(i)
    return np.concatenate(ps), np.concatenate(infos)


def revolve(profile_fn, nu, nv, centre=(0, 0, 0), uv_scale=(1.0, 1.0), flip=False):
    """Surface of revolution about +Y: profile_fn(v) -> (radius, height)."""
    c = np.asarray(centre, float)

    def fn(U, V):
        r, h = profile_fn(V)
        ang = U * 2 * np.pi
        return np.stack([c[0] + r * np.cos(ang), c[1] + h, c[2] + r * np.sin(ang)], axis=-1)
    return grid_surface(fn, nu, nv, uv_scale, flip)


# ---- textures / environment --------------------------------------------------------------------
def procedural_texture(size, seed, kind="stone"):
    """RGBA8 [size,size,4]; built at low resolution and upsampled so 26 x 1024^2 stays cheap."""
    rng = np.random.default_rng(seed)
    base = 128
    y, x = np.meshgrid(np.arange(base), np.arange(base), indexing="ij")
    p = np.stack([x / 8.0, y / 8.0, np.full_like(x, seed % 17, dtype=float)], axis=-1)
    n = fbm3(p, seed, 3)
    tint = rng.uniform(0.35, 1.0, 3)
    if kind == "brick":
        bx, by = (x // 16 + (y // 8) % 2 * 0.5) % 1, (y % 8) / 8.0
        mortar = ((x + (y // 8 % 2) * 8) % 16 < 1) | (y % 8 < 1)
        v = np.where(mortar, 0.75, 0.45 + 0.4 * n)
    elif kind == "fabric":
        v = 0.55 + 0.25 * np.sin(x * 0.8) * np.sin(y * 0.8) + 0.2 * n
    elif kind == "normal":
        gx = np.gradient(n, axis=1) * 6.0
        gy = np.gradient(n, axis=0) * 6.0
        nz = 1.0 / np.sqrt(gx * gx + gy * gy + 1.0)
        rgb = np.stack([-gx * nz * 0.5 + 0.5, -gy * nz * 0.5 + 0.5, nz * 0.5 + 0.5], axis=-1)
        img = np.clip(rgb * 255.0 + 0.5, 0, 255).astype(np.uint8)
        img = np.concatenate([img, np.full((base, base, 1), 255, np.uint8)], axis=-1)
        rep = size // base
        return np.ascontiguousarray(np.kron(img, np.ones((rep, rep, 1), dtype=np.uint8))) if rep >= 1 else img[:size, :size]
    elif kind == "scalar":
        v = np.clip(0.15 + 0.8 * n, 0, 1)
        tint = np.ones(3)
    else:
        v = 0.4 + 0.6 * n
    rgb = np.clip(v[..., None] * tint[None, None, :], 0, 1)
    img = np.clip(rgb * 255.0 + 0.5, 0, 255).astype(np.uint8)
    img = np.concatenate([img, np.full((base, base, 1), 255, np.uint8)], axis=-1)
    rep = max(size // base, 1)
    out = np.kron(img, np.ones((rep, rep, 1), dtype=np.uint8))
    # break the blockiness with a per-texel dither so bilinear weights matter
    dither = rng.integers(-6, 7, size=(out.shape[0], out.shape[1], 1), dtype=np.int16)
    out = np.clip(out.astype(np.int16) + np.concatenate([dither] * 3 + [np.zeros_like(dither)], axis=-1), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(out[:size, :size])


_FACE_CELLS = [(1, 2), (1, 0), (0, 1), (2, 1), (1, 1), (1, 3)]  # +X -X +Y -Y +Z -Z in the 4x3 cross (RayTracer.cpp:377-399)


def sky_cross(face, seed=0x5EED5):
    """A 4F x 3F RGBA8 cross image of a procedural sky (the input format of loadEnvironmentMap)."""
    img = np.zeros((3 * face, 4 * face, 4), dtype=np.uint8)
    t = (np.arange(face) + 0.5) / face * 2 - 1
    sc, tc = np.meshgrid(t, t, indexing="xy")  # sc along x (columns), tc along y (rows)
    sun = _normalize(np.array([0.35, 0.8, 0.45]))
    for f, (row, col) in enumerate(_FACE_CELLS):
        one = np.ones_like(sc)
        # inverse of the D3D face table: +X(sc=-z,tc=-y) -X(+z,-y) +Y(+x,+z) -Y(+x,-z) +Z(+x,-y) -Z(-x,-y)
        d = [np.stack([one, -tc, -sc], -1), np.stack([-one, -tc, sc], -1), np.stack([sc, one, tc], -1),
             np.stack([sc, -one, -tc], -1), np.stack([sc, -tc, one], -1), np.stack([-sc, -tc, -one], -1)][f]
        d = _normalize(d)
        up = d[..., 1]
        sky = np.stack([0.35 + 0.25 * (1 - up), 0.55 + 0.2 * (1 - up), 0.95 * np.ones_like(up)], -1)
        ground = np.stack([0.42 + 0 * up, 0.40 + 0 * up, 0.36 + 0 * up], -1)
        clouds = fbm3(d * 3.0 + 7.0, seed, 3)[..., None]
        col_ = np.where(up[..., None] > 0, sky * (0.75 + 0.5 * clouds), ground * (0.8 + 0.4 * clouds))
        s = np.clip((d @ sun - 0.995) / 0.005, 0, 1)[..., None]
        col_ = np.clip(col_ + s, 0, 1)
        rgba = np.concatenate([np.clip(col_ * 255 + 0.5, 0, 255).astype(np.uint8), np.full(up.shape + (1,), 255, np.uint8)], -1)
        img[row * face:(row + 1) * face, col * face:(col + 1) * face] = rgba
    return img


# ---- scene descriptions --------------------------------------------------------------------------
@dataclass
class MeshEntity:
    mesh: int
    material: dict = field(default_factory=dict)
    position: tuple = (0, 0, 0)
    rotation: tuple = (0, 0, 0)
    scale: tuple = (1, 1, 1)


@dataclass
class SceneDescription:
    name: str
    meshes: list = field(default_factory=list)          # [(positions, info)]
    textures: list = field(default_factory=list)        # [rgba8 HxWx4]
    entities: list = field(default_factory=list)        # [MeshEntity]
    spheres: list = field(default_factory=list)         # [(position, radius, material dict)]
    dir_lights: list = field(default_factory=list)      # [(rotation, colour, intensity, effective_angle)]
    point_lights: list = field(default_factory=list)    # [(position, colour, intensity, radius)]
    spot_lights: list = field(default_factory=list)     # [(position, rotation, colour, intensity, radius, max_angle)]
    env_cross: object = None
    camera: dict = field(default_factory=lambda: dict(position=(0, 0, 0), rotation=(0, 0, 0), fov=90.0, near_z=0.1, far_z=1000.0))
    bvh: tuple = (30, 5)      # RayTracer.cpp:97
    octree: tuple = (2, 50)   # Application.h:30-31

    @property
    def num_triangles(self):
        return int(sum(len(p) for p, _ in self.meshes))


def test_scene_spheres():
    """Config 1: values read from the ImGui panel of Pics/before.PNG, Y negated."""
    s = SceneDescription("TestScene")
    base = dict(roughness=1.0, metallic=0.0, specular=1.0, emissionPower=0.0, transparency=0.0, indexOfRefraction=1.0)
    s.spheres = [
        ((-81.3, -1109.9, 18.0), 1109.8, dict(base, albedo=(255 / 255, 179 / 255, 128 / 255))),
        ((-4.0, 0.4, 20.0), 3.2, dict(base, albedo=(1, 1, 1))),
        ((105.5, 71.6, 118.0), 81.0, dict(base, albedo=(1, 1, 1), emissionColour=(223 / 255, 168 / 255, 81 / 255), emissionPower=4.95)),
        ((106.0, 700.6, -290.1), 680.0, dict(base, albedo=(0, 0, 0))),
    ]
    s.camera = dict(position=(1.2, 4.1, 10.1), rotation=(19.4, -10.9, 0.0), fov=90.0, near_z=0.1, far_z=1000.0)
    return s


def _knot(U, scale=1.0):
    t = U * 2 * np.pi
    r = 2.0 + np.cos(3 * t)
    return np.stack([r * np.cos(2 * t), np.sin(3 * t) * 1.2, r * np.sin(2 * t)], axis=-1) * scale


def dragon_proxy(num_triangles=80000, seed=0xD5A60, size=20.0):
    """Config 2: a displaced (2,3) torus-knot tube (2*nu*nv triangles) on a ground quad."""
    nv = {80000: 20, 871414: 41, 8000000: 128}.get(num_triangles)
    if nv is None:
        nv = 16
        while num_triangles % (2 * nv):
            nv += 1
    nu = num_triangles // (2 * nv)
    assert 2 * nu * nv == num_triangles

    def fn(U, V):
        c = _knot(U)
        e = 1e-3
        tang = _normalize(_knot(U + e) - _knot(U - e))
        ref = np.broadcast_to(np.array([0.0, 1.0, 0.0]), tang.shape)
        n1 = _normalize(np.cross(tang, ref))
        n2 = np.cross(tang, n1)
        ang = V * 2 * np.pi
        radial = n1 * np.cos(ang)[..., None] + n2 * np.sin(ang)[..., None]
        # periodic 3-octave value-noise displacement of the tube radius (sampled on the unit circle in u and v)
        q = np.stack([np.cos(U * 2 * np.pi) * 6 + np.cos(ang) * 2, np.sin(U * 2 * np.pi) * 6 + np.sin(ang) * 2, np.sin(U * 4 * np.pi) * 3], -1)
        r = 0.45 + 0.35 * fbm3(q, seed, 3)
        return (c + radial * r[..., None]) * (size / 3.5)

    # OKRT_SCENE_CACHE=<dir>: the generated mesh is kept on disk (profiling runs start the same bench command five times; the
    # 8 M-triangle mesh takes over a minute to generate)
    cache = os.environ.get("OKRT_SCENE_CACHE")
    files = [os.path.join(cache, f"dragon_{num_triangles}_{seed:x}_{size}_{k}.npy") for k in ("pos", "info")] if cache else []
    if files and all(os.path.exists(f) for f in files):
        pos, info = np.load(files[0]), np.load(files[1])
    else:
        pos, info = grid_surface(fn, nu, nv, (64.0, 2.0))
        if files:
            os.makedirs(cache, exist_ok=True)
            for f, a in zip(files, (pos, info)):
                np.save(f + ".tmp.npy", a)
                os.replace(f + ".tmp.npy", f)
    s = SceneDescription(f"dragon-proxy-{num_triangles}")
    s.meshes.append((pos, info))
    lo = float(pos[..., 1].min())
    ext = size * 4
    ground = np.array([[[-ext, lo, -ext], [ext, lo, -ext], [ext, lo, ext]], [[-ext, lo, -ext], [ext, lo, ext], [-ext, lo, ext]]], dtype=F32)
    ginfo = np.zeros((2, 3, 11), F32)
    ginfo[..., 1] = 1.0     # normal +Y
    ginfo[..., 5] = 1.0     # tangent +X
    ginfo[..., 10] = 1.0    # bitangent +Z
    ginfo[0, :, 3:5] = [[0, 0], [1, 0], [1, 1]]
    ginfo[1, :, 3:5] = [[0, 0], [1, 1], [0, 1]]
    s.meshes.append((ground, ginfo.reshape(2, 33)))
    mat = dict(albedo=(0.8, 0.45, 0.2), roughness=1.0, metallic=0.0, specular=0.0, transparency=0.0)
    s.entities = [MeshEntity(0, mat), MeshEntity(1, dict(albedo=(0.6, 0.6, 0.6), roughness=1.0, specular=0.0))]
    s.env_cross = sky_cross(512)
    radius = float(np.abs(pos).max())
    s.camera = dict(position=(1.5 * radius, 0.35 * radius, 0.0), rotation=(12.0, -90.0, 0.0), fov=90.0, near_z=0.1, far_z=1000.0)
    return s


def glass_revolver_proxy(tex_size=1024, env_face=1024, detail=1.0):
    """Config 3: double-walled glass of revolution (~10k tris) + textured blob (~40k tris) + IBL."""
    s = SceneDescription("glass-revolver-proxy")

    def outer(V):
        h = V * 9.0
        r = 2.2 + 0.9 * np.sin(V * 2.6) - 0.5 * V
        return np.where(V < 0.04, V / 0.04 * r, r), h

    def inner(V):
        h = 0.35 + V * 8.65
        r = 1.95 + 0.9 * np.sin(V * 2.6) - 0.5 * V
        return np.where(V < 0.04, V / 0.04 * r, r), h

    nu, nv = int(64 * detail), int(40 * detail)
    po, io = revolve(outer, nu, nv, centre=(-6, -4, 0))
    pi_, ii = revolve(inner, nu, nv, centre=(-6, -4, 0), flip=True)
    s.meshes.append((np.concatenate([po, pi_]), np.concatenate([io, ii])))

    def blob(U, V):
        th, ph = U * 2 * np.pi, V * np.pi
        d = np.stack([np.sin(ph) * np.cos(th), np.cos(ph), np.sin(ph) * np.sin(th)], -1)
        r = 3.0 + 1.6 * fbm3(d * 2.0 + 3.0, 0xBEEF, 3) + 0.8 * np.abs(np.sin(3 * th)) * np.sin(ph)
        return d * r[..., None] * np.array([1.8, 0.8, 0.7]) + np.array([5.0, -1.0, 0.0])

    nbu, nbv = int(200 * detail), int(100 * detail)
    s.meshes.append(grid_surface(blob, nbu, nbv, (4.0, 2.0)))
    fl = -4.0

    def floor(U, V):
        return np.stack([(U - 0.5) * 80, np.full_like(U, fl), (V - 0.5) * 80], -1)
    s.meshes.append(grid_surface(floor, 8, 8, (8.0, 8.0), flip=True))
    s.textures = [procedural_texture(tex_size, 11, "stone"), procedural_texture(tex_size, 12, "scalar"),
                  procedural_texture(tex_size, 13, "scalar"), procedural_texture(tex_size, 14, "normal")]
    s.entities = [
        MeshEntity(0, dict(albedo=(1, 1, 1), roughness=0.0, specular=1.0, transparency=1.0, indexOfRefraction=1.5)),
        # albedo/roughness/metallic/normal = slices 0-3, as wired in Application.cpp:36-40,79-82
        MeshEntity(1, dict(albedo=(1, 1, 1), albedoTextureId=0, roughness=0.1, roughnessTextureId=1, metallic=0.2, metallicTextureId=2, normalMapIdx=3, specular=1.0)),
        MeshEntity(2, dict(albedo=(0.7, 0.7, 0.72), roughness=0.6, specular=1.0)),
    ]
    s.env_cross = sky_cross(env_face)
    s.camera = dict(position=(0.0, 3.0, -22.0), rotation=(8.0, 0.0, 0.0), fov=70.0, near_z=0.1, far_z=1000.0)
    return s


SPONZA_TRIANGLES = 262267
SPONZA_MESHES = 390
SPONZA_TEXTURES = 26


def sponza_proxy(tex_size=1024, env_face=1024, tri_scale=1.0):
    """Config 4: procedural atrium, 262 267 triangles in 390 mesh entities (identity transforms,
    geometry baked in world space like Application.cpp:44 bakes the 0.07 scale), 26 textures.

    `tri_scale` < 1 thins the tessellation for quick tests (mesh/texture counts stay)."""
    s = SceneDescription("sponza-proxy")
    meshes = []   # (pos, info, texture id, kind)
    X0, X1, Y0, Y1, Z0, Z1 = -130.0, 125.0, -12.0, 95.0, -80.0, 75.0

    def d(n):
        return max(1, int(round(n * tri_scale ** 0.5)))

    def add(p_i, tex, extra=None):
        meshes.append((p_i[0], p_i[1], tex, extra or {}))

    def relief(fn_flat, normal_axis, amp, seed):
        def fn(U, V):
            P = fn_flat(U, V)
            P[..., normal_axis] += amp * (fbm3(np.stack([P[..., 0], P[..., 1], P[..., 2]], -1) * 0.25, seed, 3) - 0.5)
            return P
        return fn

    # floor (1) + ceiling ring (4) + outer walls (4 x 2 storeys = 8)
    add(grid_surface(relief(lambda U, V: np.stack([X0 + U * (X1 - X0), np.full_like(U, Y0), Z0 + V * (Z1 - Z0)], -1), 1, 0.4, 1), d(120), d(75), (24, 16), flip=True), 0)
    for k, (xa, xb, za, zb) in enumerate([(X0, X1, Z0, Z0 + 30), (X0, X1, Z1 - 30, Z1), (X0, X0 + 30, Z0 + 30, Z1 - 30), (X1 - 30, X1, Z0 + 30, Z1 - 30)]):
        add(grid_surface(lambda U, V, xa=xa, xb=xb, za=za, zb=zb: np.stack([xa + U * (xb - xa), np.full_like(U, Y1), za + V * (zb - za)], -1), d(40), d(24), (8, 4)), 1)
    for storey in range(2):
        ya, yb = (Y0, 40.0) if storey == 0 else (40.0, Y1)
        for k in range(4):
            if k < 2:
                z = Z0 if k == 0 else Z1
                fn = relief(lambda U, V, z=z, ya=ya, yb=yb: np.stack([X0 + U * (X1 - X0), ya + V * (yb - ya), np.full_like(U, z)], -1), 2, 1.2, 10 + k)
                add(grid_surface(fn, d(100), d(24), (16, 4), flip=(k == 0)), 2 + storey)
            else:
                x = X0 if k == 2 else X1
                fn = relief(lambda U, V, x=x, ya=ya, yb=yb: np.stack([np.full_like(U, x), ya + V * (yb - ya), Z0 + U * (Z1 - Z0)], -1), 0, 1.2, 10 + k)
                add(grid_surface(fn, d(100), d(16), (12, 4), flip=(k == 3)), 2 + storey)

    # arcades: 2 sides x 2 storeys x 14 columns, each column = base + shaft + capital (168 meshes)
    ncol = 14
    col_x = np.linspace(X0 + 22, X1 - 22, ncol)
    for side, zc in enumerate((-34.0, 30.0)):
        for storey in range(2):
            yb0 = Y0 if storey == 0 else 42.0
            hgt = 40.0 if storey == 0 else 34.0
            for cx in col_x:
                add(box_mesh((cx - 3.2, yb0, zc - 3.2), (cx + 3.2, yb0 + 2.5, zc + 3.2), d(3)), 4)
                shaft = lambda V, hgt=hgt: (2.1 - 0.35 * V + 0.12 * np.cos(V * 40), 2.5 + V * (hgt - 6.0))
                add(revolve(shaft, d(24), d(14), centre=(cx, yb0, zc), uv_scale=(2, 6)), 5 + storey)
                cap = lambda V: (1.8 + 1.6 * V ** 2, V * 3.5)
                add(revolve(cap, d(24), d(6), centre=(cx, yb0 + hgt - 3.5, zc), uv_scale=(2, 1)), 7)
    # arches between columns: 2 sides x 2 storeys x 13 (52 meshes)
    for side, zc in enumerate((-34.0, 30.0)):
        for storey in range(2):
            ytop = (Y0 + 40.0) if storey == 0 else 76.0
            for k in range(ncol - 1):
                xa, xb = col_x[k], col_x[k + 1]

                def arch(U, V, xa=xa, xb=xb, ytop=ytop, zc=zc):
                    ang = U * np.pi
                    r = (xb - xa) * 0.5 - 1.0
                    cxm = (xa + xb) * 0.5
                    return np.stack([cxm - r * np.cos(ang), ytop - 9.0 + r * 0.55 * np.sin(ang), zc - 3.0 + V * 6.0], -1)
                add(grid_surface(arch, d(34), d(8), (4, 1)), 8 + storey)
    # galleries' floor slabs (2) + balustrades (2 sides x 13 = 26)
    for side, (za, zb) in enumerate(((Z0, -30.0), (26.0, Z1))):
        add(box_mesh((X0, 40.0, za), (X1, 42.0, zb), d(10), 6.0), 10)
    for side, zc in enumerate((-31.0, 27.0)):
        for k in range(ncol - 1):
            xa, xb = col_x[k] + 2.5, col_x[k + 1] - 2.5
            add(box_mesh((xa, 42.0, zc - 0.6), (xb, 47.0, zc + 0.6), d(4), 2.0), 11)
    # hanging cloth banners: 12 high-resolution meshes
    ban_x = np.linspace(X0 + 40, X1 - 40, 6)
    for side, zc in enumerate((-27.0, 23.0)):
        for k, bx in enumerate(ban_x):
            def cloth(U, V, bx=bx, zc=zc, k=k):
                x = bx - 7 + U * 14
                y = 74.0 - V * 44.0
                z = zc + 1.5 * np.sin(U * 9 + k) * V + 2.5 * np.sin(V * 5 + U * 3 + k) * V
                return np.stack([x, y, z], -1)
            add(grid_surface(cloth, d(24), d(60), (1, 3)), 12 + (k % 3), dict(kind="fabric"))
    # long curtains on the upper gallery: 10
    for side, zc in enumerate((-36.5, 32.5)):
        for k in range(5):
            xa = X0 + 30 + k * 40.0

            def curtain(U, V, xa=xa, zc=zc, k=k):
                x = xa + U * 30
                y = 78.0 - V * 34.0
                z = zc + 1.2 * np.sin(U * 40 + k) * (0.3 + V)
                return np.stack([x, y, z], -1)
            add(grid_surface(curtain, d(60), d(24), (3, 3)), 15 + (k % 3), dict(kind="fabric"))
    # ornaments: lion-head blobs (2), vases (16 = body) with plants (16), roof beams (12), lamps (8)
    for k, x in enumerate((X0 + 4.0, X1 - 4.0)):
        def head(U, V, x=x, k=k):
            th, ph = U * 2 * np.pi, V * np.pi
            dd = np.stack([np.sin(ph) * np.cos(th), np.cos(ph), np.sin(ph) * np.sin(th)], -1)
            r = 6.0 + 3.0 * fbm3(dd * 2.5 + k * 5.0, 77 + k, 3)
            return dd * r[..., None] + np.array([x, 22.0, -2.0])
        add(grid_surface(head, d(80), d(40), (2, 1)), 18)
    vase_pos = [(x, z) for x in np.linspace(X0 + 35, X1 - 35, 8) for z in (-18.0, 14.0)]
    for (vx, vz) in vase_pos:
        body = lambda V: (2.0 + 1.8 * np.sin(V * 3.0) * (1 - 0.5 * V), V * 9.0)
        add(revolve(body, d(24), d(12), centre=(vx, Y0, vz), uv_scale=(2, 1)), 19)

        def plant(U, V, vx=vx, vz=vz):
            th, ph = U * 2 * np.pi, V * np.pi
            dd = np.stack([np.sin(ph) * np.cos(th), np.cos(ph), np.sin(ph) * np.sin(th)], -1)
            r = 3.5 + 2.5 * fbm3(dd * 4.0 + vx, 31, 3)
            return dd * r[..., None] + np.array([vx, Y0 + 13.0, vz])
        add(grid_surface(plant, d(28), d(14), (3, 2)), 20)
    for k, x in enumerate(np.linspace(X0 + 20, X1 - 20, 12)):
        add(box_mesh((x - 1.5, Y1 - 6.0, -30.0), (x + 1.5, Y1 - 3.0, 26.0), d(6), 4.0), 21)
    for k, x in enumerate(np.linspace(X0 + 50, X1 - 50, 4)):
        for zc in (-24.0, 20.0):
            lamp = lambda V: (0.3 + 1.6 * np.sin(V * np.pi) ** 2, V * 5.0)
            add(revolve(lamp, d(20), d(10), centre=(x, 50.0, zc)), 22)
    # door frames / plinths to reach the entity count
    k = 0
    while len(meshes) < SPONZA_MESHES - 1:
        x = X0 + 12 + (k % 30) * 8.0
        z = (-60.0, 55.0)[(k // 30) % 2]
        y = Y0 + (k // 60) * 44.0
        add(box_mesh((x, y, z), (x + 5.0, y + 3.0 + (k % 5), z + 5.0), d(2), 1.0), 4 + (k % 3))
        k += 1
    # filler strip so the total is exactly 262 267 triangles at tri_scale 1 (a low wall along +x)
    have = sum(len(m[0]) for m in meshes)
    target = int(SPONZA_TRIANGLES * tri_scale) if tri_scale != 1.0 else SPONZA_TRIANGLES
    remain = max(target - have, 2)
    nu = remain // 2
    strip, sinfo = grid_surface(lambda U, V: np.stack([X0 + 5 + U * (X1 - X0 - 10), Y0 + 0.2 + V * 3.0, np.full_like(U, Z1 - 8.0)], -1), nu, 1, (64, 1))
    if remain % 2:  # odd remainder: append one more triangle
        strip = np.concatenate([strip, strip[-1:] + np.array([0, 3.2, 0], F32)])
        sinfo = np.concatenate([sinfo, sinfo[-1:]])
    add((strip, sinfo), 23)
    assert len(meshes) == SPONZA_MESHES, len(meshes)

    kinds = ["stone", "stone", "brick", "brick", "stone", "stone", "stone", "stone", "brick", "brick", "stone", "stone",
             "fabric", "fabric", "fabric", "fabric", "fabric", "fabric", "stone", "stone", "stone", "stone", "stone", "brick",
             "scalar", "scalar"]
    assert len(kinds) == SPONZA_TEXTURES
    s.textures = [procedural_texture(tex_size, 100 + i, kd) for i, kd in enumerate(kinds)]
    for i, (p, inf, tex, extra) in enumerate(meshes):
        s.meshes.append((p, inf))
        mat = dict(albedo=(1, 1, 1), albedoTextureId=tex)  # defaults roughness 1 / specular 1 (Components.h:45-47)
        if tex in (12, 15, 19):  # three materials carry a `specular` slice, like map_Ks in sponza.mtl
            mat["specularTextureId"] = 24 + (tex % 2)
        s.entities.append(MeshEntity(i, mat))
    s.env_cross = sky_cross(env_face)
    s.camera = dict(position=(60.0, 0.0, 0.0), rotation=(0.0, -90.0, 0.0), fov=90.0, near_z=0.1, far_z=1000.0)  # Application.cpp:67-70
    return s


def small_mesh_scene(seed=3, n_objects=6, detail=10, with_textures=True, with_lights=False, transforms=False, tex_size=32, octree=(2, 2),
                     scales=None, bake_scale=False):
    """A tiny mixed scene for parity tests: blobs, a glass ball, a floor, spheres, optional lights.

    `scales` (with transforms=True): per-object Transform.scale, cycled over the blobs — a float for a uniform scale or a
    3-tuple (Components.h:7-23; quirk Q3). `bake_scale=True` multiplies the same factors into the mesh vertices instead
    and leaves the entity at unit scale: the same world-space geometry without the reference's local-vs-world cull."""
    rng = np.random.default_rng(seed)
    s = SceneDescription(f"small-{seed}")
    if with_textures:
        s.textures = [procedural_texture(tex_size, 5, "stone"), procedural_texture(tex_size * 2, 6, "scalar"),
                      procedural_texture(tex_size, 7, "normal"), procedural_texture(tex_size // 2, 8, "brick")]
    for k in range(n_objects):
        c = rng.uniform(-8, 8, 3) * np.array([1, 0.4, 1])

        def blob(U, V, c=c, k=k):
            th, ph = U * 2 * np.pi, V * np.pi
            dd = np.stack([np.sin(ph) * np.cos(th), np.cos(ph), np.sin(ph) * np.sin(th)], -1)
            r = 1.5 + 1.0 * fbm3(dd * 2.0 + k, 50 + k, 2)
            base = np.zeros(3) if transforms else c
            return dd * r[..., None] + base
        sc3 = (1.0, 1.0, 1.0)
        if transforms and scales:
            v = scales[k % len(scales)]
            sc3 = tuple(float(x) for x in v) if isinstance(v, (tuple, list)) else (float(v),) * 3
        if bake_scale:
            def blob_scaled(U, V, blob=blob, sc3=sc3):
                return blob(U, V) * np.asarray(sc3)
            s.meshes.append(grid_surface(blob_scaled, 2 * detail, detail, (2, 1)))
        else:
            s.meshes.append(grid_surface(blob, 2 * detail, detail, (2, 1)))
        mat = dict(albedo=tuple(rng.uniform(0.2, 1.0, 3)), roughness=float(rng.uniform(0, 1)), metallic=float(rng.uniform(0, 1)),
                   specular=float(rng.uniform(0, 1)))
        if with_textures and k % 2 == 0:
            mat.update(albedoTextureId=k % 4, roughnessTextureId=1, normalMapIdx=2)
        if k == 1:
            mat.update(transparency=1.0, indexOfRefraction=1.5, roughness=0.0)
        if k == 2:
            mat.update(emissionColour=(1.0, 0.8, 0.5), emissionPower=3.0)
        ent = MeshEntity(k, mat)
        if transforms:
            ent.position = tuple(c)
            ent.rotation = tuple(rng.uniform(-40, 40, 3))
            if not bake_scale:
                ent.scale = sc3
        s.entities.append(ent)

    def floor(U, V):
        return np.stack([(U - 0.5) * 40, np.full_like(U, -4.0), (V - 0.5) * 40], -1)
    s.meshes.append(grid_surface(floor, 4, 4, (4, 4), flip=True))
    s.entities.append(MeshEntity(n_objects, dict(albedo=(0.7, 0.7, 0.7), albedoTextureId=3 if with_textures else 0xFFFFFFFF)))
    s.spheres = [((0.0, 6.0, 4.0), 2.0, dict(albedo=(0.9, 0.3, 0.3), roughness=0.3, specular=1.0)),
                 ((-6.0, 1.0, -3.0), 1.5, dict(albedo=(1, 1, 1), transparency=1.0, indexOfRefraction=1.4, roughness=0.0))]
    if with_lights:
        s.point_lights = [((4.0, 8.0, -2.0), (1.0, 0.9, 0.8), 2.0, 1.0)]
        s.spot_lights = [((-5.0, 9.0, 5.0), (60.0, 30.0, 0.0), (0.6, 0.7, 1.0), 3.0, 1.2, 75.0)]
        s.dir_lights = [((35.0, 20.0, 0.0), (1.0, 1.0, 0.9), 0.5, 30.0)]
    s.env_cross = sky_cross(16)
    s.camera = dict(position=(0.0, 3.0, -18.0), rotation=(6.0, 0.0, 0.0), fov=80.0, near_z=0.1, far_z=1000.0)
    s.octree = octree
    return s


# ---- instantiate on the host layer ----------------------------------------------------------------
def instantiate(desc, width, height, device=0, api=None, fns=None, host=None):
    """Build ResourceManager / Scene / RayTracer for `desc`. Returns (raytracer, scene, resources)."""
    if host is None:
        from . import host as host_mod
    else:
        host_mod = host
    res = host_mod.ResourceManager()
    for pos, info in desc.meshes:
        res.add_mesh(pos, info)
    for tex in desc.textures:
        res.add_texture(tex)
    scene = host_mod.Scene()
    for e in desc.entities:
        ent = scene.create_entity(e.position, e.rotation, e.scale)
        scene.add_mesh(ent, e.mesh, host_mod.Material(**e.material))
    for pos, radius, mat in desc.spheres:
        ent = scene.create_entity(pos)
        scene.add_sphere(ent, radius, host_mod.Material(**mat))
    for rot, colour, intensity, angle in desc.dir_lights:
        scene.add_dir_light(scene.create_entity(rotation=rot), colour, intensity, angle)
    for pos, colour, intensity, radius in desc.point_lights:
        scene.add_point_light(scene.create_entity(pos), colour, intensity, radius)
    for pos, rot, colour, intensity, radius, max_angle in desc.spot_lights:
        scene.add_spot_light(scene.create_entity(pos, rot), colour, intensity, radius, max_angle)
    cam = desc.camera
    scene.add_camera(scene.create_entity(cam["position"], cam["rotation"]), cam["fov"], cam["near_z"], cam["far_z"])

    rt = host_mod.RayTracer(device=device, api=api, fns=fns)
    rt.initiate(width, height, res, desc.env_cross)
    if desc.bvh != (30, 5):
        rt.loadMeshAndBvhData(*desc.bvh)
    rt.setScene(scene)
    rt.createOctTree(scene, *desc.octree)
    return rt, scene, res


def load_meshes_as_entities(resources, scene, file_path, textures_path="", scale=1.0, host=None):
    """Application::loadMeshesAsEntities (Application.cpp:131-152) for OBJ/MTL files: imports the objects, loads the
    textures the materials name (deduplicated by file name like ResourceManager::findOrLoadTexture), and creates one
    entity per object with its texture ids wired into the material. Returns the object descriptions."""
    import os
    from PIL import Image
    if host is None:
        from . import host as host_mod
    else:
        host_mod = host
    objs = resources.import_obj(file_path, scale)
    loaded = getattr(resources, "_texture_ids", None)
    if loaded is None:
        loaded = resources._texture_ids = {}

    def tex(name):
        if not name:
            return 0xFFFFFFFF
        key = os.path.basename(name.replace("\\\\", "/"))
        if key not in loaded:
            path = os.path.join(textures_path, name) if textures_path else os.path.join(os.path.dirname(file_path), name)
            img = np.asarray(Image.open(path).convert("RGBA"), dtype=np.uint8)  # stbi_load(..., STBI_rgb_alpha)
            loaded[key] = resources.add_texture(img)
        return loaded[key]

    for o in objs:
        ent = scene.create_entity()
        mat = host_mod.Material(albedoTextureId=tex(o["albedo"]), roughnessTextureId=tex(o["roughness"]), metallicTextureId=tex(o["metallic"]),
                                specularTextureId=tex(o["specular"]), normalMapIdx=tex(o["normal"]))
        scene.add_mesh(ent, o["meshId"], mat)
        o["entity"] = ent
    return objs

"""The MI355X traversal structure (okrt_fast_build.cpp) checked on the CPU: built from the host layer's tables and
traced by a host-side walker with the kernel's node/leaf decoding, against brute force over the same triangles."""
import ctypes as C

import numpy as np
import pytest

from conftest import abi, scenes


def brute_force(tp, orig, dirs):
    """Closest Moeller-Trumbore hit (reference edge convention) over a triangle soup, float64."""
    dn = dirs / np.linalg.norm(dirs, axis=1, keepdims=True)
    best = np.full(len(orig), np.inf)
    p0, p1, p2 = tp[:, 0].astype(np.float64), tp[:, 1].astype(np.float64), tp[:, 2].astype(np.float64)
    e1, e2 = p0 - p2, p1 - p2
    for r in range(len(orig)):
        c1 = np.cross(dn[r], e2)
        det = (e1 * c1).sum(1)
        ok = np.abs(det) > 1e-6
        idet = 1.0 / np.where(ok, det, 1.0)
        mv = orig[r] - p2
        u = (mv * c1).sum(1) * idet
        c2 = np.cross(mv, e1)
        v = (c2 @ dn[r]) * idet
        t = (e2 * c2).sum(1) * idet
        hit = ok & (u >= -1e-6) & (v >= -1e-6) & (u + v <= 1) & (t > 1e-9)
        if hit.any():
            best[r] = t[hit].min()
    return best


def fast_closest(rt, orig, dirs):
    lib = C.CDLL(abi.device_library_path())
    T = {k: rt.table(k) for k in (0, 2, 5, 7)}
    n = len(orig)
    dist, tri, inst, stats = np.zeros(n, np.float32), np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(4, np.uint32)
    P = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    o32, d32 = np.ascontiguousarray(orig, np.float32), np.ascontiguousarray(dirs, np.float32)
    lib.okrt_debug_fast_closest_hit(P(T[0]), len(T[0]), P(T[2]), len(T[2]), P(T[5]), len(T[5]), P(T[7]), len(T[7]), n,
                                    P(o32), P(d32), P(dist), P(tri), P(inst), P(stats))
    return dist, tri, inst, stats, T


@pytest.mark.parametrize("transforms", [False, True])
def test_fast_structure_finds_the_brute_force_hit(oracle, transforms):
    desc = scenes.small_mesh_scene(seed=21, n_objects=6, detail=8, with_textures=False, transforms=transforms)
    rt, sc, res = oracle.instantiate(desc, 16, 16)
    rng = np.random.default_rng(1)
    n = 600
    orig = np.tile(np.array([0.0, 3.0, -18.0]), (n, 1)) + rng.normal(0, 0.5, (n, 3))
    dirs = rng.normal(size=(n, 3)); dirs[:, 2] = np.abs(dirs[:, 2]) + 1.0; dirs[:, 1] *= 0.4
    dist, tri, inst, stats, T = fast_closest(rt, orig, dirs)
    tp = T[0].view(np.float32).reshape(-1, 3, 3)
    assert stats[1] == len(tp) and stats[3] == (0 if transforms else 1)
    if transforms:  # world-space soup = M * p per instance (what the structure bakes)
        meshes = T[7]
        world = np.zeros_like(tp, dtype=np.float64)
        for m in meshes:
            M = m[:64].view(np.float32).reshape(4, 4).astype(np.float64)  # row j of memory = row j of G
            s, e = m[128:136].view(np.uint32)
            pts = tp[s:e].reshape(-1, 3).astype(np.float64)
            world[s:e] = (pts @ M[:3, :3].T + M[:3, 3]).reshape(-1, 3, 3)
        tp_world = world
    else:
        tp_world = tp
    ref = brute_force(tp_world, orig, dirs)
    hit = np.isfinite(ref)
    assert (hit == (dist < 3e38)).all()
    assert 0.15 < hit.mean() < 0.95
    assert np.allclose(dist[hit], ref[hit], rtol=2e-4, atol=2e-4)
    # the reported triangle really is at that distance
    assert (tri[hit] < len(tp)).all() and (inst[hit] < len(T[7])).all()
    rt.close()


def test_fast_structure_degenerate_inputs(oracle):
    # no meshes at all: one empty root, every ray misses
    rt, sc, res = oracle.instantiate(scenes.test_scene_spheres(), 16, 16)
    dist, tri, inst, stats, T = fast_closest(rt, np.zeros((4, 3)), np.array([[0, 0, 1.0]] * 4))
    assert stats[0] == 1 and stats[1] == 0 and (dist > 3e38).all() and (tri == 0xFFFFFFFF).all()
    rt.close()
    # a single triangle (one leaf under the root), and many coincident triangles (median splits)
    for soup in (np.array([[[0, 0, 5], [4, 0, 5], [0, 4, 5]]], np.float32),
                 np.tile(np.array([[[0, 0, 5], [4, 0, 5], [0, 4, 5]]], np.float32), (37, 1, 1))):
        desc = scenes.SceneDescription("degenerate")
        desc.meshes = [(soup, np.zeros((len(soup), 33), np.float32))]
        desc.entities = [scenes.MeshEntity(0, {})]
        rt, sc, res = oracle.instantiate(desc, 16, 16)
        dist, tri, inst, stats, T = fast_closest(rt, np.array([[1.0, 1.0, 0.0], [9.0, 9.0, 0.0]]), np.array([[0, 0, 1.0], [0, 0, 1.0]]))
        assert stats[1] == len(soup) and dist[0] == 5.0 and dist[1] > 3e38
        rt.close()


def test_fast_structure_with_coordinates_that_overflow_the_collapse_costs(oracle):
    """Boxes with extents around 1e19 have areas around 1e38: every sum in the SAH-optimal collapse (okrt_fast_build.cpp:
    Emitter::solve) overflows to inf, no candidate split is ever "cheaper", and the forest walk must still terminate with at
    most four children per node (a malformed OBJ or table can supply such numbers; runs under `make asan` too)."""
    rng = np.random.default_rng(5)
    centres = rng.uniform(-1e19, 1e19, (400, 1, 3))
    soup = (centres + rng.uniform(-3e18, 3e18, (400, 3, 3))).astype(np.float32)
    soup[:8] *= np.float32(1e-19)  # and a few ordinary triangles among them
    desc = scenes.SceneDescription("huge")
    desc.meshes = [(soup, np.zeros((len(soup), 33), np.float32))]
    desc.entities = [scenes.MeshEntity(0, {})]
    rt, sc, res = oracle.instantiate(desc, 16, 16)
    orig = np.zeros((3, 3))
    dist, tri, inst, stats, T = fast_closest(rt, orig, np.array([[0, 0, 1.0], [1.0, 0, 0], [0, 1.0, 0]]))
    assert stats[1] == len(soup) and stats[0] >= 1  # every triangle referenced, a tree came out
    assert np.isfinite(dist[dist < 3e38]).all()
    rt.close()

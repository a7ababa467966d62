"""The device-side builder of the traversal structure (OKRT_OPT_BUILDER = 1, csrc/okrt_fast_build_device.hip; SURVEY §8f
rank 2) against the host builder and the oracle.

What must hold for ANY correct structure, whoever built it:
  * the triangles are exactly the (source triangle, instance) pairs the reference shader reaches, each once, baked to
    world space with the same bits as the host build;
  * every node slot but the root and every triangle slot is referenced exactly once;
  * every quantised child box contains all triangles below it (conservative => the closest hit is unchanged);
so renders agree with the reference-order traversal / the oracle up to equal-distance ties."""
import numpy as np
import pytest

from conftest import abi, compare_accum, scenes

pytestmark = pytest.mark.gpu

LEAF = 0x80000000
EMPTY = 0xFFFFFFFF


def decode_boxes(nodes):
    """nodes uint32 [n,16] -> (lo, hi) float64 [n,4,3] child boxes, child refs [n,4]."""
    origin = nodes[:, 0:3].copy().view(np.float32).astype(np.float64)
    exps = nodes[:, 3]
    child = nodes[:, 4:8]
    step = np.stack([np.ldexp(1.0, ((exps >> (8 * a)) & 255).astype(np.int64) - 127) for a in range(3)], axis=1)
    lo = np.zeros((len(nodes), 4, 3))
    hi = np.zeros((len(nodes), 4, 3))
    for a in range(3):
        for k in range(4):
            lo[:, k, a] = origin[:, a] + ((nodes[:, 8 + a] >> (8 * k)) & 255) * step[:, a]
            hi[:, k, a] = origin[:, a] + ((nodes[:, 12 + a] >> (8 * k)) & 255) * step[:, a]
    return lo, hi, child


def tri_vertices(tris):
    f = tris.view(np.float32)
    p2 = f[:, 0:3].astype(np.float64)
    p0 = p2 + f[:, 4:7]
    p1 = p2 + f[:, 8:11]
    return np.stack([p0, p1, p2], axis=1)  # [m,3,3], up to the rounding of E = p - p2


def check_structure(nodes, tris):
    n, m = len(nodes), len(tris)
    lo, hi, child = decode_boxes(nodes)
    verts = tri_vertices(tris)
    tlo, thi = verts.min(1), verts.max(1)
    node_refs = np.zeros(n, np.int64)
    tri_refs = np.zeros(m, np.int64)
    blo = np.full((n, 3), np.inf)
    bhi = np.full((n, 3), -np.inf)
    for i in range(n - 1, -1, -1):  # children always live above their parent
        for k in range(4):
            c = int(child[i, k])
            if c == EMPTY:
                continue
            if c & LEAF:
                first, count = c & 0x07FFFFFF, ((c >> 27) & 15) + 1
                assert count <= 15 and first + count <= m
                tri_refs[first:first + count] += 1
                clo, chi = tlo[first:first + count].min(0), thi[first:first + count].max(0)
            else:
                assert i < c < n, (i, c)
                node_refs[c] += 1
                clo, chi = blo[c], bhi[c]
            # tolerance: the edge vectors E1/E2 are rounded differences, so p0/p1 are recovered to ~1 ulp
            eps = 1e-5 * max(1.0, np.abs(clo).max(), np.abs(chi).max())
            assert (clo >= lo[i, k] - eps).all() and (chi <= hi[i, k] + eps).all(), (i, k, clo, lo[i, k], chi, hi[i, k])
            blo[i] = np.minimum(blo[i], clo)
            bhi[i] = np.maximum(bhi[i], chi)
    assert (tri_refs == 1).all()
    assert (node_refs[1:] == 1).all() and node_refs[0] == 0


def build(rt, builder):
    rt.ctx.set_option(abi.OPT_BUILDER, builder)
    rt.ctx.commit()
    info = rt.ctx.build_info()
    assert info["builder"] == builder
    nodes, tris = rt.ctx.fast_structure()
    return info, nodes, tris


def sorted_tris(tris):
    key = tris[:, 3].astype(np.uint64) << np.uint64(32) | tris[:, 7].astype(np.uint64)  # srcTri, instance
    order = np.argsort(key, kind="stable")
    return key[order], tris[order]


@pytest.mark.parametrize("scene", ["identity", "transforms", "single", "sponza_small", "dragon80k"])
def test_device_structure_is_valid_and_covers_the_same_triangles(scene):
    if scene == "identity":
        desc = scenes.small_mesh_scene(seed=5, n_objects=8, detail=16, with_lights=False)
    elif scene == "transforms":
        desc = scenes.small_mesh_scene(seed=13, n_objects=7, detail=12, transforms=True, with_lights=True, octree=(0, 50))
    elif scene == "single":
        desc = scenes.small_mesh_scene(seed=2, n_objects=1, detail=1, with_lights=False)
    elif scene == "sponza_small":
        desc = scenes.sponza_proxy(tex_size=64, env_face=64, tri_scale=0.1)
    else:
        desc = scenes.dragon_proxy(80000)
    rt, sc, res = scenes.instantiate(desc, 64, 36)
    hinfo, hnodes, htris = build(rt, abi.BUILDER_HOST)
    dinfo, dnodes, dtris = build(rt, abi.BUILDER_DEVICE)
    assert dinfo["num_triangles"] == hinfo["num_triangles"] == len(dtris)
    hk, ht = sorted_tris(htris)
    dk, dt = sorted_tris(dtris)
    assert (hk == dk).all()
    assert (ht[:, :11] == dt[:, :11]).all()  # baked vertices / edges: same bits as the host build
    check_structure(dnodes, dtris)
    check_structure(hnodes, htris)
    assert dinfo["max_depth"] >= 2 or dinfo["num_triangles"] <= 4
    rt.close()


@pytest.mark.parametrize("scene", ["identity", "transforms", "glass"])
def test_device_built_render_vs_oracle(oracle, scene):
    if scene == "identity":
        desc, mb = scenes.small_mesh_scene(seed=5, n_objects=8, detail=16, with_lights=True), 1
    elif scene == "transforms":
        desc, mb = scenes.small_mesh_scene(seed=13, n_objects=7, detail=12, transforms=True, with_lights=True, octree=(0, 50)), 1
    else:
        desc, mb = scenes.glass_revolver_proxy(tex_size=128, env_face=64, detail=0.5), 8
    w, h = 320, 180
    rt, sc, res = scenes.instantiate(desc, w, h)
    rt.ctx.set_option(abi.OPT_BUILDER, abi.BUILDER_DEVICE)
    rt.ctx.set_option(abi.OPT_MAX_BOUNCES, mb)
    rt.render()
    ga = rt.ctx.read_accum()
    assert rt.ctx.build_info()["builder"] == abi.BUILDER_DEVICE
    rt.close()
    ort, _, _ = oracle.instantiate(desc, w, h)
    ort.ctx.set_option(abi.OPT_MAX_BOUNCES, mb)
    ort.render()
    ca = ort.ctx.read_accum()
    ort.close()
    frac, rmse, nanmm, _ = compare_accum(ga, ca)
    if scene == "transforms":
        assert frac >= 0.99 and rmse <= 0.03, (frac, rmse)
    else:
        assert frac >= 0.995 and rmse <= 0.02 and nanmm <= 1e-5, (frac, rmse, nanmm)


def test_device_and_host_built_renders_agree_bit_for_bit_up_to_ties():
    desc = scenes.sponza_proxy(tex_size=64, env_face=64, tri_scale=0.25)
    rt, sc, res = scenes.instantiate(desc, 640, 360)
    accs = []
    for b in (abi.BUILDER_HOST, abi.BUILDER_DEVICE):
        rt.ctx.set_option(abi.OPT_BUILDER, b)
        rt.resetAccumulation()
        rt.renderFrames(2)
        accs.append(rt.ctx.read_accum())
    same = (accs[0] == accs[1]).all(-1) | (np.isnan(accs[0]).any(-1) & np.isnan(accs[1]).any(-1))
    assert same.mean() >= 0.999, same.mean()
    rt.close()


def test_instance_edit_rebuilds_on_the_device():
    """Moving an entity re-uploads t7 (+ octree): the device builder re-bakes and re-clusters without touching the
    per-mesh tables, and the moved object shows up where the reference-order traversal puts it."""
    desc = scenes.small_mesh_scene(seed=21, n_objects=5, detail=10, with_lights=False, octree=(0, 50))
    rt, sc, res = scenes.instantiate(desc, 256, 144)
    rt.ctx.set_option(abi.OPT_BUILDER, abi.BUILDER_DEVICE)
    rt.render()
    before = rt.ctx.read_accum().copy()
    sc.set_transform(0, position=(3.0, 1.5, -2.0), rotation=(0.0, 30.0, 0.0))  # entity 0 = the first mesh of the description
    rt.createOctTree(sc, 0, 50)
    images = []
    for trav in (abi.TRAVERSAL_FAST, abi.TRAVERSAL_REFERENCE):
        rt.ctx.set_option(abi.OPT_TRAVERSAL, trav)
        rt.resetAccumulation()
        rt.render()
        images.append(rt.ctx.read_accum().copy())
    assert not np.array_equal(np.nan_to_num(before), np.nan_to_num(images[0]))
    same = (images[0] == images[1]).all(-1) | (np.isnan(images[0]).any(-1) & np.isnan(images[1]).any(-1))
    l2 = np.sqrt(((images[0][..., :3] - images[1][..., :3]) ** 2).sum(-1))
    assert same.mean() >= 0.99 and (l2[np.isfinite(l2)] <= 1e-3).mean() >= 0.995, same.mean()
    rt.close()


def test_empty_and_tiny_scenes_build_on_the_device():
    desc = scenes.small_mesh_scene(seed=2, n_objects=1, detail=1, with_lights=False)
    rt, sc, res = scenes.instantiate(desc, 64, 36)
    info, nodes, tris = build(rt, abi.BUILDER_DEVICE)
    assert info["num_nodes"] >= 1 and info["num_triangles"] == desc.num_triangles
    rt.render()
    rt.close()

"""Oracle and host layer against the committed fixtures (tests/golden/, made by make_golden.py).

ref_host.npz holds outputs of the REAL reference host sources (BvhBuilder.cpp, Components.h + glm) compiled in
the build container; oracle_images.npz pins the oracle's images (the reference ships none: parity unpinned)."""
import os
import sys

import numpy as np
import pytest

from conftest import abi, host, scenes

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden  # noqa: E402

REF = np.load(os.path.join(HERE, "golden", "ref_host.npz"))
IMG = np.load(os.path.join(HERE, "golden", "oracle_images.npz"))


def test_struct_sizes_match_the_reference_headers():
    s = REF["struct_sizes"]
    assert list(s[:6]) == [24, 36, 132, 68, 232, 84] and s[7] == 44
    assert (s[6] & 255, (s[6] >> 8) & 255, (s[6] >> 16) & 255) == (32, 32, 48)
    assert abi.TABLE_STRIDE == {0: 36, 1: 132, 2: 36, 5: 48, 6: 84, 7: 232, 8: 32, 9: 32, 10: 48}


@pytest.mark.parametrize("name", ["soup96", "grid96", "soup96_leaf1"])
def test_bvh_builder_equals_reference_builder(name):
    tris = REF[f"bvh_{name}_tris"]
    leaf, depth = [int(x) for x in REF[f"bvh_{name}_params"]]
    nodes, order = host.bvh_build(tris, leaf, depth)
    ref_nodes = REF[f"bvh_{name}_nodes"]
    assert nodes.shape == ref_nodes.shape
    assert np.array_equal(nodes.view(np.float32)[:, :6], ref_nodes.view(np.float32)[:, :6])  # boxes (== treats -0 as 0)
    assert np.array_equal(nodes[:, 6:], ref_nodes[:, 6:])  # triStart, triEnd, firstChild
    assert np.array_equal(order, REF[f"bvh_{name}_order"])


def test_camera_equals_reference_calculateProjectionData(oracle):
    for cin, cout in zip(REF["camera_in"], REF["camera_out"]):
        sc = host.Scene()
        sc.add_camera(sc.create_entity(cin[0:3], cin[3:6]), float(cin[6]), float(cin[7]), float(cin[8]))
        rt = host.RayTracer(api=oracle.api, fns=oracle.fns)
        rt.initiate(int(cin[9]), int(cin[10]), host.ResourceManager())
        rt.setScene(sc)
        rt.calculateProjectionData()
        rd = rt.render_data
        got = np.concatenate([np.array(rd.viewPlaneDims), np.array(rd.cameraInverseProjectionMatrix), np.array(rd.cameraInverseViewMatrix),
                              np.array(rd.cameraPosition), [rd.cameraNearZ], np.array(rd.cameraUpDir), np.array(rd.cameraRightDir)]).astype(np.float32)
        assert np.array_equal(got, cout[:44]), np.abs(got - cout[:44]).max()
        rt.close()


def test_instance_matrices_equal_reference():
    import ctypes as C
    lib = host.host_library()
    for tin, tout in zip(REF["xform_in"], REF["xform_out"]):
        m, inv = (C.c_float * 16)(), (C.c_float * 16)()
        lib.okh_instance_matrices((C.c_float * 3)(*tin[0:3]), (C.c_float * 3)(*tin[3:6]), (C.c_float * 3)(*tin[6:9]), m, inv)
        assert np.array_equal(np.array(m, np.float32), tout[:16])
        assert np.array_equal(np.array(inv, np.float32), tout[16:])


@pytest.mark.parametrize("name", sorted(make_golden.golden_scenes()))
def test_oracle_reproduces_golden_images(oracle, name):
    desc, w, h, frames, mb = make_golden.golden_scenes()[name]
    rt, _, _ = oracle.instantiate(desc, w, h)
    rt.ctx.set_option(abi.OPT_MAX_BOUNCES, mb)
    rt.ctx.set_option(abi.OPT_COUNTERS, 1)
    for _ in range(frames):
        rt.render()
    acc = rt.ctx.read_accum()
    gold = IMG[name]
    same = (acc == gold).all(-1) | (np.isnan(acc).any(-1) & np.isnan(gold).any(-1))
    # the scene generators run numpy float64 trig; allow a stray last-ulp vertex difference across machines
    assert same.mean() >= 0.999, same.mean()
    c = rt.ctx.counters()
    gc = dict(zip(sorted(c), IMG[name + "_counters"].tolist()))
    for key in ("pixels", "segments", "n_node", "n_tri", "mesh_hits"):
        assert abs(c[key] - gc[key]) <= 0.001 * max(gc[key], 1) + 2, (key, c[key], gc[key])
    assert (acc[..., 3] == frames).all()  # accumulationBuffer += float4(light, 1)  (:528)
    rt.close()

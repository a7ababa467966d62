"""The OBJ/MTL reader (host/obj_loader.cpp) against the importer conventions it documents (Importer.cpp:15-56,77-120;
assimp's ConvertToLeftHanded = MakeLeftHanded + FlipUVs + FlipWindingOrder, postprocess.h:610-624)."""
import os

import numpy as np
import pytest

from conftest import abi, host, scenes

OBJ = """# two objects, a quad and a triangle, two materials
mtllib test.mtl
o plate
v 0 0 0
v 2 0 0
v 2 0 2
v 0 0 2
vt 0 0
vt 1 0
vt 1 1
vt 0 1
vn 0 1 0
usemtl wood
f 1/1/1 2/2/1 3/3/1 4/4/1
o spike
v 0 1 0
v 1 1 0
v 0 2 0
usemtl bare
f -3 -2 -1
"""
MTL = """newmtl wood
Kd 1 1 1
map_Kd -s 1 1 1 wood_albedo.png
map_Ks wood_spec.png
map_bump wood_bump.png
disp wood_disp.png
newmtl bare
Kd 0.5 0.5 0.5
"""


@pytest.fixture
def obj_dir(tmp_path):
    from PIL import Image
    (tmp_path / "test.obj").write_text(OBJ)
    (tmp_path / "test.mtl").write_text(MTL)
    rng = np.random.default_rng(0)
    for name in ("wood_albedo.png", "wood_spec.png", "wood_disp.png"):
        Image.fromarray(rng.integers(0, 256, (8, 8, 3), dtype=np.uint8)).save(tmp_path / name)
    return tmp_path


def test_conventions(obj_dir):
    res = host.ResourceManager()
    objs = res.import_obj(str(obj_dir / "test.obj"), 2.0)
    assert [(o["name"], o["numTris"]) for o in objs] == [("plate", 2), ("spike", 1)]
    assert objs[0]["albedo"] == "wood_albedo.png" and objs[0]["specular"] == "wood_spec.png"
    assert objs[0]["normal"] == "wood_disp.png"          # no NORMALS map -> DISPLACEMENT (Importer.cpp:110-113); map_bump is ignored
    assert objs[1]["albedo"] == "" and objs[1]["normal"] == ""
    rt_scene = host.Scene()
    ents = scenes.load_meshes_as_entities(res, rt_scene, str(obj_dir / "test.obj"), "", 2.0)  # second import: new meshes, cached textures
    assert len(res._texture_ids) == 3 and ents[0]["meshId"] == 2


def test_geometry_is_left_handed_flipped_and_scaled(obj_dir, oracle):
    res = host.ResourceManager()
    res.import_obj(str(obj_dir / "test.obj"), 2.0)
    sc = host.Scene()
    sc.add_mesh(sc.create_entity(), 0)
    sc.add_mesh(sc.create_entity(), 1)
    sc.add_camera(sc.create_entity())
    rt = host.RayTracer(api=oracle.api, fns=oracle.fns)
    rt.initiate(8, 8, res)
    rt.setScene(sc)
    rt.createOctTree(sc, 0, 50)
    md = rt.table(100).view(np.uint32)
    tp = rt.table(0).view(np.float32).reshape(-1, 3, 3)
    ti = rt.table(1).view(np.float32).reshape(-1, 3, 11)
    plate = tp[md[0, 0]:md[0, 1]].reshape(-1, 3)
    # scale 2 (Importer.cpp:40), z negated (MakeLeftHanded)
    assert set(map(tuple, np.unique(plate, axis=0))) == {(0, 0, 0), (4, 0, 0), (4, 0, -4), (0, 0, -4)}
    info = ti[md[0, 0]:md[0, 1]].reshape(-1, 11)
    assert np.allclose(info[:, 0:3], [0, 1, 0])                       # normal (z of (0,1,0) negated is still 0)
    pos_uv = {tuple(p): tuple(i[3:5]) for p, i in zip(plate, info)}
    assert pos_uv[(0, 0, 0)] == (0, 1) and pos_uv[(4, 0, -4)] == (1, 0)  # FlipUVs: v -> 1 - v
    # tangent = +u direction = +x, bitangent = +v direction = +z before the flips -> (0,0,-1) after MakeLeftHanded
    assert np.allclose(info[:, 5:8], [1, 0, 0], atol=1e-6) and np.allclose(info[:, 8:11], [0, 0, -1], atol=1e-6)
    # FlipWindingOrder: OBJ faces are CCW seen from +y in a right-handed frame; after z-negation + index reversal the
    # geometric normal (cross of the edges in table order) still points along the stored vertex normal's side
    tri = tp[md[0, 0]]
    assert np.cross(tri[1] - tri[0], tri[2] - tri[0])[1] != 0
    # the spike has no vn / vt: face normal generated, uv (0,0), zero tangent frame (Importer.cpp:43-45)
    sinfo = ti[md[1, 0]:md[1, 1]].reshape(-1, 11)
    assert np.allclose(np.abs(sinfo[:, 0:3]), [0, 0, 1]) and (sinfo[:, 3:] == 0).all()
    rt.close()


@pytest.mark.skipif(not os.path.exists("/root/reference/resources/meshes/ikea_glass.obj"), reason="reference checkout only in the build container")
def test_reference_ikea_glass_obj_loads():
    res = host.ResourceManager()
    objs = res.import_obj("/root/reference/resources/meshes/ikea_glass.obj", 10.0)  # Application.cpp:42
    assert len(objs) == 3 and sum(o["numTris"] for o in objs) == 10386  # 5310 faces (quads + triangles), fan-triangulated


def test_errors(tmp_path):
    res = host.ResourceManager()
    with pytest.raises(abi.OkrtError):
        res.import_obj(str(tmp_path / "missing.obj"))
    (tmp_path / "bad.obj").write_text("v 0 0 0\nf 1 2 3\n")
    with pytest.raises(abi.OkrtError, match="out of range"):
        res.import_obj(str(tmp_path / "bad.obj"))
    # relative texture / normal indices that point before the first element stay negative after resolution
    for face in ("f 1/-100 2/1 3/1", "f 1/1/-7 2/1/1 3/1/1"):
        (tmp_path / "neg.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvn 0 0 1\n" + face + "\n")
        with pytest.raises(abi.OkrtError, match="out of range"):
            res.import_obj(str(tmp_path / "neg.obj"))

"""Malformed-input sweep of the two file-facing pieces of the C++ host layer (host/obj_loader.cpp, host/png_writer.cpp):
whatever a file holds, the importer returns objects or a clean error — never a crash, an out-of-range read or an
unparseable description — and the PNG encoder refuses empty images and encodes every odd size correctly. `make asan`
runs this file (with the rest of the CPU suite) against -fsanitize=address,undefined builds."""
import ctypes as C
import os
import random

import numpy as np
import pytest

from conftest import abi, host

GOOD_OBJ = """mtllib m.mtl
o a
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
vt 0 0
vt 1 0
vt 1 1
vn 0 0 1
usemtl x
f 1/1/1 2/2/1 3/3/1 4/1/1
o b
usemtl y
f -4 -3 -2
"""
GOOD_MTL = """newmtl x
Kd 1 1 1
map_Kd -s 1 1 1 tex.png
newmtl y
disp d.png
"""
TRI = "v 0 0 0\nv 1 0 0\nv 0 1 0\n"


def try_import(path):
    res = host.ResourceManager()
    try:
        objs = res.import_obj(str(path))
    except abi.OkrtError as e:
        return None, str(e)
    for o in objs:  # a successful import describes real meshes
        assert o["numTris"] > 0 and isinstance(o["name"], str)
    return objs, None


def test_hand_picked_malformed_obj_files(tmp_path):
    (tmp_path / "m.mtl").write_text(GOOD_MTL)
    cases = {
        "empty": "",
        "comments_only": "# nothing\n\n#\n",
        "no_faces": TRI,
        "index_zero": TRI + "f 0 1 2\n",
        "index_too_big": TRI + "f 1 2 4\n",
        "index_huge": TRI + "f 1 2 99999999999999999999999\n",
        "index_negative_too_far": TRI + "f -1 -2 -4\n",
        "index_negative_huge": TRI + "f -1 -2 -99999999999999999999\n",
        "uv_index_out_of_range": TRI + "vt 0 0\nf 1/1 2/2 3/9\n",
        "normal_index_out_of_range": TRI + "f 1//1 2//1 3//1\n",
        "garbage_tokens": "v a b c\nvt x\nvn\nf 1/x/y 2/// 3/-/-\nf\nf 1\nf 1 2\n",
        "truncated_vertex": "v 0 0\nv 1\nv\nv 0 1 0\nf 1 2 3\nf 1 2 4\n",
        "truncated_mid_line": GOOD_OBJ[: len(GOOD_OBJ) - 7],
        "missing_mtl": GOOD_OBJ.replace("m.mtl", "does_not_exist.mtl"),
        "mtllib_without_name": "mtllib\n" + GOOD_OBJ,
        "usemtl_unknown": GOOD_OBJ.replace("usemtl x", "usemtl nobody"),
        "nan_inf": "v nan inf -inf\nv 1e39 -1e39 1e-50\nv 0 1 0\nf 1 2 3\n",
        "degenerate_faces": "v 0 0 0\nv 0 0 0\nv 0 0 0\nvt 0 0\nf 1/1 2/1 3/1\nf 1 1 1\n",
        "very_long_line": TRI + "f " + " ".join(["1", "2", "3"] * 20000) + "\n",
        "crlf": GOOD_OBJ.replace("\n", "\r\n"),
        "many_slashes": TRI + "f 1//// 2/1/1/1/1 3/\n",
    }
    ok = 0
    for name, text in cases.items():
        p = tmp_path / f"{name}.obj"
        p.write_text(text, newline="")
        objs, err = try_import(p)
        assert (objs is None) != (err is None), name
        ok += objs is not None
    assert ok >= 6  # the recoverable ones import (missing .mtl, CRLF, a long fan, ...), the rest say why not
    assert try_import(tmp_path / "absent.obj")[1] is not None
    (tmp_path / "good.obj").write_text(GOOD_OBJ)  # the good file, for reference
    objs, err = try_import(tmp_path / "good.obj")
    assert err is None and [(o["name"], o["numTris"]) for o in objs] == [("a", 2), ("b", 1)]
    assert objs[0]["albedo"] == "tex.png" and objs[1]["normal"] == "d.png"


def test_names_that_are_not_utf8_or_need_escaping(tmp_path):
    """Object and texture names come straight from the files: bytes that are not UTF-8, quotes, backslashes and control
    characters must come back as a parseable description (escaped as Latin-1 code points), not as an exception."""
    (tmp_path / "n.mtl").write_bytes(b"newmtl m\nmap_Kd wei" + bytes([0xDF]) + b"\\dir\\\"tex\".png\n")
    (tmp_path / "n.obj").write_bytes(b"mtllib n.mtl\no caf" + bytes([0xE9, 0x01]) + b"\"quoted\"\\\n" + TRI.encode() + b"usemtl m\nf 1 2 3\n")
    objs, err = try_import(tmp_path / "n.obj")
    assert err is None and len(objs) == 1
    assert objs[0]["name"] == "caf" + chr(0xE9) + chr(1) + "\"quoted\"\\" and objs[0]["albedo"] == "wei" + chr(0xDF) + "\\dir\\\"tex\".png"


@pytest.mark.parametrize("seed", range(6))
def test_random_mutations_of_a_valid_obj_never_crash(tmp_path, seed):
    """Byte-level fuzz: deletions, duplications, bit flips and spliced garbage on a valid OBJ / MTL pair (300 variants per
    seed). Every variant imports or fails with a message."""
    rng = random.Random(seed)
    junk = [b"/", b"-", b"999999999", b"-999999999", bytes([0]), bytes([0xFF, 0xFE]), b"f ", b"v ", b"\n", b" 1e400 ", b"usemtl ", b"mtllib ", b"//", b"o "]
    imported = 0
    for _ in range(300):
        obj, mtl = bytearray(GOOD_OBJ.encode()), bytearray(GOOD_MTL.encode())
        for buf in (obj, mtl):
            for _ in range(rng.randint(1, 6)):
                pos = rng.randrange(len(buf) + 1)
                op = rng.randrange(4)
                if op == 0 and len(buf) > 1:
                    del buf[pos % len(buf): pos % len(buf) + rng.randint(1, 12)]
                elif op == 1:
                    buf[pos:pos] = buf[max(0, pos - rng.randint(1, 20)): pos]
                elif op == 2 and len(buf):
                    buf[pos % len(buf)] ^= 1 << rng.randrange(8)
                else:
                    buf[pos:pos] = rng.choice(junk)
        (tmp_path / "m.mtl").write_bytes(bytes(mtl))
        p = tmp_path / "f.obj"
        p.write_bytes(bytes(obj))
        objs, err = try_import(p)
        assert (objs is None) != (err is None)
        imported += objs is not None
    assert 0 < imported < 300  # both outcomes were exercised


def test_png_writer_sizes_and_refusals(tmp_path):
    from PIL import Image
    lib = host.host_library()
    rng = np.random.default_rng(1)
    # odd sizes; rows longer and images larger than one 65535-byte stored block
    for (h, w) in [(1, 1), (1, 7), (7, 1), (3, 5), (1, 16384), (16385, 1), (255, 257), (129, 127)]:
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        path = str(tmp_path / f"s{h}x{w}.png")
        host.write_png(path, img)
        back = np.asarray(Image.open(path))
        assert back.shape == img.shape and np.array_equal(back, img), (h, w)
    one = np.zeros((1, 1, 4), np.uint8)
    px = one.ctypes.data_as(C.c_void_p)
    for (w, h) in [(0, 0), (0, 5), (5, 0)]:  # empty images are refused, nothing is written
        path = tmp_path / f"z{w}x{h}.png"
        assert lib.okh_write_png(os.fsencode(str(path)), w, h, px) == 0 and not path.exists()
    assert lib.okh_write_png(os.fsencode(str(tmp_path / "n.png")), 4, 4, None) == 0  # no pixels
    assert lib.okh_write_png(None, 1, 1, px) == 0                                    # no path
    assert lib.okh_write_png(os.fsencode(str(tmp_path / "no_dir" / "x.png")), 1, 1, px) == 0

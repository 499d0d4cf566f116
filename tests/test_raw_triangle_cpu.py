"""PLY / GLB formats of a triangle model (diff_recon_hip/raw_triangle.py, mirror of the reference's RawTriangle IO,
src/diff_recon/models/raw_triangle.py:124-223).  plyfile / trimesh are not in the image, so the checks are against the two
format specifications (header grammar, chunk / accessor arithmetic) and round trips."""
import json
import struct

import numpy as np
import pytest

from diff_recon_hip.raw_triangle import RGB2SH, SH2RGB, RawTriangle, read_glb, read_ply_vertex_element


def _model(P, degree, seed=0):
    rng = np.random.default_rng(seed)
    K = (degree + 1) ** 2
    return RawTriangle(rng.normal(size=(P, 3, 3)).astype(np.float32), rng.normal(size=(P, 1)).astype(np.float32),
                       (0.3 * rng.normal(size=(P, 3 * K))).astype(np.float32))


@pytest.mark.parametrize("degree", [0, 1, 3])
def test_ply_round_trip_and_header(tmp_path, degree):
    m = _model(257, degree)
    p = tmp_path / "a" / "model.ply"  # the parent directory is created, raw_triangle.py:160
    m.savePLY(str(p), save_extra=True)
    head = open(p, "rb").read().split(b"end_header\n")[0].decode().splitlines()
    assert head[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 257"]
    names = [l.split()[2] for l in head[3:]]
    assert names[:13] == ["x1", "y1", "z1", "x2", "y2", "z2", "x3", "y3", "z3", "opacity", "f_dc_0", "f_dc_1", "f_dc_2"]
    assert names[13:] == [f"f_rest_{i}" for i in range(3 * (degree + 1) ** 2 - 3)]
    assert all(l.split()[:2] == ["property", "float"] for l in head[3:])
    assert p.stat().st_size == len("\n".join(head)) + len("\nend_header\n") + 257 * 4 * len(names)
    r = RawTriangle(ply_path=str(p))
    for k in ("vertex", "opacity", "shs"):
        assert np.array_equal(getattr(r, k), getattr(m, k)) and getattr(r, k).dtype == np.float32, k
    assert r.shDegree() == degree
    m.savePLY(str(tmp_path / "dc.ply"))  # default: DC only (raw_triangle.py:165-172)
    assert np.array_equal(RawTriangle(ply_path=str(tmp_path / "dc.ply")).shs, m.shs[:, :3])


def test_ply_reader_takes_other_encodings(tmp_path):
    m = _model(5, 0, seed=3)
    rows = np.concatenate([m.vertex.reshape(-1, 9), m.opacity, m.shs], axis=1)
    names = ["x1", "y1", "z1", "x2", "y2", "z2", "x3", "y3", "z3", "opacity", "f_dc_0", "f_dc_1", "f_dc_2"]
    head = "ply\nformat {}\ncomment written by hand\nelement vertex 5\n" + "".join(f"property float {n}\n" for n in names) + "element face 0\nproperty list uchar int vertex_indices\nend_header\n"
    (tmp_path / "be.ply").write_bytes(head.format("binary_big_endian 1.0").encode() + rows.astype(">f4").tobytes())
    (tmp_path / "asc.ply").write_bytes(head.format("ascii 1.0").encode() + "\n".join(" ".join(repr(float(x)) for x in r) for r in rows).encode() + b"\n")
    for f in ("be.ply", "asc.ply"):
        r = RawTriangle(ply_path=str(tmp_path / f))
        assert np.array_equal(r.vertex, m.vertex) and np.array_equal(r.opacity, m.opacity) and np.array_equal(r.shs, m.shs), f
    el = read_ply_vertex_element(str(tmp_path / "be.ply"))
    assert list(el) == names
    assert RawTriangle().loadPLY(str(tmp_path / "missing.ply")) is None  # warns and returns, raw_triangle.py:125-127
    m0 = RawTriangle(np.zeros((0, 3, 3), np.float32), np.zeros((0, 1), np.float32), np.zeros((0, 3), np.float32))
    m0.savePLY(str(tmp_path / "empty.ply"))
    assert not (tmp_path / "empty.ply").exists()  # save_empty=False


@pytest.mark.parametrize("save_back", [True, False])
def test_glb_container_and_round_trip(tmp_path, save_back):
    m = _model(100, 1, seed=7)
    p = tmp_path / "m.glb"
    m.saveGLB(str(p), save_back=save_back)
    raw = p.read_bytes()
    magic, version, total = struct.unpack_from("<4sII", raw, 0)
    assert (magic, version, total) == (b"glTF", 2, len(raw)) and total % 4 == 0
    jlen, jkind = struct.unpack_from("<I4s", raw, 12)
    assert jkind == b"JSON" and jlen % 4 == 0
    blen, bkind = struct.unpack_from("<I4s", raw, 20 + jlen)
    assert bkind == b"BIN\x00" and blen % 4 == 0 and 28 + jlen + blen == total
    doc, binary = read_glb(str(p))
    assert doc["asset"]["version"] == "2.0" and doc["buffers"][0]["byteLength"] == len(binary)
    prim = doc["meshes"][0]["primitives"][0]
    acc = doc["accessors"]
    assert doc["meshes"][0]["name"] == "geometry_0" and prim["mode"] == 4
    assert acc[prim["attributes"]["POSITION"]]["count"] == 300 and acc[prim["attributes"]["COLOR_0"]]["count"] == 300
    assert acc[prim["indices"]]["count"] == 300 * (2 if save_back else 1)
    for a in acc:  # every accessor fits its view, every view fits the buffer, views are 4-byte aligned
        v = doc["bufferViews"][a["bufferView"]]
        item = {5121: 1, 5125: 4, 5126: 4}[a["componentType"]] * {"SCALAR": 1, "VEC3": 3, "VEC4": 4}[a["type"]]
        assert a["count"] * item <= v["byteLength"] and v["byteOffset"] % 4 == 0 and v["byteOffset"] + v["byteLength"] <= len(binary)
    pos = np.frombuffer(binary, "<f4", 900, doc["bufferViews"][0]["byteOffset"]).reshape(-1, 3)
    assert np.allclose(acc[0]["min"], pos.min(axis=0)) and np.allclose(acc[0]["max"], pos.max(axis=0))  # required for POSITION
    idx = np.frombuffer(binary, "<u4", acc[prim["indices"]]["count"], doc["bufferViews"][2]["byteOffset"]).reshape(-1, 3)
    assert np.array_equal(idx[:100], np.arange(300).reshape(-1, 3))
    if save_back:
        assert np.array_equal(idx[100:], idx[:100, ::-1])  # raw_triangle.py:195-197
    r = RawTriangle(glb_path=str(p))
    assert np.array_equal(r.vertex.astype(np.float32), m.vertex)
    # colours and opacity go through 8 bits (what trimesh's face colours do as well): half a step of 1/255
    assert np.abs(SH2RGB(r.shs) - np.clip(SH2RGB(m.shs[:, :3]), 0, 1)).max() <= 0.5 / 255 + 1e-6
    sig = lambda x: 1 / (1 + np.exp(-x))
    assert np.abs(sig(r.opacity) - np.clip(sig(m.opacity), 1e-5, 1 - 1e-5)).max() <= 0.5 / 255 + 1e-5
    assert np.allclose(RGB2SH(SH2RGB(m.shs)), m.shs, atol=1e-6)

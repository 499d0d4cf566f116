"""On-disk formats of a triangle model ("on-disk formats come with it").

Mirror of the IO half of the reference's `RawTriangle` (src/diff_recon/models/raw_triangle.py:12-33, 124-223): the same class
name, constructor keywords, attributes (`vertex (P, 3, 3)`, `opacity (P, 1)` = the pre-sigmoid parameter, `shs (P, 3 K)` with
the DC triple first) and method names `loadPLY / savePLY / saveGLB / loadGLB`, so that checkpoints move between the two code
bases.  The reference goes through `plyfile` and `trimesh`; neither is a dependency here -- both formats are written and parsed
directly from their public specifications with numpy:

  * PLY (Turk, "The PLY Polygon File Format"): one element `vertex` with float32 properties
    x1 y1 z1 x2 y2 z2 x3 y3 z3 opacity f_dc_0 f_dc_1 f_dc_2 [f_rest_0 ...]   (raw_triangle.py:137-147, 161-172),
    written `binary_little_endian 1.0` like plyfile's default; the reader also takes big-endian, ascii and other scalar types;
  * GLB (Khronos glTF 2.0 binary container): one mesh `geometry_0` with un-shared vertices, triangle indices (front and, by
    default, back faces, raw_triangle.py:190-197) and a per-vertex COLOR_0 = (clip(SH2RGB(f_dc), 0, 1), sigmoid(opacity)) as
    normalised unsigned bytes -- what trimesh writes for face colours and what `loadGLB` reads back (:211-223).

Parity status: UNPINNED against plyfile / trimesh files (neither library is in the image, so no file written by the reference
could be produced here); the tests check the two specifications (header grammar, chunk and accessor arithmetic) and round trips."""
from __future__ import annotations

import json
import os
import struct
from pathlib import Path

import numpy as np

C0 = 0.28209479177387814  # src/diff_recon/utils/sh_utils.py:24


def SH2RGB(sh):  # sh_utils.py:107-108
    return sh * C0 + 0.5


def RGB2SH(rgb):  # sh_utils.py:103-104
    return (rgb - 0.5) / C0


_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
              "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def read_ply_vertex_element(path) -> dict:
    """The first element of a PLY file as {property name: 1-D array}.  List properties are not supported (the format above has none)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_first = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if count is None:
                    count, in_first = int(tok[2]), True
                else:
                    in_first = False
            elif tok[0] == "property" and in_first:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt is None or count is None:
            raise ValueError(f"{path}: PLY header without format / element")
        if fmt == "ascii":
            rows = np.loadtxt(f, max_rows=count, ndmin=2) if count else np.zeros((0, len(props)))
            return {n: rows[:, i].astype(t) for i, (n, t) in enumerate(props)}
        order = {"binary_little_endian": "<", "binary_big_endian": ">"}[fmt]
        dt = np.dtype([(n, order + t) for n, t in props])
        data = np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)
        return {n: np.ascontiguousarray(data[n]) for n, _ in props}


class RawTriangle:
    def __init__(self, vertex: np.ndarray = None, opacity: np.ndarray = None, shs: np.ndarray = None, *, ply_path: str = None,
                 glb_path: str = None) -> None:
        self.vertex, self.opacity, self.shs = vertex, opacity, shs
        if ply_path is not None:
            self.loadPLY(ply_path)
        if glb_path is not None:
            self.loadGLB(glb_path)

    def __len__(self):
        return len(self.vertex) if self.vertex is not None else 0

    def shDegree(self):  # raw_triangle.py:52-54
        return int(np.sqrt(self.shs.shape[1] / 3) - 1)

    # ---- PLY ---------------------------------------------------------------------------------------------------------------
    def loadPLY(self, path):  # raw_triangle.py:124-154
        if not os.path.exists(path):
            print(f"[Warning] File {path} does not exist! From loadPLY function in RawTriangle class.")
            return
        self.ply_path = path
        try:
            el = read_ply_vertex_element(path)
        except Exception as e:  # the reference reports and carries on
            print(f"Error reading {path}: {e}")
            return
        vertex = np.stack([el[n] for n in ("x1", "y1", "z1", "x2", "y2", "z2", "x3", "y3", "z3")], axis=1).astype(np.float32).reshape(-1, 3, 3)
        opacity = np.asarray(el["opacity"])[..., np.newaxis].astype(np.float32)
        f_dc = np.stack([el[f"f_dc_{i}"] for i in range(3)], axis=1)
        extra = sorted((n for n in el if n.startswith("f_rest_")), key=lambda n: int(n.split("_")[-1]))
        shs = np.concatenate([f_dc] + ([np.stack([el[n] for n in extra], axis=1)] if extra else []), axis=1).astype(np.float32)
        assert len(vertex) == len(opacity) == len(shs)
        assert len(extra) in [((d + 1) ** 2 - 1) * 3 for d in range(4)]
        self.vertex, self.opacity, self.shs = vertex, opacity, shs
        return self

    def savePLY(self, path, save_empty=False, save_extra=False):  # raw_triangle.py:156-181
        if not save_empty and len(self) == 0:
            return
        Path(path).parent.mkdir(parents=True, exist_ok=True)
        names = ["x1", "y1", "z1", "x2", "y2", "z2", "x3", "y3", "z3", "opacity"] + [f"f_dc_{i}" for i in range(3)]
        cols = [self.vertex.reshape(-1, 9), self.opacity.reshape(-1, 1), self.shs[:, :3]]
        if save_extra:
            names += [f"f_rest_{i}" for i in range(self.shs.shape[1] - 3)]
            cols.append(self.shs[:, 3:])
        rows = np.ascontiguousarray(np.concatenate(cols, axis=1), dtype="<f4")
        header = ["ply", "format binary_little_endian 1.0", f"element vertex {len(rows)}"] + [f"property float {n}" for n in names] + ["end_header"]
        with open(path, "wb") as f:
            f.write(("\n".join(header) + "\n").encode("ascii"))
            f.write(rows.tobytes())

    # ---- GLB ---------------------------------------------------------------------------------------------------------------
    def saveGLB(self, path, save_empty=False, save_back=True, process=False):  # raw_triangle.py:183-209
        if not save_empty and len(self) == 0:
            return
        if process:
            raise NotImplementedError("process=True (trimesh's vertex merging) has no counterpart here; the reference's default is False")
        Path(path).parent.mkdir(parents=True, exist_ok=True)
        P = len(self)
        pos = np.ascontiguousarray(self.vertex.reshape(-1, 3), dtype="<f4")
        rgba = np.concatenate([np.clip(SH2RGB(self.shs[:, :3]), 0, 1), 1 / (1 + np.exp(-self.opacity.reshape(-1, 1)))], axis=1)
        col = np.ascontiguousarray(np.repeat(np.round(rgba * 255).astype(np.uint8), 3, axis=0))  # one colour per face -> its three vertices
        faces = np.arange(P * 3, dtype="<u4").reshape(-1, 3)
        if save_back:
            faces = np.concatenate([faces, faces[:, ::-1]], axis=0)
        idx = np.ascontiguousarray(faces.reshape(-1), dtype="<u4")
        blobs, views, off = [], [], 0
        for arr, target in ((pos, 34962), (col, 34962), (idx, 34963)):
            raw = arr.tobytes()
            views.append({"buffer": 0, "byteOffset": off, "byteLength": len(raw), "target": target})
            raw += b"\x00" * (-len(raw) % 4)
            blobs.append(raw)
            off += len(raw)
        doc = {
            "asset": {"version": "2.0", "generator": "diff_recon_hip.raw_triangle"},
            "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"name": "geometry_0", "mesh": 0}],
            "meshes": [{"name": "geometry_0", "primitives": [{"attributes": {"POSITION": 0, "COLOR_0": 1}, "indices": 2, "mode": 4}]}],
            "buffers": [{"byteLength": off}], "bufferViews": views,
            "accessors": [
                {"bufferView": 0, "componentType": 5126, "count": int(len(pos)), "type": "VEC3",
                 "min": [float(x) for x in (pos.min(axis=0) if len(pos) else np.zeros(3))], "max": [float(x) for x in (pos.max(axis=0) if len(pos) else np.zeros(3))]},
                {"bufferView": 1, "componentType": 5121, "normalized": True, "count": int(len(col)), "type": "VEC4"},
                {"bufferView": 2, "componentType": 5125, "count": int(len(idx)), "type": "SCALAR"},
            ],
        }
        js = json.dumps(doc, separators=(",", ":")).encode("utf-8")
        js += b" " * (-len(js) % 4)
        binary = b"".join(blobs)
        with open(path, "wb") as f:
            f.write(struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(binary)))
            f.write(struct.pack("<I4s", len(js), b"JSON") + js)
            f.write(struct.pack("<I4s", len(binary), b"BIN\x00") + binary)

    def loadGLB(self, path):  # raw_triangle.py:211-223
        if not os.path.exists(path):
            print(f"[Warning] File {path} does not exist! From loadGLB function in RawTriangle class.")
        self.glb_path = path
        doc, binary = read_glb(path)
        mesh = next((m for m in doc["meshes"] if m.get("name") == "geometry_0"), doc["meshes"][0])
        prim = mesh["primitives"][0]
        pos = _accessor(doc, binary, prim["attributes"]["POSITION"]).astype(np.float64)
        col = _accessor(doc, binary, prim["attributes"]["COLOR_0"])
        acc = doc["accessors"][prim["attributes"]["COLOR_0"]]
        rgba = col.astype(np.float64) / {5121: 255.0, 5123: 65535.0}.get(acc["componentType"], 1.0)
        if rgba.shape[1] == 3:
            rgba = np.concatenate([rgba, np.ones((len(rgba), 1))], axis=1)
        triangles = pos.reshape(-1, 3, 3)  # the vertices are not shared: face i owns vertices 3 i .. 3 i + 2 (front faces come first)
        face_rgba = rgba[::3][: len(triangles)]
        eps = 1e-5
        self.vertex = triangles
        self.opacity = -np.log(1 / np.clip(face_rgba[:, 3:], eps, 1 - eps) - 1)
        self.shs = RGB2SH(face_rgba[:, :3])
        return self


def read_glb(path):
    """(JSON document, BIN chunk) of a glTF 2.0 binary file."""
    raw = open(path, "rb").read()
    magic, version, total = struct.unpack_from("<4sII", raw, 0)
    if magic != b"glTF" or version != 2 or total != len(raw):
        raise ValueError(f"{path}: not a glTF 2.0 binary container")
    off, doc, binary = 12, None, b""
    while off < total:
        n, kind = struct.unpack_from("<I4s", raw, off)
        chunk = raw[off + 8:off + 8 + n]
        if kind == b"JSON":
            doc = json.loads(chunk.decode("utf-8"))
        elif kind == b"BIN\x00":
            binary = chunk
        off += 8 + n
    if doc is None:
        raise ValueError(f"{path}: GLB without a JSON chunk")
    return doc, binary


def _accessor(doc, binary, i) -> np.ndarray:
    a = doc["accessors"][i]
    v = doc["bufferViews"][a["bufferView"]]
    comp = {5120: "i1", 5121: "u1", 5122: "<i2", 5123: "<u2", 5125: "<u4", 5126: "<f4"}[a["componentType"]]
    width = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4}[a["type"]]
    start = v.get("byteOffset", 0) + a.get("byteOffset", 0)
    stride = v.get("byteStride", 0)
    item = np.dtype(comp).itemsize * width
    if stride and stride != item:
        rows = [np.frombuffer(binary, dtype=comp, count=width, offset=start + k * stride) for k in range(a["count"])]
        return np.stack(rows) if rows else np.zeros((0, width), comp)
    return np.frombuffer(binary, dtype=comp, count=a["count"] * width, offset=start).reshape(a["count"], width)

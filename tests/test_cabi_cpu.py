"""CPU tests of the drop-in boundary (no GPU, no compute calls): libts2d.so loads and exports every symbol that
include/ts2d.h declares; the Python package mirrors the reference's interface (names, field order, errors)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ts2d.h")
HEADERS = [os.path.join(ROOT, "include", n) for n in sorted(os.listdir(os.path.join(ROOT, "include"))) if n.endswith(".h")]


@pytest.fixture(scope="module")
def lib(hip_lib_built):
    return ctypes.CDLL(hip_lib_built)


def _declared_functions():
    names = set()
    for h in HEADERS:  # every header under include/: ts2d.h (rasterizer), ts_loss.h (photometric loss), ts_knn.h (nearest neighbours)
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b((?:ts2d|tsl|tsk|tsm|tso)_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_header_declares_the_expected_entry_points():
    names = _declared_functions()
    for must in ("ts2d_forward_bin", "ts2d_forward_render", "ts2d_backward", "ts2d_geometry_state_bytes",
                 "ts2d_binning_state_bytes", "ts2d_image_state_bytes", "ts2d_backward_scratch_bytes", "ts2d_last_error",
                 "ts2d_version", "ts2d_forward_speculative", "ts2d_binning_capacity", "ts2d_instance_capacity_hint", "ts2d_sh_grad_expand", "tsl_workspace_bytes",
                 "tsl_photometric_forward", "tsl_photometric_backward", "tsk_workspace_bytes", "tsk_mean_dist3",
                 "tsk_nearest_other", "tsm_training_statistic", "tso_adam_step"):
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    for name in _declared_functions():
        assert hasattr(lib, name), f"libts2d.so does not export {name}"


def test_product_library_reads_no_environment_and_has_one_blend_path(hip_lib_built):
    """libts2d.so carries one blend kernel family per variant (the lane-group kernels) and no run-time switch: the measurement
    kernels of earlier rounds and their TS2D_BLEND / TS2D_BWD / TS2D_ABLATE variables exist only in tools/bin/libts2d_lab.so."""
    import subprocess
    blob = open(hip_lib_built, "rb").read()
    for name in (b"TS2D_BLEND", b"TS2D_BWD", b"TS2D_ABLATE"):
        assert name not in blob, name
    syms = subprocess.run(["nm", "--defined-only", hip_lib_built], capture_output=True, text=True).stdout  # static table: the launchers are not exported
    launchers = sorted(set(re.findall(r"ts_launch_render\w*?(?=RK)", syms)))
    assert launchers, "nm found no blend launchers"
    assert all("group" in n for n in launchers), launchers


def test_product_library_exports_only_the_declared_c_abi(hip_lib_built):
    """libts2d.so is built with -fvisibility=hidden: its dynamic symbol table holds the entry points that include/*.h declares and nothing
    else of ours -- no internal ts_* C++ symbol, no test hook (ts2d_test_*, ts2d_debug_*, ts2d_lab_*: csrc/ts2d_lab.h, lab library only),
    no rocPRIM."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", hip_lib_built], capture_output=True, text=True).stdout
    exported = [l.split()[-1] for l in out.splitlines() if l.split()[-2:-1] and l.split()[-2] in ("T", "D", "B", "R")]
    ours = [n for n in exported if not n.startswith("__hip_")]  # __hip_cuid_*: the toolchain's per-object markers
    declared = set(_declared_functions())
    assert set(ours) == declared, (sorted(set(ours) - declared), sorted(declared - set(ours)))
    everything = subprocess.run(["nm", "-D", hip_lib_built], capture_output=True, text=True).stdout
    assert "rocprim" not in everything.lower()
    assert not re.search(r"ts2d_(test|debug|lab)_", everything)


def test_binning_capacity_inverts_the_size_query(lib):
    """The layout of a binning buffer follows from its size: capacity(bytes(N)) >= N, bytes(capacity(b)) <= b, monotone."""
    lib.ts2d_binning_state_bytes.restype = ctypes.c_size_t
    lib.ts2d_binning_state_bytes.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32]
    lib.ts2d_binning_capacity.restype = ctypes.c_int64
    lib.ts2d_binning_capacity.argtypes = [ctypes.c_size_t, ctypes.c_int32, ctypes.c_int32]
    last = -1
    for n in (0, 1, 63, 4096, 4097, 262_145, 4_610_735, 50_000_000):
        b = lib.ts2d_binning_state_bytes(n, 1920, 1080)
        c = lib.ts2d_binning_capacity(b, 1920, 1080)
        assert c >= n and lib.ts2d_binning_state_bytes(c, 1920, 1080) <= b
        assert lib.ts2d_binning_state_bytes(c + 1, 1920, 1080) > b  # the LARGEST count that fits
        assert c >= last
        last = c
    assert lib.ts2d_binning_capacity(0, 1920, 1080) == 0 and lib.ts2d_binning_capacity(100, 64, 64) == 0


def test_state_size_queries_are_monotone_and_aligned(lib):
    lib.ts2d_geometry_state_bytes.restype = ctypes.c_size_t
    lib.ts2d_geometry_state_bytes.argtypes = [ctypes.c_int32]
    lib.ts2d_image_state_bytes.restype = ctypes.c_size_t
    lib.ts2d_image_state_bytes.argtypes = [ctypes.c_int32, ctypes.c_int32]
    lib.ts2d_backward_scratch_bytes.restype = ctypes.c_size_t
    lib.ts2d_backward_scratch_bytes.argtypes = [ctypes.c_int32]
    a, b = lib.ts2d_geometry_state_bytes(1000), lib.ts2d_geometry_state_bytes(2000)
    assert 1000 * 64 <= a < b  # at least the 64-byte render record per triangle
    assert lib.ts2d_image_state_bytes(1920, 1080) >= 1920 * 1080 * 8 + 120 * 68 * 8
    assert lib.ts2d_backward_scratch_bytes(1000) >= 1000 * 64
    assert lib.ts2d_geometry_state_bytes(0) < 4096


def test_null_arguments_are_rejected_not_crashed(lib):
    lib.ts2d_forward_bin.restype = ctypes.c_int
    lib.ts2d_last_error.restype = ctypes.c_char_p
    rc = lib.ts2d_forward_bin(None, None, 0, None, None, None, None)
    assert rc == 1 and b"null" in lib.ts2d_last_error()


def test_triangle_count_beyond_the_id_bits_is_refused(lib):
    """The instance lists keep four bits of each value for the quadrant mask (csrc/ts2d_support.h): 2^28 triangles are a capacity error of the
    argument check, before any memory is touched."""
    from diff_triangle_rasterization_2D import _C
    lib.ts2d_forward_bin.restype = ctypes.c_int
    lib.ts2d_last_error.restype = ctypes.c_char_p
    cam, geom = _C._Camera(), _C._Geometry()
    cam.width, cam.height = 64, 64
    geom.P, geom.C, geom.M, geom.gamma = 1 << 28, 3, 0, 1.0
    n = ctypes.c_int64(0)
    rc = lib.ts2d_forward_bin(ctypes.byref(cam), ctypes.byref(geom), 0, None, None, ctypes.byref(n), None)
    assert rc == 3 and b"2^28" in lib.ts2d_last_error()  # TS2D_ERR_CAPACITY


# ---- Python surface: same names / order / errors as R2D/diff_triangle_rasterization_2D/__init__.py ----------
REFERENCE_SETTINGS_FIELDS = ("image_width", "image_height", "tanfovx", "tanfovy", "viewmatrix", "projmatrix", "campos",
                             "sh_degree", "gamma", "scale_modifier", "background_depth", "background", "back_culling",
                             "rich_info", "debug")  # reference __init__.py:28-46


def _settings(**kw):
    from diff_triangle_rasterization_2D import TriangleRasterizationSettings
    base = dict(image_width=32, image_height=32, tanfovx=0.3, tanfovy=0.3, viewmatrix=torch.eye(4), projmatrix=torch.eye(4),
                campos=torch.zeros(3), sh_degree=0, gamma=1.0, scale_modifier=1.0, background_depth=10.0,
                background=torch.zeros(3), back_culling=False, rich_info=True, debug=False)
    base.update(kw)
    return TriangleRasterizationSettings(**base)


def test_settings_namedtuple_matches_reference(hip_lib_built):
    from diff_triangle_rasterization_2D import TriangleRasterizationSettings
    assert TriangleRasterizationSettings._fields == REFERENCE_SETTINGS_FIELDS


def test_rasterizer_module_surface(hip_lib_built):
    from diff_triangle_rasterization_2D import TriangleRasterizer
    rs = _settings()
    r = TriangleRasterizer(rs)
    assert isinstance(r, torch.nn.Module) and r.raster_settings is rs  # read by triangle_renderer.py:77
    v, c2d, op = torch.rand(4, 3, 3), torch.zeros(4, 2), torch.rand(4, 1)
    with pytest.raises(Exception, match="excatly one"):  # reference __init__.py:180-181 (sic)
        r(v, c2d, op)
    with pytest.raises(Exception, match="excatly one"):
        r(v, c2d, op, shs=torch.rand(4, 1, 3), feature=torch.rand(4, 3))


def test_argument_checks_mirror_reference_errors(hip_lib_built):
    """extension_interface.cu:53-81: the same conditions raise RuntimeError with the same messages; tensors on the CPU
    are refused loudly (there is no CPU fallback in the product path)."""
    from diff_triangle_rasterization_2D import _C
    E = torch.Tensor([])

    def call(vertex=None, shs=None, feature=E, opacity=None, gamma=1.0, background=None, view=None):
        vertex = torch.rand(4, 3, 3) if vertex is None else vertex
        shs = torch.rand(4, 1, 3) if shs is None else shs
        opacity = torch.rand(4, 1) if opacity is None else opacity
        background = torch.zeros(3) if background is None else background
        view = torch.eye(4) if view is None else view
        return _C.rasterize_triangles(32, 32, 0.3, 0.3, view, torch.eye(4), torch.zeros(3), 0, gamma, 1.0, 10.0, background,
                                      vertex, shs, feature, opacity, False, True, False)

    with pytest.raises(RuntimeError, match=r"vertex must have dimensions \(num_points, 3, 3\)"):
        call(vertex=torch.rand(4, 3, 2))
    with pytest.raises(RuntimeError, match="num_channels can't be larger than MAX_CHANNELS"):
        call(shs=E, feature=torch.rand(4, 5), background=torch.zeros(5))
    with pytest.raises(RuntimeError, match="background must have the same number of channels"):
        call(background=torch.zeros(2))
    with pytest.raises(RuntimeError, match="gamma must be larger than 0"):
        call(gamma=-1.0)
    with pytest.raises(RuntimeError, match="input tensors must be contiguous"):
        call(view=torch.arange(16.0).view(4, 4).t())  # a transposed view, like camera.py:112 before .contiguous()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        call()


def test_missing_library_fails_loudly(tmp_path, hip_lib_built):
    """The product must not degrade silently when the HIP library is absent."""
    import shutil
    import subprocess
    import sys
    pkg_src = os.path.join(ROOT, "triangle-splatting_amd", "diff_triangle_rasterization_2D")
    pkg = tmp_path / "diff_triangle_rasterization_2D"
    shutil.copytree(pkg_src, pkg, ignore=shutil.ignore_patterns("*.so", "__pycache__"))
    r = subprocess.run([sys.executable, "-c", "import diff_triangle_rasterization_2D"], cwd=tmp_path, capture_output=True,
                       text=True, env={**os.environ, "PYTHONPATH": str(tmp_path)})
    assert r.returncode != 0 and "libts2d.so" in r.stderr and "no CPU fallback" in r.stderr


def test_compiled_reference_side_binding_loads(hip_lib_built):
    """bindings/_ts2d_torch_C.so (torch C++ extension with the reference's ext.cpp signatures, linked against libts2d.so) builds
    without a GPU and exports the reference's two entry points (R2D/ext.cpp:4-9) plus, since round 6, the package's own two (`*_ex`: the same
    calls with the variant / capacity / preallocated-output arguments the package adds; diff_triangle_rasterization_2D/_C.py prefers them over ctypes)."""
    import importlib.util
    import torch  # noqa: F401
    spec = importlib.util.spec_from_file_location("ts2d_build_ext", os.path.join(ROOT, "triangle-splatting_amd", "bindings", "build_torch_ext.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    so = mod.build()
    spec = importlib.util.spec_from_file_location("_ts2d_torch_C", so)
    ext = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ext)
    assert sorted(n for n in dir(ext) if not n.startswith("_")) == ["rasterize_triangles", "rasterize_triangles_backward", "rasterize_triangles_backward_ex",
                                                                    "rasterize_triangles_ex"]
    from diff_triangle_rasterization_2D import _C
    assert _C.binding() == ("ctypes" if (os.environ.get("TS2D_BINDING") == "ctypes" or os.environ.get("TS2D_LIBRARY_PATH")) else "compiled")
    with pytest.raises(RuntimeError):  # CPU tensors: the checks pass, the library refuses host pointers or the device guard raises
        z = torch.zeros
        ext.rasterize_triangles(8, 8, 0.3, 0.3, z(4, 4), z(4, 4), z(3), 0, 1.0, 1.0, 1.0, z(3), z(2, 3, 3), z(2, 1, 3), torch.empty(0), z(2, 1),
                                False, True, False)


def test_ctypes_structures_match_the_c_headers(tmp_path, hip_lib_built):
    """The ctypes mirrors of the C ABI's structs (diff_triangle_rasterization_2D/_C.py, diff_recon_hip/optim.py) against the headers themselves:
    a probe compiled with gcc from include/*.h prints sizeof / offsetof of every field; a field added to a header but not to its mirror (or
    the other way round) fails here, without a GPU."""
    import subprocess
    from diff_triangle_rasterization_2D import _C
    from diff_recon_hip import optim
    mirrors = {"ts2d_camera": _C._Camera, "ts2d_geometry": _C._Geometry, "ts2d_forward_out": _C._ForwardOut, "ts2d_loss_grads": _C._LossGrads,
               "ts2d_backward_out": _C._BackwardOut, "ts2d_state": _C._State, "tso_adam_slice": optim._Slice}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "ts2d.h"', '#include "ts_optim.h"', 'int main(void) {']
    for cname, mirror in mirrors.items():
        lines.append(f'printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in mirror._fields_:
            lines.append(f'printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append("return 0; }")
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    got = {tuple(l.split()[:2]): int(l.split()[2]) for l in out.splitlines()}
    for cname, mirror in mirrors.items():
        assert got[(cname, "size")] == ctypes.sizeof(mirror), cname
        for fname, _ in mirror._fields_:
            assert got[(cname, fname)] == getattr(mirror, fname).offset, (cname, fname)
    # and the header has no field the mirror lacks: the sizes above already say so for trailing fields; count the declarators for the rest
    for cname, mirror, header in [("ts2d_geometry", _C._Geometry, "ts2d.h"), ("tso_adam_slice", optim._Slice, "ts_optim.h")]:
        text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", header)).read(), flags=re.S)
        body = re.search(r"typedef struct " + cname + r"\s*\{(.*?)\}\s*" + cname + r"\s*;", text, flags=re.S).group(1)
        declared = [n for stmt in body.split(";") for n in re.findall(r"\**\s*([A-Za-z_]\w*)\s*(?:,|$)", stmt.strip().split(None, 1)[-1] if stmt.strip() else "")]
        assert len(declared) == len(mirror._fields_), (cname, declared, [f for f, _ in mirror._fields_])


def test_binning_state_bytes_grow_with_the_capacity_and_invert(lib):
    """The forward and the backward both carve the binning state from the BUFFER'S SIZE (ts2d_binning_capacity inverts ts2d_binning_state_bytes by
    bisection), so the bytes must never shrink when the capacity grows -- also across the capacities at which the instance sort changes its chunk
    length (csrc/ts2d_common.h: ts_instance_chunk; the tables are sized for the shortest chunks at every capacity for exactly this reason)."""
    lib.ts2d_binning_state_bytes.restype = ctypes.c_size_t
    lib.ts2d_binning_state_bytes.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32]
    lib.ts2d_binning_capacity.restype = ctypes.c_int64
    lib.ts2d_binning_capacity.argtypes = [ctypes.c_size_t, ctypes.c_int32, ctypes.c_int32]
    for W, H in ((1920, 1080), (256, 256), (37, 5)):
        caps = sorted(set(list(range(0, 70000, 997)) + list(range(2_499_000, 2_501_001, 125)) + [1 << k for k in range(4, 27)] +
                          [(1 << k) + 1 for k in range(4, 27)] + [4_610_000, 12_600_000, 23_000_000]))
        prev = -1
        for n in caps:
            b = lib.ts2d_binning_state_bytes(n, W, H)
            assert b >= 16 * n and b >= prev, (n, b, prev)
            prev = b
        for n in (0, 1, 1023, 1024, 1025, 65_537, 2_499_999, 2_500_000, 2_500_001, 4_610_000):
            b = lib.ts2d_binning_state_bytes(n, W, H)
            cap = lib.ts2d_binning_capacity(b, W, H)
            assert cap >= n and lib.ts2d_binning_state_bytes(cap, W, H) <= b, (n, b, cap)  # the largest capacity whose carving fits the buffer

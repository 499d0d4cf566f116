"""Builds the reference-side torch extension over libts2d.so (bindings/ts2d_torch_ext.cpp) in-tree with hipcc.

    python triangle-splatting_amd/bindings/build_torch_ext.py [--force]     -> bindings/_ts2d_torch_C.so (git-ignored; travels with gpurun)

The module exposes `rasterize_triangles` / `rasterize_triangles_backward` with the reference's pybind signatures
(R2D/ext.cpp:4-9); it links libts2d.so through an $ORIGIN-relative rpath.  hipcc compiles it without a GPU."""
from __future__ import annotations

import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "ts2d_torch_ext.cpp")
OUT = os.path.join(HERE, "_ts2d_torch_C.so")
LIBDIR = os.path.join(os.path.dirname(HERE), "diff_triangle_rasterization_2D")


def build(force: bool = False) -> str:
    import torch  # include / library directories only
    deps = [SRC, os.path.join(HERE, "..", "..", "include", "ts2d.h"), os.path.abspath(__file__)]
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(d) for d in deps):
        return OUT
    ti = os.path.dirname(torch.__file__)
    cmd = ["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-DTORCH_EXTENSION_NAME=_ts2d_torch_C", f"-I{ti}/include", f"-I{ti}/include/torch/csrc/api/include",
           f"-I{sysconfig.get_paths()['include']}", "-I/opt/rocm/include", "-w", "-x", "c++", SRC, "-x", "none", f"-L{ti}/lib", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10",
           "-lc10_hip", "-ltorch_python", f"-L{LIBDIR}", "-lts2d", "-Wl,-rpath,$ORIGIN/../diff_triangle_rasterization_2D", "-o", OUT]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building the torch extension failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    return OUT


if __name__ == "__main__":
    print(build("--force" in sys.argv))

"""Builds the REFERENCE's own extensions for gfx950 into oracle/_ref/ -- test infrastructure only.

    python oracle/build_ref.py            # needs /root/reference (build container only); outputs oracle/_ref/*.so

What this is: the reference's three torch extensions
    submodules/diff-triangle-rasterization-2D   -> oracle/_ref/_ref2d_C.so   (rasterize_triangles, rasterize_triangles_backward)
    submodules/diff-triangle-rasterization-3D   -> oracle/_ref/_ref3d_C.so
    submodules/simple-knn                       -> oracle/_ref/_refknn_C.so  (distCUDA2, nearestNeighbor)
plus _ref3d_scalar_C.so / _ref3d_nofma_C.so: the 3D extension again with -fno-slp-vectorize / -ffp-contract=off, and _ref2d_nofma_C.so: the 2D
extension again with -ffp-contract=off (see CODEGEN_FLAGS)
compiled from the sources WHERE THEY LIE under /root/reference, with the toolchain this image ships for exactly that
purpose: ROCm's `hipify-perl` (CUDA -> HIP source translation; the same step torch.utils.cpp_extension performs when a CUDA
extension is installed on a ROCm build of PyTorch), `hipcc`, hipCUB / rocThrust, and the installed PyTorch headers and
libraries.  Nothing of the reference is copied into the repository: the translated sources are piped into a scratch
directory under oracle/_ref/ (git-ignored), compiled, and deleted again; only the shared objects remain, and those travel
to the GPU box with the snapshot.  The reference tree is never written to (hipify-perl writes to stdout).

Every deviation from "the reference's sources as they are", all applied to the translated stream, none to the reference:
  1. hipify-perl's CUDA -> HIP renames (cuda* -> hip*, cub:: -> hipcub::, <cooperative_groups.h> -> <hip/hip_cooperative_groups.h>, ...);
  2. four include lines that have no HIP counterpart and whose contents the code does not use are dropped:
     <cooperative_groups/reduce.h>, <cub/device/device_radix_sort.cuh> (covered by <hipcub/hipcub.hpp>), and the two empty
     includes hipify-perl leaves behind for "device_launch_parameters.h";
  3. at::cuda::OptionalCUDAGuard / <c10/cuda/CUDAGuard.h> -> c10::hip::OptionalHIPGuardMasqueradingAsCUDA /
     <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>: the mapping PyTorch's own hipify applies to every extension.
No header, library, tool or generated file is written to stand in for something the image lacks.

Use: tests/test_reference_gpu.py compares the reference build with the CPU oracle (which pins the oracle) and with the HIP
product path directly; bench.py times it beside the product path ("reference_gpu").  The product never loads it.
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF = "/root/reference/submodules"
ARCH = "gfx950"

EXTENSIONS = {
    "_ref2d_C": ("diff-triangle-rasterization-2D", ["src/forward.cu", "src/backward.cu", "src/rasterizer.cu", "src/extension_interface.cu", "ext.cpp"],
                 ["src/auxiliary.h", "src/backward.h", "src/config.h", "src/extension_interface.h", "src/forward.h", "src/param_struct.h",
                  "src/rasterizer.h"]),
    "_ref3d_C": ("diff-triangle-rasterization-3D", ["src/forward.cu", "src/backward.cu", "src/rasterizer.cu", "src/extension_interface.cu", "ext.cpp"],
                 ["src/auxiliary.h", "src/backward.h", "src/config.h", "src/extension_interface.h", "src/forward.h", "src/param_struct.h",
                  "src/rasterizer.h"]),
    # the 2D extension again with -ffp-contract=off (round 5): the product's and the oracle's per-triangle kernels are compiled without
    # contraction, so against THIS build num_rendered, radii, the sorted instance list and the tile ranges are compared for equality
    # (tests/test_reference_gpu.py::test_integer_chain_equals_the_references_uncontracted_build)
    "_ref2d_nofma_C": ("diff-triangle-rasterization-2D", ["src/forward.cu", "src/backward.cu", "src/rasterizer.cu", "src/extension_interface.cu", "ext.cpp"],
                       ["src/auxiliary.h", "src/backward.h", "src/config.h", "src/extension_interface.h", "src/forward.h", "src/param_struct.h",
                        "src/rasterizer.h"]),
    # the 3D extension twice more, same sources, other code-generation switches: its per-pixel ray / plane arithmetic is so
    # ill-conditioned that WHICH products the compiler fuses into FMAs decides gradients of grazing triangles outright
    # (tests/triage/noise_floor.py, DESIGN.md "noise floor"); these builds measure the reference's distance to itself
    "_ref3d_scalar_C": ("diff-triangle-rasterization-3D", ["src/forward.cu", "src/backward.cu", "src/rasterizer.cu", "src/extension_interface.cu", "ext.cpp"],
                        ["src/auxiliary.h", "src/backward.h", "src/config.h", "src/extension_interface.h", "src/forward.h", "src/param_struct.h",
                         "src/rasterizer.h"]),
    "_ref3d_nofma_C": ("diff-triangle-rasterization-3D", ["src/forward.cu", "src/backward.cu", "src/rasterizer.cu", "src/extension_interface.cu", "ext.cpp"],
                       ["src/auxiliary.h", "src/backward.h", "src/config.h", "src/extension_interface.h", "src/forward.h", "src/param_struct.h",
                        "src/rasterizer.h"]),
    "_refknn_C": ("simple-knn", ["simple_knn.cu", "interface.cu", "ext.cpp"], ["auxiliary.h", "interface.h", "simple_knn.h"]),
}

# hipcc's defaults are -ffp-contract=fast plus the SLP vectorizer (which turns pairs of products into v_pk_mul_f32, i.e. keeps
# them OUT of FMAs): "_scalar" switches the vectorizer off (every a*b+c fuses), "_nofma" switches contraction off
CODEGEN_FLAGS = {"_ref3d_scalar_C": ["-fno-slp-vectorize"], "_ref3d_nofma_C": ["-ffp-contract=off"], "_ref2d_nofma_C": ["-ffp-contract=off"]}

DROP = ("cooperative_groups/reduce.h", "<cub/device/device_radix_sort.cuh>", '#include ""', "#include <>")


def available() -> bool:
    return os.path.isdir(REF) and shutil.which("hipify-perl", path="/opt/rocm/bin:" + os.environ.get("PATH", "")) is not None


def _translate(src: str, dst: str):
    hipify = shutil.which("hipify-perl", path="/opt/rocm/bin:" + os.environ.get("PATH", ""))
    text = subprocess.run([hipify, src], capture_output=True, text=True, check=True).stdout
    lines = [l for l in text.splitlines() if not any(d in l for d in DROP)]
    text = "\n".join(lines) + "\n"
    text = text.replace("c10/cuda/CUDAGuard.h", "ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h")
    text = text.replace("at::cuda::OptionalCUDAGuard", "c10::hip::OptionalHIPGuardMasqueradingAsCUDA")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst, "w") as f:
        f.write(text)


def _newest_source(name: str) -> float:
    sub, srcs, hdrs = EXTENSIONS[name]
    return max([os.path.getmtime(os.path.join(REF, sub, p)) for p in srcs + hdrs] + [os.path.getmtime(os.path.abspath(__file__))])


def build_one(name: str, force: bool = False) -> str:
    import torch  # only for the include / library directories
    sub, srcs, hdrs = EXTENSIONS[name]
    so = os.path.join(OUT, name + ".so")
    if not force and os.path.exists(so) and os.path.getmtime(so) >= _newest_source(name):
        return so
    work = os.path.join(OUT, "_scratch_" + name)
    shutil.rmtree(work, ignore_errors=True)
    ti = os.path.join(os.path.dirname(torch.__file__))
    try:
        for p in srcs + hdrs:
            _translate(os.path.join(REF, sub, p), os.path.join(work, re.sub(r"\.cu$", ".hip", p)))
        flags = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
                 f"-DTORCH_EXTENSION_NAME={name}", f"-I{ti}/include", f"-I{ti}/include/torch/csrc/api/include",
                 f"-I{sysconfig.get_paths()['include']}", f"-I{work}", "-w", *CODEGEN_FLAGS.get(name, [])]
        objs = []

        def compile_one(p):
            src = os.path.join(work, re.sub(r"\.cu$", ".hip", p))
            obj = src + ".o"
            subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-x", "hip", "-c", src, "-o", obj], check=True, cwd=work)
            return obj

        with ThreadPoolExecutor(max_workers=5) as ex:
            objs = list(ex.map(compile_one, srcs))
        subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, f"-L{ti}/lib", "-ltorch", "-ltorch_cpu",
                        "-ltorch_hip", "-lc10", "-lc10_hip", "-ltorch_python", "-o", so], check=True)
    finally:
        shutil.rmtree(work, ignore_errors=True)  # the translated sources never outlive the build
    return so


def build(force: bool = False):
    if not available():
        raise RuntimeError("/root/reference or hipify-perl is not present: the reference build exists only as the prebuilt oracle/_ref/*.so")
    os.makedirs(OUT, exist_ok=True)
    with ThreadPoolExecutor(max_workers=len(EXTENSIONS)) as ex:  # hipcc on torch headers is slow: overlap the three extensions
        return list(ex.map(lambda n: build_one(n, force), EXTENSIONS))


if __name__ == "__main__":
    for so in build("--force" in sys.argv):
        print(so)

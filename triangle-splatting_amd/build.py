"""Builds libts2d.so (the C-ABI HIP library of include/ts2d.h) for gfx950 with hipcc, in-tree.

    python triangle-splatting_amd/build.py [--force] [--verbose] [--lab]

Output: triangle-splatting_amd/diff_triangle_rasterization_2D/libts2d.so (git-ignored, travels with gpurun).
--lab builds tools/bin/libts2d_lab.so instead: the same objects + the measurement kernels of earlier rounds (tools/lab/: render.hip, render3d.hip,
render_q8.hip, lab_hooks.hip) and api.hip compiled with -DTS2D_LAB, which reads TS2D_BLEND / TS2D_BWD / TS2D_ABLATE.  The product library contains
one blend path per variant and reads no environment; only tools/ and tests/ load the lab library (TS2D_LIBRARY_PATH, see _C.py).
hipcc cross-compiles without a GPU.  Per-file flags matter:
  * preprocess.hip is built with -ffp-contract=off (bit-comparable integer state, see the file header);
  * render.hip uses the default fast contraction and hardware float atomics (-munsafe-fp-atomics).
"""
from __future__ import annotations

import argparse
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "diff_triangle_rasterization_2D")
OBJ_DIR = os.path.join(HERE, "build")
LIB = os.path.join(OUT_DIR, "libts2d.so")
ARCH = "gfx950"

COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function",
          "-Wno-unused-result", "-DNDEBUG", "-fvisibility=hidden"]  # exports = what include/*.h declares (api.hip), nothing else
COMMON += os.environ.get("TS2D_EXTRA_FLAGS", "").split()  # profiling builds: -DTS2D_ABLATION, -DTS2D_STATS (use --force)
SOURCES = {
    "preprocess.hip": ["-ffp-contract=off"],
    "preprocess3d.hip": ["-ffp-contract=off"],
    "shgrad.hip": ["-ffp-contract=off"],
    "photometric.hip": [],
    "depth_normal.hip": ["-ffp-contract=off"],
    "aux_losses.hip": ["-ffp-contract=off"],
    "resample.hip": ["-ffp-contract=off"],
    "knn.hip": [],
    "model_update.hip": [],
    "optim.hip": ["-ffp-contract=off"],
    "binning.hip": [],
    "select.hip": [],
    # the 2D blend kernels: ONE source, two translation units (TSG_PART), so that each kernel gets the machine-scheduler strategy it measured best
    # with (round 5, profiles/r05_sched_strategies.txt: max-ilp +1.3 % for the forward, -1 % for the backward; round 6: profiles/r06_blend_ab.txt)
    "render_group.hip@fwd": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-fno-slp-vectorize", "-DTSG_PART=1", "-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "render_group.hip@bwd": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-fno-slp-vectorize", "-DTSG_PART=2"],
    "render3d_group.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-fno-slp-vectorize"],
    "api.hip": [],
}
LAB_SOURCES = {  # measurement kernels: libts2d_lab.so only
    "render.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-fno-slp-vectorize"],
    "render3d.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-fno-slp-vectorize"],
    "render_q8.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-fno-slp-vectorize"],
    "lab_hooks.hip": [],  # sort / scan test hooks + their rocPRIM comparators (csrc/ts2d_lab.h)
    "api.hip": ["-DTS2D_LAB"],
}
LAB_LIB = os.path.join(os.path.dirname(HERE), "tools", "bin", "libts2d_lab.so")
LAB_SRC = os.path.join(os.path.dirname(HERE), "tools", "lab")  # render.hip, render3d.hip, render_q8.hip, lab_hooks.hip: measurement kernels of rounds 1-3 and the
                                                                # test hooks -- out of the product's csrc/ since round 6; they include csrc's headers (-I)
HEADERS = ["ts2d_common.h", "ts2d_lab.h", "ts2d_math.h", "ts2d_wave.h", "ts2d_group.h", "ts2d_support.h", "ts2d_sh.h", "ts2d_stage.h", "ts2d_preprocess_launch.h", "ts2d_imgops.h", "ts2d_select.h", os.path.join("..", "..", "include", "ts2d.h"),
           os.path.join("..", "..", "include", "ts_loss.h"),
           os.path.join("..", "..", "include", "ts_knn.h"),
           os.path.join("..", "..", "include", "ts_model.h"),
           os.path.join("..", "..", "include", "ts_optim.h")]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libts2d.so cannot be built (no CPU fallback exists by design)")


_TOOLCHAIN_ID = {}


def _toolchain_id(cc: str) -> str:
    if cc not in _TOOLCHAIN_ID:
        r = subprocess.run([cc, "--version"], capture_output=True, text=True)
        _TOOLCHAIN_ID[cc] = hashlib.sha1((r.stdout + r.stderr).encode()).hexdigest()
    return _TOOLCHAIN_ID[cc]


def _newest_header() -> float:
    return max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)


def build(force: bool = False, verbose: bool = False, lab: bool = False) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    cc = hipcc()
    hdr_t = max(_newest_header(), os.path.getmtime(os.path.abspath(__file__)))
    jobs, objs = [], []
    sources = dict(SOURCES)
    lib = LIB
    if lab:
        build(force, verbose)  # the product's objects are shared
        sources = {k: v for k, v in SOURCES.items() if k != "api.hip"}
        sources.update({("lab/" + k): v for k, v in LAB_SOURCES.items()})
        lib = LAB_LIB
        os.makedirs(os.path.join(OBJ_DIR, "lab"), exist_ok=True)
        os.makedirs(os.path.dirname(LAB_LIB), exist_ok=True)
    for src, extra in sources.items():
        src, _, part = src.partition("@")  # "file.hip@tag": the same source compiled into file_tag.o with its own flags
        s = os.path.join(CSRC, os.path.basename(src))
        if not os.path.exists(s):  # a lab-only source
            s = os.path.join(LAB_SRC, os.path.basename(src))
            extra = list(extra) + ["-I" + CSRC]
        o = os.path.join(OBJ_DIR, src.replace(".hip", ("_" + part if part else "") + ".o"))
        objs.append(o)
        cmd = [cc, *COMMON, *extra, "-c", s, "-o", o]
        # the object is only reused when it was produced by this very command line with this very compiler: a profiling build
        # (TS2D_EXTRA_FLAGS=-DTS2D_STATS ...) or a toolchain update must not leave its objects behind for the next plain build
        stamp, key = o + ".cmd", _toolchain_id(cc) + "\n" + " ".join(cmd)
        fresh = os.path.exists(o) and os.path.exists(stamp) and open(stamp).read() == key
        if force or not fresh or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_t):
            jobs.append((cmd, stamp, key))

    def compile_one(job):
        cmd, stamp, key = job
        if os.path.exists(stamp):
            os.remove(stamp)
        run(cmd)
        with open(stamp, "w") as f:
            f.write(key)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(compile_one, jobs))
    if jobs or force or not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(o) for o in objs):
        run([cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib, *objs])
    return lib


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--lab", action="store_true")
    a = ap.parse_args()
    print(build(a.force, a.verbose, a.lab))

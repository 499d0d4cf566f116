#!/usr/bin/env python
"""Static instruction mix of the blend kernels' step loops, priced with the issue costs measured in REAL shader cycles
(tools/valu_bench3.hip, profiles/r03_valu_microbench3.txt): fma / add / mul / and / xor 2 cycles per wave instruction on a SIMD, compare /
select / min / max / shift / bit-field / mad24 / cvt / bfi / or3 / every DPP form 4, exp / rcp / log / sqrt 8.  Runs in the build container
(compiles csrc/render_group.hip to assembly); the result is committed as profiles/r03_valu_mix.json and read by bench.py, which combines it
with the SQ counters of the same kernels (profiles/blend_pmc.json).

The priced sum is an UPPER estimate of the VALU time: the microbenchmark also shows that a select interleaved with FMAs issues at the full
rate (the pair costs 4.5 cycles, not 6.4), i.e. the half-rate classes only cost their 4 cycles in runs of their own kind (the DPP reduction
network is such a run).  SQ_INSTS_VALU x 2 cycles is the LOWER bound.  bench.py reports both.

    python tools/isa_mix.py > profiles/r03_valu_mix.json
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HALF = ("v_cmp", "v_cndmask", "v_min", "v_max", "v_med3", "v_bfe", "v_bfi", "v_lshl", "v_lshr", "v_ashr", "v_mad_i32", "v_mad_u32", "v_mul_u32", "v_mul_lo", "v_mul_hi",
        "v_readlane", "v_writelane", "v_readfirstlane", "v_mbcnt", "v_cvt", "v_or3", "v_and_or", "v_lshl_add", "v_lshl_or", "v_add_lshl", "v_perm", "v_alignbit")
TRANS = ("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")


def classify(line):
    op = line.split()[0]
    if not op.startswith("v_"):
        return "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "scratch_", "flat_")) else None
    if op.startswith(TRANS):
        return "trans"
    if "_dpp" in line.split(";")[0] or "sdwa" in line or op.startswith(HALF):
        return "half"
    return "full"


def innermost_loops(lines):
    """Line ranges of the innermost loops: the compiler annotates every basic block of a loop with `in Loop: Header=BBx_y` (and the header with
    `This Inner Loop Header`); a loop = the header block + every block annotated with it (the latch may sit in front of the header)."""
    labels = [i for i, l in enumerate(lines) if re.match(r"\.LBB\d+_\d+:|; %bb\.", l.strip())]
    out = []
    for i, l in enumerate(lines):
        if "This Inner Loop Header" not in l:
            continue
        hdr = i
        while not re.match(r"\.L(BB\d+_\d+):", lines[hdr].strip()):  # the annotation may continue over several comment lines
            hdr -= 1
        name = re.match(r"\.L(BB\d+_\d+):", lines[hdr].strip()).group(1)
        blocks = [hdr] + [j for j in labels if re.search(r"in Loop: Header=" + name + r"\b", lines[j])]
        ranges = []
        for start in blocks:
            nxt = [j for j in labels if j > start]
            ranges.append((start, (nxt[0] if nxt else len(lines)) - 1))
        out.append(ranges)
    return out


def main():
    text = ""
    with tempfile.TemporaryDirectory() as tmp:
        # the product's translation units with the product's flags (triangle-splatting_amd/build.py): since round 6 the 2D forward and backward are
        # compiled separately, the forward with -amdgpu-sched-strategy=max-ilp
        units = (("render_group.hip", ["-DTSG_PART=1", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]), ("render_group.hip", ["-DTSG_PART=2"]), ("render3d_group.hip", []))
        for name, extra in units:
            src = os.path.join(ROOT, "triangle-splatting_amd", "csrc", name)
            asm = os.path.join(tmp, "k.s")
            subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-DNDEBUG", "-mllvm",
                            "-amdgpu-atomic-optimizer-strategy=None", "-fno-slp-vectorize", *extra, "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", src,
                            "-o", asm], check=True, capture_output=True)
            text += open(asm).read() + "\n"
    res = {}
    # <rich_info, gamma == 1>: the headline instantiations first (bench.py reads render_fwd / render_bwd); then gamma != 1 (pow_nonneg = v_log + v_exp
    # + the reciprocal of backward.cu:443-447: round 5) and the 3D variant's kernels
    for kernel, key, steps_per_body in (("23render_fwd_group_kernelILb1ELb1E", "render_fwd", 8), ("23render_bwd_group_kernelILb1ELb1E", "render_bwd", 1),
                                        ("23render_fwd_group_kernelILb1ELb0E", "render_fwd_gamma_ne_1", 8), ("23render_bwd_group_kernelILb1ELb0E", "render_bwd_gamma_ne_1", 1),
                                        ("25render3d_fwd_group_kernelILb1ELb1E", "render3d_fwd", 8), ("25render3d_bwd_group_kernelILb1ELb1E", "render3d_bwd", 1),
                                        ("25render3d_bwd_group_kernelILb1ELb0E", "render3d_bwd_gamma_ne_1", 1)):
        m = re.search(r"^(_ZN\S*" + kernel + r"\S*):.*?s_endpgm", text, re.S | re.M)
        lines = m.group(0).split("\n")
        loops = innermost_loops(lines)
        # the step loop is the innermost loop with the most VALU instructions
        best = None
        for ranges in loops:
            c = {"full": 0, "half": 0, "trans": 0, "salu": 0, "lds": 0, "vmem": 0}
            for a, b in ranges:
                for l in lines[a:b + 1]:
                    l = l.strip()
                    if not l or l[0] in ";." or l.endswith(":"):
                        continue
                    k = classify(l)
                    if k:
                        c[k] += 1
            if best is None or c["full"] + c["half"] > best["full"] + best["half"]:
                best = c
        n = best["full"] + best["half"] + best["trans"]
        cyc = 2 * best["full"] + 4 * best["half"] + 8 * best["trans"]
        res[key] = {"steps_per_loop_body": steps_per_body, "valu_instructions_per_body": n, **best,
                    "priced_cycles_per_body": cyc, "priced_cycles_per_valu_instruction": round(cyc / n, 3),
                    "note": "static count of the step loop's body (the forward's body is one window of up to 8 steps incl. the statistics; the backward's "
                            "includes its serialised-accumulate path)"}
    res["costs"] = {"full": 2, "half": 4, "trans": 8, "source": "tools/valu_bench3.hip (s_memtime), profiles/r03_valu_microbench3.txt"}
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()

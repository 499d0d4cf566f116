"""fp32 noise floor of the geometry gradients at full size.

    python tests/triage/noise_floor.py [P W H D [seed]]      -> gpurun_out/noise_floor_<P>.json (copy to profiles/)

Three independent fp32 evaluations of the SAME algorithm on the SAME scene:
    R = the reference's own kernels built for gfx950 (oracle/_ref/_ref2d_C.so; FMA-contracted like any nvcc/hipcc build),
    O = the CPU oracle (contraction-free, per-triangle sums accumulated in fp64),
    H = the HIP product path (libts2d.so).
For each of the pairs (O,R), (H,R), (H,O) the script records the UN-budgeted relative L2 of dL_dvertex / dL_dcenter2D /
dL_dshs / dL_dopacity / image, the same after setting aside the k worst triangles (k = 0 ... 500), and how many pixels
took a different discrete decision (n_contrib, i.e. the T <= 1e-4 termination position) or differ by more than 1e-3 of the
image range (an alpha >= 1/255 flip moves a pixel by ~4e-3 of the blended colour behind it).
The reference's private image state is parsed with the layout of R2D/src/param_struct.h:85-103 (ranges uint2[WH],
n_contrib u32[WH], final_T f32[WH], each aligned to 128 bytes)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd"), os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402
import torch  # noqa: E402

import helpers  # noqa: E402
import ref_build  # noqa: E402
import synthetic  # noqa: E402

KS = (0, 5, 10, 25, 50, 100, 205, 500)


def ref_image_state(img_buffer, W, H):
    """n_contrib and final_T out of the reference's imageBuffer (param_struct.h:85-103; ALIGNMENT = 128)."""
    base = img_buffer.data_ptr()
    al = lambda p: (p + 127) & ~127
    n = W * H
    p_ranges = al(base)
    p_nc = al(p_ranges + 8 * n)
    p_T = al(p_nc + 4 * n)
    raw = img_buffer.cpu().numpy()
    nc = raw[p_nc - base:p_nc - base + 4 * n].view(np.uint32).reshape(H, W)
    fT = raw[p_T - base:p_T - base + 4 * n].view(np.float32).reshape(H, W)
    return nc.astype(np.int64), fT


def reference_run(s):
    ref = ref_build.load("_ref2d_C")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    empty = torch.empty(0, device="cuda")
    cam = (s["tanfovx"], s["tanfovy"], t(s["viewmatrix"]), t(s["projmatrix"]), t(s["campos"]), int(s["sh_degree"]), float(s["gamma"]),
           float(s["scale_modifier"]), float(s["background_depth"]), t(s["background"]))
    vertex, opacity, shs = t(s["vertex"]), t(s["opacity"]), t(s["shs"])
    out = ref.rasterize_triangles(s["image_width"], s["image_height"], *cam, vertex, shs, empty, opacity, False, True, False)
    n, img, radii, depth, normal, csum, cmax, gb, bb, ib = out
    bw = ref.rasterize_triangles_backward(*cam, vertex, shs, empty, opacity, n, radii, gb, bb, ib, t(s["dL_dout_feature"]),
                                          t(s["dL_dout_depth"]), t(s["dL_dout_normal"]), True, False)
    torch.cuda.synchronize()
    nc, fT = ref_image_state(ib, s["image_width"], s["image_height"])
    dv, dc, dsh, df, dop = (x.cpu().numpy() for x in bw)
    return dict(num_rendered=int(n), out_feature=img.cpu().numpy(), radii=radii.cpu().numpy(), depth=depth.cpu().numpy(),
                normal=normal.cpu().numpy(), contrib_sum=csum.cpu().numpy(), contrib_max=cmax.cpu().numpy(), dL_dvertex=dv,
                dL_dcenter2D=dc, dL_dshs=dsh, dL_dopacity=dop, n_contrib=nc, final_T=fT)


def curve(a, b):
    """rel-L2 of a against b after setting aside the k worst rows (norm of the KEPT rows of b), k in KS."""
    P = a.shape[0]
    d = np.linalg.norm((a.astype(np.float64) - b).reshape(P, -1), axis=1) ** 2
    r = np.linalg.norm(b.astype(np.float64).reshape(P, -1), axis=1) ** 2
    order = np.argsort(d)
    dc, rc = np.cumsum(d[order]), np.cumsum(r[order])
    return {str(k): float(np.sqrt(dc[P - 1 - k] / rc[P - 1 - k])) for k in KS if k < P}


def compare(a, b, W, H):
    out = {}
    for k in ("out_feature", "depth", "normal", "contrib_sum", "contrib_max", "dL_dshs", "dL_dopacity"):
        out[k] = float(helpers.rel_l2(a[k], b[k]))
    for k in ("dL_dvertex", "dL_dcenter2D"):
        out[k] = curve(a[k], b[k])
    if "n_contrib" in a and "n_contrib" in b:
        out["pixels_n_contrib_differs"] = int((a["n_contrib"] != b["n_contrib"]).sum())
    rng = float(np.abs(b["out_feature"]).max())
    out["pixels_image_differs_gt_1e-3"] = int((np.abs(a["out_feature"] - b["out_feature"]).max(axis=0) > 1e-3 * rng).sum())
    out["pixels"] = W * H
    out["radii_differ"] = int((a["radii"] != b["radii"]).sum())
    out["num_rendered"] = [int(a["num_rendered"]), int(b["num_rendered"])]
    return out


def main():
    P, W, H, D = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (1_000_000, 1920, 1080, 3)))
    seed = int(sys.argv[5]) if len(sys.argv) > 5 else 42
    with_ref = os.environ.get("NOISE_FLOOR_NO_REF", "") == ""
    variant = int(os.environ.get("NOISE_FLOOR_VARIANT", "2"))  # 3: the 3D rasterizer (ray / plane barycentrics)
    s = synthetic.scene(P, W, H, D, seed=seed)
    t0 = time.time()
    of = helpers.oracle_forward(s, True, False, variant=variant)
    ob = helpers.oracle_backward(s, of, True)
    O = dict(of, **ob)
    O["n_contrib"] = of["state"].field("n_contrib").astype(np.int64).reshape(H, W)
    t1 = time.time()
    hf = helpers.hip_forward_backward(s, True, False, variant=variant)
    hf["n_contrib"] = helpers.hip_state(hf, s, "n_contrib").astype(np.int64).reshape(H, W)
    res = {"scene": f"S(P={P}, {W}x{H}, D={D}, seed={seed})", "rasterizer": f"{variant}D", "oracle_seconds": round(t1 - t0, 1), "k_set_aside": list(KS),
           "HIP_vs_oracle": compare(hf, O, W, H)}
    if with_ref:
        R = reference_run(s) if variant == 2 else ref_build.forward_backward(s, True, False, variant=3)
        res["oracle_vs_reference"] = compare(O, R, W, H)
        res["HIP_vs_reference"] = compare(hf, R, W, H)
        if variant == 3:
            # the reference against ITSELF: the same sources built with other code-generation switches (oracle/build_ref.py)
            Rs = ref_build.forward_backward(s, True, False, variant=3, build="_ref3d_scalar_C")
            Rn = ref_build.forward_backward(s, True, False, variant=3, build="_ref3d_nofma_C")
            res["reference_scalar_vs_reference"] = compare(Rs, R, W, H)
            res["reference_nofma_vs_reference"] = compare(Rn, R, W, H)
            res["reference_nofma_vs_reference_scalar"] = compare(Rn, Rs, W, H)
            res["HIP_vs_reference_scalar"] = compare(hf, Rs, W, H)
            res["oracle_vs_reference_nofma"] = compare(O, Rn, W, H)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", f"noise_floor_{P}.json" if variant == 2 else f"noise_floor3d_{P}.json")
    json.dump(res, open(path, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

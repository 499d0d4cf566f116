"""Shared helpers for the parity tests: run one scene through the CPU oracle and through the HIP path
(via the drop-in Python package, i.e. through the C ABI), and compare."""
from __future__ import annotations

import numpy as np

import synthetic
from oracle import ts2d_oracle as O


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    d = np.linalg.norm((a - b).ravel())
    n = np.linalg.norm(b.ravel())
    return d / n if n > 0 else d


def oracle_forward(s, rich_info=True, back_culling=False, use_feature=False, variant=2):
    shs = None if use_feature else s["shs"]
    feature = s["feature"] if use_feature else None
    n, img, radii, depth, normal, csum, cmax, st = O.rasterize_triangles(
        s["image_width"], s["image_height"], s["tanfovx"], s["tanfovy"], s["viewmatrix"], s["projmatrix"], s["campos"],
        s["sh_degree"], s["gamma"], s["scale_modifier"], s["background_depth"], s["background"], s["vertex"], shs,
        feature, s["opacity"], back_culling, rich_info, variant=variant)
    return dict(num_rendered=n, out_feature=img, radii=radii, depth=depth, normal=normal, contrib_sum=csum,
                contrib_max=cmax, state=st)


def oracle_backward(s, fwd, rich_info=True, use_feature=False):
    shs = None if use_feature else s["shs"]
    feature = s["feature"] if use_feature else None
    dv, dc, dsh, df, dop = O.rasterize_triangles_backward(
        s["tanfovx"], s["tanfovy"], s["viewmatrix"], s["projmatrix"], s["campos"], s["sh_degree"], s["gamma"],
        s["scale_modifier"], s["background_depth"], s["background"], s["vertex"], shs, feature, s["opacity"],
        fwd["radii"], fwd["state"], s["dL_dout_feature"], s.get("dL_dout_depth"), s.get("dL_dout_normal"), rich_info)
    return dict(dL_dvertex=dv, dL_dcenter2D=dc, dL_dshs=dsh, dL_dfeature=df, dL_dopacity=dop)


def hip_settings(s, rich_info=True, back_culling=False, debug=False, device="cuda"):
    import torch
    from diff_triangle_rasterization_2D import TriangleRasterizationSettings

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return TriangleRasterizationSettings(
        image_width=s["image_width"], image_height=s["image_height"], tanfovx=s["tanfovx"], tanfovy=s["tanfovy"],
        viewmatrix=t(s["viewmatrix"]), projmatrix=t(s["projmatrix"]), campos=t(s["campos"]), sh_degree=s["sh_degree"],
        gamma=s["gamma"], scale_modifier=s["scale_modifier"], background_depth=s["background_depth"],
        background=t(s["background"]), back_culling=back_culling, rich_info=rich_info, debug=debug)


def hip_forward_backward(s, rich_info=True, back_culling=False, use_feature=False, backward=True, device="cuda", debug=False,
                         variant=2):
    """Runs the HIP path through the drop-in autograd module.  Returns numpy outputs + grads + raw state."""
    import torch
    if variant == 3:
        from diff_triangle_rasterization_3D import TriangleRasterizer
    else:
        from diff_triangle_rasterization_2D import TriangleRasterizer

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    rs = hip_settings(s, rich_info, back_culling, debug, device)
    vertex = t(s["vertex"]).requires_grad_(True)
    opacity = t(s["opacity"]).requires_grad_(True)
    center2D = torch.zeros((vertex.shape[0], 2), device=device, requires_grad=True)
    shs = feature = None
    if use_feature:
        feature = t(s["feature"]).requires_grad_(True)
    else:
        shs = t(s["shs"]).requires_grad_(True)
    out = TriangleRasterizer(rs)(vertex, center2D, opacity, shs=shs, feature=feature)
    res = dict(out_feature=out[0].detach().cpu().numpy(), radii=out[1].cpu().numpy())
    if rich_info:
        res.update(depth=out[2].detach().cpu().numpy(), normal=out[3].detach().cpu().numpy(),
                   contrib_sum=out[4].cpu().numpy(), contrib_max=out[5].cpu().numpy())
    node = out[0].grad_fn
    res["num_rendered"] = node.num_rendered
    saved = node.saved_tensors
    res["buffers"] = saved[5:8]
    if backward:
        loss = (out[0] * t(s["dL_dout_feature"])).sum()
        if rich_info:
            loss = loss + (out[2] * t(s["dL_dout_depth"])).sum() + (out[3] * t(s["dL_dout_normal"])).sum()
        loss.backward()
        res.update(dL_dvertex=vertex.grad.cpu().numpy(), dL_dcenter2D=center2D.grad.cpu().numpy(),
                   dL_dopacity=opacity.grad.cpu().numpy())
        if use_feature:
            res["dL_dfeature"] = feature.grad.cpu().numpy()
        else:
            res["dL_dshs"] = shs.grad.cpu().numpy()
    return res


def hip_state(res, s, name):
    from diff_triangle_rasterization_2D import _C
    g, b, im = res["buffers"]
    P = s["vertex"].shape[0]
    return _C.debug_read_state(name, P, res["num_rendered"], s["image_width"], s["image_height"], g, b, im).numpy()


def grazing_mask(of, cos_limit):
    """3D variant: triangles whose plane is seen within acos(cos_limit) of edge-on (from the oracle's view-space state).
    There depth = v1.n / p_ray.n (R3D forward.cu:243-244) loses a factor 1 / cos of precision in ANY fp32 evaluation."""
    st = of["state"]
    c = (st.field("v1_view").astype(np.float64) + st.field("v2_view") + st.field("v3_view")) / 3.0
    n = st.field("normal_view").astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        cosv = np.abs((c * n).sum(1)) / (np.linalg.norm(c, axis=1) * np.linalg.norm(n, axis=1))
    return np.nan_to_num(cosv, nan=1.0) < cos_limit


def robust_rel_l2(hip, ora, budget, exclude=None, ref=None, dropped_bound=None):
    """rel-L2 over triangles (or pixels) after setting aside `exclude` (bool mask) and the `budget` largest per-row errors.
    The error is measured against the norm of the KEPT rows of `ora` (or `ref` when given).  The rows that were set aside
    must still be sane: finite, and -- when `dropped_bound` is given -- each within dropped_bound x max(its own reference
    magnitude, the median magnitude of the non-zero rows), so that a garbage row cannot hide inside the budget."""
    P = hip.shape[0]
    h = hip.astype(np.float64).reshape(P, -1)
    o = ora.astype(np.float64).reshape(P, -1)
    assert np.isfinite(h).all(), "non-finite values in the compared output"
    err = np.linalg.norm(h - o, axis=1)
    mag = np.linalg.norm(o, axis=1)
    keep = np.ones(P, bool)
    if exclude is not None:
        keep &= ~exclude
    if budget > 0:
        order = np.argsort(np.where(keep, err, -1.0))
        keep[order[P - budget:]] = False
    if dropped_bound is not None and (~keep).any():
        nz = mag[mag > 0]
        floor = float(np.median(nz)) if nz.size else 0.0
        bad = err[~keep] > dropped_bound * np.maximum(mag[~keep], floor)
        assert not bad.any(), f"{int(bad.sum())} set-aside rows are off by more than {dropped_bound}x their own scale"
    d = np.sqrt((err[keep] ** 2).sum())
    n = np.sqrt((mag[keep] ** 2).sum()) if ref is None else ref
    return d / n if n > 0 else d

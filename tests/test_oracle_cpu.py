"""CPU tests of the oracle (no GPU): golden fixtures generated from the reference's own Python, hand-checked
integer semantics, and an independent float64 autograd restatement of the forward model for the gradients.

Pinning status (see oracle/ts2d_oracle.c header): the reference ships no tests/golden vectors for the rasterizer; the
rasterizer as a whole is pinned on the GPU against the reference's own kernels built for gfx950
(tests/test_reference_gpu.py).  What can be pinned WITHOUT a GPU is pinned here: the SH colour polynomial
(sh_utils.eval_sh) and the camera/matrix convention (camera.Camera).
"""
import math
import os

import numpy as np
import pytest

import helpers
import synthetic
from oracle import ts2d_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ------------------------------------------------------------------ golden fixtures from the reference's Python
@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_colour_matches_reference_eval_sh(deg):
    g = np.load(os.path.join(GOLD, "sh_eval.npz"))
    dirs, sh = g["dirs"], g["sh"]  # (n,3) unit, (n,3,16)
    campos = np.array([0.3, -0.2, 0.1])
    pos = campos + 7.5 * dirs  # any point along the direction
    shs = np.ascontiguousarray(np.transpose(sh, (0, 2, 1)))  # rasterizer layout (n, M, 3)
    rgb, clamped = O.sh_color(deg, shs, pos, campos)
    expect = g[f"deg{deg}"] + 0.5  # forward.cu:51
    assert np.array_equal(clamped, expect < 0) or np.abs(expect[clamped != (expect < 0)]).max() < 1e-5
    np.testing.assert_allclose(rgb, np.maximum(expect, 0.0), rtol=0, atol=3e-6)


def _cams():
    g = np.load(os.path.join(GOLD, "camera.npz"))
    n = len([k for k in g.files if k.startswith("W_")])
    return [{k.rsplit("_", 1)[0]: g[k] for k in g.files if k.endswith(f"_{i}")} for i in range(n)]


def test_canonical_bench_camera_matches_reference_camera_class():
    c = _cams()[-1]  # R2D/main.cu pose built through the reference's Camera
    mine = synthetic.camera(int(c["W"]), int(c["H"]))
    np.testing.assert_allclose(mine["viewmatrix"], c["world_view_transform"], atol=1e-6)
    np.testing.assert_allclose(mine["projmatrix"], c["full_proj_transform"], rtol=2e-6, atol=1e-4)
    np.testing.assert_allclose(mine["campos"], c["camera_center"], atol=1e-3)
    assert abs(mine["tanfovx"] - float(c["tan_fovx"])) < 1e-6 and abs(mine["tanfovy"] - float(c["tan_fovy"])) < 1e-6


@pytest.mark.parametrize("i", range(6))
def test_oracle_consumes_reference_camera_convention(i):
    """Points placed in front of a reference-built camera must land at the pixel the reference convention predicts:
    pixel = ((ndc + 1) * S - 1) / 2 with ndc = (p_h @ full_proj_transform).xy / w, depth key = (p_h @ world_view).z."""
    c = _cams()[i]
    W, H = int(c["W"]), int(c["H"])
    view, proj, cam = c["world_view_transform"].astype(np.float64), c["full_proj_transform"].astype(np.float64), c["camera_center"]
    rng = np.random.default_rng(i)
    # points in view space inside the frustum, mapped back to world space with the inverse view matrix (row vectors)
    n = 64
    z = rng.uniform(3.0, 30.0, n)
    x = rng.uniform(-0.8, 0.8, n) * z * float(c["tan_fovx"])
    y = rng.uniform(-0.8, 0.8, n) * z * float(c["tan_fovy"])
    pv = np.stack([x, y, z, np.ones(n)], 1)
    pw = pv @ np.linalg.inv(view)
    tri = pw[:, None, :3] + rng.normal(0, 1e-3, (n, 3, 3)) * z[:, None, None]
    centroid = tri.mean(1)
    ph = np.concatenate([centroid, np.ones((n, 1))], 1)
    clip = ph @ proj
    ndc = clip[:, :2] / clip[:, 3:4]
    pix = ((ndc + 1.0) * np.array([W, H]) - 1.0) * 0.5
    out = O.rasterize_triangles(W, H, float(c["tan_fovx"]), float(c["tan_fovy"]), view, proj, cam, 0, 1.0, 1.0, 100.0,
                                np.zeros(3), tri, np.ones((n, 1, 3)), None, np.full((n, 1), 0.5), False, True)
    radii, st = out[2], out[7]
    assert (radii > 0).all()
    c2d = (st.field("v1_2D") + st.field("v2_2D") + st.field("v3_2D")) / 3.0
    assert np.abs(c2d - pix).max() < 0.6  # the 0.5 px low-pass dilation moves vertices, not the convention
    np.testing.assert_allclose(st.field("depth"), (ph @ view)[:, 2], rtol=1e-5)


def test_gamma_rescale_fixture_formula():
    """Caller-side constant of VanillaTS_model.py:615-617 (kept as a fixture for the later 'next' row)."""
    g = np.load(os.path.join(GOLD, "gamma_rescale.npz"))
    for gamma, ratio in zip(g["gamma"], g["ratio"]):
        beta = 1.0 / gamma
        assert abs(1.0 / math.sqrt(2.0 ** beta * beta * math.gamma(beta)) - ratio) < 1e-12


# ------------------------------------------------------------------ integer semantics, hand-checked
def test_higher_msb_matches_reference_bit_count():
    # R2D/src/rasterizer.cu:20-35: number of bits needed for the tile count (used as sort end bit)
    for n, bits in [(1, 1), (2, 2), (3, 2), (4, 3), (255, 8), (256, 9), (8160, 13), (65535, 16), (65536, 17)]:
        assert O.higher_msb(n) == bits


def _one_triangle_scene(W=64, H=48):
    s = synthetic.scene(1, W, H, 0, seed=0)
    # a triangle of ~10 px around the screen centre, at depth 1100
    px = (synthetic.CAM_DIST - 100.0) * s["tanfovx"] / (0.5 * W)  # world units per pixel at z_view = 1100
    s["vertex"] = np.array([[[-5 * px, -4 * px, 100.0], [6 * px, -3 * px, 100.0], [0.0, 7 * px, 100.0]]], np.float32)
    s["opacity"] = np.array([[0.8]], np.float32)
    return s


def test_single_triangle_state_and_image():
    s = _one_triangle_scene()
    f = helpers.oracle_forward(s)
    st = f["state"]
    assert f["radii"][0] > 0
    rmin, rmax = st.field("rect_min")[0], st.field("rect_max")[0]
    assert int(st.field("tiles_touched")[0]) == (rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == f["num_rendered"]
    # instances are emitted row-major over the rectangle and end up sorted by tile id (rasterizer.cu:63-73)
    keys = st.field("keys")
    tiles = (keys >> np.uint64(32)).astype(np.int64)
    gx = st.grid[0]
    expect = sorted(y * gx + x for y in range(rmin[1], rmax[1]) for x in range(rmin[0], rmax[0]))
    assert tiles.tolist() == expect
    assert np.all(keys.astype(np.uint64) & np.uint64(0xFFFFFFFF) == np.float32(st.field("depth")[0]).view(np.uint32))
    # inside the triangle the pixel gets alpha = min(0.99, 0.8 * exp(-0.5 ecc^2)) with ecc < 1; centre pixel: ecc ~ 0
    img, T = f["out_feature"], st.field("final_T")
    cy, cx = 24, 32
    assert 0.19 < T[cy, cx] < 0.25  # 1 - 0.8 * exp(-small)
    # far corner: untouched, equals background (0) and T == 1, n_contrib counts examined entries of its tile
    assert T[0, 0] == 1.0 and np.all(img[:, 0, 0] == 0)
    rng = st.field("ranges")
    tile_of_center = (cy // 16) * gx + cx // 16
    assert st.field("n_contrib")[cy, cx] == rng[tile_of_center, 1] - rng[tile_of_center, 0] == 1


def test_sort_is_stable_and_depth_ordered():
    s = synthetic.scene(500, 64, 64, 0, seed=4, edge_px=10.0)
    s["vertex"][250:] = s["vertex"][:250]  # exact duplicates -> equal depth keys, ties must keep ascending triangle id
    f = helpers.oracle_forward(s)
    st = f["state"]
    keys, vals = st.field("keys"), st.field("vals").astype(np.int64)
    assert np.all(np.diff(keys.astype(np.uint64).astype(object)) >= 0)
    same = keys[1:] == keys[:-1]
    assert same.any() and np.all(vals[1:][same] > vals[:-1][same])


def test_termination_counts_examined_entries():
    """Opaque stack: the pixel stops AFTER the entry that drives T <= 1e-4 and n_contrib counts every examined entry
    including skipped ones (forward.cu:296-297, 332-334).  Expected values come from a plain-Python replay of the
    reference loop for one pixel."""
    s = _one_triangle_scene()
    n = 12
    s["vertex"] = np.repeat(s["vertex"], n, 0)
    s["vertex"][:, :, 2] += np.arange(n, dtype=np.float32)[:, None] * 0.5  # distinct depths, same footprint
    s["opacity"] = np.full((n, 1), 0.999, np.float32)
    s["opacity"][n - 2] = 0.001  # second-nearest (larger world z = nearer) is below 1/255: examined but skipped
    s["shs"] = np.ones((n, 1, 3), np.float32)
    f = helpers.oracle_forward(s)
    st = f["state"]
    cy, cx = 24, 32
    gx = st.grid[0]
    tile = (cy // 16) * gx + cx // 16
    r0, r1 = st.field("ranges")[tile]
    ids = st.field("vals")[r0:r1]
    assert ids.tolist() == list(range(n - 1, -1, -1))  # nearest (largest world z) first
    T, examined, contributed = 1.0, 0, []
    for i in ids:
        examined += 1
        p = [st.field(f"v{k}_2D")[i].astype(np.float64) - np.array([cx, cy], np.float64) for k in (1, 2, 3)]
        cr = lambda a, b: a[0] * b[1] - a[1] * b[0]
        area = float(st.field("area2")[i])
        a1, a2 = cr(p[1], p[2]) / area, cr(p[2], p[0]) / area
        ecc = 1 - 3 * min(a1, a2, 1 - a1 - a2)
        if ecc < 0 or ecc > 10:
            continue
        alpha = min(0.99, float(s["opacity"][i, 0]) * math.exp(-0.5 * ecc ** 2))
        if alpha < 1 / 255:
            continue
        contributed.append(int(i))
        T *= 1 - alpha
        if T <= 1e-4:
            break
    assert examined < n  # the stack really saturates before the list ends
    assert st.field("n_contrib")[cy, cx] == examined
    assert abs(st.field("final_T")[cy, cx] - T) < 1e-7
    assert (n - 2) not in contributed and f["contrib_sum"][n - 2] == 0  # the skipped one never contributes
    assert all(f["contrib_sum"][i] > 0 for i in contributed)


def test_empty_and_error_paths():
    s = synthetic.scene(0, 32, 32, 0, seed=1)
    f = helpers.oracle_forward(s)
    assert f["num_rendered"] == 0 and f["out_feature"].shape == (3, 32, 32) and not f["out_feature"].any()
    s = synthetic.scene(4, 32, 32, 0, seed=1)
    with pytest.raises(RuntimeError):
        O.rasterize_triangles(32, 32, 0.3, 0.3, s["viewmatrix"], s["projmatrix"], s["campos"], 0, -1.0, 1.0, 10.0,
                              s["background"], s["vertex"], s["shs"], None, s["opacity"], False, True)
    with pytest.raises(RuntimeError):
        O.rasterize_triangles(32, 32, 0.3, 0.3, s["viewmatrix"], s["projmatrix"], s["campos"], 0, 1.0, 1.0, 10.0,
                              s["background"], s["vertex"][:, :2], s["shs"], None, s["opacity"], False, True)


# ------------------------------------------------------------------ independent float64 autograd restatement
def _torch_forward(vertex, shs, opacity, cam, W, H, D, gamma, bg, bg_depth, order, hits):
    """Differentiable float64 restatement of SURVEY.md Appendix A for a whole image at once.  `order` (depth order) and
    `hits` (which (triangle, pixel) pairs pass the discrete tests, [P, H*W] bool) come from a no-grad pass so that the
    graph contains only the smooth part, exactly like the reference's hand-written backward."""
    import torch

    view = torch.tensor(cam["viewmatrix"], dtype=torch.float64)
    proj = torch.tensor(cam["projmatrix"], dtype=torch.float64)
    campos = torch.tensor(cam["campos"], dtype=torch.float64)
    tx, ty = cam["tanfovx"], cam["tanfovy"]
    c = vertex.mean(1)
    ch = torch.cat([c, torch.ones_like(c[:, :1])], 1) @ proj
    ndc = ch[:, :3] / (ch[:, 3:4].abs() + 1e-8)
    cv = torch.cat([c, torch.ones_like(c[:, :1])], 1) @ view
    z = cv[:, 2]
    cx = torch.minimum(torch.maximum(cv[:, 0], -1.3 * tx * z), 1.3 * tx * z)
    cy = torch.minimum(torch.maximum(cv[:, 1], -1.3 * ty * z), 1.3 * ty * z)
    r = (vertex - c[:, None, :]) @ view[:3, :3]  # (P,3,3) view-space radial vectors
    rho = torch.stack([(r[..., 0] - r[..., 2] * (cx / z)[:, None]) / (z * tx)[:, None],
                       (r[..., 1] - r[..., 2] * (cy / z)[:, None]) / (z * ty)[:, None]], -1)  # (P,3,2)
    nrm = rho.norm(dim=-1, keepdim=True)
    cpix = torch.stack([((ndc[:, 0] + 1) * W - 1) * 0.5, ((ndc[:, 1] + 1) * H - 1) * 0.5], -1)
    p = cpix[:, None, :] + rho * (torch.tensor([0.5 * W, 0.5 * H], dtype=torch.float64) + 0.5 / nrm)  # (P,3,2)
    area = (p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1]) - (p[:, 1, 1] - p[:, 0, 1]) * (p[:, 2, 0] - p[:, 0, 0])
    # colour
    d = c - campos
    d = d / d.norm(dim=-1, keepdim=True)
    x, y, zz_ = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    C0, C1 = 0.28209479177387814, 0.4886025119029199
    rgb = C0 * shs[:, 0]
    if D > 0:
        rgb = rgb - C1 * y * shs[:, 1] + C1 * zz_ * shs[:, 2] - C1 * x * shs[:, 3]
    rgb = torch.clamp(rgb + 0.5, min=0.0)
    normal = torch.cross(r[:, 0], r[:, 1], dim=-1)
    normal = normal / normal.norm(dim=-1, keepdim=True)
    vdepth = r[..., 2] + z[:, None]
    # per pixel
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    pix = torch.stack([xs.reshape(-1), ys.reshape(-1)], -1)  # (HW,2)

    def cross2(a, b):
        return a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]

    pv = p[:, :, None, :] - pix[None, None, :, :]  # (P,3,HW,2)
    a1 = cross2(pv[:, 1], pv[:, 2]) / area[:, None]
    a2 = cross2(pv[:, 2], pv[:, 0]) / area[:, None]
    a3 = 1 - a1 - a2
    ecc = 1 - 3 * torch.minimum(torch.minimum(a1, a2), a3)
    alpha = opacity.reshape(-1, 1) * torch.exp(-0.5 * ecc.clamp(min=0) ** (2 * gamma))
    alpha = torch.where(torch.tensor(hits), alpha, torch.zeros_like(alpha))
    alpha = alpha[order]  # front to back
    Tafter = torch.cumprod(1 - alpha, 0)
    Tbefore = torch.cat([torch.ones_like(Tafter[:1]), Tafter[:-1]], 0)
    contrib = alpha * Tbefore
    img = (contrib[:, None, :] * rgb[order][:, :, None]).sum(0) + Tafter[-1][None] * torch.tensor(bg, dtype=torch.float64)[:, None]
    dep = (contrib * (vdepth[order, 0:1] * a1[order] + vdepth[order, 1:2] * a2[order] + vdepth[order, 2:3] * a3[order])).sum(0) \
        + Tafter[-1] * bg_depth
    nor = (contrib[:, None, :] * normal[order][:, :, None]).sum(0)
    return img.reshape(3, H, W), dep.reshape(H, W), nor.reshape(3, H, W)


def test_backward_matches_float64_autograd():
    """The oracle's hand-written backward (restating backward.cu) against autograd of an independent float64
    restatement of the forward model.  Scene chosen so that the reference's deliberate deviations from the true
    derivative are inactive: opacity < 0.9 (no 0.99 clamp), no centre clipping, no early termination."""
    import torch

    P, W, H, D = 40, 48, 40, 1
    s = synthetic.scene(P, W, H, D, seed=21, edge_px=9.0)
    s["opacity"] = (0.15 + 0.6 * s["opacity"]).astype(np.float32)
    s["shs"] = (s["shs"] * 0.8).astype(np.float32)
    s["background"] = np.array([0.1, 0.3, 0.2], np.float32)
    of = helpers.oracle_forward(s)
    ob = helpers.oracle_backward(s, of)
    st = of["state"]
    assert (of["radii"] > 0).all() and st.field("final_T").min() > 1e-3
    order = np.argsort(st.field("depth"), kind="stable")
    # discrete decisions taken from a float64 evaluation of the same tests
    with torch.no_grad():
        v64 = torch.tensor(s["vertex"], dtype=torch.float64)
        full = _torch_forward(v64, torch.tensor(s["shs"], dtype=torch.float64), torch.tensor(s["opacity"], dtype=torch.float64),
                              s, W, H, D, 1.0, s["background"], s["background_depth"], order, np.ones((P, H * W), bool))
    # hit mask: in-rect tiles AND 0 <= ecc <= 10 AND alpha >= 1/255, evaluated in float64
    vertex = torch.tensor(s["vertex"], dtype=torch.float64, requires_grad=True)
    shs = torch.tensor(s["shs"], dtype=torch.float64, requires_grad=True)
    opacity = torch.tensor(s["opacity"], dtype=torch.float64, requires_grad=True)

    # recompute per-pair ecc/alpha without grad to build the mask
    def masks():
        import torch as th
        with th.no_grad():
            p = np.stack([st.field("v1_2D"), st.field("v2_2D"), st.field("v3_2D")], 1).astype(np.float64)
            area = st.field("area2").astype(np.float64)
            ys, xs = np.mgrid[0:H, 0:W]
            pix = np.stack([xs.reshape(-1), ys.reshape(-1)], -1).astype(np.float64)
            pv = p[:, :, None, :] - pix[None, None]
            cr = lambda a, b: a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]
            a1 = cr(pv[:, 1], pv[:, 2]) / area[:, None]
            a2 = cr(pv[:, 2], pv[:, 0]) / area[:, None]
            a3 = 1 - a1 - a2
            ecc = 1 - 3 * np.minimum(np.minimum(a1, a2), a3)
            alpha = s["opacity"].reshape(-1, 1) * np.exp(-0.5 * np.clip(ecc, 0, None) ** 2)
            hit = (ecc >= 0) & (ecc <= 10) & (alpha >= 1 / 255)
            rmin, rmax = st.field("rect_min").astype(int), st.field("rect_max").astype(int)
            tx, ty = (pix[:, 0] // 16).astype(int), (pix[:, 1] // 16).astype(int)
            inrect = (tx[None] >= rmin[:, 0:1]) & (tx[None] < rmax[:, 0:1]) & (ty[None] >= rmin[:, 1:2]) & (ty[None] < rmax[:, 1:2])
            return hit & inrect

    hits = masks()
    img, dep, nor = _torch_forward(vertex, shs, opacity, s, W, H, D, 1.0, s["background"], s["background_depth"], order, hits)
    np.testing.assert_allclose(img.detach().numpy(), of["out_feature"], atol=2e-5)
    np.testing.assert_allclose(dep.detach().numpy(), of["depth"], rtol=2e-5, atol=1e-2)
    loss = (img * torch.tensor(s["dL_dout_feature"], dtype=torch.float64)).sum() \
        + (dep * torch.tensor(s["dL_dout_depth"], dtype=torch.float64)).sum() \
        + (nor * torch.tensor(s["dL_dout_normal"], dtype=torch.float64)).sum()
    loss.backward()
    assert helpers.rel_l2(ob["dL_dshs"], shs.grad.numpy()) < 1e-4
    assert helpers.rel_l2(ob["dL_dopacity"], opacity.grad.numpy()) < 1e-4
    assert helpers.rel_l2(ob["dL_dvertex"], vertex.grad.numpy()) < 2e-3
    del full

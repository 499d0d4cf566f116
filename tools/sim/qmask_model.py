"""numpy (float32) model of csrc/ts2d_support.h: the quadrant mask of an instance must contain every quadrant in which the blend kernels' per-pixel
test (render_group.hip: barycentrics -> ecc -> alpha >= 1/255) accepts a pixel.  Random triangles from sub-pixel slivers to image-sized ones, all
tiles around them; also reports how tight the mask is (quadrants flagged / quadrants with a hit).   python tools/sim/qmask_model.py [n] [seed]"""
import sys
import numpy as np

f32 = np.float32


def support_scale(op, g2):
    t = f32(255.0) * op
    with np.errstate(divide="ignore", invalid="ignore"):
        L = f32(2.0 * 0.6931471805599453) * np.log2(np.maximum(t, f32(1e-30)), dtype=f32)
        E = np.where(g2 == f32(2.0), np.sqrt(np.maximum(L, 0), dtype=f32), np.where(g2 < 1e-6, f32(10.0), np.exp2(np.log2(np.maximum(L, f32(1e-30)), dtype=f32) * (f32(1.0) / g2), dtype=f32)))
    E = np.minimum(E * f32(1.0005) + f32(0.002), f32(10.01)).astype(f32)
    return np.where(t >= 1.0, E, f32(-1.0)).astype(f32)


def quad_setup(v, E, pad_px=0.0):
    v1x, v1y, v2x, v2y, v3x, v3y = [v[:, i] for i in range(6)]
    area2 = ((v2x - v1x) * (v3y - v1y)).astype(f32) - ((v2y - v1y) * (v3x - v1x)).astype(f32)
    ia = (f32(1.0) / area2).astype(f32)
    q = dict(v=v, ia=ia)
    q["A1"] = (v2y - v3y) * ia; q["B1"] = (v3x - v2x) * ia
    q["A2"] = (v3y - v1y) * ia; q["B2"] = (v1x - v3x) * ia
    q["A3"] = -q["A1"] - q["A2"]; q["B3"] = -q["B1"] - q["B2"]
    m = (f32(1.0) - E) * f32(1.0 / 3.0)
    for k in "123":
        A, B = q["A" + k], q["B" + k]
        q["P" + k] = (np.maximum(f32(0), f32(7) * A) + np.maximum(f32(0), f32(7) * B) - m + (f32(2e-6) * f32(15) + f32(pad_px)) * (np.abs(A) + np.abs(B))).astype(f32)
    cx = (v1x + v2x + v3x) * f32(1.0 / 3.0); cy = (v1y + v2y + v3y) * f32(1.0 / 3.0)
    ex = np.stack([E * (v1x - cx), E * (v2x - cx), E * (v3x - cx)]); ey = np.stack([E * (v1y - cy), E * (v2y - cy), E * (v3y - cy)])
    padx = f32(0.05) + f32(pad_px) + f32(4e-7) * np.abs(cx); pady = f32(0.05) + f32(pad_px) + f32(4e-7) * np.abs(cy)
    q.update(bminx=cx + ex.min(0) - padx, bmaxx=cx + ex.max(0) + padx, bminy=cy + ey.min(0) - pady, bmaxy=cy + ey.max(0) + pady, live=E > 0)
    return q


def quadrant_mask(q, TX, TY):
    v = q["v"]
    u1x, u1y, u2x, u2y, u3x, u3y = v[:, 0] - TX, v[:, 1] - TY, v[:, 2] - TX, v[:, 3] - TY, v[:, 4] - TX, v[:, 5] - TY
    t1a, t1b, t2a, t2b, aia = u2x * u3y, u2y * u3x, u3x * u1y, u3y * u1x, np.abs(q["ia"])
    C1 = (t1a - t1b) * q["ia"]; C2 = (t2a - t2b) * q["ia"]; C3 = f32(1.0) - C1 - C2
    r1 = f32(4e-7) * (np.abs(t1a) + np.abs(t1b)) * aia; r2 = f32(4e-7) * (np.abs(t2a) + np.abs(t2b)) * aia
    k = [C1 + q["P1"] + r1, C2 + q["P2"] + r2, C3 + q["P3"] + (r1 + r2 + f32(4e-7))]
    ax = [f32(8) * q["A" + c] for c in "123"]; by = [f32(8) * q["B" + c] for c in "123"]
    x0 = q["live"] & (q["bminx"] <= TX + 7) & (q["bmaxx"] >= TX); x1 = q["live"] & (q["bminx"] <= TX + 15) & (q["bmaxx"] >= TX + 8)
    y0 = (q["bminy"] <= TY + 7) & (q["bmaxy"] >= TY); y1 = (q["bminy"] <= TY + 15) & (q["bmaxy"] >= TY + 8)
    with np.errstate(invalid="ignore"):
        ok = lambda dx, dy: np.all([k[i] + (ax[i] if dx else 0) + (by[i] if dy else 0) >= 0 for i in range(3)], axis=0)
        return (x0 & y0 & ok(0, 0)) * 1 + (x1 & y0 & ok(1, 0)) * 2 + (x0 & y1 & ok(0, 1)) * 4 + (x1 & y1 & ok(1, 1)) * 8


def quad_anchor(q, TX0, TY0, Wpx, Hpx):
    """ts2d_support.h: quad_anchor -- the affine constants of a triangle over its tile rectangle (round 5), margins bounded over the rectangle."""
    v = q["v"]
    u = [v[:, i] - (TX0 if i % 2 == 0 else TY0) for i in range(6)]
    aia = np.abs(q["ia"])
    C1 = (u[2] * u[5] - u[3] * u[4]) * q["ia"]; C2 = (u[4] * u[1] - u[5] * u[0]) * q["ia"]; C3 = f32(1.0) - C1 - C2
    U = [np.maximum(np.abs(u[i]), np.abs(u[i] - (Wpx if i % 2 == 0 else Hpx))) for i in range(6)]
    r1 = f32(4e-7) * (U[2] * U[5] + U[3] * U[4]) * aia; r2 = f32(4e-7) * (U[4] * U[1] + U[5] * U[0]) * aia
    s = [f32(6e-7) * (np.abs(q["A" + c]) * Wpx + np.abs(q["B" + c]) * Hpx) + f32(2.5e-7) * np.abs(C) for c, C in zip("123", (C1, C2, C3))]
    K = [(C1 + q["P1"] + r1 + s[0]).astype(f32), (C2 + q["P2"] + r2 + s[1]).astype(f32), (C3 + q["P3"] + (r1 + r2 + f32(4e-7)) + s[2]).astype(f32)]
    return dict(K=K, A=[q["A" + c] for c in "123"], B=[q["B" + c] for c in "123"], bminx=np.where(q["live"], q["bminx"], f32(3e38)),
                bmaxx=np.where(q["live"], q["bmaxx"], f32(-3e38)), bminy=q["bminy"], bmaxy=q["bmaxy"])


def fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)  # one rounding, like v_fma_f32


def quadrant_mask_affine(o, fx, fy, TX, TY):
    k = [fma32(o["B"][i], fy, fma32(o["A"][i], fx, o["K"][i])) for i in range(3)]
    ax = [f32(8) * o["A"][i] for i in range(3)]; by = [f32(8) * o["B"][i] for i in range(3)]
    x0 = (o["bminx"] <= TX + 7) & (o["bmaxx"] >= TX); x1 = (o["bminx"] <= TX + 15) & (o["bmaxx"] >= TX + 8)
    y0 = (o["bminy"] <= TY + 7) & (o["bmaxy"] >= TY); y1 = (o["bminy"] <= TY + 15) & (o["bmaxy"] >= TY + 8)
    with np.errstate(invalid="ignore"):
        ok = lambda dx, dy: np.all([k[i] + (ax[i] if dx else 0) + (by[i] if dy else 0) >= 0 for i in range(3)], axis=0)
        return (x0 & y0 & ok(0, 0)) * 1 + (x1 & y0 & ok(1, 0)) * 2 + (x0 & y1 & ok(0, 1)) * 4 + (x1 & y1 & ok(1, 1)) * 8


def pixel_hits(v, ia, op, g2, TX, TY, dtype):
    """(n, 4) bool: some pixel of quadrant q accepted by the per-pixel test, evaluated like render_group.hip (origin-relative vertices, then the pixel)."""
    n = len(v)
    hit = np.zeros((n, 4), bool)
    lx, ly = np.meshgrid(np.arange(8), np.arange(8))
    lx, ly = lx.ravel().astype(dtype), ly.ravel().astype(dtype)
    for qi in range(4):
        OX, OY = (TX + 8 * (qi & 1)).astype(dtype), (TY + 8 * (qi >> 1)).astype(dtype)
        u = [(v[:, i].astype(dtype) - (OX if i % 2 == 0 else OY)).astype(dtype) for i in range(6)]
        p = [(u[i][:, None] - (lx if i % 2 == 0 else ly)[None, :]).astype(dtype) for i in range(6)]
        iad = ia.astype(dtype)[:, None]
        with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
            a1 = ((p[2] * p[5] - p[3] * p[4]) * iad).astype(dtype); a2 = ((p[4] * p[1] - p[5] * p[0]) * iad).astype(dtype)
            a3 = (1 - a1 - a2).astype(dtype)
            ecc = (1 - 3 * np.minimum(np.minimum(a1, a2), a3)).astype(dtype)
            pw = np.power(np.maximum(ecc, 0), g2.astype(dtype)[:, None]).astype(dtype)
            alpha = np.minimum(0.99, op.astype(dtype)[:, None] * np.exp2(pw * dtype(-0.7213475204444817))).astype(dtype)
            h = (ecc >= 0) & (ecc <= 10) & (alpha >= dtype(1.0 / 255.0))
        hit[:, qi] = h.any(1)
    return hit


PAD3D = 0.02


def quad_setup_3d(V, E, tanx, tany, W, H, Nn=None, rect=None):
    """ts2d_support.h: quad_setup_3d -- the view-space triangle V (n, 3, 3) scaled by E about its centroid, projected to pixels, then the 2D setup
    with E = 1; triangles it cannot be trusted on (a scaled vertex near / behind the camera, a projection thinner than 1e-3 px) flag every quadrant."""
    n = len(V)
    c = (V[:, 0] + V[:, 1] + V[:, 2]) * f32(1.0 / 3.0)
    Wv = (c[:, None, :] + E[:, None, None] * (V - c[:, None, :])).astype(f32)
    zmin = f32(0.05) * c[:, 2]
    ok = (E > 0) & (c[:, 2] > 0) & np.all(Wv[:, :, 2] >= zmin[:, None], axis=1)
    if rect is not None:  # the horizon of the triangle's plane crossing the rectangle (see ts2d_support.h): p_ray . n at the corners
        px0, py0, px1, py1 = [r.astype(f32) for r in rect]
        rx = [f32(tanx) * ((f32(2) * p - f32(W) + f32(1)) / f32(W)) for p in (px0, px1)]; ry = [f32(tany) * ((f32(2) * p - f32(H) + f32(1)) / f32(H)) for p in (py0, py1)]
        d = np.stack([(x * Nn[:, 0] + y * Nn[:, 1] + Nn[:, 2]).astype(f32) for x in rx for y in ry])
        lo, hi = d.min(0), d.max(0); big = np.maximum(np.abs(lo), np.abs(hi))
        ok &= (lo > f32(1e-3) * big) | (hi < f32(-1e-3) * big)
    kx, ky, ox, oy = f32(0.5 * W / tanx), f32(0.5 * H / tany), f32(0.5 * W - 0.5), f32(0.5 * H - 0.5)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        iz = (f32(1.0) / Wv[:, :, 2]).astype(f32)
        sx = (Wv[:, :, 0] * iz * kx + ox).astype(f32); sy = (Wv[:, :, 1] * iz * ky + oy).astype(f32)
        area2 = (sx[:, 1] - sx[:, 0]) * (sy[:, 2] - sy[:, 0]) - (sy[:, 1] - sy[:, 0]) * (sx[:, 2] - sx[:, 0])
        span = np.maximum(np.maximum(np.abs(sx[:, 1] - sx[:, 0]), np.abs(sx[:, 2] - sx[:, 0])), np.maximum(np.abs(sy[:, 1] - sy[:, 0]), np.abs(sy[:, 2] - sy[:, 0])))
        ok &= (np.abs(area2) > f32(1e-3) * span) & (span < f32(1e7))
    v2 = np.stack([sx[:, 0], sy[:, 0], sx[:, 1], sy[:, 1], sx[:, 2], sy[:, 2]], 1).astype(f32)
    v2[~ok] = np.array([0, 0, 1, 0, 0, 1], f32)  # any finite triangle: its constants are overwritten below
    with np.errstate(over="ignore", invalid="ignore"):
        q = quad_setup(v2, np.ones(n, f32), PAD3D)
    for k in ("A1", "A2", "A3", "B1", "B2", "B3", "ia"):
        q[k] = np.where(ok, q[k], f32(0)).astype(f32)
    for k in ("P1", "P2", "P3"):
        q[k] = np.where(ok, q[k], f32(1e30)).astype(f32)
    q["v"] = np.where(ok[:, None], q["v"], f32(0)).astype(f32)
    q["bminx"] = np.where(ok, q["bminx"], f32(-3e38)); q["bminy"] = np.where(ok, q["bminy"], f32(-3e38))
    q["bmaxx"] = np.where(ok, q["bmaxx"], f32(3e38)); q["bmaxy"] = np.where(ok, q["bmaxy"], f32(3e38))
    q["live"] = np.ones(n, bool)
    return q, ok


def pixel_hits_3d(V, N, g2, TX, TY, tanx, tany, W, H, dtype):
    """(n, 4) bool: some pixel of quadrant q passes the 3D BACKWARD's per-pixel test (render3d_group.hip hit3 + G >= 1/255, a superset of the
    forward's alpha test): ray / plane intersection, barycentrics in 3D (R3D forward.cu:238-256, backward.cu:328-351)."""
    n = len(V)
    hit = np.zeros((n, 4), bool)
    lx, ly = np.meshgrid(np.arange(8), np.arange(8))
    lx, ly = lx.ravel().astype(dtype), ly.ravel().astype(dtype)
    V = V.astype(dtype); N = N.astype(dtype)
    inn = (1.0 / (N * N).sum(1)).astype(dtype)[:, None]
    d0 = (V[:, 0] * N).sum(1).astype(dtype)[:, None]
    for qi in range(4):
        px = (TX.astype(dtype) + 8 * (qi & 1))[:, None] + lx[None, :]; py = (TY.astype(dtype) + 8 * (qi >> 1))[:, None] + ly[None, :]
        rx = (dtype(tanx) * ((2 * px - W + 1) / dtype(W))).astype(dtype); ry = (dtype(tany) * ((2 * py - H + 1) / dtype(H))).astype(dtype)
        with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
            prn = (rx * N[:, 0:1] + ry * N[:, 1:2] + N[:, 2:3]).astype(dtype)
            depth = (d0 / prn).astype(dtype)
            pvx, pvy, pvz = depth * rx, depth * ry, depth
            P = [[(V[:, k, 0:1] - pvx).astype(dtype), (V[:, k, 1:2] - pvy).astype(dtype), (V[:, k, 2:3] - pvz).astype(dtype)] for k in range(3)]
            cross = lambda a, b: [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]
            dotn = lambda c: (c[0] * N[:, 0:1] + c[1] * N[:, 1:2] + c[2] * N[:, 2:3]).astype(dtype)
            a1 = (dotn(cross(P[1], P[2])) * inn).astype(dtype); a2 = (dotn(cross(P[2], P[0])) * inn).astype(dtype)
            a3 = (1 - a1 - a2).astype(dtype)
            ecc = (1 - 3 * np.minimum(np.minimum(a1, a2), a3)).astype(dtype)
            pw = np.power(np.maximum(ecc, 0), g2.astype(dtype)[:, None]).astype(dtype)
            G = np.exp2(pw * dtype(-0.7213475204444817)).astype(dtype)
            h = (np.abs(prn) >= 1e-8) & (ecc >= 0) & (ecc <= 10) & (G >= dtype(1.0 / 255.0)) & (px < W) & (py < H) & (px >= 0) & (py >= 0)
        hit[:, qi] = h.any(1)
    return hit


def main3d():
    """3D variant: random view-space triangles (face-on to edge-on, near to far, a few pixels to hundreds), every tile around their projection."""
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    W, H, tanx, tany = 1920, 1080, 0.8, 0.45
    z = np.exp(rng.uniform(np.log(1.5), np.log(60.0), n))
    cpx, cpy = rng.uniform(-30, W + 30, n), rng.uniform(-30, H + 30, n)
    c = np.stack([tanx * ((2 * cpx - W + 1) / W) * z, tany * ((2 * cpy - H + 1) / H) * z, z], 1)
    size_px = np.exp(rng.uniform(np.log(0.3), np.log(400), n))
    size = size_px * z * (2 * tanx / W)
    # an orthonormal frame with a random normal; grazing triangles (normal nearly perpendicular to the view direction) over-represented
    nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    graz = rng.random(n) < 0.35
    view = c / np.linalg.norm(c, axis=1, keepdims=True)
    nrm[graz] -= (nrm[graz] * view[graz]).sum(1, keepdims=True) * view[graz] * rng.uniform(0.9, 1.0, (graz.sum(), 1))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    t1 = np.cross(nrm, rng.normal(size=(n, 3))); t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
    t2 = np.cross(nrm, t1)
    ang = rng.uniform(0, 2 * np.pi, (n, 3)); rad = size[:, None] * rng.uniform(0.1, 1.0, (n, 3))
    V = np.stack([c + rad[:, k, None] * (np.cos(ang[:, k, None]) * t1 + np.sin(ang[:, k, None]) * t2) for k in range(3)], 1).astype(f32)
    Nn = np.cross(V[:, 1] - V[:, 0], V[:, 2] - V[:, 0]).astype(f32)  # unnormalised, as the reference keeps it
    keep = (V[:, :, 2].min(1) > 0.2) & (np.linalg.norm(Nn, axis=1) > 1e-12)
    V, Nn, c, size_px = V[keep], Nn[keep], c[keep], size_px[keep]
    n = len(V)
    g2 = (2 * rng.choice([1.0, 1.0, 2.0, 8.0, 50.0], n)).astype(f32)
    E = support_scale(np.ones(n, f32), g2)
    cen = (V[:, 0] + V[:, 1] + V[:, 2]) / 3
    cxp = ((cen[:, 0] / cen[:, 2] / tanx + 1) * W - 1) / 2; cyp = ((cen[:, 1] / cen[:, 2] / tany + 1) * H - 1) / 2
    cxi, cyi = np.floor(cxp / 16), np.floor(cyp / 16)
    reach = np.ceil(np.minimum(size_px * 4.0 * E.max(), 200) / 16).astype(int) + 1
    TX0 = ((cxi - reach) * 16).astype(f32); TY0 = ((cyi - reach) * 16).astype(f32)
    ext = (2 * reach * 16).astype(f32)
    q, ok = quad_setup_3d(V, E, tanx, tany, W, H, Nn, (TX0 - 1, TY0 - 1, TX0 + ext + 16, TY0 + ext + 16))
    with np.errstate(over="ignore", invalid="ignore"):
        anchor = quad_anchor(q, TX0, TY0, ext, ext)
    flagged = hits = missed32 = missed64 = aff_missed = aff_flagged = 0
    worst = None
    R = int(reach.max())
    for dy in range(-R, R + 1):
        for dx in range(-R, R + 1):
            sel = (np.abs(dx) <= reach) & (np.abs(dy) <= reach)
            if not sel.any():
                continue
            idx = np.nonzero(sel)[0]
            TX = ((cxi[idx] + dx) * 16).astype(f32); TY = ((cyi[idx] + dy) * 16).astype(f32)
            inimg = (TX > -16) & (TX < W) & (TY > -16) & (TY < H)
            idx, TX, TY = idx[inimg], TX[inimg], TY[inimg]
            if len(idx) == 0:
                continue
            sub = {k_: (val[idx] if isinstance(val, np.ndarray) else val) for k_, val in q.items()}
            with np.errstate(over="ignore", invalid="ignore"):
                m = quadrant_mask(sub, TX, TY)
                suba = {k_: ([x[idx] for x in val] if isinstance(val, list) else val[idx]) for k_, val in anchor.items()}
                ma = quadrant_mask_affine(suba, (TX - TX0[idx]).astype(f32), (TY - TY0[idx]).astype(f32), TX, TY)
            mb = np.stack([(m >> b) & 1 for b in range(4)], 1).astype(bool)
            mab = np.stack([(ma >> b) & 1 for b in range(4)], 1).astype(bool)
            h32 = pixel_hits_3d(V[idx], Nn[idx], g2[idx], TX, TY, tanx, tany, W, H, np.float32)
            h64 = pixel_hits_3d(V[idx], Nn[idx], g2[idx], TX, TY, tanx, tany, W, H, np.float64)
            bad32, bad64 = h32 & ~mb, h64 & ~mb
            if (bad32.any() or (h32 & ~mab).any()) and worst is None:
                j = np.nonzero((bad32 | (h32 & ~mab)).any(1))[0][0]
                worst = (V[idx[j]].tolist(), float(g2[idx[j]]), float(TX[j]), float(TY[j]), int(m[j]), int(ma[j]), h32[j].tolist())
            missed32 += int(bad32.sum()); missed64 += int(bad64.sum()); flagged += int((mb & ok[idx][:, None]).sum()); hits += int(((h32 | h64) & ok[idx][:, None]).sum())
            aff_missed += int((h32 & ~mab).sum()) + int((h64 & ~mab).sum()); aff_flagged += int((mab & ok[idx][:, None]).sum())
    print(f"3D triangles {n} ({int((~ok).sum())} flag every quadrant): quadrants flagged {flagged}, with a hit {hits} (tightness {hits / max(flagged, 1):.3f}); missed fp32 {missed32}, fp64 {missed64}")
    print(f"3D affine form over the rectangle: quadrants flagged {aff_flagged} (tightness {hits / max(aff_flagged, 1):.3f}); missed {aff_missed}")
    if worst:
        print("first miss:", worst)
    return 1 if (missed32 or missed64 or aff_missed) else 0


def main():
    if len(sys.argv) > 3 and sys.argv[3] == "3d":
        return main3d()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    c = rng.uniform(-40, 1960, (n, 2))
    size = np.exp(rng.uniform(np.log(0.2), np.log(600), n))
    ang = rng.uniform(0, 2 * np.pi, (n, 3))
    rad = size[:, None] * rng.uniform(0.05, 1.0, (n, 3))
    sliver = rng.random(n) < 0.3  # nearly collinear vertices
    ang[sliver, 1] = ang[sliver, 0] + np.pi + rng.normal(0, 1e-3, sliver.sum()); ang[sliver, 2] = ang[sliver, 0] + rng.normal(0, 1e-3, sliver.sum())
    v = np.zeros((n, 6))
    for k in range(3):
        v[:, 2 * k] = c[:, 0] + rad[:, k] * np.cos(ang[:, k]); v[:, 2 * k + 1] = c[:, 1] + rad[:, k] * np.sin(ang[:, k])
    v = v.astype(f32)
    op = np.where(rng.random(n) < 0.2, rng.uniform(0.0035, 0.0045, n), rng.uniform(0.0, 1.0, n)).astype(f32)
    op[rng.random(n) < 0.05] = 1.0
    g2 = (2 * rng.choice([0.5, 1.0, 1.0, 2.0, 8.0, 50.0], n)).astype(f32)
    E = support_scale(op, g2)
    q = quad_setup(v, E)
    area_ok = np.abs(1.0 / q["ia"].astype(np.float64)) >= 1e-8
    flagged = hits = missed32 = missed64 = 0
    aff_flagged = aff_missed32 = aff_missed64 = 0
    worst = None
    cxi, cyi = np.floor(c[:, 0] / 16), np.floor(c[:, 1] / 16)
    # the tile window tested around a triangle = its "rectangle" for the affine form (the emission kernel anchors at the rectangle's first tile);
    # big triangles get a window of up to 41 x 41 tiles so that the affine step is exercised over hundreds of pixels
    reach = np.ceil(np.minimum(size * 3.0, np.where(size > 100, 310, 80)) / 16).astype(int) + 1
    TX0 = ((cxi - reach) * 16).astype(f32); TY0 = ((cyi - reach) * 16).astype(f32)
    ext = (2 * reach * 16).astype(f32)
    anchor = quad_anchor(q, TX0, TY0, ext, ext)
    R = int(reach.max())
    for dy in range(-R, R + 1):
        for dx in range(-R, R + 1):
            sel = (np.abs(dx) <= reach) & (np.abs(dy) <= reach) & area_ok
            if not sel.any():
                continue
            idx = np.nonzero(sel)[0]
            TX = ((cxi[idx] + dx) * 16).astype(f32); TY = ((cyi[idx] + dy) * 16).astype(f32)
            sub = {k_: (val[idx] if isinstance(val, np.ndarray) else val) for k_, val in q.items()}
            m = quadrant_mask(sub, TX, TY)
            mb = np.stack([(m >> b) & 1 for b in range(4)], 1).astype(bool)
            suba = {k_: ([x[idx] for x in val] if isinstance(val, list) else val[idx]) for k_, val in anchor.items()}
            ma = quadrant_mask_affine(suba, (TX - TX0[idx]).astype(f32), (TY - TY0[idx]).astype(f32), TX, TY)
            mab = np.stack([(ma >> b) & 1 for b in range(4)], 1).astype(bool)
            h32 = pixel_hits(v[idx], q["ia"][idx], op[idx], g2[idx], TX, TY, np.float32)
            h64 = pixel_hits(v[idx], q["ia"][idx], op[idx], g2[idx], TX, TY, np.float64)
            bad32, bad64 = h32 & ~mb, h64 & ~mb
            if bad32.any() and worst is None:
                j = np.nonzero(bad32.any(1))[0][0]
                worst = (v[idx[j]].tolist(), float(op[idx[j]]), float(g2[idx[j]]), float(TX[j]), float(TY[j]), int(m[j]), h32[j].tolist())
            missed32 += int(bad32.sum()); missed64 += int(bad64.sum()); flagged += int(mb.sum()); hits += int((h32 | h64).sum())
            abad32, abad64 = h32 & ~mab, h64 & ~mab
            if abad32.any() and worst is None:
                j = np.nonzero(abad32.any(1))[0][0]
                worst = ("affine", v[idx[j]].tolist(), float(op[idx[j]]), float(g2[idx[j]]), float(TX[j]), float(TY[j]), int(ma[j]), h32[j].tolist())
            aff_missed32 += int(abad32.sum()); aff_missed64 += int(abad64.sum()); aff_flagged += int(mab.sum())
    print(f"triangles {n}: quadrants flagged {flagged}, with a hit {hits} (tightness {hits / max(flagged, 1):.3f}); missed fp32 {missed32}, fp64 {missed64}")
    print(f"affine form over the rectangle: quadrants flagged {aff_flagged} (tightness {hits / max(aff_flagged, 1):.3f}); missed fp32 {aff_missed32}, fp64 {aff_missed64}")
    if worst:
        print("first miss:", worst)
    return 1 if (missed32 or aff_missed32) else 0


if __name__ == "__main__":
    sys.exit(main())

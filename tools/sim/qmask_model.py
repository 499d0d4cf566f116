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


def quad_setup(v, E):
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
        q["P" + k] = (np.maximum(f32(0), f32(7) * A) + np.maximum(f32(0), f32(7) * B) - m + f32(2e-6) * f32(15) * (np.abs(A) + np.abs(B))).astype(f32)
    cx = (v1x + v2x + v3x) * f32(1.0 / 3.0); cy = (v1y + v2y + v3y) * f32(1.0 / 3.0)
    ex = np.stack([E * (v1x - cx), E * (v2x - cx), E * (v3x - cx)]); ey = np.stack([E * (v1y - cy), E * (v2y - cy), E * (v3y - cy)])
    padx = f32(0.05) + f32(4e-7) * np.abs(cx); pady = f32(0.05) + f32(4e-7) * np.abs(cy)
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


def main():
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
    worst = None
    cxi, cyi = np.floor(c[:, 0] / 16), np.floor(c[:, 1] / 16)
    reach = np.ceil(np.minimum(size * 3.0, 80) / 16).astype(int) + 1
    for dy in range(-6, 7):
        for dx in range(-6, 7):
            sel = (np.abs(dx) <= reach) & (np.abs(dy) <= reach) & area_ok
            if not sel.any():
                continue
            idx = np.nonzero(sel)[0]
            TX = ((cxi[idx] + dx) * 16).astype(f32); TY = ((cyi[idx] + dy) * 16).astype(f32)
            sub = {k_: (val[idx] if isinstance(val, np.ndarray) else val) for k_, val in q.items()}
            m = quadrant_mask(sub, TX, TY)
            mb = np.stack([(m >> b) & 1 for b in range(4)], 1).astype(bool)
            h32 = pixel_hits(v[idx], q["ia"][idx], op[idx], g2[idx], TX, TY, np.float32)
            h64 = pixel_hits(v[idx], q["ia"][idx], op[idx], g2[idx], TX, TY, np.float64)
            bad32, bad64 = h32 & ~mb, h64 & ~mb
            if bad32.any() and worst is None:
                j = np.nonzero(bad32.any(1))[0][0]
                worst = (v[idx[j]].tolist(), float(op[idx[j]]), float(g2[idx[j]]), float(TX[j]), float(TY[j]), int(m[j]), h32[j].tolist())
            missed32 += int(bad32.sum()); missed64 += int(bad64.sum()); flagged += int(mb.sum()); hits += int((h32 | h64).sum())
    print(f"triangles {n}: quadrants flagged {flagged}, with a hit {hits} (tightness {hits / max(flagged, 1):.3f}); missed fp32 {missed32}, fp64 {missed64}")
    if worst:
        print("first miss:", worst)
    return 1 if missed32 else 0


if __name__ == "__main__":
    sys.exit(main())

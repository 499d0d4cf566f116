// group_sim.c -- CPU model of the lane-group blend loop (render_group.hip) for choosing the group shape: replays the forward blend
// of every 8x8 quadrant over its tile list and counts, for several lane-group shapes, the (entry, block) pairs that survive the
// conservative cull, the lockstep wave steps (max list length over the groups of a wave, per 64-entry batch) and the blended pairs.
//   gcc -O2 -fopenmp tools/sim/group_sim.c -o tools/bin/group_sim -lm ; tools/bin/group_sim dump.bin
// dump.bin (tools/sim/dump_scene.py): int32 P, N, W, H ; float v1[P][2], v2[P][2], v3[P][2], opacity[P] ; uint32 vals[N] ; uint32 ranges[T][2]
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int bw, bh; const char *name; } Shape;
static const Shape shapes[] = {{8, 8, "1 x 8x8"}, {4, 8, "2 x 4x8"}, {4, 4, "4 x 4x4"}, {4, 2, "8 x 4x2 (w4 h2)"}, {2, 4, "8 x 2x4 (w2 h4)"}, {2, 2, "16 x 2x2"}, {4, 1, "16 x 4x1"}};
#define NS ((int)(sizeof(shapes) / sizeof(shapes[0])))

int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb");
    int32_t hdr[4];
    if (!f || fread(hdr, 4, 4, f) != 4) return 1;
    const int P = hdr[0], N = hdr[1], W = hdr[2], H = hdr[3];
    const int gx = (W + 15) / 16, gy = (H + 15) / 16, T = gx * gy;
    float *v1 = malloc(8 * (size_t)P), *v2 = malloc(8 * (size_t)P), *v3 = malloc(8 * (size_t)P), *op = malloc(4 * (size_t)P);
    uint32_t *vals = malloc(4 * (size_t)N), *ranges = malloc(8 * (size_t)T);
    if (fread(v1, 8, P, f) != (size_t)P || fread(v2, 8, P, f) != (size_t)P || fread(v3, 8, P, f) != (size_t)P || fread(op, 4, P, f) != (size_t)P ||
        fread(vals, 4, N, f) != (size_t)N || fread(ranges, 8, T, f) != (size_t)T) return 2;
    double surv[NS] = {0}, steps[NS] = {0}, surv_exact[NS] = {0}, steps_exact[NS] = {0};
    double pairs = 0, batches = 0, entries = 0, eq_surv = 0;
#pragma omp parallel for schedule(dynamic, 8) reduction(+ : surv[:NS], steps[:NS], surv_exact[:NS], steps_exact[:NS], pairs, batches, entries, eq_surv)
    for (int tile = 0; tile < T; tile++)
    {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int q = 0; q < 4; q++)
        {
            const int X0 = tx * 16 + (q & 1) * 8, Y0 = ty * 16 + (q >> 1) * 8;
            float Tt[64];
            uint64_t done = 0;
            for (int l = 0; l < 64; l++)
            {
                Tt[l] = 1.0f;
                if (X0 + (l & 7) >= W || Y0 + (l >> 3) >= H) done |= 1ull << l;
            }
            for (uint32_t base = r0; base < r1 && done != ~0ull; base += 64)
            {
                const int n = (int)((r1 - base) < 64 ? (r1 - base) : 64);
                int len[NS][64], lenx[NS][64];
                memset(len, 0, sizeof len); memset(lenx, 0, sizeof lenx);
                const uint64_t alive0 = ~done;
                int anywork = 0;
                entries += n;
                for (int e = 0; e < n; e++)
                {
                    const uint32_t id = vals[base + e];
                    const float u1x = v1[2 * id] - X0, u1y = v1[2 * id + 1] - Y0, u2x = v2[2 * id] - X0, u2y = v2[2 * id + 1] - Y0;
                    const float u3x = v3[2 * id] - X0, u3y = v3[2 * id + 1] - Y0;
                    const float area2 = (u2x - u1x) * (u3y - u1y) - (u2y - u1y) * (u3x - u1x);
                    const float ia = 1.0f / area2, o = op[id];
                    // exact geometric hit mask
                    uint64_t geo = 0;
                    float alpha[64];
                    for (int l = 0; l < 64; l++)
                    {
                        const float fx = (float)(l & 7), fy = (float)(l >> 3);
                        const float p1x = u1x - fx, p1y = u1y - fy, p2x = u2x - fx, p2y = u2y - fy, p3x = u3x - fx, p3y = u3y - fy;
                        const float a1 = (p2x * p3y - p2y * p3x) * ia, a2 = (p3x * p1y - p3y * p1x) * ia, a3 = 1.0f - a1 - a2;
                        const float mn = fminf(fminf(a1, a2), a3), ecc = 1.0f - 3.0f * mn;
                        const float al = fminf(0.99f, o * expf(-0.5f * ecc * ecc));
                        alpha[l] = al;
                        if (ecc >= 0.0f && ecc <= 10.0f && al >= 1.0f / 255.0f) geo |= 1ull << l;
                    }
                    // conservative cull (block_cull of render_group.hip, any block shape)
                    const float t = 255.0f * o;
                    float E = -1.0f;
                    if (t >= 1.0f) { E = sqrtf(2.0f * logf(t)); E = fminf(E * 1.0005f + 0.002f, 10.01f); }
                    const float C1 = (u2x * u3y - u2y * u3x) * ia, A1 = (u2y - u3y) * ia, B1 = (u3x - u2x) * ia;
                    const float C2 = (u3x * u1y - u3y * u1x) * ia, A2 = (u3y - u1y) * ia, B2 = (u1x - u3x) * ia;
                    const float A3 = -A1 - A2, B3 = -B1 - B2, C3 = 1.0f - C1 - C2;
                    const float cx = (u1x + u2x + u3x) / 3.0f, cy = (u1y + u2y + u3y) / 3.0f;
                    const float bminx = cx + E * fminf(fminf(u1x - cx, u2x - cx), u3x - cx) - 0.05f, bmaxx = cx + E * fmaxf(fmaxf(u1x - cx, u2x - cx), u3x - cx) + 0.05f;
                    const float bminy = cy + E * fminf(fminf(u1y - cy, u2y - cy), u3y - cy) - 0.05f, bmaxy = cy + E * fmaxf(fmaxf(u1y - cy, u2y - cy), u3y - cy) + 0.05f;
                    const float m = (1.0f - E) / 3.0f;
                    int any_q = 0;
                    for (int s = 0; s < NS; s++)
                    {
                        const int bw = shapes[s].bw, bh = shapes[s].bh, nbx = 8 / bw, nby = 8 / bh;
                        for (int b = 0; b < nbx * nby; b++)
                        {
                            const int bx = (b % nbx) * bw, by = (b / nbx) * bh;
                            uint64_t gm = 0;
                            for (int yy = 0; yy < bh; yy++) gm |= (((1ull << bw) - 1ull) << bx) << (8 * (by + yy));
                            if (!(alive0 & gm)) continue;
                            int ov = E > 0.0f && bminx <= bx + bw - 1 && bmaxx >= bx && bminy <= by + bh - 1 && bmaxy >= by;
                            if (ov)
                            {
                                const float k1 = C1 + A1 * bx + B1 * by + fmaxf(0.0f, (bw - 1) * A1) + fmaxf(0.0f, (bh - 1) * B1) - m + 1e-6f * (fabsf(C1) + 7.0f * (fabsf(A1) + fabsf(B1)));
                                const float k2 = C2 + A2 * bx + B2 * by + fmaxf(0.0f, (bw - 1) * A2) + fmaxf(0.0f, (bh - 1) * B2) - m + 1e-6f * (fabsf(C2) + 7.0f * (fabsf(A2) + fabsf(B2)));
                                const float k3 = C3 + A3 * bx + B3 * by + fmaxf(0.0f, (bw - 1) * A3) + fmaxf(0.0f, (bh - 1) * B3) - m + 1e-6f * (fabsf(C3) + 7.0f * (fabsf(A3) + fabsf(B3)));
                                ov = k1 >= 0.0f && k2 >= 0.0f && k3 >= 0.0f;
                            }
                            if (ov) { len[s][b]++; surv[s]++; if (s == 2) any_q = 1; }
                            if (geo & gm) { lenx[s][b]++; surv_exact[s]++; }
                        }
                    }
                    eq_surv += any_q;
                    anywork |= any_q;
                    // blend
                    uint64_t hit = geo & ~done;
                    pairs += __builtin_popcountll(hit);
                    for (int l = 0; l < 64; l++)
                        if ((hit >> l) & 1)
                        {
                            Tt[l] *= 1.0f - alpha[l];
                            if (Tt[l] <= 0.0001f) done |= 1ull << l;
                        }
                }
                batches += anywork;
                for (int s = 0; s < NS; s++)
                {
                    int mx = 0, mxx = 0;
                    for (int b = 0; b < 64; b++) { if (len[s][b] > mx) mx = len[s][b]; if (lenx[s][b] > mxx) mxx = lenx[s][b]; }
                    steps[s] += mx; steps_exact[s] += mxx;
                }
            }
        }
    }
    // ---- decoupled queues: every group walks its own queue across batch boundaries; table ring of R rows ----
    for (int R = 32; R <= 128; R += (R < 64 ? 16 : 64))
    for (int MINFREE = 8; MINFREE <= 24; MINFREE += 16)
    for (int LOW = 1; LOW <= 4; LOW *= 4)
    {
    double dsteps[NS] = {0}, dbatches[NS] = {0}, sstep[4][NS] = {{0}}, spass[4][NS] = {{0}};
#pragma omp parallel for schedule(dynamic, 8) reduction(+ : dsteps[:NS], dbatches[:NS], sstep[:4][:NS], spass[:4][:NS])
    for (int tile = 0; tile < T; tile++)
    {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        const int L = (int)(r1 - r0);
        if (L == 0) continue;
        uint64_t *geo = malloc(8 * (size_t)L);
        float *alpha = malloc(4 * 64 * (size_t)L);
        uint64_t *cm = malloc(8 * (size_t)L); // per shape: conservative block mask (bit b = block b)
        int *rowof = malloc(4 * (size_t)L);
        for (int q = 0; q < 4; q++)
        {
            const int X0 = tx * 16 + (q & 1) * 8, Y0 = ty * 16 + (q >> 1) * 8;
            for (int s = 0; s < NS; s++)
            {
                const int bw = shapes[s].bw, bh = shapes[s].bh, nbx = 8 / bw, nby = 8 / bh, nb = nbx * nby;
                uint64_t gmask[64];
                for (int b = 0; b < nb; b++)
                {
                    const int bx = (b % nbx) * bw, by = (b / nbx) * bh;
                    uint64_t gm = 0;
                    for (int yy = 0; yy < bh; yy++) gm |= (((1ull << bw) - 1ull) << bx) << (8 * (by + yy));
                    gmask[b] = gm;
                }
                for (int e = 0; e < L; e++)
                {
                    const uint32_t id = vals[r0 + e];
                    const float u1x = v1[2 * id] - X0, u1y = v1[2 * id + 1] - Y0, u2x = v2[2 * id] - X0, u2y = v2[2 * id + 1] - Y0;
                    const float u3x = v3[2 * id] - X0, u3y = v3[2 * id + 1] - Y0;
                    const float area2 = (u2x - u1x) * (u3y - u1y) - (u2y - u1y) * (u3x - u1x);
                    const float ia = 1.0f / area2, o = op[id];
                    if (s == 0)
                    {
                        uint64_t g = 0;
                        for (int l = 0; l < 64; l++)
                        {
                            const float fx = (float)(l & 7), fy = (float)(l >> 3);
                            const float p1x = u1x - fx, p1y = u1y - fy, p2x = u2x - fx, p2y = u2y - fy, p3x = u3x - fx, p3y = u3y - fy;
                            const float a1 = (p2x * p3y - p2y * p3x) * ia, a2 = (p3x * p1y - p3y * p1x) * ia, a3 = 1.0f - a1 - a2;
                            const float mn = fminf(fminf(a1, a2), a3), ecc = 1.0f - 3.0f * mn;
                            const float al = fminf(0.99f, o * expf(-0.5f * ecc * ecc));
                            alpha[64 * (size_t)e + l] = al;
                            if (ecc >= 0.0f && ecc <= 10.0f && al >= 1.0f / 255.0f) g |= 1ull << l;
                        }
                        geo[e] = g;
                    }
                    const float t = 255.0f * o;
                    float E = -1.0f;
                    if (t >= 1.0f) { E = sqrtf(2.0f * logf(t)); E = fminf(E * 1.0005f + 0.002f, 10.01f); }
                    const float C1 = (u2x * u3y - u2y * u3x) * ia, A1 = (u2y - u3y) * ia, B1 = (u3x - u2x) * ia;
                    const float C2 = (u3x * u1y - u3y * u1x) * ia, A2 = (u3y - u1y) * ia, B2 = (u1x - u3x) * ia;
                    const float A3 = -A1 - A2, B3 = -B1 - B2, C3 = 1.0f - C1 - C2;
                    const float cx = (u1x + u2x + u3x) / 3.0f, cy = (u1y + u2y + u3y) / 3.0f;
                    const float bminx = cx + E * fminf(fminf(u1x - cx, u2x - cx), u3x - cx) - 0.05f, bmaxx = cx + E * fmaxf(fmaxf(u1x - cx, u2x - cx), u3x - cx) + 0.05f;
                    const float bminy = cy + E * fminf(fminf(u1y - cy, u2y - cy), u3y - cy) - 0.05f, bmaxy = cy + E * fmaxf(fmaxf(u1y - cy, u2y - cy), u3y - cy) + 0.05f;
                    const float m = (1.0f - E) / 3.0f;
                    uint64_t c = 0;
                    for (int b = 0; b < nb; b++)
                    {
                        const int bx = (b % nbx) * bw, by = (b / nbx) * bh;
                        int ov = E > 0.0f && bminx <= bx + bw - 1 && bmaxx >= bx && bminy <= by + bh - 1 && bmaxy >= by;
                        if (ov)
                        {
                            const float k1 = C1 + A1 * bx + B1 * by + fmaxf(0.0f, (bw - 1) * A1) + fmaxf(0.0f, (bh - 1) * B1) - m + 1e-6f * (fabsf(C1) + 7.0f * (fabsf(A1) + fabsf(B1)));
                            const float k2 = C2 + A2 * bx + B2 * by + fmaxf(0.0f, (bw - 1) * A2) + fmaxf(0.0f, (bh - 1) * B2) - m + 1e-6f * (fabsf(C2) + 7.0f * (fabsf(A2) + fabsf(B2)));
                            const float k3 = C3 + A3 * bx + B3 * by + fmaxf(0.0f, (bw - 1) * A3) + fmaxf(0.0f, (bh - 1) * B3) - m + 1e-6f * (fabsf(C3) + 7.0f * (fabsf(A3) + fabsf(B3)));
                            ov = k1 >= 0.0f && k2 >= 0.0f && k3 >= 0.0f;
                        }
                        if (ov) c |= 1ull << b;
                    }
                    cm[e] = c;
                }
                // replay
                float Tt[64];
                uint64_t done = 0;
                for (int l = 0; l < 64; l++) { Tt[l] = 1.0f; if (X0 + (l & 7) >= W || Y0 + (l >> 3) >= H) done |= 1ull << l; }
                static __thread int queue[64][4096];
                int qh[64], qt[64];
                for (int b = 0; b < nb; b++) qh[b] = qt[b] = 0;
                int next = 0, rows_alloc = 0, rows_retired = 0;
                for (;;)
                {
                    // produce
                    for (;;)
                    {
                        if (next >= L || done == ~0ull) break;
                        if (rows_alloc - rows_retired + MINFREE > R) break;
                        int need = 0;
                        for (int b = 0; b < nb; b++) if ((~done & gmask[b]) && qt[b] - qh[b] < LOW) need = 1;
                        if (!need) break;
                        const int n = L - next < 64 ? L - next : 64;
                        int e;
                        for (e = next; e < next + n; e++)
                        {
                            int any = 0;
                            for (int b = 0; b < nb; b++) if (((cm[e] >> b) & 1) && (~done & gmask[b])) any = 1;
                            if (any && rows_alloc - rows_retired >= R) break; // ring full: the rest of this batch is culled again later
                            for (int b = 0; b < nb; b++)
                                if (((cm[e] >> b) & 1) && (~done & gmask[b])) { if (qt[b] - qh[b] < 4096) queue[b][(qt[b]++) & 4095] = e; }
                            rowof[e] = any ? rows_alloc++ : -1;
                        }
                        next = e;
                        dbatches[s]++;
                    }
                    int k = 1 << 30, mx = 0;
                    for (int b = 0; b < nb; b++) { const int len = qt[b] - qh[b]; if (len > 0 && len < k) k = len; if (len > mx) mx = len; }
                    if (mx == 0) { if (next >= L || done == ~0ull) break; else continue; }
                    if (k > 64) k = 64;
                    // if production is blocked only by the table being full, k = min over nonempty queues (already); if the list is exhausted, drain
                    for (int st = 0; st < k; st++)
                        for (int b = 0; b < nb; b++)
                            if (qt[b] - qh[b] > 0)
                            {
                                const int e = queue[b][(qh[b]++) & 4095];
                                const uint64_t hit = geo[e] & gmask[b] & ~done;
                                for (int l = 0; l < 64; l++)
                                    if ((hit >> l) & 1) { Tt[l] *= 1.0f - alpha[64 * (size_t)e + l]; if (Tt[l] <= 0.0001f) done |= 1ull << l; }
                            }
                    dsteps[s] += k;
                    int oldest = rows_alloc;
                    for (int b = 0; b < nb; b++) if (qt[b] - qh[b] > 0) { const int r = rowof[queue[b][qh[b] & 4095]]; if (r < oldest) oldest = r; }
                    rows_retired = oldest;
                }
                // streaming compaction (lockstep passes of exactly NRs rows, batches culled until NRs survivors are queued)
                if (R == 128 && LOW == 4 && MINFREE == 24)
                for (int ni = 0; ni < 4; ni++)
                {
                    const int NRs = ni == 0 ? 32 : ni == 1 ? 48 : ni == 2 ? 64 : 96;
                    float T2[64];
                    uint64_t dn = 0;
                    for (int l = 0; l < 64; l++) { T2[l] = 1.0f; if (X0 + (l & 7) >= W || Y0 + (l >> 3) >= H) dn |= 1ull << l; }
                    static __thread int pend[8192];
                    static __thread uint64_t pmask[8192];
                    int np = 0, nx = 0;
                    while ((nx < L || np > 0) && dn != ~0ull)
                    {
                        while (np < NRs && nx < L)
                        {
                            const int n = L - nx < 64 ? L - nx : 64;
                            uint64_t alive = 0;
                            for (int b = 0; b < nb; b++) if (~dn & gmask[b]) alive |= 1ull << b;
                            for (int e = nx; e < nx + n; e++) if (cm[e] & alive) { pend[np] = e; pmask[np++] = cm[e] & alive; }
                            nx += n;
                        }
                        const int take = np < NRs ? np : NRs;
                        int len[64] = {0}, mx = 0;
                        for (int i = 0; i < take; i++) for (int b = 0; b < nb; b++) if ((pmask[i] >> b) & 1) len[b]++;
                        for (int b = 0; b < nb; b++) if (len[b] > mx) mx = len[b];
                        sstep[ni][s] += mx; spass[ni][s]++;
                        for (int i = 0; i < take; i++)
                        {
                            const int e = pend[i];
                            uint64_t gm = 0;
                            for (int b = 0; b < nb; b++) if ((pmask[i] >> b) & 1) gm |= gmask[b];
                            const uint64_t hit = geo[e] & gm & ~dn;
                            for (int l = 0; l < 64; l++)
                                if ((hit >> l) & 1) { T2[l] *= 1.0f - alpha[64 * (size_t)e + l]; if (T2[l] <= 0.0001f) dn |= 1ull << l; }
                        }
                        memmove(pend, pend + take, sizeof(int) * (np - take));
                        memmove(pmask, pmask + take, 8 * (np - take));
                        np -= take;
                    }
                }
            }
        }
        free(geo); free(alpha); free(cm); free(rowof);
    }
    printf("decoupled queues, table ring R = %d rows, produce below %d queued when %d rows free:\n", R, LOW, MINFREE);
    for (int s = 0; s < NS; s++) printf("  %-18s wave steps %12.0f  lane occ %.3f  produce batches %.0f\n", shapes[s].name, dsteps[s], pairs / (64.0 * dsteps[s]), dbatches[s]);
    if (R == 128 && LOW == 4 && MINFREE == 24)
        for (int ni = 0; ni < 4; ni++)
        {
            printf("streaming compaction, lockstep passes of NR = %d rows:\n", ni == 0 ? 32 : ni == 1 ? 48 : ni == 2 ? 64 : 96);
            for (int s = 0; s < NS; s++) printf("  %-18s wave steps %12.0f  lane occ %.3f  passes %.0f\n", shapes[s].name, sstep[ni][s], pairs / (64.0 * sstep[ni][s]), spass[ni][s]);
        }
    }
    printf("P %d N %d  %dx%d : list entries visited %.0f, batches with work %.0f, (entry, quadrant) survivors %.0f, blended pairs %.0f\n", P, N, W, H, entries, batches, eq_surv, pairs);
    printf("%-18s %14s %12s %8s | %14s %12s %8s\n", "groups", "cull survivors", "wave steps", "lane occ", "exact survivors", "wave steps", "lane occ");
    for (int s = 0; s < NS; s++)
        printf("%-18s %14.0f %12.0f %8.3f | %14.0f %12.0f %8.3f\n", shapes[s].name, surv[s], steps[s], pairs / (64.0 * steps[s]), surv_exact[s], steps_exact[s],
               pairs / (64.0 * steps_exact[s]));
    return 0;
}

// api.hip -- the C ABI of libts2d.so (include/ts2d.h): validation, state carving, host sequencing.
//
// Host sequencing mirrors Rasterizer::forward / Rasterizer::backward (R2D/src/rasterizer.cu:101-267, 269-358)
// and the argument checks of rasterizeTrianglesForward / Backward (R2D/src/extension_interface.cu:53-81,193-199).
// Differences by design: everything is enqueued on the caller's stream (the reference uses the legacy default
// stream), the only host synchronisation is the num_rendered read-back, outputs need no pre-zeroing, and the
// zero-filled scratch of the backward is one 64-byte-per-triangle gradient record array.
// The library is built with -fvisibility=hidden: only what include/*.h declares (and, in the lab build, csrc/ts2d_lab.h) is exported.
#pragma GCC visibility push(default)
#include "../../include/ts2d.h"
#include "../../include/ts_loss.h"
#include "../../include/ts_knn.h"
#include "../../include/ts_model.h"
#include "../../include/ts_optim.h"
#ifdef TS2D_LAB
#include "ts2d_lab.h"
#endif
#pragma GCC visibility pop
#include "ts2d_common.h"
#include <atomic>
#include <chrono>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace
{
thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define TS_HIP(expr)                                                                                                   \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) return fail(TS2D_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));              \
    } while (0)

// R2D's CHECK_CUDA(debug) (auxiliary.h:358-367): with the debug flag, synchronise and surface errors per kernel.
#define TS_CHECK(flags, stream, what)                                                                                  \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e_ = hipGetLastError();                                                                             \
        if (e_ == hipSuccess && ((flags)&TS2D_FLAG_DEBUG)) e_ = hipStreamSynchronize(stream);                          \
        if (e_ != hipSuccess) return fail(TS2D_ERR_HIP, "%s: %s", what, hipGetErrorString(e_));                        \
    } while (0)

// ---- optional per-kernel timing with HIP events on the caller's stream ---------------------------------------
struct ProfRow { std::string name; double ms = 0; int64_t launches = 0; };
struct ProfPending { int row; hipEvent_t a, b; };
std::mutex g_prof_mu;
bool g_prof_on = false;
std::string g_prof_only; // when non-empty, only scopes with exactly this name are timed
std::vector<ProfRow> g_prof_rows;
std::vector<ProfPending> g_prof_pending;
std::vector<hipEvent_t> g_prof_free;

struct ProfScope
{
    hipStream_t s;
    int row = -1;
    hipEvent_t a = nullptr, b = nullptr;
    ProfScope(const char *name, hipStream_t stream) : s(stream)
    {
        if (!g_prof_on) return;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (!g_prof_only.empty() && g_prof_only != name) return;
        for (size_t i = 0; i < g_prof_rows.size(); i++)
            if (g_prof_rows[i].name == name) row = (int)i;
        if (row < 0) { g_prof_rows.push_back({name, 0, 0}); row = (int)g_prof_rows.size() - 1; }
        auto get = [&]() { hipEvent_t e; if (!g_prof_free.empty()) { e = g_prof_free.back(); g_prof_free.pop_back(); } else (void)hipEventCreate(&e); return e; };
        a = get(); b = get();
        (void)hipEventRecord(a, s);
    }
    ~ProfScope()
    {
        if (row < 0) return;
        (void)hipEventRecord(b, s);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof_pending.push_back({row, a, b});
    }
};

void prof_drain()
{
    for (auto &p : g_prof_pending)
    {
        float ms = 0;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess)
        {
            g_prof_rows[p.row].ms += ms;
            g_prof_rows[p.row].launches += 1;
        }
        g_prof_free.push_back(p.a);
        g_prof_free.push_back(p.b);
    }
    g_prof_pending.clear();
}

#ifdef TS2D_LAB
bool g_lab_all_quadrants = false; // ts2d_lab_force_all_quadrants (csrc/ts2d_lab.h)
bool g_lab_side_stream = false; // ts2d_lab_side_stream: the SH colours on a side stream (measured, not adopted: see SideLane)
int g_lab_colour_blocks = 0;    // ts2d_lab_colour_blocks: resident single-wave workgroups of the colour kernel (0 = TS_COLOUR_BLOCKS)
#endif

int validate(const ts2d_camera *cam, const ts2d_geometry *geom, uint32_t flags)
{
    if (!cam || !geom) return fail(TS2D_ERR_INVALID, "null camera/geometry");
    if (cam->width <= 0 || cam->height <= 0) return fail(TS2D_ERR_INVALID, "image size must be positive");
    if (cam->width > 65535 * TS_TILE || cam->height > 65535 * TS_TILE) return fail(TS2D_ERR_INVALID, "image too large");
    if (geom->P < 0) return fail(TS2D_ERR_INVALID, "P must be >= 0");
    if (geom->P > (int)TS_ID_MASK) return fail(TS2D_ERR_CAPACITY, "more than 2^28 - 1 triangles: the instance values keep four bits for the quadrant mask");
    if (geom->C > TS2D_MAX_CHANNELS) // extension_interface.cu:65-68
        return fail(TS2D_ERR_INVALID, "feature's num_channels can't be larger than MAX_CHANNELS");
    if (geom->C < 1) return fail(TS2D_ERR_INVALID, "need at least one colour channel");
    if (geom->gamma < 0.0f) return fail(TS2D_ERR_INVALID, "gamma must be larger than 0"); // extension_interface.cu:73-76
    if (flags & TS2D_FLAG_USE_SHS)
    {
        if (geom->C != 3) return fail(TS2D_ERR_INVALID, "SH mode renders 3 channels");
        if (geom->sh_degree < 0 || geom->sh_degree > 3) return fail(TS2D_ERR_INVALID, "sh_degree must be in 0..3");
        if ((geom->sh_degree + 1) * (geom->sh_degree + 1) > geom->M)
            return fail(TS2D_ERR_INVALID, "shs holds fewer coefficients than sh_degree needs");
        if (geom->P > 0 && !geom->shs) return fail(TS2D_ERR_INVALID, "shs is null");
    }
    else if (geom->P > 0 && !geom->feature) return fail(TS2D_ERR_INVALID, "feature is null");
    if (geom->P > 0 && (!geom->vertex || !geom->opacity)) return fail(TS2D_ERR_INVALID, "vertex/opacity is null");
    if (!cam->viewmatrix || !cam->projmatrix || !cam->campos || !geom->background)
        return fail(TS2D_ERR_INVALID, "camera matrices / background are null");
    return TS2D_OK;
}

PreprocessArgs make_pre(const ts2d_camera *cam, const ts2d_geometry *geom, uint32_t flags)
{
    PreprocessArgs a;
    a.W = cam->width; a.H = cam->height; a.P = geom->P; a.D = geom->sh_degree; a.M = geom->M; a.C = geom->C;
    a.grid_x = (cam->width + TS_TILE - 1) / TS_TILE; a.grid_y = (cam->height + TS_TILE - 1) / TS_TILE;
    a.rich_info = flags & TS2D_FLAG_RICH_INFO; a.use_shs = flags & TS2D_FLAG_USE_SHS;
    a.back_culling = flags & TS2D_FLAG_BACK_CULLING;
    a.tan_fovx = cam->tan_fovx; a.tan_fovy = cam->tan_fovy;
    a.viewmatrix = cam->viewmatrix; a.projmatrix = cam->projmatrix; a.campos = cam->campos;
    a.vertex = geom->vertex; a.shs = geom->shs; a.feature = geom->feature; a.opacity = geom->opacity;
    return a;
}

RenderArgs make_render(const ts2d_camera *cam, const ts2d_geometry *geom, uint32_t flags)
{
    RenderArgs r;
    r.W = cam->width; r.H = cam->height; r.C = geom->C;
    r.grid_x = (cam->width + TS_TILE - 1) / TS_TILE; r.grid_y = (cam->height + TS_TILE - 1) / TS_TILE;
    r.gamma = geom->gamma; r.background_depth = geom->background_depth;
    r.background_depth_dev = geom->background_depth_dev;
    r.background = geom->background;
    r.rich_info = flags & TS2D_FLAG_RICH_INFO;
    r.ablate = r.bwd_mfma = r.legacy_blend = 0;
#ifdef TS2D_LAB
    // libts2d_lab.so only (tools/build_lab.py; never the product library): measurement / triage kernels selected by environment
    // variables, read ONCE per process.  TS2D_BLEND=wave: round 1's whole-quadrant kernels (render.hip, render3d.hip);
    // TS2D_BLEND=q8: round 3's queue kernels (render_q8.hip); TS2D_BWD=mfma: render_bwd with f32 MFMA sums; TS2D_ABLATE: ablations.
    static const int s_ablate = [] { const char *e = getenv("TS2D_ABLATE"); return e ? atoi(e) : 0; }();
    static const int s_mfma = [] { const char *e = getenv("TS2D_BWD"); return (e && strcmp(e, "mfma") == 0) ? 1 : 0; }();
    static const int s_legacy = [] { const char *e = getenv("TS2D_BLEND"); return (e && strcmp(e, "wave") == 0) ? 1 : (e && strcmp(e, "q8") == 0) ? 3 : 0; }();
    r.ablate = s_ablate;
    r.bwd_mfma = s_mfma;
    r.legacy_blend = s_legacy;
#endif
    return r;
}
#ifdef TS2D_LAB
// ---- a side stream for the SH colours: LAB LIBRARY ONLY, a measured negative result (round 6, profiles/r06_side_stream.txt) --------------------
// one forward = an HBM-bound per-triangle kernel in front of a chain of latency-bound launches (depth sort, scan, emission,
// tile sort: 2-3 waves per SIMD, the HBM mostly idle), and 228 of the 327 bytes per triangle that kernel moves (the SH row -> r g b) are first
// read by the blend kernel.  Built: the per-triangle kernel without the colours (PRE_NOCOLOUR, 72 -> 34 us) and a colour kernel on a library-owned
// stream, forked behind it and joined in front of the blend kernel, throttled by its grid.  Measured at the headline, product and variant
// alternating on one box: NO grid wins -- the chain's kernels are chains of dependent memory round trips, and any background stream of bytes
// stretches every one of them (grid 512 = 2.3 TB/s beside them: depth sort 40 -> 67 us, census 21 -> 31, emission 53 -> 60: step 1.581 against
// 1.563; grid 128: the colours arrive 0.17 ms late).  A lowest-priority stream made EVERY later kernel of the process slower (preprocess_bwd +13 %)
// and a capture of the forked stream into a HIP graph crashed in hipStreamEndCapture on this stack.  The product keeps ONE launch on ONE stream.
struct SideLane { hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
#ifndef TS_COLOUR_BLOCKS
#define TS_COLOUR_BLOCKS 512 /* resident single-wave workgroups of the colour kernel: the throttle (2 per compute unit) */
#endif
SideLane *acquire_side_lane()
{
    constexpr int LANES = 4, MAXDEV = 16;
    static SideLane lanes[MAXDEV][LANES];
    static std::atomic<unsigned> next[MAXDEV];
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
    SideLane &l = lanes[dev][next[dev].fetch_add(1) % LANES];
    if (!l.s)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!l.s)
        {
            hipStream_t st = nullptr; // default priority: see above
            if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            if (hipEventCreateWithFlags(&l.fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&l.join, hipEventDisableTiming) != hipSuccess)
            {
                (void)hipGetLastError();
                (void)hipStreamDestroy(st);
                return nullptr;
            }
            l.s = st;
        }
    }
    return &l;
}
#else
struct SideLane; // (lab library only)
#endif

// Early read-back of the instance count (binning.hip, count_instances_kernel): a pinned host word + an event per call in flight
struct EarlyCount
{
    unsigned long long *host = nullptr;
    hipEvent_t ev = nullptr;
};
int forward_bin_impl(const ts2d_camera *cam, const ts2d_geometry *geom, uint32_t flags, int32_t *radii, const ts2d_state *state, hipStream_t s,
                     EarlyCount *early, SideLane **pending = nullptr);
bool acquire_early_count(EarlyCount &e)
{
    constexpr int SLOTS = 64, MAXDEV = 16; // calls that may be between their launch and their wait at the same time, per device
    static unsigned long long *ring = [] {
        void *p = nullptr; // pinned and mapped into every device's address space
        return hipHostMalloc(&p, (size_t)MAXDEV * SLOTS * sizeof(unsigned long long), hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess
                   ? (unsigned long long *)p : nullptr;
    }();
    static hipEvent_t events[MAXDEV][SLOTS] = {}; // an event belongs to the device it was created on
    static std::atomic<unsigned> next{0};
    static std::mutex mu;
    int dev = 0;
    if (!ring || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return false;
    const unsigned i = next.fetch_add(1) % SLOTS;
    {
        std::lock_guard<std::mutex> lk(mu);
        // hipEventBlockingSync: a host thread that has watched the pinned word in vain for its bounded time SLEEPS on the event instead of
        // spinning inside the runtime (eight ranks = eight cores otherwise); the wake-up latency is off the GPU's critical path,
        // the rest of the forward is queued by then
#ifndef TS2D_COUNT_EVENT_FLAGS
#define TS2D_COUNT_EVENT_FLAGS (hipEventDisableTiming | hipEventBlockingSync)
#endif
        if (!events[dev][i] && hipEventCreateWithFlags(&events[dev][i], TS2D_COUNT_EVENT_FLAGS) != hipSuccess) return false;
    }
    e.host = ring + (size_t)dev * SLOTS + i;
    e.ev = events[dev][i];
    __atomic_store_n(e.host, ~0ull, __ATOMIC_RELEASE); // "not there yet": the count is < 2^63, the publishing block overwrites this
    return true;
}

inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#endif
}

// Waits for the instance count of ONE forward: the publishing block stores it (system scope, fine-grained pinned memory) a few microseconds
// before its kernel retires.  The host watches the word itself for a bounded time (no runtime call between the store and this thread: this is
// the case when the GPU is already past the publishing kernel, or about to be) and then sleeps on the event recorded behind that kernel -- it
// does not burn a core while the GPU works through a queue of earlier launches, and a faulted launch surfaces as an error.
int wait_early_count(const EarlyCount &early, unsigned long long *n_out)
{
    // Bounded by TIME (round 6): 250 us.  The count is published ~50-120 us after the per-triangle kernel starts (that kernel + the depth sort's first
    // two launches).  A fixed iteration count (20 000 pauses = 20-60 us, rounds 3-5) was just long enough for the ctypes binding, whose own overhead
    // delayed the wait; the compiled binding reaches this loop earlier, ran out of spins on small scenes and paid the sleep's wake-up latency on the
    // critical path of EVERY step (10 k triangles: 0.30 ms per step against 0.20 through ctypes, profiles/r06_binding.txt).  A host thread still
    // never spins unbounded: behind 250 us it sleeps on the event (eight ranks must not burn eight cores on a long queue).
    unsigned long long n = ~0ull;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; spin++)
    {
        n = __atomic_load_n(early.host, __ATOMIC_ACQUIRE);
        if (n != ~0ull) break;
        cpu_relax();
        if ((spin & 63u) == 63u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(250)) break;
    }
    if (n == ~0ull)
    {
        const hipError_t q = hipEventSynchronize(early.ev);
        if (q != hipSuccess) return fail(TS2D_ERR_HIP, "waiting for the instance count: %s", hipGetErrorString(q));
        n = __atomic_load_n(early.host, __ATOMIC_ACQUIRE);
        if (n == ~0ull) return fail(TS2D_ERR_HIP, "the instance count was not published");
    }
    *n_out = n;
    return TS2D_OK;
}

// ---- capacity hints for the speculative forward ------------------------------------------------------------------------------------
// What the last forwards of a given (device, variant, image size) rendered, per triangle: the next call's binning buffer is sized for 1.25 x
// the recent maximum.  A wrong guess costs one cheap overflow (nothing is emitted) and the reference's sequence for that call, never a result.
// Slots are keyed by (device, variant, image size, CALLER KEY): train and evaluation cameras of one size, or two models in one process, keep
// separate histories when the caller names them (ts2d_set_capacity_hint_key, thread-local, default 0).
struct CapacityHint { int dev, variant, W, H; unsigned long long key; double per_triangle; unsigned long long stamp; };
thread_local unsigned long long t_hint_key = 0;
std::atomic<unsigned long long> g_speculative_overflows{0};
std::mutex g_hint_mu;
std::vector<CapacityHint> g_hints;
unsigned long long g_hint_clock = 0;

void record_instance_count(int variant, int W, int H, int P, unsigned long long n)
{
    int dev = 0;
    if (P <= 0 || hipGetDevice(&dev) != hipSuccess) return;
    const double per = (double)n / (double)P;
    std::lock_guard<std::mutex> lk(g_hint_mu);
    CapacityHint *slot = nullptr;
    for (auto &h : g_hints)
        if (h.dev == dev && h.variant == variant && h.W == W && h.H == H && h.key == t_hint_key) slot = &h;
    if (!slot)
    {
        if (g_hints.size() < 32) { g_hints.push_back({dev, variant, W, H, t_hint_key, 0.0, 0}); slot = &g_hints.back(); }
        else // recycle the entry that was used longest ago
        {
            slot = &g_hints[0];
            for (auto &h : g_hints)
                if (h.stamp < slot->stamp) slot = &h;
            *slot = {dev, variant, W, H, t_hint_key, 0.0, 0};
        }
    }
    slot->per_triangle = per > 0.97 * slot->per_triangle ? per : 0.97 * slot->per_triangle; // a decaying maximum over the recent views
    slot->stamp = ++g_hint_clock;
}
} // namespace

extern "C" {

const char *ts2d_version(void) { return "ts2d 0.1 (gfx950)"; }
const char *ts2d_last_error(void) { return g_err.c_str(); }

size_t ts2d_geometry_state_bytes(int32_t P)
{
    GeometryStateView v;
    return ts_carve_geometry(nullptr, P, v);
}
size_t ts2d_binning_state_bytes(int64_t N, int32_t W, int32_t H)
{
    BinningStateView v;
    return ts_carve_binning(nullptr, N, W, H, v);
}
size_t ts2d_image_state_bytes(int32_t W, int32_t H)
{
    ImageStateView v;
    return ts_carve_image(nullptr, W, H, v);
}
size_t ts2d_backward_scratch_bytes(int32_t P) { return (size_t)(P > 0 ? P : 0) * TS_GRAD_FLOATS * sizeof(float) + TS_ALIGN; }
int64_t ts2d_binning_capacity(size_t bytes, int32_t W, int32_t H)
{
    const int64_t c = ts_binning_capacity(bytes, W, H);
    return c < 0 ? 0 : c;
}
int64_t ts2d_instance_capacity_hint(int32_t P, int32_t W, int32_t H, uint32_t flags)
{
    int dev = 0;
    if (P <= 0 || hipGetDevice(&dev) != hipSuccess) return 0;
    const int variant = (flags & TS2D_FLAG_3D) ? 3 : 2;
    std::lock_guard<std::mutex> lk(g_hint_mu);
    for (auto &h : g_hints)
        if (h.dev == dev && h.variant == variant && h.W == W && h.H == H && h.key == t_hint_key)
        {
            h.stamp = ++g_hint_clock;
            const double want = 1.25 * h.per_triangle * (double)P + 4096.0;
            if (want >= 2147483647.0) return 0x7fffffffll;
            // eight sizes per octave: views whose counts differ by a few per cent ask the caller's allocator for the SAME block
            int64_t w = (int64_t)want, step = 1;
            while ((step << 4) <= w) step <<= 1;
            w = (w + step - 1) / step * step;
            return w > 0x7fffffffll ? 0x7fffffffll : w;
        }
    return 0;
}

int ts2d_forward_bin(const ts2d_camera *cam, const ts2d_geometry *geom, uint32_t flags, int32_t *radii,
                     const ts2d_state *state, int64_t *num_rendered, void *stream)
{
    if (int rc = validate(cam, geom, flags)) return rc;
    if (!state || !num_rendered) return fail(TS2D_ERR_INVALID, "null state/num_rendered");
    hipStream_t s = (hipStream_t)stream;
    *num_rendered = 0;
    const int P = geom->P;
    if (P == 0) return TS2D_OK; // extension_interface.cu:130
    if (!radii) return fail(TS2D_ERR_INVALID, "radii is null");
    if (!state->geometry || state->geometry_bytes < ts2d_geometry_state_bytes(P))
        return fail(TS2D_ERR_CAPACITY, "geometry state buffer too small: %zu < %zu", state->geometry_bytes,
                    ts2d_geometry_state_bytes(P));
    // The count is summed and copied right after preprocess; the depth sort and the scan are queued behind the copy, and the host
    // waits for the COPY only (the reference's blocking cudaMemcpy, rasterizer.cu:191, sits after its scan)
    EarlyCount early;
    const bool have_early = acquire_early_count(early);
    if (int rc = forward_bin_impl(cam, geom, flags, radii, state, s, have_early ? &early : nullptr)) return rc;
    unsigned long long n = 0;
    if (have_early)
    {
        if (int rc = wait_early_count(early, &n)) return rc;
    }
    else
    {
        GeometryStateView g;
        ts_carve_geometry((char *)state->geometry, P, g);
        TS_HIP(hipMemcpyAsync(&n, ts_instance_count_dev(g, P), sizeof(n), hipMemcpyDeviceToHost, s));
        TS_HIP(hipStreamSynchronize(s));
    }
    if (n > 0x7fffffffull) // the reference's int num_rendered wraps here; instance slots are 32-bit
        return fail(TS2D_ERR_CAPACITY, "%llu tile instances exceed the 2^31 - 1 the instance list can address", n);
    *num_rendered = (int64_t)n;
    record_instance_count((flags & TS2D_FLAG_3D) ? 3 : 2, cam->width, cam->height, P, n);
    return TS2D_OK;
}

} // extern "C" (reopened below, after the internal helpers)

namespace
{
// Everything after the instance count is known -- on the host (n_dev == nullptr, N exact: the reference's sequence) or only on
// the device (n_dev != nullptr, N = the capacity the binning state was carved for).
int forward_render_impl(const ts2d_camera *cam, const ts2d_geometry *geom, uint32_t flags, int64_t N, const unsigned long long *n_dev,
                        const ts2d_state *state, const ts2d_forward_out *out, hipStream_t s, SideLane *pending = nullptr)
{
    const bool rich = flags & TS2D_FLAG_RICH_INFO;
    const int P = geom->P, W = cam->width, H = cam->height;
    GeometryStateView g{};
    BinningStateView b{};
    ImageStateView im{};
    if (P > 0) ts_carve_geometry((char *)state->geometry, P, g);
    if (N > 0)
    {
        // the layout follows from the buffer's size (ts_binning_capacity): the backward finds it again without being told the capacity
        ts_carve_binning((char *)state->binning, ts_binning_capacity(state->binning_bytes, W, H), W, H, b);
        if (!n_dev) ts_binning_set_count(b, N); // the host knows the count: no launch covers more than it
    }
    ts_carve_image((char *)state->image, W, H, im);
    const RenderArgs r = make_render(cam, geom, flags);
    const int ntiles = r.grid_x * r.grid_y;
    if (n_dev) n_dev = ts_instance_count_dev(g, P);

    // tile ranges (rasterizer.cu:223) and the contribution statistics are cleared by the emission kernel, not by memsets
    if (P > 0)
    {
        {
            // quadrant masks for the blend kernels of both variants (ts2d_support.h, ts2d_common.h: QuadMaskArgs)
            QuadMaskArgs quad{(flags & TS2D_FLAG_3D) ? 3 : 2, fmaxf(0.0f, 2.0f * geom->gamma), cam->tan_fovx, cam->tan_fovy, W, H, 1.0f / (float)W, 1.0f / (float)H};
#ifdef TS2D_LAB
            if (g_lab_all_quadrants) quad.variant = 0; // every instance reaches every quadrant: what the masks must not change (tests/test_qmask_gpu.py)
#endif
            ProfScope ps("emit_keys", s);
            ts_launch_emit_keys(P, r.grid_x, ntiles, g, b, im, rich ? out->contrib_sum : nullptr, rich ? out->contrib_max : nullptr,
                                n_dev ? N : -1, im.status, quad, s);
        }
        TS_CHECK(flags, s, "emit_keys");
    }
    else ts_launch_zero_words((uint32_t *)im.ranges, 2 * (size_t)ntiles, s); // no triangles: nobody else clears the ranges (a kernel: see ts2d_backward)
    if (N > 0)
    {
        {
            ProfScope ps("tile_sort", s);
            ts_sort_pairs(b, N, n_dev, ntiles, s); // tile bits only, see binning.hip
        }
        TS_CHECK(flags, s, "tile_sort");
        {
            ProfScope ps("tile_ranges", s);
            ts_launch_tile_ranges(N, n_dev, b, im, s);
        }
        TS_CHECK(flags, s, "tile_ranges");
    }
#ifdef TS2D_LAB
    if (pending) TS_HIP(hipStreamWaitEvent(s, pending->join, 0)); // the SH colours: the blend kernel is their first reader
#endif
    {
        ProfScope ps("render_fwd", s);
#ifdef TS2D_LAB
        if ((flags & TS2D_FLAG_3D) && r.legacy_blend == 1)
            ts_launch_render3d_fwd(r, cam->tan_fovx, cam->tan_fovy, g, b, im, out->out_feature, out->depth, out->normal,
                                   out->contrib_sum, out->contrib_max, s);
        else if (!(flags & TS2D_FLAG_3D) && r.legacy_blend == 1)
            ts_launch_render_fwd(r, g, b, im, out->out_feature, out->depth, out->normal, out->contrib_sum, out->contrib_max, s);
        else if (!(flags & TS2D_FLAG_3D) && r.legacy_blend == 3)
            ts_launch_render_fwd_q8(r, g, b, im, out->out_feature, out->depth, out->normal, out->contrib_sum, out->contrib_max, s);
        else
#endif
        if (flags & TS2D_FLAG_3D)
            ts_launch_render3d_fwd_group(r, cam->tan_fovx, cam->tan_fovy, g, b, im, out->out_feature, out->depth, out->normal,
                                         out->contrib_sum, out->contrib_max, s);
        else
            ts_launch_render_fwd_group(r, g, b, im, out->out_feature, out->depth, out->normal, out->contrib_sum, out->contrib_max, s);
    }
    TS_CHECK(flags, s, "render_fwd");
    return TS2D_OK;
}

int check_forward_args(const ts2d_camera *cam, const ts2d_geometry *geom, uint32_t flags, int64_t N, const ts2d_state *state,
                       const ts2d_forward_out *out)
{
    if (int rc = validate(cam, geom, flags)) return rc;
    if (!state || !out || !out->out_feature) return fail(TS2D_ERR_INVALID, "null state/output");
    const bool rich = flags & TS2D_FLAG_RICH_INFO;
    if (rich && (!out->depth || !out->normal || (geom->P > 0 && (!out->contrib_sum || !out->contrib_max))))
        return fail(TS2D_ERR_INVALID, "rich_info outputs are null");
    const int P = geom->P, W = cam->width, H = cam->height;
    if (N < 0) return fail(TS2D_ERR_INVALID, "num_rendered < 0");
    if (N > 0x7fffffffll) return fail(TS2D_ERR_CAPACITY, "the instance list addresses at most 2^31 - 1 instances");
    if (!state->image || state->image_bytes < ts2d_image_state_bytes(W, H)) return fail(TS2D_ERR_CAPACITY, "image state buffer too small");
    if (N > 0 && (!state->binning || ts_binning_capacity(state->binning_bytes, W, H) < N))
        return fail(TS2D_ERR_CAPACITY, "binning state buffer too small");
    if (P > 0 && (!state->geometry || state->geometry_bytes < ts2d_geometry_state_bytes(P)))
        return fail(TS2D_ERR_CAPACITY, "geometry state buffer too small");
    return TS2D_OK;
}

// preprocess + depth order + instance count on the device (no host read)
int forward_bin_impl(const ts2d_camera *cam, const ts2d_geometry *geom, uint32_t flags, int32_t *radii, const ts2d_state *state, hipStream_t s,
                     EarlyCount *early, SideLane **pending)
{
    if (pending) *pending = nullptr;
    const int P = geom->P;
    GeometryStateView g;
    ts_carve_geometry((char *)state->geometry, P, g);
    const PreprocessArgs a = make_pre(cam, geom, flags);
    SideLane *lane = nullptr;
#ifdef TS2D_LAB
    // lab library only: the SH colours on a side stream beside the ordering chain (see SideLane: measured, not adopted)
    if (g_lab_side_stream && P >= 131072 && !(flags & TS2D_FLAG_DEBUG) && ts_preprocess_fwd_splittable(a)) lane = acquire_side_lane();
#endif
    {
        ProfScope ps("preprocess_fwd", s);
        if (flags & TS2D_FLAG_3D) ts_launch_preprocess3d_fwd(a, radii, g, s, lane ? 1 : 0);
        else ts_launch_preprocess_fwd(a, radii, g, s, lane ? 1 : 0);
    }
    TS_CHECK(flags, s, "preprocess_fwd");
#ifdef TS2D_LAB
    if (lane)
    {
        TS_HIP(hipEventRecord(lane->fork, s)); // behind the per-triangle kernel: the colour kernel reads its tile counts and writes into its records
        TS_HIP(hipStreamWaitEvent(lane->s, lane->fork, 0));
        {
            ProfScope ps("preprocess_colour", lane->s);
            ts_launch_preprocess_colour(a, g, (flags & TS2D_FLAG_3D) ? 3 : 2, g_lab_colour_blocks > 0 ? g_lab_colour_blocks : TS_COLOUR_BLOCKS, lane->s);
        }
        TS_HIP(hipEventRecord(lane->join, lane->s));
    }
#endif
    {
        // the first histogram of the depth sort also sums the instance count and writes it to the pinned host word itself: the host
        // waits for the event behind THIS launch only and allocates the binning buffer while the rest of the sort runs
        ProfScope ps("depth_census", s);
        ts_sort_by_depth_begin(g, P, early ? early->host : nullptr, s);
    }
    if (early) TS_HIP(hipEventRecord(early->ev, s));
    {
        ProfScope ps("depth_sort", s);
        ts_sort_by_depth_finish(g, P, s);
    }
    TS_CHECK(flags, s, "depth_sort");
    {
        ProfScope ps("scan", s);
        ts_scan_offsets(g, P, s);
    }
    TS_CHECK(flags, s, "scan");
#ifdef TS2D_LAB
    if (lane)
    {
        if (pending) *pending = lane; // the caller queues the rest of the forward on `s` and joins in front of the blend kernel
        else TS_HIP(hipStreamWaitEvent(s, lane->join, 0)); // two-call form: the join sits in front of whatever `s` runs next
    }
#endif
    return TS2D_OK;
}
} // namespace

extern "C" {

int ts2d_forward_render(const ts2d_camera *cam, const ts2d_geometry *geom, uint32_t flags, int64_t N,
                        const ts2d_state *state, const ts2d_forward_out *out, void *stream)
{
    if (int rc = check_forward_args(cam, geom, flags, N, state, out)) return rc;
    return forward_render_impl(cam, geom, flags, N, nullptr, state, out, (hipStream_t)stream);
}

int ts2d_forward(const ts2d_camera *cam, const ts2d_geometry *geom, uint32_t flags, int32_t *radii, const ts2d_state *state,
                 int64_t instance_capacity, const ts2d_forward_out *out, void *stream)
{
    if (int rc = check_forward_args(cam, geom, flags, instance_capacity, state, out)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (geom->P == 0) return forward_render_impl(cam, geom, flags, 0, nullptr, state, out, s);
    if (!radii) return fail(TS2D_ERR_INVALID, "radii is null");
    if (instance_capacity <= 0) return fail(TS2D_ERR_INVALID, "instance_capacity must be positive");
    SideLane *pending = nullptr;
    if (int rc = forward_bin_impl(cam, geom, flags, radii, state, s, nullptr, &pending)) return rc;
    static const unsigned long long on_device = 0; // any non-null marker: forward_render_impl resolves the real address
    return forward_render_impl(cam, geom, flags, instance_capacity, &on_device, state, out, s, pending);
}

int ts2d_forward_speculative(const ts2d_camera *cam, const ts2d_geometry *geom, uint32_t flags, int32_t *radii, const ts2d_state *state,
                             const ts2d_forward_out *out, int64_t *num_rendered, void *stream)
{
    if (!num_rendered) return fail(TS2D_ERR_INVALID, "null num_rendered");
    *num_rendered = 0;
    if (int rc = check_forward_args(cam, geom, flags, 0, state, out)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int P = geom->P, W = cam->width, H = cam->height;
    if (P == 0) return forward_render_impl(cam, geom, flags, 0, nullptr, state, out, s); // extension_interface.cu:130: background only
    if (!radii) return fail(TS2D_ERR_INVALID, "radii is null");
    int64_t cap = state->binning ? ts_binning_capacity(state->binning_bytes, W, H) : 0;
    // a buffer too small for even an empty binning state would queue no render, and a scene of zero instances would then leave the outputs
    // unwritten without tripping the caller's  num_rendered > capacity  test (0 > 0): refuse it; no buffer at all (NULL) is the
    // documented "first call" form (== ts2d_forward_bin, the caller then runs ts2d_forward_render)
    if (state->binning && cap <= 0) return fail(TS2D_ERR_CAPACITY, "binning state buffer too small for any instance (pass NULL to size it from num_rendered)");
    EarlyCount early;
    const bool have_early = acquire_early_count(early);
    SideLane *pending = nullptr;
    if (int rc = forward_bin_impl(cam, geom, flags, radii, state, s, have_early ? &early : nullptr, cap > 0 ? &pending : nullptr)) return rc;
    if (cap > 0)
    {
        // everything behind the count is queued for the CAPACITY before the host has seen the count: the GPU never waits for the host
        static const unsigned long long on_device = 0; // any non-null marker: forward_render_impl resolves the real address
        if (int rc = forward_render_impl(cam, geom, flags, cap, &on_device, state, out, s, pending)) return rc;
    }
    unsigned long long n = 0;
    if (have_early)
    {
        if (int rc = wait_early_count(early, &n)) return rc;
    }
    else // no pinned word (allocation refused): the count is copied behind everything that was queued -- slower, same results
    {
        GeometryStateView g;
        ts_carve_geometry((char *)state->geometry, P, g);
        TS_HIP(hipMemcpyAsync(&n, ts_instance_count_dev(g, P), sizeof(n), hipMemcpyDeviceToHost, s));
        TS_HIP(hipStreamSynchronize(s));
    }
    if (n > 0x7fffffffull) // the reference's int num_rendered wraps here; instance slots are 32-bit
        return fail(TS2D_ERR_CAPACITY, "%llu tile instances exceed the 2^31 - 1 the instance list can address", n);
    *num_rendered = (int64_t)n;
    record_instance_count((flags & TS2D_FLAG_3D) ? 3 : 2, W, H, P, n);
    if (cap > 0 && (int64_t)n > cap) g_speculative_overflows.fetch_add(1, std::memory_order_relaxed); // the caller now renders a second time
    return TS2D_OK;
}

void ts2d_set_capacity_hint_key(uint64_t key) { t_hint_key = key; }
uint64_t ts2d_speculative_overflow_count(void) { return g_speculative_overflows.load(std::memory_order_relaxed); }

int ts2d_forward_status(const ts2d_state *state, int32_t P, int32_t width, int32_t height, int32_t *overflowed, int64_t *num_rendered,
                        void *stream)
{
    if (!state || !state->image || !overflowed || !num_rendered) return fail(TS2D_ERR_INVALID, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    *overflowed = 0;
    *num_rendered = 0;
    if (P <= 0) return TS2D_OK;
    if (!state->geometry) return fail(TS2D_ERR_INVALID, "null geometry state");
    GeometryStateView g{};
    ImageStateView im{};
    ts_carve_geometry((char *)state->geometry, P, g);
    ts_carve_image((char *)state->image, width, height, im);
    unsigned long long n = 0;
    TS_HIP(hipMemcpyAsync(overflowed, im.status, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    TS_HIP(hipMemcpyAsync(&n, ts_instance_count_dev(g, P), sizeof(n), hipMemcpyDeviceToHost, s));
    TS_HIP(hipStreamSynchronize(s));
    *num_rendered = (int64_t)n;
    return TS2D_OK;
}

int ts2d_backward(const ts2d_camera *cam, const ts2d_geometry *geom, uint32_t flags, int64_t N, const int32_t *radii,
                  const ts2d_state *state, const ts2d_loss_grads *loss, void *scratch, size_t scratch_bytes,
                  const ts2d_backward_out *out, void *stream)
{
    return ts2d_backward_ranged(cam, geom, flags, N, radii, state, loss, scratch, scratch_bytes, out, 1, nullptr, stream);
}

int32_t ts2d_backward_range_rows(int32_t P, int32_t num_ranges)
{
    if (P <= 0 || num_ranges < 1) return 0;
    return ((P + num_ranges - 1) / num_ranges + 63) & ~63;
}

int ts2d_backward_ranged(const ts2d_camera *cam, const ts2d_geometry *geom, uint32_t flags, int64_t N, const int32_t *radii,
                         const ts2d_state *state, const ts2d_loss_grads *loss, void *scratch, size_t scratch_bytes,
                         const ts2d_backward_out *out, int32_t num_ranges, void *const *range_done_events, void *stream)
{
    if (num_ranges < 1 || num_ranges > 64) return fail(TS2D_ERR_INVALID, "num_ranges must be in 1..64");
    if (int rc = validate(cam, geom, flags)) return rc;
    if (!state || !loss || !out) return fail(TS2D_ERR_INVALID, "null state/loss/output");
    const bool rich = flags & TS2D_FLAG_RICH_INFO, use_shs = flags & TS2D_FLAG_USE_SHS;
    hipStream_t s = (hipStream_t)stream;
    const int P = geom->P, W = cam->width, H = cam->height;
    if (P == 0) return TS2D_OK; // extension_interface.cu:242
    if (!loss->dL_dout_feature) return fail(TS2D_ERR_INVALID, "upstream gradients are null");
    if (rich && ((loss->dL_dout_depth == nullptr) != (loss->dL_dout_normal == nullptr)))
        return fail(TS2D_ERR_INVALID, "dL_dout_depth and dL_dout_normal: both or neither");
    // RICH_INFO state, but no gradient arrives on the depth and normal images (a training iteration whose loss reads the colours only: the reference's
    // autograd then hands its kernel two images of zeros).  With dd = dn = 0 every depth / normal term of the pixel kernel is an exact zero
    // (backward.cu:419-437), i.e. it is the colour-only kernel over the same records: that one runs, grad_rec columns 10..15 stay at the zeros they
    // were cleared to, and the per-triangle kernel below is unchanged.
    const bool colour_only = rich && !loss->dL_dout_depth;
    const bool factored = use_shs && (flags & TS2D_FLAG_SH_FACTORED);
    if (!out->dL_dvertex || !out->dL_dcenter2D || !out->dL_dfeature || !out->dL_dopacity || (use_shs && !factored && !out->dL_dshs))
        return fail(TS2D_ERR_INVALID, "gradient outputs are null");
    if (!radii) return fail(TS2D_ERR_INVALID, "radii is null");
    if (!scratch || scratch_bytes < ts2d_backward_scratch_bytes(P)) return fail(TS2D_ERR_CAPACITY, "backward scratch too small");
    if (!state->geometry || state->geometry_bytes < ts2d_geometry_state_bytes(P) || !state->image ||
        state->image_bytes < ts2d_image_state_bytes(W, H) ||
        (N > 0 && (!state->binning || ts_binning_capacity(state->binning_bytes, W, H) < N)))
        return fail(TS2D_ERR_CAPACITY, "state buffers too small");
    GeometryStateView g{};
    BinningStateView b{};
    ImageStateView im{};
    ts_carve_geometry((char *)state->geometry, P, g);
    if (N > 0) ts_carve_binning((char *)state->binning, ts_binning_capacity(state->binning_bytes, W, H), W, H, b); // as the forward carved it
    ts_carve_image((char *)state->image, W, H, im);
    RenderArgs r = make_render(cam, geom, flags);
    if (colour_only) r.rich_info = false; // selects the pixel kernel's template only (the launchers dispatch on it)
    float *grad_rec = (float *)ts_align_up((size_t)scratch);

    {
        ProfScope ps("zero_grad_records", s);
        // rasterizer.cu:290-300.  A KERNEL, not hipMemsetAsync: captured into a HIP graph (GraphedStep, bench.py --hip-graph) the memset became a
        // memset NODE, and on this ROCm (7.2, torch 2.10) replays then produced gradients off by up to 1e28 -- the node does not run where the
        // stream order put it -- while every kernel-only capture is exact (profiles/r05_graph_memset_triage.txt: same script, two libraries)
        ts_launch_zero_words((uint32_t *)grad_rec, (size_t)TS_GRAD_FLOATS * (size_t)P, s);
    }
    if (N > 0)
    {
        ProfScope ps("render_bwd", s);
#ifdef TS2D_LAB
        if ((flags & TS2D_FLAG_3D) && r.legacy_blend == 1)
            ts_launch_render3d_bwd(r, cam->tan_fovx, cam->tan_fovy, g, b, im, loss->dL_dout_feature, loss->dL_dout_depth,
                                   loss->dL_dout_normal, grad_rec, s);
        else if (!(flags & TS2D_FLAG_3D) && (r.legacy_blend == 1 || r.bwd_mfma))
            ts_launch_render_bwd(r, g, b, im, loss->dL_dout_feature, loss->dL_dout_depth, loss->dL_dout_normal, grad_rec, s);
        else if (!(flags & TS2D_FLAG_3D) && r.legacy_blend == 3)
            ts_launch_render_bwd_q8(r, g, b, im, loss->dL_dout_feature, loss->dL_dout_depth, loss->dL_dout_normal, grad_rec, s);
        else
#endif
        if (flags & TS2D_FLAG_3D)
            ts_launch_render3d_bwd_group(r, cam->tan_fovx, cam->tan_fovy, g, b, im, loss->dL_dout_feature, loss->dL_dout_depth,
                                         loss->dL_dout_normal, grad_rec, s);
        else
            ts_launch_render_bwd_group(r, g, b, im, loss->dL_dout_feature, loss->dL_dout_depth, loss->dL_dout_normal, grad_rec, s);
    }
    TS_CHECK(flags, s, "render_bwd");
    {
        ProfScope ps("preprocess_bwd", s);
        const PreprocessArgs a = make_pre(cam, geom, flags);
        // The per-triangle kernel in num_ranges launches over consecutive triangle ranges (boundaries at multiples of 64 = its workgroups; every
        // per-triangle array is addressed through its base pointer, so a range is the same launch on shifted pointers) with an event behind each:
        // the rows [first, first + count) of every gradient output are final when its event fires, and an exchange of range k can start while
        // range k + 1 runs (parallel.GradBucket.reduce_ranges_async).  num_ranges = 1: the one launch of rounds 1-4.
        const int per = ((P + num_ranges - 1) / num_ranges + 63) & ~63;
        for (int k = 0; k < num_ranges; k++)
        {
            const int first = k * per, count = first < P ? (P - first < per ? P - first : per) : 0;
            if (count > 0)
            {
                PreprocessArgs ak = a;
                ak.P = count;
                ak.vertex += 9 * (size_t)first;
                if (ak.shs) ak.shs += (size_t)first * a.M * 3;
                if (ak.feature) ak.feature += (size_t)first * a.C;
                ak.opacity += first;
                GeometryStateView gk = g;
                gk.clamped += first; // the per-triangle state the backward reads besides the gradient records: the clamp flags and, in the
                if (gk.rec) gk.rec += 4 * (size_t)first; // 3D variant, the triangle's render record
                float *dshs = (factored || !out->dL_dshs) ? nullptr : out->dL_dshs + (size_t)first * a.M * 3;
                float *dfeat = out->dL_dfeature ? out->dL_dfeature + (size_t)first * a.C : nullptr;
                if (flags & TS2D_FLAG_3D)
                    ts_launch_preprocess3d_bwd(ak, radii + first, gk, grad_rec + TS_GRAD_FLOATS * (size_t)first, out->dL_dvertex + 9 * (size_t)first,
                                               out->dL_dcenter2D + 2 * (size_t)first, dshs, dfeat, out->dL_dopacity + first, s);
                else
                    ts_launch_preprocess_bwd(ak, radii + first, gk, grad_rec + TS_GRAD_FLOATS * (size_t)first, out->dL_dvertex + 9 * (size_t)first,
                                             out->dL_dcenter2D + 2 * (size_t)first, dshs, dfeat, out->dL_dopacity + first, s);
            }
            if (range_done_events && range_done_events[k]) TS_HIP(hipEventRecord((hipEvent_t)range_done_events[k], s));
        }
    }
    TS_CHECK(flags, s, "preprocess_bwd");
    return TS2D_OK;
}

int ts2d_sh_grad_expand(int32_t P, int32_t sh_degree, int32_t M, int32_t num_views, const float *vertex, const float *campos,
                        const float *dL_dcolor, float *dL_dshs, void *stream)
{
    if (P < 0 || num_views < 0) return fail(TS2D_ERR_INVALID, "P / num_views must be >= 0");
    if (sh_degree < 0 || sh_degree > 3) return fail(TS2D_ERR_INVALID, "sh_degree must be in 0..3");
    if ((sh_degree + 1) * (sh_degree + 1) > M) return fail(TS2D_ERR_INVALID, "shs holds fewer coefficients than sh_degree needs");
    if (P == 0) return TS2D_OK;
    if (!vertex || !dL_dshs || (num_views > 0 && (!campos || !dL_dcolor))) return fail(TS2D_ERR_INVALID, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    {
        ProfScope ps("sh_grad_expand", s);
        ts_launch_sh_grad_expand(P, sh_degree, M, num_views, vertex, campos, dL_dcolor, dL_dshs, s);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(TS2D_ERR_HIP, "sh_grad_expand: %s", hipGetErrorString(e));
    return TS2D_OK;
}

// ---- include/ts_loss.h ------------------------------------------------------------------------------------------------
size_t tsl_workspace_bytes(int32_t channels, int32_t height, int32_t width) { return ts_loss_workspace_bytes(channels, height, width); }

static int loss_args_ok(const float *image, const float *gt, int32_t C, int32_t H, int32_t W, const void *ws, size_t ws_bytes)
{
    if (C <= 0 || H <= 0 || W <= 0) return fail(TS2D_ERR_INVALID, "image dimensions must be positive");
    if (C > 65535 || (H + 15) / 16 > 65535) return fail(TS2D_ERR_INVALID, "image too large");
    if (!image || !gt) return fail(TS2D_ERR_INVALID, "null image");
    if (!ws || ws_bytes < ts_loss_workspace_bytes(C, H, W)) return fail(TS2D_ERR_CAPACITY, "loss workspace too small");
    return TS2D_OK;
}

int tsl_photometric_forward(const float *image, const float *gt, int32_t channels, int32_t height, int32_t width, float w_l1,
                            float w_ssim, int32_t need_grad, void *workspace, size_t workspace_bytes, float *out, void *stream)
{
    if (int rc = loss_args_ok(image, gt, channels, height, width, workspace, workspace_bytes)) return rc;
    if (!out) return fail(TS2D_ERR_INVALID, "null output");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("photometric_fwd", s);
    TS_HIP(ts_loss_forward(image, gt, channels, height, width, w_l1, w_ssim, need_grad != 0, workspace, out, s));
    return TS2D_OK;
}

int tsl_photometric_backward(const float *image, const float *gt, int32_t channels, int32_t height, int32_t width, float w_l1,
                             float w_ssim, const void *workspace, size_t workspace_bytes, const float *grad_out, float *dL_dimage,
                             void *stream)
{
    if (int rc = loss_args_ok(image, gt, channels, height, width, workspace, workspace_bytes)) return rc;
    if (!dL_dimage) return fail(TS2D_ERR_INVALID, "null output");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("photometric_bwd", s);
    TS_HIP(ts_loss_backward(image, gt, channels, height, width, w_l1, w_ssim, workspace, grad_out, dL_dimage, s));
    return TS2D_OK;
}

size_t tsl_depth_normal_workspace_bytes(int32_t height, int32_t width, double scale_factor)
{
    return ts_depth_normal_workspace_bytes(height, width, scale_factor);
}

static int depth_normal_args_ok(const float *depth, const float *normal, int32_t H, int32_t W, float tan_fovx, float tan_fovy, double scale,
                                const void *ws, size_t ws_bytes)
{
    if (H <= 0 || W <= 0) return fail(TS2D_ERR_INVALID, "height and width must be positive");
    if ((int64_t)H * W > (int64_t)16 * 1000 * 1000) return fail(TS2D_ERR_INVALID, "quantile() input tensor is too large"); // torch.quantile's own limit
    if (!(tan_fovx > 0.0f) || !(tan_fovy > 0.0f)) return fail(TS2D_ERR_INVALID, "tan_fovx / tan_fovy must be positive");
    if (scale > 0.0 && scale != 1.0 && ((int)floor((double)H * scale) < 1 || (int)floor((double)W * scale) < 1))
        return fail(TS2D_ERR_INVALID, "scale_factor leaves no pixel");
    if (!depth || !normal || !ws) return fail(TS2D_ERR_INVALID, "null pointer");
    if (ws_bytes < ts_depth_normal_workspace_bytes(H, W, scale)) return fail(TS2D_ERR_CAPACITY, "workspace too small");
    return TS2D_OK;
}

int tsl_depth_normal_forward(const float *depth, const float *normal, int32_t height, int32_t width, float tan_fovx, float tan_fovy,
                             double scale_factor, float quantile, void *workspace, size_t workspace_bytes, float *out, void *stream)
{
    if (int rc = depth_normal_args_ok(depth, normal, height, width, tan_fovx, tan_fovy, scale_factor, workspace, workspace_bytes)) return rc;
    if (!out) return fail(TS2D_ERR_INVALID, "null output");
    if (!(quantile >= 0.0f && quantile <= 1.0f)) return fail(TS2D_ERR_INVALID, "quantile() q values must be in the range [0, 1]");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("depth_normal_fwd", s);
    TS_HIP(ts_depth_normal_forward(depth, normal, height, width, tan_fovx, tan_fovy, scale_factor, quantile, workspace, out, s));
    return TS2D_OK;
}

int tsl_depth_normal_backward(const float *depth, const float *normal, int32_t height, int32_t width, float tan_fovx, float tan_fovy,
                              double scale_factor, const void *workspace, size_t workspace_bytes, const float *grad_out, float *dL_ddepth,
                              float *dL_dnormal, void *stream)
{
    if (int rc = depth_normal_args_ok(depth, normal, height, width, tan_fovx, tan_fovy, scale_factor, workspace, workspace_bytes)) return rc;
    if (!dL_ddepth && !dL_dnormal) return TS2D_OK;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("depth_normal_bwd", s);
    TS_HIP(ts_depth_normal_backward(depth, normal, height, width, tan_fovx, tan_fovy, scale_factor, workspace, grad_out, dL_ddepth, dL_dnormal, s));
    return TS2D_OK;
}

// ---- the down-sampler of render_up_scale (resample.hip) ----------------------------------------------------------------------------
static int downsample_args_ok(const void *a, const void *b, int32_t C, int32_t H, int32_t W, int32_t h, int32_t w)
{
    if (C <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0) return fail(TS2D_ERR_INVALID, "dimensions must be positive");
    if (H % h != 0 || W % w != 0 || H / h < 2 || W / w < 2) return fail(TS2D_ERR_INVALID, "the down-sampler takes integer factors >= 2 (H = f h, W = g w)");
    if (!a || !b) return fail(TS2D_ERR_INVALID, "null pointer");
    return TS2D_OK;
}
int tsl_downsample_forward(const float *in, int32_t C, int32_t H, int32_t W, int32_t h, int32_t w, float *out, void *stream)
{
    if (int rc = downsample_args_ok(in, out, C, H, W, h, w)) return rc;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("downsample_fwd", s);
    TS_HIP(ts_downsample_forward(in, C, H, W, h, w, out, s));
    return TS2D_OK;
}
int tsl_downsample_backward(const float *grad_out, int32_t C, int32_t H, int32_t W, int32_t h, int32_t w, float *grad_in, void *stream)
{
    if (int rc = downsample_args_ok(grad_out, grad_in, C, H, W, h, w)) return rc;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("downsample_bwd", s);
    TS_HIP(ts_downsample_backward(grad_out, C, H, W, h, w, grad_in, s));
    return TS2D_OK;
}

static int downsample_planes_ok(int32_t n, const float *const *a, float *const *b, int32_t H, int32_t W, int32_t h, int32_t w)
{
    if (n < 0) return fail(TS2D_ERR_INVALID, "num_planes must be >= 0");
    if (n == 0) return TS2D_OK;
    if (!a || !b) return fail(TS2D_ERR_INVALID, "null pointer");
    for (int k = 0; k < n; k++)
        if (int rc = downsample_args_ok(a[k], b[k], 1, H, W, h, w)) return rc;
    return TS2D_OK;
}
int tsl_downsample_forward_planes(int32_t n, const float *const *in_planes, int32_t H, int32_t W, int32_t h, int32_t w, float *const *out_planes, void *stream)
{
    if (int rc = downsample_planes_ok(n, in_planes, out_planes, H, W, h, w)) return rc;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("downsample_fwd", s);
    TS_HIP(ts_downsample_forward_planes(n, in_planes, H, W, h, w, out_planes, s));
    return TS2D_OK;
}
int tsl_downsample_backward_planes(int32_t n, const float *const *grad_out_planes, int32_t H, int32_t W, int32_t h, int32_t w, float *const *grad_in_planes,
                                   void *stream)
{
    if (int rc = downsample_planes_ok(n, grad_out_planes, grad_in_planes, H, W, h, w)) return rc;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("downsample_bwd", s);
    TS_HIP(ts_downsample_backward_planes(n, grad_out_planes, H, W, h, w, grad_in_planes, s));
    return TS2D_OK;
}

// ---- DoGLoss / SmoothnessLoss (aux_losses.hip) ------------------------------------------------------------------------------------
size_t tsl_aux_loss_workspace_bytes(int32_t channels, int32_t height, int32_t width, double scale_factor)
{
    return ts_aux_loss_workspace_bytes(channels, height, width, scale_factor);
}
static int aux_args_ok(int32_t C, int32_t H, int32_t W, double scale, const void *ws, size_t ws_bytes, bool need_ws)
{
    if (C <= 0 || C > 8) return fail(TS2D_ERR_INVALID, "channels must be in 1..8");
    if (H <= 0 || W <= 0) return fail(TS2D_ERR_INVALID, "height and width must be positive");
    if ((int64_t)H * W > (int64_t)16 * 1000 * 1000) return fail(TS2D_ERR_INVALID, "quantile() input tensor is too large");
    if (scale > 0.0 && scale != 1.0 && ((int)floor((double)H * scale) < 1 || (int)floor((double)W * scale) < 1))
        return fail(TS2D_ERR_INVALID, "scale_factor leaves no pixel");
    if (need_ws && (!ws || ws_bytes < ts_aux_loss_workspace_bytes(C, H, W, scale))) return fail(TS2D_ERR_CAPACITY, "workspace too small");
    return TS2D_OK;
}
int tsl_dog_mask(const float *gt, int32_t C, int32_t H, int32_t W, double sigma1, int32_t ksize1, double sigma2, int32_t ksize2, int32_t invert,
                 double scale_factor, void *workspace, size_t workspace_bytes, float *mask, void *stream)
{
    if (int rc = aux_args_ok(C, H, W, scale_factor, workspace, workspace_bytes, true)) return rc;
    if (!gt || !mask) return fail(TS2D_ERR_INVALID, "null pointer");
    if (!(sigma1 > 0.0) || !(sigma2 > 0.0) || ksize1 < 1 || ksize2 < ksize1 || ksize2 > 33 || !(ksize1 & 1) || !(ksize2 & 1))
        return fail(TS2D_ERR_INVALID, "need 0 < sigma, odd kernel sizes with ksize1 <= ksize2 <= 33");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("dog_mask", s);
    TS_HIP(ts_dog_mask(gt, C, H, W, sigma1, ksize1, sigma2, ksize2, invert, scale_factor, workspace, mask, s));
    return TS2D_OK;
}
int tsl_smoothness_mask(const float *gt, int32_t C, int32_t H, int32_t W, double scale_factor, float quantile, void *workspace, size_t workspace_bytes,
                        float *mask, void *stream)
{
    if (int rc = aux_args_ok(C, H, W, scale_factor, workspace, workspace_bytes, true)) return rc;
    if (!gt || !mask) return fail(TS2D_ERR_INVALID, "null pointer");
    if (!(quantile >= 0.0f && quantile <= 1.0f)) return fail(TS2D_ERR_INVALID, "quantile() q values must be in the range [0, 1]");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("smoothness_mask", s);
    TS_HIP(ts_smoothness_mask(gt, C, H, W, scale_factor, quantile, workspace, mask, s));
    return TS2D_OK;
}
int tsl_masked_l1_forward(const float *image, const float *gt, const float *mask, int32_t C, int32_t H, int32_t W, void *workspace, size_t workspace_bytes,
                          float *out, void *stream)
{
    if (int rc = aux_args_ok(C, H, W, 1.0, workspace, workspace_bytes, true)) return rc;
    if (!image || !gt || !mask || !out) return fail(TS2D_ERR_INVALID, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("masked_l1_fwd", s);
    TS_HIP(ts_masked_l1_forward(image, gt, mask, C, H, W, workspace, out, s));
    return TS2D_OK;
}
int tsl_masked_l1_backward(const float *image, const float *gt, const float *mask, int32_t C, int32_t H, int32_t W, const float *grad_out,
                           float *dL_dimage, void *stream)
{
    if (int rc = aux_args_ok(C, H, W, 1.0, nullptr, 0, false)) return rc;
    if (!image || !gt || !mask || !dL_dimage) return fail(TS2D_ERR_INVALID, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("masked_l1_bwd", s);
    TS_HIP(ts_masked_l1_backward(image, gt, mask, C, H, W, grad_out, dL_dimage, s));
    return TS2D_OK;
}
int tsl_scharr_smoothness_forward(const float *image, const float *mask, int32_t C, int32_t H, int32_t W, void *workspace, size_t workspace_bytes,
                                  float *out, void *stream)
{
    if (int rc = aux_args_ok(C, H, W, 1.0, workspace, workspace_bytes, true)) return rc;
    if (!image || !mask || !out) return fail(TS2D_ERR_INVALID, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("smoothness_fwd", s);
    TS_HIP(ts_scharr_smoothness_forward(image, mask, C, H, W, workspace, out, s));
    return TS2D_OK;
}
int tsl_scharr_smoothness_backward(const float *image, const float *mask, int32_t C, int32_t H, int32_t W, void *workspace, size_t workspace_bytes,
                                   const float *grad_out, float *dL_dimage, void *stream)
{
    if (int rc = aux_args_ok(C, H, W, 1.0, workspace, workspace_bytes, true)) return rc;
    if (!image || !mask || !dL_dimage) return fail(TS2D_ERR_INVALID, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("smoothness_bwd", s);
    TS_HIP(ts_scharr_smoothness_backward(image, mask, C, H, W, workspace, grad_out, dL_dimage, s));
    return TS2D_OK;
}

// ---- include/ts_knn.h -------------------------------------------------------------------------------------------------
size_t tsk_workspace_bytes(int32_t P) { return ts_knn_workspace_bytes(P); }

int tsk_mean_dist3(int32_t P, const float *points, float *mean_dist2, void *workspace, size_t workspace_bytes, void *stream)
{
    if (P < 0) return fail(TS2D_ERR_INVALID, "P must be >= 0");
    if (P == 0) return TS2D_OK;
    if (!points || !mean_dist2) return fail(TS2D_ERR_INVALID, "null pointer");
    if (!workspace || workspace_bytes < ts_knn_workspace_bytes(P)) return fail(TS2D_ERR_CAPACITY, "knn workspace too small");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("knn_mean_dist3", s);
    TS_HIP(ts_knn_mean_dist3(P, points, mean_dist2, workspace, s));
    return TS2D_OK;
}

int tsk_nearest_other(int32_t P, int32_t batch_size, const float *points, uint32_t *nearest, void *workspace,
                      size_t workspace_bytes, void *stream)
{
    if (P < 0) return fail(TS2D_ERR_INVALID, "P must be >= 0");
    if (batch_size <= 0) return fail(TS2D_ERR_INVALID, "batch_size must be greater than 0"); // interface.cu:30-33
    if (P % batch_size != 0) return fail(TS2D_ERR_INVALID, "num_points % batch_size must be 0");
    if (P == 0) return TS2D_OK;
    if (!points || !nearest) return fail(TS2D_ERR_INVALID, "null pointer");
    if (!workspace || workspace_bytes < ts_knn_workspace_bytes(P)) return fail(TS2D_ERR_CAPACITY, "knn workspace too small");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("knn_nearest_other", s);
    TS_HIP(ts_knn_nearest_other(P, batch_size, points, nearest, workspace, s));
    return TS2D_OK;
}

// ---- include/ts_optim.h -----------------------------------------------------------------------------------------------
int tso_adam_step(const tso_adam_slice *slices, int32_t num_slices, double beta1, double beta2, double eps, void *stream)
{
    if (num_slices < 0 || num_slices > TSO_MAX_SLICES) return fail(TS2D_ERR_INVALID, "num_slices must be in 0..%d", TSO_MAX_SLICES);
    if (num_slices > 0 && !slices) return fail(TS2D_ERR_INVALID, "null slices");
    if (!(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0)) return fail(TS2D_ERR_INVALID, "betas must be in [0, 1)"); // torch/optim/adam.py
    if (!(eps >= 0.0)) return fail(TS2D_ERR_INVALID, "Invalid epsilon value");
    for (int i = 0; i < num_slices; i++)
    {
        const tso_adam_slice &s = slices[i];
        if (s.count < 0) return fail(TS2D_ERR_INVALID, "slice %d: count < 0", i);
        if (s.count > 0 && (!s.param || !s.grad || !s.exp_avg || !s.exp_avg_sq)) return fail(TS2D_ERR_INVALID, "slice %d: null pointer", i);
        if (s.period < 0 || s.split < 0 || (s.period > 0 && s.split > s.period) || s.index0 < 0) return fail(TS2D_ERR_INVALID, "slice %d: bad period / split / index0", i);
        if (!(s.bias2_sqrt > 0.0f)) return fail(TS2D_ERR_INVALID, "slice %d: bias2_sqrt must be positive (step >= 1)", i);
    }
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps("adam_step", st);
    TS_HIP(ts_optim_adam_step(slices, num_slices, beta1, beta2, eps, st));
    return TS2D_OK;
}

int tso_adam_step_sh_factored(const tso_sh_factored_step *a, double beta1, double beta2, double eps, void *stream)
{
    if (!a) return fail(TS2D_ERR_INVALID, "null step");
    if (!(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0)) return fail(TS2D_ERR_INVALID, "betas must be in [0, 1)");
    if (!(eps >= 0.0)) return fail(TS2D_ERR_INVALID, "Invalid epsilon value");
    if (a->P < 0 || a->V < 1) return fail(TS2D_ERR_INVALID, "P must be >= 0 and V >= 1");
    if (a->M != 1 && a->M != 4 && a->M != 9 && a->M != 16) return fail(TS2D_ERR_INVALID, "M must be 1, 4, 9 or 16");
    if (a->sh_degree < 0 || (a->sh_degree + 1) * (a->sh_degree + 1) > a->M) return fail(TS2D_ERR_INVALID, "sh_degree does not fit M");
    if (a->P == 0) return TS2D_OK;
    if (!a->vertex || !a->campos || !a->dL_dcolor || !a->param_dc || !a->exp_avg_dc || !a->exp_avg_sq_dc) return fail(TS2D_ERR_INVALID, "null pointer");
    if (a->M > 1 && (!a->param_rest || !a->exp_avg_rest || !a->exp_avg_sq_rest)) return fail(TS2D_ERR_INVALID, "null f_rest pointer");
    if (a->dc_stride < 3 || (a->M > 1 && a->rest_stride < 3 * (a->M - 1))) return fail(TS2D_ERR_INVALID, "row strides too small");
    if (!(a->bias2_sqrt_dc > 0.0f) || (a->M > 1 && !(a->bias2_sqrt_rest > 0.0f))) return fail(TS2D_ERR_INVALID, "bias2_sqrt must be positive (step >= 1)");
    if (a->num_rows < 0 || a->num_rows > TSO_SH_ROW_SLICES) return fail(TS2D_ERR_INVALID, "num_rows must be in 0..%d", TSO_SH_ROW_SLICES);
    for (int r = 0; r < a->num_rows; r++)
    {
        if (!a->rows[r].param || !a->rows[r].grad || !a->rows[r].exp_avg || !a->rows[r].exp_avg_sq) return fail(TS2D_ERR_INVALID, "row slice %d: null pointer", r);
        if (a->rows[r].floats_per_row < 1) return fail(TS2D_ERR_INVALID, "row slice %d: floats_per_row must be >= 1", r);
        if (!(a->rows[r].bias2_sqrt > 0.0f)) return fail(TS2D_ERR_INVALID, "row slice %d: bias2_sqrt must be positive (step >= 1)", r);
    }
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps("adam_step_sh_factored", st);
    TS_HIP(ts_optim_adam_step_sh_factored(*a, beta1, beta2, eps, st));
    return TS2D_OK;
}

// ---- include/ts_model.h -----------------------------------------------------------------------------------------------
int tsm_training_statistic(int32_t P, int32_t num_views, const int32_t *radii, const float *center2D_grad, const float *contrib_sum,
                           const float *contrib_max, float *gradient_accum, float *gradient_denom, float *max_radii2D,
                           float *contrib_sum_state, float *contrib_max_state, float *contrib_denom, void *stream)
{
    if (P < 0 || num_views < 0) return fail(TS2D_ERR_INVALID, "P / num_views must be >= 0");
    if (P == 0 || num_views == 0) return TS2D_OK;
    if (!radii || !center2D_grad || !gradient_accum || !gradient_denom || !max_radii2D || !contrib_denom)
        return fail(TS2D_ERR_INVALID, "null pointer");
    if ((contrib_sum == nullptr) != (contrib_max == nullptr)) return fail(TS2D_ERR_INVALID, "contrib_sum and contrib_max go together");
    if (contrib_sum && (!contrib_sum_state || !contrib_max_state)) return fail(TS2D_ERR_INVALID, "null contribution state");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("training_statistic", s);
    TS_HIP(ts_model_training_statistic(P, num_views, radii, center2D_grad, contrib_sum, contrib_max, gradient_accum, gradient_denom,
                                       max_radii2D, contrib_sum_state, contrib_max_state, contrib_denom, s));
    return TS2D_OK;
}

size_t tsm_select_scratch_bytes(int32_t P) { return ts_model_select_scratch_bytes(P); }

int tsm_select_rows(int32_t P, const uint8_t *mask, int32_t match, uint32_t *pos, void *scratch, size_t scratch_bytes, uint32_t *count,
                    void *stream)
{
    if (P < 0 || !count) return fail(TS2D_ERR_INVALID, "P must be >= 0 and count non-null");
    *count = 0;
    if (P == 0) return TS2D_OK;
    if (!mask || !pos) return fail(TS2D_ERR_INVALID, "null pointer");
    if (!scratch || scratch_bytes < ts_model_select_scratch_bytes(P)) return fail(TS2D_ERR_CAPACITY, "select scratch too small");
    TS_HIP(ts_model_select_rows(P, mask, match, pos, (uint32_t *)scratch, count, (hipStream_t)stream));
    return TS2D_OK;
}

static int rows_ok(int64_t rows, int32_t row_bytes, const void *a, const void *b, const void *c)
{
    if (rows < 0 || row_bytes <= 0 || (row_bytes & 3)) return fail(TS2D_ERR_INVALID, "rows must be >= 0 and row_bytes a positive multiple of 4");
    if (rows > 0 && (!a || !b || !c)) return fail(TS2D_ERR_INVALID, "null pointer");
    return TS2D_OK;
}
int tsm_scatter_rows(int64_t rows, int32_t row_bytes, const uint32_t *pos, const void *src, void *dst, int64_t dst_row0, void *stream)
{
    if (int rc = rows_ok(rows, row_bytes, pos, src, dst)) return rc;
    TS_HIP(ts_model_scatter_rows(rows, row_bytes / 4, pos, src, dst, dst_row0, (hipStream_t)stream));
    return TS2D_OK;
}
int tsm_gather_rows(int64_t rows, int32_t row_bytes, const uint32_t *idx, const void *src, void *dst, int64_t dst_row0, void *stream)
{
    if (int rc = rows_ok(rows, row_bytes, idx, src, dst)) return rc;
    TS_HIP(ts_model_gather_rows(rows, row_bytes / 4, idx, src, dst, dst_row0, (hipStream_t)stream));
    return TS2D_OK;
}
int tsm_grow_classify(int32_t P, const float *vertex, float *gradient_accum, float *gradient_denom, float min_view_count, float grad_threshold,
                      float split_scale_threshold, uint8_t *code, void *stream)
{
    if (P < 0) return fail(TS2D_ERR_INVALID, "P must be >= 0");
    if (P > 0 && (!vertex || !gradient_accum || !gradient_denom || !code)) return fail(TS2D_ERR_INVALID, "null pointer");
    TS_HIP(ts_model_grow_classify(P, vertex, gradient_accum, gradient_denom, min_view_count, grad_threshold, split_scale_threshold, code,
                                  (hipStream_t)stream));
    return TS2D_OK;
}
int tsm_split_vertex(int32_t n_split, const uint32_t *parents, const float *vertex, float *child1, float *child2, void *stream)
{
    if (n_split < 0) return fail(TS2D_ERR_INVALID, "n_split must be >= 0");
    if (n_split > 0 && (!parents || !vertex || !child1 || !child2)) return fail(TS2D_ERR_INVALID, "null pointer");
    TS_HIP(ts_model_split_vertex(n_split, parents, vertex, child1, child2, (hipStream_t)stream));
    return TS2D_OK;
}
int tsm_update_mask(int32_t P, int32_t mode, const float *opacity, const float *vertex, const float *max_radii2D, float a, float b, uint8_t *mask,
                    void *stream)
{
    if (P < 0 || mode < 0 || mode > 3) return fail(TS2D_ERR_INVALID, "bad P / mode");
    if (P > 0 && (!mask || (mode <= 1 && !opacity) || (mode >= 2 && !vertex) || (mode == 2 && !max_radii2D)))
        return fail(TS2D_ERR_INVALID, "null pointer");
    TS_HIP(ts_model_update_mask(P, mode, opacity, vertex, max_radii2D, a, b, mask, (hipStream_t)stream));
    return TS2D_OK;
}
int tsm_clip(int32_t P, int32_t mode, const uint8_t *mask, float value, float *param, float *exp_avg, float *exp_avg_sq, void *stream)
{
    if (P < 0 || mode < 0 || mode > 1) return fail(TS2D_ERR_INVALID, "bad P / mode");
    if (P > 0 && (!mask || !param || ((exp_avg == nullptr) != (exp_avg_sq == nullptr)))) return fail(TS2D_ERR_INVALID, "null pointer");
    TS_HIP(ts_model_clip(P, mode, mask, value, param, exp_avg, exp_avg_sq, (hipStream_t)stream));
    return TS2D_OK;
}
int tsm_opacity_reset(int32_t P, float reset_value, float *opacity, float *exp_avg, float *exp_avg_sq, void *stream)
{
    if (P < 0) return fail(TS2D_ERR_INVALID, "P must be >= 0");
    if (P > 0 && (!opacity || ((exp_avg == nullptr) != (exp_avg_sq == nullptr)))) return fail(TS2D_ERR_INVALID, "null pointer");
    TS_HIP(ts_model_opacity_reset(P, reset_value, opacity, exp_avg, exp_avg_sq, (hipStream_t)stream));
    return TS2D_OK;
}

int tsm_max_vertex_distance(int32_t n_vertices, const float *vertex, const float *camera_center, float *out, void *stream)
{
    if (n_vertices < 0) return fail(TS2D_ERR_INVALID, "n_vertices must be >= 0");
    if (!out || (n_vertices > 0 && (!vertex || !camera_center))) return fail(TS2D_ERR_INVALID, "null pointer");
    TS_HIP(ts_model_max_distance(n_vertices, vertex, camera_center, out, (hipStream_t)stream));
    return TS2D_OK;
}

#ifdef TS2D_LAB // csrc/ts2d_lab.h: diagnostics and comparators that only tools/bin/libts2d_lab.so carries
int ts2d_debug_read_state(const ts2d_state *state, int32_t P, int64_t N, int32_t W, int32_t H, int32_t field, void *dst,
                          size_t dst_bytes, void *stream)
{
    if (!state || !dst) return fail(TS2D_ERR_INVALID, "null state/dst");
    hipStream_t s = (hipStream_t)stream;
    GeometryStateView g{};
    BinningStateView b{};
    ImageStateView im{};
    if (P > 0 && state->geometry) ts_carve_geometry((char *)state->geometry, P, g);
    if (N > 0 && state->binning) ts_carve_binning((char *)state->binning, ts_binning_capacity(state->binning_bytes, W, H), W, H, b);
    if (state->image) ts_carve_image((char *)state->image, W, H, im);
    const int gx = (W + TS_TILE - 1) / TS_TILE, gy = (H + TS_TILE - 1) / TS_TILE;
    const void *src = nullptr;
    size_t bytes = 0;
    std::vector<float> recs;
    auto need_recs = [&]() -> int {
        recs.resize((size_t)P * TS_REC_FLOATS);
        if (P == 0) return TS2D_OK;
        TS_HIP(hipMemcpyAsync(recs.data(), g.rec, recs.size() * sizeof(float), hipMemcpyDeviceToHost, s));
        TS_HIP(hipStreamSynchronize(s));
        return TS2D_OK;
    };
    auto from_recs = [&](std::initializer_list<int> cols, bool area) -> int {
        if (int rc = need_recs()) return rc;
        const size_t n = area ? 1 : cols.size();
        if (dst_bytes < (size_t)P * n * sizeof(float)) return fail(TS2D_ERR_CAPACITY, "dst too small");
        float *o = (float *)dst;
        for (int i = 0; i < P; i++)
        {
            const float *r = &recs[(size_t)i * TS_REC_FLOATS];
            if (area) o[i] = (r[2] - r[0]) * (r[5] - r[1]) - (r[3] - r[1]) * (r[4] - r[0]);
            else { size_t c = 0; for (int col : cols) o[(size_t)i * n + c++] = r[col]; }
        }
        return TS2D_OK;
    };
    switch (field)
    {
    case 0: return from_recs({0, 1, 2, 3, 4, 5}, false);
    case 1: return from_recs({}, true);
    case 2: return from_recs({10, 11, 12}, false);
    case 3: return from_recs({13, 14, 15}, false);
    case 5: return from_recs({7, 8, 9}, false);
    case 4: src = g.depth; bytes = (size_t)P * 4; break;
    case 6: src = g.clamped; bytes = (size_t)P; break;
    case 7: src = g.offsets; bytes = (size_t)P * 4; break;
    case 8: src = g.tiles_touched; bytes = (size_t)P * 4; break;
    case 9:
    {
        std::vector<uint2> rc((size_t)P);
        if (dst_bytes < (size_t)P * 16) return fail(TS2D_ERR_CAPACITY, "dst too small");
        if (P > 0)
        {
            TS_HIP(hipMemcpyAsync(rc.data(), g.rect, (size_t)P * sizeof(uint2), hipMemcpyDeviceToHost, s));
            TS_HIP(hipStreamSynchronize(s));
        }
        uint32_t *o = (uint32_t *)dst;
        for (int i = 0; i < P; i++)
        {
            o[4 * i] = rc[i].x & 0xffffu; o[4 * i + 1] = rc[i].x >> 16;
            o[4 * i + 2] = rc[i].y & 0xffffu; o[4 * i + 3] = rc[i].y >> 16;
        }
        return TS2D_OK;
    }
    case 10: // the reference's 64-bit key (tile << 32 | depth bits) of every sorted instance, rebuilt on the host
    {
        if (dst_bytes < (size_t)N * 8) return fail(TS2D_ERR_CAPACITY, "dst too small");
        std::vector<uint32_t> tile((size_t)N), vals((size_t)N), depth((size_t)P);
        if (N > 0)
        {
            TS_HIP(hipMemcpyAsync(tile.data(), b.tile, (size_t)N * 4, hipMemcpyDeviceToHost, s));
            TS_HIP(hipMemcpyAsync(vals.data(), b.vals, (size_t)N * 4, hipMemcpyDeviceToHost, s));
            TS_HIP(hipMemcpyAsync(depth.data(), g.depth, (size_t)P * 4, hipMemcpyDeviceToHost, s));
            TS_HIP(hipStreamSynchronize(s));
        }
        uint64_t *o = (uint64_t *)dst;
        for (int64_t i = 0; i < N; i++) o[i] = ((uint64_t)tile[i] << 32) | depth[vals[i] & TS_ID_MASK]; // id bits (the top four: quadrant mask, ts2d_support.h)
        return TS2D_OK;
    }
    case 11: src = b.vals; bytes = (size_t)N * 4; break;
    case 12: src = im.ranges; bytes = (size_t)gx * gy * 8; break;
    case 13: src = im.n_contrib; bytes = (size_t)W * H * 4; break;
    case 14: src = im.final_T; bytes = (size_t)W * H * 4; break;
    case 15: src = b.k[(b.passes & 1) ^ 1]; bytes = (size_t)N * 4; break; // the ping-pong partner of the sorted list
    case 16: src = b.v[(b.passes & 1) ^ 1]; bytes = (size_t)N * 4; break;
    case 17: // triangle ids in depth order: the 4th pass's output, or the 3rd's when the census found the top key byte constant
    {
        uint32_t top_const = 0;
        if (P > 0)
        {
            TS_HIP(hipMemcpyAsync(&top_const, g.top_const, 4, hipMemcpyDeviceToHost, s));
            TS_HIP(hipStreamSynchronize(s));
        }
        src = top_const ? g.sv[0] : g.sv[1];
        bytes = (size_t)P * 4;
        break;
    }
    case 18: src = g.rec; bytes = (size_t)P * TS_REC_FLOATS * 4; break; // raw 64-byte render records (2D or 3D layout)
    default: return fail(TS2D_ERR_INVALID, "unknown field %d", field);
    }
    if (dst_bytes < bytes) return fail(TS2D_ERR_CAPACITY, "dst too small");
    if (bytes)
    {
        TS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
        TS_HIP(hipStreamSynchronize(s));
    }
    return TS2D_OK;
}

// ts2d_test_sort_pairs / ts2d_test_inclusive_scan_rocprim: tools/lab/lab_hooks.hip (the rocPRIM comparators live there, outside the product's objects)
void ts2d_lab_force_ticket_passes(int on) { ts_force_ticket_passes(on != 0); }
void ts2d_lab_force_all_quadrants(int on) { g_lab_all_quadrants = on != 0; }
void ts2d_lab_side_stream(int on) { g_lab_side_stream = on != 0; }
void ts2d_lab_colour_blocks(int blocks) { g_lab_colour_blocks = blocks; }
void ts2d_lab_force_depth_pass4(int on) { ts_force_depth_pass4(on != 0); }
void ts2d_lab_depth_split(int mode, int bucket_cap) { ts_lab_depth_split(mode, bucket_cap); }
#endif // TS2D_LAB

void ts2d_profile_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
}
void ts2d_profile_only(const char *name)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_only = name ? name : "";
}
void ts2d_profile_reset(void)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof_drain();
    g_prof_rows.clear();
}
int ts2d_profile_read(int32_t index, char *name, size_t name_bytes, double *total_ms, int64_t *launches)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof_drain();
    if (index < 0 || (size_t)index >= g_prof_rows.size()) return TS2D_ERR_INVALID;
    const ProfRow &r = g_prof_rows[index];
    if (name && name_bytes) { strncpy(name, r.name.c_str(), name_bytes - 1); name[name_bytes - 1] = 0; }
    if (total_ms) *total_ms = r.ms;
    if (launches) *launches = r.launches;
    return TS2D_OK;
}
} // extern "C"

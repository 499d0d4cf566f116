// ts2d_torch_ext.cpp -- the reference-side binding of libts2d.so: a torch C++ extension with EXACTLY the two entry points
// the reference exports through pybind (R2D/ext.cpp:4-9) and their signatures (R2D/src/extension_interface.h:7-62):
//     rasterize_triangles(...)           replaces rasterizeTrianglesForward   (R2D/src/extension_interface.cu:19-152)
//     rasterize_triangles_backward(...)  replaces rasterizeTrianglesBackward  (R2D/src/extension_interface.cu:154-260)
// A maintainer who keeps the reference's build (setup.py + ext.cpp) swaps extension_interface.cu / rasterizer.cu / forward.cu /
// backward.cu for this one file and links -lts2d; `diff_triangle_rasterization_2D/__init__.py` of the reference then works
// unchanged on top of it.  Argument checks, error texts, output shapes / dtypes and ownership mirror the reference; device
// memory comes from torch's allocator, kernels are enqueued on torch's current stream.
// Built by bindings/build_torch_ext.py (hipcc, in-tree); exercised by tests/test_binding_gpu.py.
#include <torch/extension.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h> // ROCm builds of torch: guard / stream types behind the "cuda" device type
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <optional>
#include <tuple>
#include <vector>

#include "../../include/ts2d.h"

namespace
{
struct Shape
{
    int P, H, W, C, M;
    bool use_shs;
};

Shape derive(const torch::Tensor &vertex, const torch::Tensor &shs, const torch::Tensor &feature, int H, int W)
{
    Shape s;
    s.P = (int)vertex.size(0);
    s.H = H;
    s.W = W;
    s.use_shs = feature.dim() <= 1 || (feature.size(0) == 0 && shs.size(0) > 0); // extension_interface.cu:44
    s.C = s.use_shs ? 3 : (int)feature.size(1);
    s.M = (shs.dim() >= 2 && shs.size(0) != 0) ? (int)shs.size(1) : 0;
    return s;
}

const float *fptr(const torch::Tensor &t) { return t.numel() ? t.data_ptr<float>() : nullptr; }
float *fptr_mut(torch::Tensor &t) { return t.numel() ? t.data_ptr<float>() : nullptr; }

void check(int rc, const char *what)
{
    if (rc != TS2D_OK) AT_ERROR(what, ": ", ts2d_last_error());
}
} // namespace

// The package's own entry point (diff_triangle_rasterization_2D/_C.py, which prefers this module over its ctypes binding since round 6: a forward +
// backward through ctypes costs 0.3-0.4 ms of host time, what bounds every scene below ~100 k triangles).  The reference's signature plus what the
// package adds to it: variant (2 / 3 = the 3D rasterizer, TS2D_FLAG_3D), instance_capacity (> 0: the sync-free ts2d_forward; 0: the speculative
// forward), background_depth_dev (the model's 0-dim device tensor handed over as a pointer instead of a synchronising float).
std::tuple<int64_t, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
rasterizeTrianglesForwardEx(const int image_width, const int image_height, const float tan_fovx, const float tan_fovy,
                            const torch::Tensor &viewmatrix_, const torch::Tensor &projmatrix_, const torch::Tensor &campos_, const int sh_degree,
                            const float gamma, const float scale_modifier, const float background_depth, const torch::Tensor &background_,
                            const torch::Tensor &vertex_, const torch::Tensor &shs_, const torch::Tensor &feature_, const torch::Tensor &opacity_,
                            const bool back_culling, const bool rich_info, const bool debug, const int variant, const int64_t instance_capacity,
                            const std::optional<torch::Tensor> &background_depth_dev)
{
    // R3D/src/extension_interface.cu:82-92 takes .contiguous() of every input where the 2D module raises
    const bool v3 = variant == 3;
    const torch::Tensor viewmatrix = v3 ? viewmatrix_.contiguous() : viewmatrix_, projmatrix = v3 ? projmatrix_.contiguous() : projmatrix_;
    const torch::Tensor campos = v3 ? campos_.contiguous() : campos_, background = v3 ? background_.contiguous() : background_;
    const torch::Tensor vertex = v3 ? vertex_.contiguous() : vertex_, shs = v3 ? shs_.contiguous() : shs_;
    const torch::Tensor feature = v3 ? feature_.contiguous() : feature_, opacity = v3 ? opacity_.contiguous() : opacity_;
    // extension_interface.cu:53-81
    if (vertex.ndimension() != 3 || vertex.size(1) != 3 || vertex.size(2) != 3) AT_ERROR("vertex must have dimensions (num_points, 3, 3)");
    const Shape s = derive(vertex, shs, feature, image_height, image_width);
    if (!s.use_shs && feature.ndimension() != 2) AT_ERROR("feature must have dimensions (num_points, num_channels)");
    if (s.use_shs && shs.ndimension() != 3) AT_ERROR("shs must have dimensions (num_points, (1 + sh_degree) ** 2, 3)");
    if (s.C > TS2D_MAX_CHANNELS) AT_ERROR("feature's num_channels can't be larger than MAX_CHANNELS");
    if (s.C != background.size(0)) AT_ERROR("background must have the same number of channels as feature");
    if (gamma < 0.0f) AT_ERROR("gamma must be larger than 0");
    for (const torch::Tensor *t : {&viewmatrix, &projmatrix, &campos, &background, &vertex, &shs, &feature, &opacity})
        if (!t->is_contiguous()) AT_ERROR("input tensors must be contiguous");
    if (!vertex.is_cuda()) AT_ERROR("diff_triangle_rasterization_2D (MI355X build) needs tensors on a HIP device; there is no CPU fallback");
    for (const torch::Tensor *t : {&viewmatrix, &projmatrix, &campos, &background, &vertex, s.use_shs ? &shs : &feature, &opacity})
        if (t->numel() > 0 && t->scalar_type() != torch::kFloat32) AT_ERROR("expected scalar type Float");

    c10::hip::OptionalHIPGuardMasqueradingAsCUDA device_guard(vertex.device());
    void *stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
    auto f32 = vertex.options().dtype(torch::kFloat32), i32 = vertex.options().dtype(torch::kInt32), u8 = vertex.options().dtype(torch::kByte);
    const int P = s.P, H = s.H, W = s.W;
    // the library writes every element when P > 0 (culled triangles get explicit zeros): no zero-fill pass
    auto alloc = [&](std::vector<int64_t> shape, const torch::TensorOptions &o) { return P == 0 ? torch::zeros(shape, o) : torch::empty(shape, o); };
    torch::Tensor out_feature = alloc({s.C, H, W}, f32), radii = alloc({P}, i32);
    torch::Tensor depth = rich_info ? alloc({H, W}, f32) : torch::empty({0}, f32);
    torch::Tensor normal = rich_info ? alloc({3, H, W}, f32) : torch::empty({0}, f32);
    torch::Tensor contrib_sum = rich_info ? alloc({P}, f32) : torch::empty({0}, f32);
    torch::Tensor contrib_max = rich_info ? alloc({P}, f32) : torch::empty({0}, f32);
    torch::Tensor geometryBuffer = torch::empty({0}, u8), binningBuffer = torch::empty({0}, u8), imageBuffer = torch::empty({0}, u8);
    int64_t num_rendered = 0;
    if (P != 0) // extension_interface.cu:130
    {
        ts2d_camera cam{W, H, tan_fovx, tan_fovy, fptr(viewmatrix), fptr(projmatrix), fptr(campos)};
        const float *bg_dev = nullptr;
        if (background_depth_dev.has_value() && background_depth_dev->defined())
        {
            if (!background_depth_dev->is_cuda() || background_depth_dev->scalar_type() != torch::kFloat32 || background_depth_dev->numel() != 1)
                AT_ERROR("background_depth must be a float or a one-element float32 tensor on the HIP device");
            bg_dev = background_depth_dev->data_ptr<float>();
        }
        ts2d_geometry geom{P, sh_degree, s.M, s.C, gamma, scale_modifier, bg_dev ? 0.0f : background_depth, fptr(background), fptr(vertex),
                           s.use_shs ? fptr(shs) : nullptr, s.use_shs ? nullptr : fptr(feature), fptr(opacity), bg_dev};
        const uint32_t flags = (back_culling ? TS2D_FLAG_BACK_CULLING : 0u) | (rich_info ? TS2D_FLAG_RICH_INFO : 0u) |
                               (debug ? TS2D_FLAG_DEBUG : 0u) | (s.use_shs ? TS2D_FLAG_USE_SHS : 0u) | (v3 ? TS2D_FLAG_3D : 0u);
        geometryBuffer = torch::empty({(int64_t)ts2d_geometry_state_bytes(P)}, u8);
        imageBuffer = torch::empty({(int64_t)ts2d_image_state_bytes(W, H)}, u8);
        if (instance_capacity > 0) // sync-free forward: nothing is read back; num_rendered = the capacity (it sizes the state for the backward)
        {
            binningBuffer = torch::empty({(int64_t)ts2d_binning_state_bytes(instance_capacity, W, H)}, u8);
            ts2d_state st{geometryBuffer.data_ptr(), (size_t)geometryBuffer.numel(), binningBuffer.data_ptr(), (size_t)binningBuffer.numel(),
                          imageBuffer.data_ptr(), (size_t)imageBuffer.numel()};
            ts2d_forward_out out{fptr_mut(out_feature), fptr_mut(depth), fptr_mut(normal), fptr_mut(contrib_sum), fptr_mut(contrib_max)};
            check(ts2d_forward(&cam, &geom, flags, radii.data_ptr<int>(), &st, instance_capacity, &out, stream), "rasterize_triangles");
            return std::make_tuple(instance_capacity, out_feature, radii, depth, normal, contrib_sum, contrib_max, geometryBuffer, binningBuffer, imageBuffer);
        }
        // Rasterizer::forward (rasterizer.cu:101-267) with its num_rendered read-back off the GPU's critical path: the binning buffer is sized
        // from what recent forwards rendered, everything is queued for that capacity, then the host waits for the exact count only
        const int64_t guess = ts2d_instance_capacity_hint(P, W, H, flags);
        if (guess > 0) binningBuffer = torch::empty({(int64_t)ts2d_binning_state_bytes(guess, W, H)}, u8);
        ts2d_state st{geometryBuffer.data_ptr(), (size_t)geometryBuffer.numel(), guess > 0 ? binningBuffer.data_ptr() : nullptr,
                      (size_t)binningBuffer.numel(), imageBuffer.data_ptr(), (size_t)imageBuffer.numel()};
        ts2d_forward_out out{fptr_mut(out_feature), fptr_mut(depth), fptr_mut(normal), fptr_mut(contrib_sum), fptr_mut(contrib_max)};
        check(ts2d_forward_speculative(&cam, &geom, flags, radii.data_ptr<int>(), &st, &out, &num_rendered, stream), "rasterize_triangles");
        if (guess <= 0 || num_rendered > ts2d_binning_capacity(st.binning_bytes, W, H))
        {
            // no history yet, or the guess was too small (nothing was emitted): the second half with the exact size (rasterizer.cu:195-266)
            binningBuffer = torch::empty({(int64_t)ts2d_binning_state_bytes(num_rendered, W, H)}, u8);
            st.binning = binningBuffer.data_ptr();
            st.binning_bytes = (size_t)binningBuffer.numel();
            check(ts2d_forward_render(&cam, &geom, flags, num_rendered, &st, &out, stream), "rasterize_triangles");
        }
    }
    return std::make_tuple(num_rendered, out_feature, radii, depth, normal, contrib_sum, contrib_max, geometryBuffer, binningBuffer, imageBuffer);
}

// R2D/ext.cpp:6 -- the reference's signature, nothing added
std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
rasterizeTrianglesForward(const int image_width, const int image_height, const float tan_fovx, const float tan_fovy,
                          const torch::Tensor &viewmatrix, const torch::Tensor &projmatrix, const torch::Tensor &campos, const int sh_degree,
                          const float gamma, const float scale_modifier, const float background_depth, const torch::Tensor &background,
                          const torch::Tensor &vertex, const torch::Tensor &shs, const torch::Tensor &feature, const torch::Tensor &opacity,
                          const bool back_culling, const bool rich_info, const bool debug)
{
    auto r = rasterizeTrianglesForwardEx(image_width, image_height, tan_fovx, tan_fovy, viewmatrix, projmatrix, campos, sh_degree, gamma, scale_modifier,
                                         background_depth, background, vertex, shs, feature, opacity, back_culling, rich_info, debug, 2, 0, std::nullopt);
    return std::make_tuple((int)std::get<0>(r), std::get<1>(r), std::get<2>(r), std::get<3>(r), std::get<4>(r), std::get<5>(r), std::get<6>(r), std::get<7>(r),
                           std::get<8>(r), std::get<9>(r));
}

// The package's backward: the reference's signature plus variant, sh_factored (TS2D_FLAG_SH_FACTORED: no dense dL_dshs, the fourth result is the
// clamp-masked colour gradient), preallocated outputs (parallel.GradBucket: the kernels write straight into the exchange bucket), the background
// depth as a device pointer, and the event handles of a ranged backward (ts2d_backward_ranged).  dL_dshs is an undefined tensor (None) when
// sh_factored.
std::tuple<torch::Tensor, torch::Tensor, std::optional<torch::Tensor>, torch::Tensor, torch::Tensor>
rasterizeTrianglesBackwardEx(const float tan_fovx, const float tan_fovy, const torch::Tensor &viewmatrix_, const torch::Tensor &projmatrix_,
                             const torch::Tensor &campos_, const int sh_degree, const float gamma, const float scale_modifier,
                             const float background_depth, const torch::Tensor &background_, const torch::Tensor &vertex_, const torch::Tensor &shs_,
                             const torch::Tensor &feature_, const torch::Tensor &opacity_, const int64_t num_rendered, const torch::Tensor &radii_,
                             const torch::Tensor &geometryBuffer, const torch::Tensor &binningBuffer, const torch::Tensor &imageBuffer,
                             const torch::Tensor &dL_dout_feature_, const torch::Tensor &dL_dout_depth_, const torch::Tensor &dL_dout_normal_,
                             const bool rich_info, const bool debug, const int variant, const bool sh_factored_,
                             const std::optional<torch::Tensor> &out_vertex, const std::optional<torch::Tensor> &out_center2D,
                             const std::optional<torch::Tensor> &out_color, const std::optional<torch::Tensor> &out_opacity,
                             const std::optional<torch::Tensor> &background_depth_dev, const std::vector<int64_t> &range_events)
{
    const bool v3 = variant == 3; // R3D/src/extension_interface.cu:186-206: .contiguous() instead of the 2D module's error
    const torch::Tensor viewmatrix = v3 ? viewmatrix_.contiguous() : viewmatrix_, projmatrix = v3 ? projmatrix_.contiguous() : projmatrix_;
    const torch::Tensor campos = v3 ? campos_.contiguous() : campos_, background = v3 ? background_.contiguous() : background_;
    const torch::Tensor vertex = v3 ? vertex_.contiguous() : vertex_, shs = v3 ? shs_.contiguous() : shs_;
    const torch::Tensor feature = v3 ? feature_.contiguous() : feature_, opacity = v3 ? opacity_.contiguous() : opacity_, radii = v3 ? radii_.contiguous() : radii_;
    const torch::Tensor dL_dout_feature = v3 ? dL_dout_feature_.contiguous() : dL_dout_feature_, dL_dout_depth = v3 ? dL_dout_depth_.contiguous() : dL_dout_depth_;
    const torch::Tensor dL_dout_normal = v3 ? dL_dout_normal_.contiguous() : dL_dout_normal_;
    const Shape s = derive(vertex, shs, feature, (int)dL_dout_feature.size(1), (int)dL_dout_feature.size(2)); // extension_interface.cu:182-183
    for (const torch::Tensor *t : {&viewmatrix, &projmatrix, &campos, &background, &vertex, &shs, &feature, &opacity, &radii, &geometryBuffer,
                                   &binningBuffer, &imageBuffer, &dL_dout_feature, &dL_dout_depth, &dL_dout_normal})
        if (!t->is_contiguous()) AT_ERROR("input tensors must be contiguous"); // extension_interface.cu:193-199
    if (!vertex.is_cuda()) AT_ERROR("diff_triangle_rasterization_2D (MI355X build) needs tensors on a HIP device; there is no CPU fallback");
    for (const torch::Tensor *t : {&dL_dout_feature, &dL_dout_depth, &dL_dout_normal})
        if (t->numel() > 0 && t->scalar_type() != torch::kFloat32) AT_ERROR("expected scalar type Float");
    c10::hip::OptionalHIPGuardMasqueradingAsCUDA device_guard(vertex.device());
    void *stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
    auto opts = vertex.options();
    const int P = s.P;
    auto alloc = [&](std::vector<int64_t> shape) { return P == 0 ? torch::zeros(shape, opts) : torch::empty(shape, opts); };
    // a preallocated output (the exchange bucket's view) or a fresh tensor
    auto placed = [&](const std::optional<torch::Tensor> &given, std::vector<int64_t> shape, const char *name) {
        if (!given.has_value() || !given->defined()) return alloc(shape);
        if (given->sizes().vec() != shape || !given->is_contiguous() || given->scalar_type() != torch::kFloat32 || given->device() != vertex.device())
            AT_ERROR("preallocated gradient output '", name, "' must be a contiguous float32 tensor of the gradient's shape on the rasterizer's device");
        if (P == 0) given->zero_();
        return *given;
    };
    const bool sh_factored = sh_factored_ && s.use_shs;
    torch::Tensor dL_dvertex = placed(out_vertex, {P, 3, 3}, "vertex"), dL_dcenter2D = placed(out_center2D, {P, 2}, "center2D");
    torch::Tensor dL_dopacity = placed(out_opacity, {P, 1}, "opacity");
    torch::Tensor dL_dshs = sh_factored ? torch::Tensor() : (s.use_shs ? placed(out_color, {P, s.M, 3}, "color") : torch::zeros({P, s.M, 3}, opts));
    torch::Tensor dL_dfeature = s.use_shs ? alloc({P, s.C}) : placed(out_color, {P, s.C}, "color");
    if (P != 0) // extension_interface.cu:242
    {
        ts2d_camera cam{s.W, s.H, tan_fovx, tan_fovy, fptr(viewmatrix), fptr(projmatrix), fptr(campos)};
        const float *bg_dev = nullptr;
        if (background_depth_dev.has_value() && background_depth_dev->defined())
        {
            if (!background_depth_dev->is_cuda() || background_depth_dev->scalar_type() != torch::kFloat32 || background_depth_dev->numel() != 1)
                AT_ERROR("background_depth must be a float or a one-element float32 tensor on the HIP device");
            bg_dev = background_depth_dev->data_ptr<float>();
        }
        ts2d_geometry geom{P, sh_degree, s.M, s.C, gamma, scale_modifier, bg_dev ? 0.0f : background_depth, fptr(background), fptr(vertex),
                           s.use_shs ? fptr(shs) : nullptr, s.use_shs ? nullptr : fptr(feature), fptr(opacity), bg_dev};
        const uint32_t flags = (rich_info ? TS2D_FLAG_RICH_INFO : 0u) | (debug ? TS2D_FLAG_DEBUG : 0u) | (s.use_shs ? TS2D_FLAG_USE_SHS : 0u) |
                               (v3 ? TS2D_FLAG_3D : 0u) | (sh_factored ? TS2D_FLAG_SH_FACTORED : 0u);
        ts2d_state st{geometryBuffer.data_ptr(), (size_t)geometryBuffer.numel(), binningBuffer.numel() ? binningBuffer.data_ptr() : nullptr,
                      (size_t)binningBuffer.numel(), imageBuffer.data_ptr(), (size_t)imageBuffer.numel()};
        ts2d_loss_grads loss{fptr(dL_dout_feature), rich_info ? fptr(dL_dout_depth) : nullptr, rich_info ? fptr(dL_dout_normal) : nullptr};
        torch::Tensor scratch = torch::empty({(int64_t)ts2d_backward_scratch_bytes(P)}, opts.dtype(torch::kByte));
        ts2d_backward_out bo{fptr_mut(dL_dvertex), fptr_mut(dL_dcenter2D), dL_dshs.defined() ? fptr_mut(dL_dshs) : nullptr, fptr_mut(dL_dfeature),
                             fptr_mut(dL_dopacity)};
        if (!range_events.empty())
        {
            std::vector<void *> ev(range_events.size());
            for (size_t i = 0; i < ev.size(); i++) ev[i] = (void *)(intptr_t)range_events[i];
            check(ts2d_backward_ranged(&cam, &geom, flags, num_rendered, radii.data_ptr<int>(), &st, &loss, scratch.data_ptr(), (size_t)scratch.numel(), &bo,
                                       (int32_t)ev.size(), ev.data(), stream),
                  "rasterize_triangles_backward");
        }
        else
            check(ts2d_backward(&cam, &geom, flags, num_rendered, radii.data_ptr<int>(), &st, &loss, scratch.data_ptr(), (size_t)scratch.numel(), &bo,
                                stream),
                  "rasterize_triangles_backward"); // rasterizer.cu:269-358
    }
    return std::make_tuple(dL_dvertex, dL_dcenter2D, dL_dshs.defined() ? std::optional<torch::Tensor>(dL_dshs) : std::nullopt, dL_dfeature, dL_dopacity);
}

// R2D/ext.cpp:8 -- the reference's signature, nothing added
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
rasterizeTrianglesBackward(const float tan_fovx, const float tan_fovy, const torch::Tensor &viewmatrix, const torch::Tensor &projmatrix,
                           const torch::Tensor &campos, const int sh_degree, const float gamma, const float scale_modifier,
                           const float background_depth, const torch::Tensor &background, const torch::Tensor &vertex, const torch::Tensor &shs,
                           const torch::Tensor &feature, const torch::Tensor &opacity, const int num_rendered, const torch::Tensor &radii,
                           const torch::Tensor &geometryBuffer, const torch::Tensor &binningBuffer, const torch::Tensor &imageBuffer,
                           const torch::Tensor &dL_dout_feature, const torch::Tensor &dL_dout_depth, const torch::Tensor &dL_dout_normal,
                           const bool rich_info, const bool debug)
{
    auto r = rasterizeTrianglesBackwardEx(tan_fovx, tan_fovy, viewmatrix, projmatrix, campos, sh_degree, gamma, scale_modifier, background_depth, background, vertex,
                                          shs, feature, opacity, num_rendered, radii, geometryBuffer, binningBuffer, imageBuffer, dL_dout_feature, dL_dout_depth,
                                          dL_dout_normal, rich_info, debug, 2, false, std::nullopt, std::nullopt, std::nullopt, std::nullopt, std::nullopt, {});
    return std::make_tuple(std::get<0>(r), std::get<1>(r), *std::get<2>(r), std::get<3>(r), std::get<4>(r));
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) // R2D/ext.cpp:4-9
{
    m.def("rasterize_triangles", &rasterizeTrianglesForward);
    m.def("rasterize_triangles_backward", &rasterizeTrianglesBackward);
    // the package's own entry points (not part of the reference's surface)
    m.def("rasterize_triangles_ex", &rasterizeTrianglesForwardEx);
    m.def("rasterize_triangles_backward_ex", &rasterizeTrianglesBackwardEx);
}

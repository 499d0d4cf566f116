"""End-to-end training loop on a synthetic multi-view target, using only this repository's drop-in pieces the way the reference's
trainer + model use their own (src/diff_recon/trainers/VanillaTS_trainer.py:60-130, src/diff_recon/models/VanillaTS_model.py:560-694):

    render_view (argument construction of VanillaTSModel.forward)  ->  TriangleRenderer  ->  2D or 3D HIP rasterizer
    photometric_loss (fused L1 + SSIM)  ->  backward through the rasterizer  ->  Adam (diff_recon_hip.FusedAdam: one fused launch)
    model_update(iteration) in the reference's order (:560-575): training statistic, densification, opacity pruning / clipping,
    scale pruning / clipping, contribution pruning, opacity reset, gamma schedule, SH-degree schedule
    -- every structural update through the native row operators of include/ts_model.h (diff_recon_hip/model_update.py).

`views_per_step` views are rendered per optimisation step and their gradients summed, which is what one step of image-parallel
training computes across ranks (BASELINE.json configs[3]: 8 views per step over 8 GPUs; here the views run one after the other
on one GPU -- the cross-rank exchange itself is covered by tests/test_multigpu_gpu.py and bench.py --gpus N).

A "ground truth" is rendered from a hidden set of triangles from several cameras; a perturbed, sparser copy is optimised.
    python examples/train_synthetic.py [--rasterizer 2D|3D] [--iters 400] [--triangles 20000] [--views 4]
"""
import argparse
import math
import os
import sys
import time
from types import SimpleNamespace as NS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd")]

import numpy as np
import torch

import synthetic
import diff_recon_hip as D
from diff_recon_hip import DensificationStats, DepthNormalLoss, photometric_loss, render_view
from diff_triangle_rasterization_2D.parallel import ShGradSink, factored_sh_grads


class Camera:
    """The attributes of the reference's Camera that the renderer reads (src/diff_recon/utils/camera.py:70-117)."""

    def __init__(self, s, device, shift=(0.0, 0.0, 0.0)):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        view = s["viewmatrix"].copy()
        sh = np.asarray(shift, np.float32)
        view[3, :3] -= sh * np.array([-1, 1, -1], np.float32)  # translate the camera centre by `shift`
        proj = (view @ synthetic.projection_matrix(s["tanfovx"], s["tanfovy"]).T).astype(np.float32)
        self.image_width, self.image_height = s["image_width"], s["image_height"]
        self.tan_fovx, self.tan_fovy = s["tanfovx"], s["tanfovy"]
        self.world_view_transform, self.full_proj_transform = t(view), t(proj)
        self.camera_center = t(np.array([0, 0, synthetic.CAM_DIST], np.float32) + sh)
        self.device = device


from diff_recon_hip.schedulers import exponential_scheduler  # noqa: E402  (mirror of src/diff_recon/utils/scheduler.py)


class SyntheticModel(DensificationStats):
    """The state VanillaTSModel carries through training, under the reference's attribute names: four per-triangle parameters in
    named Adam groups (:119-135), the six statistics arrays (:196-201, inherited), gamma / active_sh_degree and the schedulers of
    _setup_model_update_utils (:150-194)."""

    def __init__(self, vertex, f_dc, f_rest, raw_opacity, iters, max_sh_degree, single_sh=False):
        super().__init__(vertex.shape[0], vertex.device)
        self._vertex, self._opacity = torch.nn.Parameter(vertex), torch.nn.Parameter(raw_opacity)
        groups = [{"params": [self._vertex], "lr": 0.03, "name": "vertex"}, {"params": [self._opacity], "lr": 0.05, "name": "opacity"}]
        if single_sh:
            # ONE (P, M, 3) colour tensor with the two learning rates of the reference's f_dc / f_rest groups inside it (FusedAdam: lr for the
            # first 3 floats of every triangle's 3 M, lr_tail for the rest): no torch.cat per forward, no split of its gradient per backward
            self._shs = torch.nn.Parameter(torch.cat([f_dc, f_rest], 1).contiguous())
            M = self._shs.shape[1]
            groups.append({"params": [self._shs], "lr": 0.01, "lr_tail": 0.0005, "tail_period": 3 * M, "tail_split": 3, "name": "shs"})
        else:
            self._f_dc, self._f_rest = torch.nn.Parameter(f_dc), torch.nn.Parameter(f_rest)
            groups += [{"params": [self._f_dc], "lr": 0.01, "name": "f_dc"}, {"params": [self._f_rest], "lr": 0.0005, "name": "f_rest"}]
        self.single_sh = single_sh
        # the reference's torch.optim.Adam(l, lr=0.0, eps=1e-15) (VanillaTS_model.py:108-124) as one fused launch per step (include/ts_optim.h)
        self.optimizer = D.FusedAdam(groups, lr=0.0, eps=1e-15)
        self.max_sh_degree, self.active_sh_degree, self.gamma = max_sh_degree, 0, 1.0
        self.scene_bbox, self.ste_threshold = None, None
        q = max(iters // 4, 1)
        every = lambda a, b, k, **kw: NS(start_iter=a, end_iter=b, hold_iter=b, interval_iter=k, **kw)
        self.config = NS(model_update=NS(
            statistic=NS(start_iter=0, end_iter=iters),  # the window in which _training_statistic accumulates (VanillaTS_model.py:348-350)
            densification=every(q // 2, 3 * q, max(q // 3, 1), min_view_count=8, split_num=2, split_scale_threshold=60.0),
            opacity_pruning=every(q, iters, max(q // 2, 1)), opacity_clipping=every(q, iters, max(q // 2, 1)),
            scale_pruning=every(q, iters, q, radii_threshold=200.0, scale_threshold=200.0), scale_clipping=every(q, iters, max(q // 2, 1)),
            contribution_pruning=every(2 * q, iters, q, min_view_count=8, target_point_num=int(0.8 * vertex.shape[0]), prune_ratio=0.3,
                                       max_prune_ratio=0.3, contrib_max_ratio=0.5, sparsity_retain_ratio=0.2, downsample_iteration=[],
                                       downsample_point_num=[]),
            opacity_reset=every(2 * q, 2 * q + 1, 2 * q + 1, reset_value=0.7),
            gamma_schedule=NS(start_iter=q, end_iter=iters), sh_schedule=NS(one_up_iters=[q, 2 * q, 3 * q])))
        self.grad_threshold_scheduler = exponential_scheduler(1.2e-5, 8e-6, 3 * q)
        self.opacity_pruning_scheduler = exponential_scheduler(0.02, 0.05, iters - q)
        self.opacity_clipping_scheduler = exponential_scheduler(0.999, 0.99, iters - q)
        self.scale_max_scheduler = exponential_scheduler(120.0, 100.0, iters - q)  # world units: the scene's mean side is ~55
        self.gamma_scheduler = exponential_scheduler(1.0, 4.0, iters - q)  # the reference's configs go 1 -> 50 over 30 k iterations
        self.log = []

    def model_update(self, iteration, render_pkgs):
        """VanillaTSModel.model_update (:567-581), same order (diff_recon_hip.model_update.run_model_update)."""
        for name, res in D.run_model_update(self, iteration, render_pkgs):
            self.log.append((iteration, name, res, self._vertex.shape[0]))


def train(rasterizer="2D", iters=200, triangles=20000, width=256, height=192, seed=0, views=2, views_per_step=2, log=print, updates=True,
          w_geometry=0.0, single_sh=False, init_from_pcd=False, factored_sh=False):
    """w_geometry > 0 adds the depth / normal consistency term of the *_VanillaTS_mesh.yaml configurations (geometry_loss: w_geometry 0.05,
    scale_factor 0.5, from iteration start_iter on; VanillaTS_trainer.py:30-31,64-65,84,111) -- the producer of dL_dout_depth / dL_dout_normal.
    factored_sh: the backward passes hand the optimizer (dL_dRGB, camera centre) per view instead of writing the dense dL_dshs, and FusedAdam
    steps the colour parameters from those (include/ts_optim.h: tso_adam_step_sh_factored) -- the same numbers, 12 M bytes per triangle less
    written and as many less read."""
    dev = torch.device("cuda")
    geometry_loss = DepthNormalLoss(scale_factor=0.5) if w_geometry > 0 else None
    g_start_iter = iters // 2  # the reference's configs start it at half of the schedule (15 000 of 30 000)
    D_sh = 2
    s = synthetic.scene(triangles, width, height, D_sh, seed=seed, edge_px=10.0)
    cams = [Camera(s, dev, (6.0 * v, -3.0 * v, 0.0)) for v in range(views)]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    bg = torch.zeros(3)
    kw = dict(bg_color=bg, max_sh_degree=D_sh, rasterizer_type=rasterizer)
    with torch.no_grad():  # hidden targets
        gts = [render_view(c, t(s["vertex"]), t(s["shs"][:, :1]), t(s["shs"][:, 1:]), torch.logit(t(s["opacity"]).clamp(0.05, 0.95)), is_training=False,
                           gamma=1.0, active_sh_degree=D_sh, **kw)["render"].clamp(0, 1) for c in cams]
    g = torch.Generator(device="cuda").manual_seed(seed + 1)
    keep = torch.rand(triangles, device=dev, generator=g) < 0.7  # start sparser than the target: densification has work to do
    vertex = (t(s["vertex"]) + 1.5 * torch.randn(s["vertex"].shape, device=dev, generator=g))[keep].contiguous()
    n0 = vertex.shape[0]
    if init_from_pcd:
        # the way the reference's trainer starts (VanillaTSModel.create_from_pcd, VanillaTS_model.py:830-917): a point cloud -- here the perturbed
        # centroids with grey colours and no normals, like a COLMAP cloud without them -- becomes equilateral triangles sized by the distance to the
        # three nearest neighbours (simple_knn.distCUDA2)
        init = D.create_from_pcd(vertex.mean(dim=1), torch.full((n0, 3), 0.5, device=dev), None, max_sh_degree=D_sh, init_opacity=0.5)
        m = SyntheticModel(init["_vertex"], init["_f_dc"], init["_f_rest"], init["_opacity"], iters, D_sh, single_sh=single_sh)
    else:
        m = SyntheticModel(vertex, torch.full((n0, 1, 3), 0.5, device=dev), torch.zeros((n0, (D_sh + 1) ** 2 - 1, 3), device=dev),
                           torch.zeros((n0, 1), device=dev), iters, D_sh, single_sh=single_sh)
    losses, t0 = [], time.perf_counter()
    for it in range(1, iters + 1):
        m.optimizer.zero_grad(set_to_none=True)
        pkgs, total = [], torch.zeros((), device=dev)  # the loss stays on the device: no host synchronisation inside an iteration
        sink = ShGradSink()
        for k in range(views_per_step):  # the views of one step: gradients are summed, like ranks' gradients in image-parallel training
            v = (it * views_per_step + k) % views
            colour = dict(shs=m._shs) if m.single_sh else {}
            pkg = render_view(cams[v], m._vertex, None if m.single_sh else m._f_dc, None if m.single_sh else m._f_rest, m._opacity, is_training=True,
                              gamma=m.gamma, active_sh_degree=m.active_sh_degree, **colour, **kw)
            loss = photometric_loss(pkg["render"], gts[v], 0.8, 0.2)  # w_L1 = 1 - w_ssim, VanillaTS_trainer.py:72,111
            if geometry_loss is not None and it > g_start_iter:
                loss = loss + w_geometry * geometry_loss(pkg["depth"], pkg["normal"], cams[v].tan_fovx, cams[v].tan_fovy)
            with factored_sh_grads(sink, enabled=factored_sh):
                loss.backward()
            pkgs.append(pkg)
            total += loss.detach()
        if factored_sh:
            colour = dict(shs=m._shs) if m.single_sh else dict(f_dc=m._f_dc, f_rest=m._f_rest)
            m.optimizer.step(sh_factors=D.ShFactors(sink, m._vertex, m.active_sh_degree, **colour))
        else:
            m.optimizer.step()
        if updates:
            m.model_update(it, pkgs)
        else:
            for pkg in pkgs:
                m.update(pkg)
        losses.append(total / views_per_step)
        if log and (it % 50 == 0 or it == 1 or it == iters):
            log(f"iter {it:4d}  loss {float(losses[-1]):.5f}  triangles {m._vertex.shape[0]}  gamma {m.gamma:.2f}  sh {m.active_sh_degree}")
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / iters
    return [float(x) for x in torch.stack(losses).cpu()], m, sec


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rasterizer", default="2D", choices=["2D", "3D"])
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--triangles", type=int, default=20000)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--single-sh-tensor", action="store_true", help="one (P, M, 3) colour parameter with two learning rates instead of f_dc + f_rest")
    ap.add_argument("--init-from-pcd", action="store_true", help="start from diff_recon_hip.create_from_pcd (point cloud -> distCUDA2 -> equilateral triangles) like the reference's trainer")
    ap.add_argument("--factored-sh", action="store_true", help="Adam on the SH coefficients from the factored gradient (dL_dRGB per view) instead of the dense dL_dshs")
    ap.add_argument("--w-geometry", type=float, default=0.0, help="weight of the depth / normal consistency loss (0.05 in the *_VanillaTS_mesh configs)")
    a = ap.parse_args()
    losses, m, sec = train(a.rasterizer, a.iters, a.triangles, views=a.views, w_geometry=a.w_geometry, single_sh=a.single_sh_tensor, init_from_pcd=a.init_from_pcd,
                           factored_sh=a.factored_sh)
    for row in m.log:
        print("  update", row)
    print(f"{a.rasterizer}: loss {losses[0]:.5f} -> {losses[-1]:.5f} in {a.iters} iterations, {sec * 1e3:.2f} ms/iteration (incl. Python)")

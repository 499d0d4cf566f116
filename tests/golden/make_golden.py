"""Generates the golden fixtures under tests/golden/ from the REFERENCE's own Python code.

Runs only in the build container (it imports /root/reference by file path; the reference never travels):

    python tests/golden/make_golden.py

Fixtures are data only (inputs + expected outputs):
  sh_eval.npz : unit directions, SH coefficients, degree -> `eval_sh` of src/diff_recon/utils/sh_utils.py:41-100
                (the reference's independent Python statement of the SH colour polynomial that the CUDA
                 computeRGBFromSH, R2D/src/forward.cu:9-59, evaluates).
  camera.npz  : (R, T, FoVx, FoVy, W, H) -> world_view_transform, full_proj_transform, camera_center, tan_fov of
                src/diff_recon/utils/camera.py:70-117 (the matrix convention the rasterizer consumes).
  gamma_rescale.npz : gamma -> the triangle rescale ratio of src/diff_recon/models/VanillaTS_model.py:615-617
                (restated formula 1/sqrt(2^beta * beta * Gamma(beta)), beta = 1/gamma, evaluated with scipy like the
                 reference; the model class itself is not importable here).
  model_update.npz : one model state (300 triangles: the four per-triangle parameters, their Adam moments after a real optimizer
                step, the six densification statistics) and the state after each of the reference's own update methods
                _prune_points, _densification (+ _grow_points), _opacity_pruning, _opacity_clipping, _scale_pruning,
                _scale_clipping, _opacity_reset, _contribution_pruning (src/diff_recon/models/VanillaTS_model.py:214-537),
                executed by importing the reference's VanillaTSModel class (see load_reference_model_class).
  photometric.npz : image pairs, (w_L1, w_ssim) -> L1, ssimLoss, img_loss and d img_loss / d image (torch autograd) of
                src/diff_recon/trainers/trainer_utils.py:9-103,323-324 combined as VanillaTS_trainer.py:80-81,111.
                trainer_utils.py imports two third-party packages at module level that this image lacks and that the
                loss code never touches (torchmetrics' LPIPS class, simple_knn); empty placeholder modules are
                registered for those two names so the file imports -- every line that produces the vectors
                (GaussianSmoothing2D, SSIM, SSIMLoss, L1) is the reference's own.
"""
import importlib.util
import sys
import types
import math
import os

import numpy as np
import torch

REF = "/root/reference/src/diff_recon/utils"
HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_trainer_utils():
    for name, attrs in (("torchmetrics", {}), ("torchmetrics.image", {}),
                        ("torchmetrics.image.lpip", {"LearnedPerceptualImagePatchSimilarity": lambda **kw: None}),
                        ("simple_knn", {"nearestNeighbor": None})):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
    spec = importlib.util.spec_from_file_location(
        "ref_trainer_utils", "/root/reference/src/diff_recon/trainers/trainer_utils.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def photometric(rng):
    tu = load_trainer_utils()
    out = {}
    cases = [(3, 37, 53, 0.8, 0.2), (3, 64, 48, 0.5, 0.5), (1, 20, 70, 1.0, 0.0), (3, 12, 9, 0.0, 1.0)]
    for i, (C, H, W, w1, ws) in enumerate(cases):
        gt = rng.uniform(0, 1, size=(C, H, W)).astype(np.float32)
        # smooth-ish render: gt blurred by noise mix, so SSIM is neither ~0 nor ~1
        img = np.clip(0.6 * gt + 0.4 * rng.uniform(0, 1, size=(C, H, W)), 0, 1).astype(np.float32)
        res = {}
        for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            x = torch.tensor(img, dtype=dt, requires_grad=True)
            g = torch.tensor(gt, dtype=dt)
            ssim_mod = tu.SSIMLoss()
            if dt == torch.float64:  # same code, evaluated in double: only the dtype of the (float32-built) window changes
                ssim_mod.ssim.window.kernel = ssim_mod.ssim.window.kernel.double()
            l1 = tu.L1(x, g)
            sl = ssim_mod(x, g)
            loss = w1 * l1 + ws * sl
            loss.backward()
            res[tag] = (float(l1), float(sl), float(loss), x.grad.numpy().astype(np.float64))
        out.update({f"img{i}": img, f"gt{i}": gt, f"w{i}": np.array([w1, ws]),
                    f"l1_{i}": res["f64"][0], f"ssim_loss_{i}": res["f64"][1], f"loss_{i}": res["f64"][2],
                    f"grad_{i}": res["f64"][3], f"loss_f32_{i}": res["f32"][2], f"grad_f32_{i}": res["f32"][3].astype(np.float32)})
    out["n"] = len(cases)
    np.savez_compressed(os.path.join(HERE, "photometric.npz"), **out)


def main():
    sh_utils = load("sh_utils")
    camera = load("camera")
    rng = np.random.default_rng(20250927)
    photometric(np.random.default_rng(77))

    # ---- SH colour polynomial
    n = 256
    dirs = rng.standard_normal((n, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    sh = rng.uniform(-1.0, 1.0, size=(n, 3, 16))  # eval_sh wants [..., C, coeffs]
    out = {}
    for deg in range(4):
        res = sh_utils.eval_sh(deg, torch.from_numpy(sh), torch.from_numpy(dirs))
        out[f"deg{deg}"] = res.numpy()
    np.savez(os.path.join(HERE, "sh_eval.npz"), dirs=dirs, sh=sh, **out)

    # ---- camera convention
    cams = []
    for i in range(6):
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        R = camera.qvec2rotmat(q)
        T = rng.uniform(-5, 5, size=3)
        fovx = float(rng.uniform(0.4, 1.4))
        W, H = int(rng.integers(64, 2000)), int(rng.integers(64, 1200))
        # FoVy derived from the aspect ratio the way Camera does when only an image is given (camera.py:106-107)
        fovy = math.atan(math.tan(fovx / 2) * (H / W)) * 2 if i % 2 else float(rng.uniform(0.3, 1.2))
        cam = camera.Camera(R=R, T=T, FoVx=fovx, FoVy=fovy, image_width=W, image_height=H)
        cams.append(dict(R=R, T=T, FoVx=fovx, FoVy=cam.FoVy, W=W, H=H,
                         world_view_transform=cam.world_view_transform.contiguous().numpy(),
                         full_proj_transform=cam.full_proj_transform.contiguous().numpy(),
                         camera_center=cam.camera_center.numpy(), tan_fovx=cam.tan_fovx, tan_fovy=cam.tan_fovy))
    # the canonical bench camera of R2D/main.cu:12-22 (our synthetic.camera() must reproduce it)
    R = np.diag([-1.0, 1.0, -1.0]); T = np.array([0.0, 0.0, 1200.0]); W, H = 1920, 1080
    fovx = 2 * math.atan(0.3148); fovy = 2 * math.atan(0.3148 * H / W)
    cam = camera.Camera(R=R, T=T, FoVx=fovx, FoVy=fovy, image_width=W, image_height=H)
    cams.append(dict(R=R, T=T, FoVx=fovx, FoVy=fovy, W=W, H=H,
                     world_view_transform=cam.world_view_transform.contiguous().numpy(),
                     full_proj_transform=cam.full_proj_transform.contiguous().numpy(),
                     camera_center=cam.camera_center.numpy(), tan_fovx=cam.tan_fovx, tan_fovy=cam.tan_fovy))
    np.savez(os.path.join(HERE, "camera.npz"), **{f"{k}_{i}": np.asarray(c[k]) for i, c in enumerate(cams) for k in c})

    # ---- gamma rescale ratio (VanillaTS_model.py:615-617)
    import scipy.special
    gammas = np.array([1.0, 1.5, 2.0, 5.0, 10.0, 25.0, 50.0])
    beta = 1.0 / gammas
    ratio = 1.0 / np.sqrt(2.0 ** beta * beta * scipy.special.gamma(beta))
    np.savez(os.path.join(HERE, "gamma_rescale.npz"), gamma=gammas, ratio=ratio)
    print("wrote", sorted(os.listdir(HERE)))





# ---- model_update.npz: the reference's own VanillaTSModel update methods, executed on CPU tensors ------------------------------------
def load_reference_model_class():
    """Imports src/diff_recon/models/VanillaTS_model.py without running the package __init__ (which pulls in trainers, datasets and
    tensorboard): empty package shells with the real paths, a no-op Logger, and empty placeholders for the mesh / point-cloud IO
    packages this image lacks (plyfile, trimesh, ...: imported at module level by point_cloud.py / raw_triangle.py; nothing used here
    touches them).  The drop-in rasterizer / simple_knn packages stand in for the CUDA
    extensions the model module imports -- none of their functions is called by the methods exercised below."""
    import importlib
    root = "/root/reference/src/diff_recon"
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "triangle-splatting_amd"))

    def pkg(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        m.__package__ = name
        sys.modules[name] = m

    pkg("diff_recon", root)
    for sub in ("models", "utils", "renderer"):
        pkg("diff_recon." + sub, os.path.join(root, sub))

    class Logger:
        def info(self, *a, **k):
            pass
        warning = info

    lg = types.ModuleType("diff_recon.utils.logger")
    lg.Logger, lg.stdout_logger = Logger, Logger()
    sys.modules["diff_recon.utils.logger"] = lg
    ply = types.ModuleType("plyfile")
    ply.PlyData = ply.PlyElement = None
    sys.modules["plyfile"] = ply
    for third_party in ("trimesh", "pygltflib", "open3d"):  # mesh / GLB IO of raw_triangle.py and point_cloud.py: imported, never called here
        if third_party not in sys.modules:
            try:
                importlib.import_module(third_party)
            except ModuleNotFoundError:
                sys.modules[third_party] = types.ModuleType(third_party)
    while True:
        try:
            return importlib.import_module("diff_recon.models.VanillaTS_model").VanillaTSModel, Logger
        except ModuleNotFoundError as e:  # any further IO-only third-party module this image lacks
            if e.name is None or e.name.startswith("diff_recon"):
                raise
            sys.modules[e.name] = types.ModuleType(e.name)


def create_from_pcd(rng):
    """create_from_pcd.npz: the reference's own VanillaTSModel.create_from_pcd on CPU tensors (VanillaTS_model.py:830-917).  Its one CUDA leaf,
    distCUDA2 (exact mean squared distance to the three nearest neighbours, submodules/simple-knn), is replaced by the float64 k-d tree of
    oracle/ts_knn_oracle.py -- the same exact search -- because this container has no GPU; everything else is the reference's code.  Random numbers
    come from torch's global CPU generator, seeded per case (stored), so that the HIP-side test can replay the stream."""
    from types import SimpleNamespace as NS
    Model, Logger = load_reference_model_class()
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import ts_knn_oracle
    import diff_recon.models.VanillaTS_model as VM

    def ipd(pc):
        d2 = ts_knn_oracle.mean_dist3(pc.detach().cpu().numpy().astype(np.float64))
        return torch.from_numpy(np.asarray(d2, np.float32)).clamp_(min=1e-10).sqrt()
    VM.inter_point_distance = ipd
    PC = VM.PointCloud
    out, cases = {}, {
        "plain": dict(n=400, zero_normals=False, bbox=None, sampling=NS(sample_method="direct", n_sample_inside=None, n_sample_outside=None, grid_size_inside=None,
                                                                       grid_size_outside=None, init_opacity=0.5, duplicate_count=1), back_culling=False, max_sh=3, seed=11),
        "twins_dup": dict(n=300, zero_normals=True, bbox=(2.0, 2.0, 2.0, 8.0, 8.0, 8.0), sampling=NS(sample_method="direct", n_sample_inside=None, n_sample_outside=None,
                                                                                                   grid_size_inside=None, grid_size_outside=None, init_opacity=0.1, duplicate_count=3),
                          back_culling=True, max_sh=0, seed=12),
        "grid": dict(n=2000, zero_normals=False, bbox=(0.0, 0.0, 10.0, 6.0), sampling=NS(sample_method="grid", n_sample_inside=300, n_sample_outside=None, grid_size_inside=None,
                                                                                       grid_size_outside=1.5, init_opacity=0.3, duplicate_count=1), back_culling=False, max_sh=2, seed=13),
    }
    for name, c in cases.items():
        pts = rng.random((c["n"], 3)) * 10
        cols = rng.random((c["n"], 3))
        nrm = np.zeros_like(pts) if c["zero_normals"] else rng.normal(size=(c["n"], 3))
        nrm[: c["n"] // 10, :2] = 0.0  # some normals along +-z: the `up x normal` fallback (:897)
        m = Model.__new__(Model)
        torch.nn.Module.__init__(m) if isinstance(m, torch.nn.Module) else None
        m.device, m.logger, m.scene_bbox, m.back_culling, m.max_sh_degree = torch.device("cpu"), Logger(), c["bbox"], c["back_culling"], c["max_sh"]
        m.config = NS(sampling=c["sampling"])
        m._training_setup = lambda: None
        torch.manual_seed(c["seed"])
        m.create_from_pcd(PC(points=pts, colors=cols, normals=nrm))
        out.update({f"{name}/points": pts, f"{name}/colors": cols, f"{name}/normals": nrm, f"{name}/seed": np.int64(c["seed"]),
                    f"{name}/bbox": np.asarray(c["bbox"] if c["bbox"] is not None else [], np.float64), f"{name}/back_culling": np.bool_(c["back_culling"]),
                    f"{name}/max_sh": np.int64(c["max_sh"]), f"{name}/init_opacity": np.float64(c["sampling"].init_opacity),
                    f"{name}/duplicate_count": np.int64(c["sampling"].duplicate_count), f"{name}/method": np.int64({"direct": 0, "random": 1, "grid": 2}[c["sampling"].sample_method]),
                    f"{name}/n_sample_inside": np.int64(c["sampling"].n_sample_inside or -1), f"{name}/grid_size_outside": np.float64(c["sampling"].grid_size_outside or -1.0),
                    f"{name}/vertex": m._vertex.detach().numpy(), f"{name}/opacity": m._opacity.detach().numpy(), f"{name}/f_dc": m._f_dc.detach().numpy(),
                    f"{name}/f_rest": m._f_rest.detach().numpy()})
    np.savez_compressed(os.path.join(HERE, "create_from_pcd.npz"), **out)
    print("create_from_pcd.npz:", {k: out[f"{k}/vertex"].shape for k in cases})


def model_update(rng):
    from types import SimpleNamespace as NS
    from copy import deepcopy
    Model, Logger = load_reference_model_class()
    P, M = 300, 4

    def fresh():
        g = torch.Generator().manual_seed(1234)
        m = Model.__new__(Model)
        torch.nn.Module.__init__(m) if isinstance(m, torch.nn.Module) else None
        m.device = torch.device("cpu")
        m.logger = Logger()
        m.scene_bbox = None
        m.ste_threshold = None
        centre = torch.rand((P, 1, 3), generator=g) * 10
        size = torch.rand((P, 1, 1), generator=g) ** 3 * 1.5 + 0.02
        m._vertex = torch.nn.Parameter(centre + torch.randn((P, 3, 3), generator=g) * size)
        m._opacity = torch.nn.Parameter(torch.randn((P, 1), generator=g) * 3)
        m._f_dc = torch.nn.Parameter(torch.rand((P, 1, 3), generator=g))
        m._f_rest = torch.nn.Parameter(torch.rand((P, M - 1, 3), generator=g))
        groups = [{"params": [getattr(m, "_" + n)], "lr": 0.01, "name": n} for n in ("vertex", "opacity", "f_dc", "f_rest")]
        m.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        for n in ("vertex", "opacity", "f_dc", "f_rest"):  # one real Adam step populates exp_avg / exp_avg_sq
            getattr(m, "_" + n).grad = torch.randn(getattr(m, "_" + n).shape, generator=g)
        m.optimizer.step()
        m.optimizer.zero_grad(set_to_none=True)
        m.gradient_accum = torch.rand((P,), generator=g) * 5
        m.gradient_denom = torch.randint(0, 12, (P,), generator=g).float()
        m.max_radii2D = torch.rand((P,), generator=g) * 60
        m.contrib_sum = torch.rand((P,), generator=g) * 9
        m.contrib_max = torch.rand((P,), generator=g)
        m.contrib_denom = torch.randint(0, 9, (P,), generator=g).float()
        it = NS(start_iter=0, end_iter=1000, hold_iter=1000, interval_iter=100)
        m.config = NS(model_update=NS(
            densification=NS(**vars(it), min_view_count=4, split_num=2, split_scale_threshold=0.6),
            opacity_pruning=NS(**vars(it)), opacity_clipping=NS(**vars(it)),
            scale_pruning=NS(**vars(it), radii_threshold=50.0, scale_threshold=1.2),
            scale_clipping=NS(**vars(it)), opacity_reset=NS(**vars(it), reset_value=0.3),
            contribution_pruning=NS(**vars(it), min_view_count=3, target_point_num=150, prune_ratio=0.5, max_prune_ratio=0.6, contrib_max_ratio=0.4,
                                    sparsity_retain_ratio=0.0, downsample_iteration=[], downsample_point_num=[])))
        m.grad_threshold_scheduler = lambda step: 0.21
        m.opacity_pruning_scheduler = lambda step: 0.25
        m.opacity_clipping_scheduler = lambda step: 0.9
        m.scale_max_scheduler = lambda step: 0.8
        return m

    def snapshot(m, prefix, out):
        for n in ("vertex", "opacity", "f_dc", "f_rest"):
            p = getattr(m, "_" + n)
            out[f"{prefix}/{n}"] = p.detach().numpy().copy()
            st = m.optimizer.state[p]
            out[f"{prefix}/{n}.exp_avg"] = st["exp_avg"].numpy().copy()
            out[f"{prefix}/{n}.exp_avg_sq"] = st["exp_avg_sq"].numpy().copy()
        for n in ("gradient_accum", "gradient_denom", "max_radii2D", "contrib_sum", "contrib_max", "contrib_denom"):
            out[f"{prefix}/{n}"] = getattr(m, n).numpy().copy()

    out = {}
    snapshot(fresh(), "input", out)
    prune_mask = torch.rand((P,), generator=torch.Generator().manual_seed(5)) < 0.3
    out["prune_mask"] = prune_mask.numpy()
    cases = {
        "prune_points": lambda m: m._prune_points(prune_mask),
        "densification": lambda m: m._densification(100),
        "opacity_pruning": lambda m: m._opacity_pruning(100),
        "opacity_clipping": lambda m: m._opacity_clipping(100),
        "scale_pruning": lambda m: m._scale_pruning(100),
        "scale_clipping": lambda m: m._scale_clipping(100),
        "opacity_reset": lambda m: m._opacity_reset(100),
        "contribution_pruning": lambda m: m._contribution_pruning(100),
    }
    for name, fn in cases.items():
        m = fresh()
        with torch.no_grad():
            fn(m)
        snapshot(m, name, out)
    # _training_statistic (:347-363), three consecutive iterations of render outputs
    m = fresh()
    m.config.model_update.statistic = NS(start_iter=0, end_iter=1000)
    g = torch.Generator().manual_seed(99)
    for it in range(3):
        radii = torch.randint(-2, 40, (P,), generator=g).clamp(min=0).int()
        c2d = torch.zeros((P, 2), requires_grad=True)
        c2d.grad = torch.randn((P, 2), generator=g)
        pkg = {"center2D": c2d, "visible_mask": radii > 0, "radii": radii, "contrib_sum": torch.rand((P,), generator=g) * 3,
               "contrib_max": torch.rand((P,), generator=g)}
        for k in ("radii", "contrib_sum", "contrib_max"):
            out[f"statistic_in{it}/{k}"] = pkg[k].numpy().copy()
        out[f"statistic_in{it}/center2D_grad"] = c2d.grad.numpy().copy()
        with torch.no_grad():
            m._training_statistic(it + 1, pkg)
    snapshot(m, "training_statistic", out)
    # the statistic WINDOW (:348-350): model_update over the same three iterations with statistic.start_iter = 1, end_iter = 2 -- only
    # iteration 2 may move the accumulators; every other rule of the sequence is switched off so that the snapshot isolates the window
    m = fresh()
    for name in ("densification", "opacity_pruning", "opacity_clipping", "scale_pruning", "scale_clipping", "contribution_pruning", "opacity_reset",
                 "gamma_schedule", "sh_schedule"):
        setattr(m.config.model_update, name, None)
    m.config.model_update.statistic = NS(start_iter=1, end_iter=2)
    for it in range(3):
        c2d = torch.zeros((P, 2), requires_grad=True)
        c2d.grad = torch.from_numpy(out[f"statistic_in{it}/center2D_grad"])
        radii = torch.from_numpy(out[f"statistic_in{it}/radii"])
        pkg = {"center2D": c2d, "visible_mask": radii > 0, "radii": radii, "contrib_sum": torch.from_numpy(out[f"statistic_in{it}/contrib_sum"]),
               "contrib_max": torch.from_numpy(out[f"statistic_in{it}/contrib_max"])}
        m.model_update(it + 1, pkg)
    snapshot(m, "statistic_window", out)
    np.savez_compressed(os.path.join(HERE, "model_update.npz"), **out)
    print("model_update.npz:", {k: out[f"{k}/vertex"].shape[0] for k in cases})



# ---- schedules.npz: the reference's value schedules (utils/scheduler.py) and _set_gamma / _set_sh_degree ----------------------------
def schedules():
    sched = load("scheduler")
    steps = np.array([-5, 0, 1, 2, 7, 50, 99, 100, 101, 250, 499, 500, 501, 2999, 3000, 10_000, 29_999, 30_000, 40_000])
    out = {"steps": steps}
    exp_cases = [(1.6e-4, 1.6e-6, 30_000, 0, 1.0), (1.0, 50.0, 25_000, 0, 1.0), (0.02, 0.05, 500, 0, 1.0), (1e-2, 1e-4, 3000, 100, 0.01),
                 (5.0, 0.5, 100, 250, 0.5)]
    out["exp_cases"] = np.array(exp_cases, dtype=np.float64)
    out["exp_values"] = np.array([[float(sched.exponential_scheduler(a, b, int(m), int(d), dm)(int(t))) for t in steps] for a, b, m, d, dm in exp_cases])
    out["step_values_a"] = np.array([sched.step_scheduler([1.0, 2.0, 4.0, 8.0], [100, 500, 3000])(int(t)) for t in steps])
    out["step_values_b"] = np.array([sched.step_scheduler([3.0, 2.0, 1.0], [1, 100, 501])(int(t)) for t in steps])
    es_cases = [(1.0, 50.0, 25_000, 10, 0, 1.0), (1e-2, 1e-4, 3000, 4, 100, 0.01)]
    out["es_cases"] = np.array(es_cases, dtype=np.float64)
    out["es_values"] = np.array([[float(sched.exponential_step_scheduler(a, b, int(m), int(n), int(d), dm)(int(t))) for t in steps]
                                 for a, b, m, n, d, dm in es_cases])
    # _set_gamma / _set_sh_degree on the reference's model object (methods only; no tensors involved)
    Model, Logger = load_reference_model_class()
    NS = types.SimpleNamespace
    m = Model.__new__(Model)
    m.config = NS(model_update=NS(gamma_schedule=NS(start_iter=500, end_iter=25_500), sh_schedule=NS(one_up_iters=[1000, 2000, 3000, 9000])))
    m.gamma_scheduler = sched.exponential_scheduler(1.0, 50.0, 25_000)
    m.max_sh_degree = 3
    m.gamma, m.active_sh_degree = 1.0, 0
    iters = np.array([1, 499, 500, 501, 1000, 1001, 2001, 3001, 9001, 13_000, 25_500, 25_501, 30_000])
    g, d = [], []
    for it in iters:
        m._set_gamma(int(it))
        m._set_sh_degree(int(it))
        g.append(float(m.gamma))
        d.append(int(m.active_sh_degree))
    out.update(model_iters=iters, model_gamma=np.array(g), model_sh_degree=np.array(d))
    np.savez(os.path.join(HERE, "schedules.npz"), **out)
    print("schedules.npz:", out["exp_values"].shape, out["es_values"].shape)

# ---- depth_normal.npz: the reference's DepthNormalLoss (trainer_utils.py:204-257) + torch autograd ------------------------------------
def depth_normal():
    tu = load_trainer_utils()
    rng = np.random.default_rng(11)
    out = {}
    cases = [(47, 64, 0.5, 0.9, 0.31, 0.23), (64, 48, 0.5, 0.9, 0.5, 0.66), (33, 41, None, 0.9, 0.4, 0.3), (40, 56, 0.25, 0.7, 0.31, 0.2),
             # non-dyadic factors on sizes divisible by 10: floor(H * s) and 1 / s must be formed in double like F.interpolate does (H = 40,
             # s = 0.7: 28 rows; a float32 0.7 gives 27)
             (40, 50, 0.7, 0.9, 0.35, 0.28), (30, 60, 0.3, 0.8, 0.3, 0.15)]
    out["cases"] = np.array([[H, W, -1.0 if s is None else s, q, tx, ty] for H, W, s, q, tx, ty in cases], np.float64)
    for i, (H, W, s, q, tx, ty) in enumerate(cases):
        # a smooth depth map with a few discontinuities (what a rendered scene looks like), strictly positive
        yy, xx = np.meshgrid(np.linspace(0, 1, H), np.linspace(0, 1, W), indexing="ij")
        depth = 5.0 + 2.0 * np.sin(3 * xx + 1) * np.cos(2 * yy) + 1.5 * (xx + yy > 1.1) + 0.05 * rng.standard_normal((H, W))
        normal = rng.standard_normal((3, H, W)) * 0.3 + np.array([0.1, -0.2, -1.0])[:, None, None]
        normal[:, 0, 0] = 0.0  # a pixel without any coverage: exercises the eps branch of F.normalize
        d = torch.tensor(depth, dtype=torch.float32, requires_grad=True)
        n = torch.tensor(normal, dtype=torch.float32, requires_grad=True)
        loss = tu.DepthNormalLoss(scale_factor=s, depth_grad_filter_quantile=q)(d, n, tx, ty)
        loss.backward()
        out[f"depth{i}"], out[f"normal{i}"] = d.detach().numpy(), n.detach().numpy()
        out[f"loss{i}"] = np.float32(loss.item())
        out[f"ddepth{i}"], out[f"dnormal{i}"] = d.grad.numpy(), n.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "depth_normal.npz"), **out)
    print("depth_normal.npz:", [float(out[f"loss{i}"]) for i in range(len(cases))])


def aux_losses():
    """DoGLoss / SmoothnessLoss of trainer_utils.py:105-201: the reference's classes + torch autograd on seeded image pairs.  Stored: the inputs, the
    masks the classes form from the target (their private _dog_mask / _low_grad_mask), the losses and the gradients with respect to the image."""
    tu = load_trainer_utils()
    rng = np.random.default_rng(23)
    out = {}
    cases = [(3, 48, 64, 0.5, 90, 0.3), (3, 37, 53, 0.5, 80, 0.5), (1, 40, 40, 0.25, 90, 0.3), (3, 30, 50, None, 20, 0.7), (2, 50, 40, 0.7, 90, 0.3)]
    out["cases"] = np.array([[C, H, W, -1.0 if s is None else s, f, q] for C, H, W, s, f, q in cases], np.float64)
    for i, (C, H, W, s, freq, q) in enumerate(cases):
        yy, xx = np.meshgrid(np.linspace(0, 1, H), np.linspace(0, 1, W), indexing="ij")
        base = 0.5 + 0.3 * np.sin(7 * xx + 2 * yy) * np.cos(5 * yy) + 0.2 * (xx > 0.55) - 0.15 * (yy > 0.4)
        gt = np.clip(base[None] + 0.05 * rng.standard_normal((C, H, W)) + 0.1 * np.arange(C)[:, None, None], 0, 1)
        img = np.clip(gt + 0.08 * rng.standard_normal((C, H, W)), 0, 1)
        img[:, :3, :3] = gt[:, :3, :3]  # pixels where image == target: torch's sign(0) = 0
        # (no exactly flat patch: where the Scharr response cancels to an exact 0 torch's norm backward gives 0, but the reference's float32
        # convolution leaves ~1e-9 of rounding noise there instead, and its gradient is then a unit vector in the direction of that noise --
        # not a property anything can be pinned to; the kernels' exact-0 behaviour is tested on a constant image in tests/test_loss_gpu.py)
        g = torch.tensor(gt, dtype=torch.float32)
        dog = tu.DoGLoss(freq=freq, scale_factor=s) if s is not None else tu.DoGLoss(freq=freq, scale_factor=1.0)
        smo = tu.SmoothnessLoss(quantile=q, scale_factor=s) if s is not None else tu.SmoothnessLoss(quantile=q, scale_factor=1.0)
        x = torch.tensor(img, dtype=torch.float32, requires_grad=True)
        l1 = dog(x, g)
        l1.backward()
        out[f"img{i}"], out[f"gt{i}"] = x.detach().numpy(), g.numpy()
        out[f"dog_mask{i}"] = dog._dog_mask(g[None])[0, 0].numpy()
        out[f"dog_loss{i}"], out[f"dog_grad{i}"] = np.float32(l1.item()), x.grad.numpy().copy()
        x.grad = None
        l2 = smo(x, g)
        l2.backward()
        out[f"smooth_mask{i}"] = smo._low_grad_mask(g[None])[0, 0].numpy()
        out[f"smooth_loss{i}"], out[f"smooth_grad{i}"] = np.float32(l2.item()), x.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "aux_losses.npz"), **out)
    print("aux_losses.npz:", [(float(out[f"dog_loss{i}"]), float(out[f"smooth_loss{i}"])) for i in range(len(cases))])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "depth_normal":
        depth_normal()
    elif len(sys.argv) > 1 and sys.argv[1] == "aux_losses":
        aux_losses()
    elif len(sys.argv) > 1 and sys.argv[1] == "create_from_pcd":
        create_from_pcd(np.random.default_rng(7))  # only this fixture (round 6)
    elif len(sys.argv) > 1 and sys.argv[1] == "model_update":
        model_update(np.random.default_rng(1))  # only this fixture (the others are unchanged since round 1)
    elif len(sys.argv) > 1 and sys.argv[1] == "schedules":
        schedules()
    else:
        main()
        model_update(np.random.default_rng(1))
        schedules()

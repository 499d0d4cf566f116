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


if __name__ == "__main__":
    main()

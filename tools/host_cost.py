"""Host time (enqueue only, GPU idle before each call) of the pieces of one eager training iteration at the mesh93k shape -- where the 0.8 ms of
Python / dispatch per iteration go (the iteration is host-bound there: 0.75-0.85 ms eager against 0.55 under graph replay)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd"), os.path.join(ROOT, "tools")]
import numpy as np, torch
import synthetic, diff_recon_hip as D
from bench_train_step import Camera
from diff_triangle_rasterization_2D.parallel import ShGradSink, factored_sh_grads
dev = torch.device("cuda")
P, w, h, up = 93000, 800, 800, 2
s = synthetic.scene(P, w * up, h * up, 0, seed=42)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
cam = Camera(s, w, h, dev)
vertex = torch.nn.Parameter(t(s["vertex"])); shs = torch.nn.Parameter(t(s["shs"])); raw = torch.nn.Parameter(torch.logit(t(s["opacity"]).clamp(0.02, 0.98)))
opt = D.FusedAdam([{"params": [vertex], "lr": 0.0}, {"params": [raw], "lr": 0.0}, {"params": [shs], "lr": 0.0, "lr_tail": 0.0, "tail_period": 3, "tail_split": 3}], lr=0.0, eps=1e-15)
stats = D.DensificationStats(P, dev)
gt = torch.rand((3, h, w), device=dev); bg = torch.zeros(3, device=dev)
geo = D.DepthNormalLoss(scale_factor=0.5)
radii = torch.randint(0, 50, (P,), device=dev, dtype=torch.int32)
def cost(name, fn, n=100):
    for _ in range(5): fn()
    tot = 0.0
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); tot += time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{name:46s} {tot / n * 1e6:8.1f} us")
cost("radii // 2", lambda: radii // 2)
cost("torch.div(radii, 2, rounding_mode='floor')", lambda: torch.div(radii, 2, rounding_mode="floor"))
cost("radii >> 1", lambda: radii >> 1)
cost("bg.to(dev) (no-op)", lambda: bg.to(dev))
cost("background_depth", lambda: D.background_depth(vertex, cam.camera_center))
cost("torch.sigmoid(raw)", lambda: torch.sigmoid(raw))
state = {}
def fwd():
    state["pkg"] = D.render_view(cam, vertex, None, None, raw, shs=shs, bg_color=bg, gamma=1.0, active_sh_degree=0, max_sh_degree=0, is_training=True, render_up_scale=up, rasterizer_type="3D")
cost("render_view (forward, all of it)", fwd)
def loss():
    pkg = state["pkg"]
    state["loss"] = D.photometric_loss(pkg["render"], gt, 0.8, 0.2) + 0.05 * geo(pkg["depth"], pkg["normal"], cam.tan_fovx, cam.tan_fovy)
cost("photometric + depth/normal loss (forward)", lambda: (fwd(), loss()) and None)
def bwd():
    fwd(); loss()
    vertex.grad = shs.grad = raw.grad = None
    with factored_sh_grads() as sink:
        state["loss"].backward()
    state["sink"] = sink
cost("forward + losses + backward", bwd)
def full():
    bwd(); stats.update(state["pkg"]); opt.step(sh_factors=D.ShFactors(state["sink"], vertex, 0, shs=shs))
cost("whole iteration (+ statistics + optimizer)", full)
import gc
def wall(name, fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    print(f"{name:46s} {(time.perf_counter() - t0) / n * 1e6:8.1f} us per iteration (steady state, no synchronisation in between)")
wall("whole iteration, back to back", full)
gc.collect(); gc.disable()
wall("... with the garbage collector off", full)
import diff_triangle_rasterization_2D as pkg2d
pkg2d.set_instance_capacity(600000)
wall("... and the sync-free forward", full)
print("overflowed", pkg2d.forward_overflowed()[0])

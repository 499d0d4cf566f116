"""The fused Adam step (csrc/optim.hip through include/ts_optim.h; diff_recon_hip/optim.py) against torch.optim.Adam on the same tensors,
with the reference's configuration: four named groups with their own learning rates, lr = 0 default, eps = 1e-15
(src/diff_recon/models/VanillaTS_model.py:108-124), learning rates rewritten between steps (:583).

Policy: the kernel performs torch's operations in torch's order, each rounded on its own; torch's kernels may fuse a product into the
following sum.  So the moments agree to a few ulp per step and the parameters to 2e-6 of their magnitude after ten steps -- stated here,
checked below; bit-equality is not claimed.  (exp_avg is a signed sum: its bound is relative to the tensor's magnitude.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _groups(P, M, seed, dev):
    import torch
    g = torch.Generator(device=dev).manual_seed(seed)
    mk = lambda *s: torch.rand(s, device=dev, generator=g)
    return {"vertex": mk(P, 3, 3) * 10, "opacity": mk(P, 1), "f_dc": mk(P, 1, 3), "f_rest": mk(P, M - 1, 3) * 0.1}


LRS = {"vertex": 3e-2, "opacity": 5e-2, "f_dc": 1e-2, "f_rest": 5e-4}


@pytest.mark.parametrize("P", [1, 37, 4096, 100_003])  # odd sizes: slices that are not multiples of four floats / 16-byte aligned
def test_fused_adam_matches_torch_adam(P):
    import torch
    from diff_recon_hip import FusedAdam
    dev, M = "cuda", 16
    init = _groups(P, M, 5, dev)
    a = {k: v.clone().requires_grad_() for k, v in init.items()}
    b = {k: v.clone().requires_grad_() for k, v in init.items()}
    mk = lambda cls, t: cls([{"params": [t[k]], "lr": LRS[k], "name": k} for k in t], lr=0.0, eps=1e-15)
    fused, ref = mk(FusedAdam, a), mk(torch.optim.Adam, b)
    g = torch.Generator(device=dev).manual_seed(9)
    for it in range(10):
        if it == 4:
            for opt in (fused, ref):
                for group in opt.param_groups:  # update_learning_rate
                    group["lr"] = group["lr"] * 0.5
        for k in a:
            grad = (torch.rand(a[k].shape, device=dev, generator=g) - 0.5) * (1e-3 if k == "vertex" else 1.0)
            if it == 7 and k == "opacity":
                grad.zero_()  # a zero gradient: denom = sqrt(v) / bc + 1e-15 must not blow up
            a[k].grad, b[k].grad = grad.clone(), grad.clone()
        fused.step()
        ref.step()
        fused.zero_grad(set_to_none=True)
        ref.zero_grad(set_to_none=True)
    for k in a:
        sa, sb = fused.state[a[k]], ref.state[b[k]]
        assert int(sa["step"]) == int(sb["step"]) == 10
        for name in ("exp_avg", "exp_avg_sq"):
            x, y = sa[name], sb[name]
            # exp_avg_sq is a sum of non-negative terms: elementwise relative; exp_avg is a signed sum (elements cancel to ~0): a few ulp of
            # the tensor's magnitude
            atol = 1e-30 if name == "exp_avg_sq" else 5e-7 * float(y.abs().max())
            assert torch.allclose(x, y, rtol=2e-6, atol=atol), (k, name, float((x - y).abs().max()))
        scale = float(b[k].detach().abs().max())
        assert float((a[k].detach() - b[k].detach()).abs().max()) <= 2e-6 * scale + 1e-9, (k, float((a[k].detach() - b[k].detach()).abs().max()), scale)


def test_two_learning_rates_inside_one_sh_tensor():
    """group keys lr_tail / tail_period / tail_split: one (P, M, 3) tensor updated like the reference's f_dc and f_rest groups."""
    import torch
    from diff_recon_hip import FusedAdam
    dev, P, M = "cuda", 5003, 9
    init = _groups(P, M, 11, dev)
    shs = torch.cat([init["f_dc"], init["f_rest"]], 1).contiguous().requires_grad_()
    f_dc, f_rest = init["f_dc"].clone().requires_grad_(), init["f_rest"].clone().requires_grad_()
    fused = FusedAdam([{"params": [shs], "lr": 1e-2, "lr_tail": 5e-4, "tail_period": 3 * M, "tail_split": 3, "name": "shs"}], lr=0.0, eps=1e-15)
    ref = torch.optim.Adam([{"params": [f_dc], "lr": 1e-2}, {"params": [f_rest], "lr": 5e-4}], lr=0.0, eps=1e-15)
    g = torch.Generator(device=dev).manual_seed(2)
    for _ in range(5):
        grad = torch.rand(shs.shape, device=dev, generator=g) - 0.5
        shs.grad, f_dc.grad, f_rest.grad = grad, grad[:, :1].contiguous(), grad[:, 1:].contiguous()
        fused.step()
        ref.step()
    want = torch.cat([f_dc, f_rest], 1)
    assert float((shs.detach() - want.detach()).abs().max()) <= 2e-6 * float(want.detach().abs().max())


def test_state_surgery_like_the_model_update():
    """The model update replaces a parameter and its moments between steps (prune / densify, VanillaTS_model.py:214-345): the optimizer
    picks the new tensors up, `step` carried over -- like torch.optim.Adam."""
    import torch
    from diff_recon_hip import FusedAdam
    dev = "cuda"
    p = torch.rand(100, 3, device=dev).requires_grad_()
    opt = FusedAdam([{"params": [p], "lr": 1e-2, "name": "vertex"}], lr=0.0, eps=1e-15)
    ref_p = p.detach().clone().requires_grad_()
    ref = torch.optim.Adam([{"params": [ref_p], "lr": 1e-2}], lr=0.0, eps=1e-15)
    for o, q in ((opt, p), (ref, ref_p)):
        q.grad = torch.ones_like(q)
        o.step()
    keep = torch.arange(100, device=dev) % 3 != 0

    def prune(o, q):
        st = o.state.pop(q)
        new = torch.nn.Parameter(q.detach()[keep].clone())
        st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"][keep].clone(), st["exp_avg_sq"][keep].clone()
        o.param_groups[0]["params"][0] = new
        o.state[new] = st
        return new
    p2, r2 = prune(opt, p), prune(ref, ref_p)
    for o, q in ((opt, p2), (ref, r2)):
        q.grad = torch.full_like(q, 0.5)
        o.step()
    assert torch.allclose(p2, r2, rtol=2e-6, atol=1e-8) and int(opt.state[p2]["step"]) == 2


def test_cpu_tensors_and_bad_arguments_are_refused():
    import torch
    from diff_recon_hip import FusedAdam
    p = torch.rand(8).requires_grad_()
    p.grad = torch.ones(8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        FusedAdam([p], lr=1e-3).step()
    with pytest.raises(ValueError, match="Invalid epsilon"):
        FusedAdam([p], eps=-1.0)


def _sharded_worker(port, q):
    import os
    import sys
    import torch
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    sys.path[:0] = [root, os.path.join(root, "triangle-splatting_amd"), here]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from diff_recon_hip import FusedAdam, ShardedAdam
        P, M = 20_001, 4
        init = _groups(P, M, 21, dev)
        shs0 = torch.cat([init["f_dc"], init["f_rest"]], 1).contiguous()
        opt = ShardedAdam({"vertex": init["vertex"], "opacity": init["opacity"], "shs": shs0}, {"vertex": 3e-2, "opacity": 5e-2, "shs": 1e-2},
                          eps=1e-15, tails={"shs": (5e-4, 3 * M, 3)}, force_collectives=True)
        ref_t = {"vertex": init["vertex"].clone().requires_grad_(), "opacity": init["opacity"].clone().requires_grad_(),
                 "shs": shs0.clone().requires_grad_()}
        ref = FusedAdam([{"params": [ref_t["vertex"]], "lr": 3e-2}, {"params": [ref_t["opacity"]], "lr": 5e-2},
                         {"params": [ref_t["shs"]], "lr": 1e-2, "lr_tail": 5e-4, "tail_period": 3 * M, "tail_split": 3}], lr=0.0, eps=1e-15)
        g = torch.Generator(device=dev).manual_seed(4)
        for _ in range(3):  # without host synchronisation in between: the side stream orders itself against the compute stream
            for (k, t), view in zip(ref_t.items(), opt.bucket.views()):
                grad = torch.rand(t.shape, device=dev, generator=g) - 0.5
                view.copy_(grad)
                t.grad = grad
            opt.step()
            params = opt.wait()
            ref.step()
        torch.cuda.synchronize()
        q.put({k: bool(torch.equal(params[k].detach(), ref_t[k].detach())) for k in ref_t})  # the same kernel on the same numbers
    finally:
        dist.destroy_process_group()


def test_sharded_adam_on_the_hip_kernel_equals_fused_adam():
    """ShardedAdam in a group of ONE rank over RCCL with the collectives forced (reduce-scatter into the own slice, the fused kernel on
    the bucket's side stream, all-gather of the parameters): the same parameters as FusedAdam on the same gradients.  (Sharded ==
    replicated at world 2 and 3: tests/test_parallel_cpu.py over gloo.)"""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_sharded_worker, args=(port, q))
    p.start()
    p.join(600)
    assert p.exitcode == 0
    res = q.get(timeout=10)
    assert all(res.values()), res


def test_sharded_adam_end_to_end_through_the_rasterizer_backward():
    """ShardedAdam driven by a REAL rasterizer backward under `opt.bucket.capture()` -- not by gradients copied into the bucket --
    against FusedAdam fed by ordinary autograd on the same scene: after three steps (rasterize -> loss -> backward -> step) both hold the
    same vertex / opacity / SH parameters.  The colour tensor is registered as "shs" (the name of INTEGRATION.md's example and of the
    reference's rasterizer argument); it lands in the bucket's `color` slot.  (world = 1: the sharding protocol itself is
    tests/test_parallel_cpu.py's, the RCCL branch test_sharded_adam_on_the_hip_kernel_equals_fused_adam's.)"""
    import torch
    import helpers
    import synthetic
    from diff_recon_hip import FusedAdam, ShardedAdam
    from diff_triangle_rasterization_2D import TriangleRasterizer
    dev = "cuda"
    s = synthetic.scene(4000, 160, 128, 2, seed=77)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rs = helpers.hip_settings(s, True)
    lrs = {"vertex": 1e-3, "opacity": 1e-2, "shs": 5e-3}
    M = s["shs"].shape[1]
    opt = ShardedAdam({"vertex": t(s["vertex"]), "opacity": t(s["opacity"]), "shs": t(s["shs"])}, lrs, eps=1e-15, tails={"shs": (2.5e-4, 3 * M, 3)})
    assert set(opt.bucket.named_views()) == {"vertex", "opacity", "color"}
    ref = {k: t(s[k]).requires_grad_() for k in ("vertex", "opacity", "shs")}
    ropt = FusedAdam([{"params": [ref["vertex"]], "lr": lrs["vertex"]}, {"params": [ref["opacity"]], "lr": lrs["opacity"]},
                      {"params": [ref["shs"]], "lr": lrs["shs"], "lr_tail": 2.5e-4, "tail_period": 3 * M, "tail_split": 3}], lr=0.0, eps=1e-15)
    gi, gd, gn = t(s["dL_dout_feature"]), t(s["dL_dout_depth"]), t(s["dL_dout_normal"])

    def loss_of(p):
        c2d = torch.zeros((p["vertex"].shape[0], 2), device=dev, requires_grad=True)
        out = TriangleRasterizer(rs)(p["vertex"], c2d, p["opacity"], shs=p["shs"])
        return (out[0] * gi).sum() + (out[2] * gd).sum() + (out[3] * gn).sum()

    for _ in range(3):
        with opt.bucket.capture():
            loss_of(opt.params).backward()
        assert all(p.grad is None for p in opt.params.values())  # the gradients live in the bucket, not in .grad
        opt.step()
        params = opt.wait()
        ropt.zero_grad(set_to_none=True)
        loss_of(ref).backward()
        ropt.step()
    torch.cuda.synchronize()
    for k in ref:
        # same kernels on both sides; the only freedom is the order of the backward's atomic adds
        scale = float(ref[k].detach().abs().max())
        assert float((params[k].detach() - ref[k].detach()).abs().max()) <= 1e-4 * scale, k
        assert not torch.equal(params[k].detach(), t(s[k]))  # ... and the parameters did move


def test_sharded_adam_refuses_what_a_capture_cannot_fill():
    """Unknown tensor names raise (a capture writes vertex / opacity / center2D / colour only); under capture the rasterizer's inputs must be
    the optimizer's own leaves -- an activation between parameter and rasterizer raises instead of training on the wrong gradient."""
    import torch
    import helpers
    import synthetic
    from diff_recon_hip import ShardedAdam
    from diff_triangle_rasterization_2D import TriangleRasterizer
    dev = "cuda"
    s = synthetic.scene(500, 64, 64, 0, seed=3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    with pytest.raises(ValueError, match="not gradient slots"):
        ShardedAdam({"vertex": t(s["vertex"]), "sh_coefficients": t(s["shs"])}, {"vertex": 1e-3, "sh_coefficients": 1e-3})
    with pytest.raises(ValueError, match="same gradient slot"):
        ShardedAdam({"shs": t(s["shs"]), "color": t(s["shs"])}, {"shs": 1e-3, "color": 1e-3})
    opt = ShardedAdam({"vertex": t(s["vertex"]), "opacity": t(s["opacity"]), "shs": t(s["shs"])}, {"vertex": 1e-3, "opacity": 1e-3, "shs": 1e-3})
    c2d = torch.zeros((500, 2), device=dev, requires_grad=True)
    out = TriangleRasterizer(helpers.hip_settings(s, True))(opt.params["vertex"], c2d, torch.sigmoid(opt.params["opacity"]), shs=opt.params["shs"])
    with opt.bucket.capture():
        with pytest.raises(RuntimeError, match="is not the parameter the bucket's optimizer owns"):
            out[0].sum().backward()


@pytest.mark.parametrize("layout,max_deg,deg,views", [("one", 3, 3, 1), ("one", 3, 1, 2), ("one", 3, 2, 1), ("one", 1, 0, 2), ("one", 1, 1, 1), ("one", 2, 1, 2),
                                                       ("split", 3, 2, 1), ("split", 1, 1, 3), ("one", 0, 0, 1), ("split", 0, 0, 2)])
def test_adam_from_factored_sh_gradients_equals_adam_on_the_dense_gradient(layout, max_deg, deg, views):
    """include/ts_optim.h, tso_adam_step_sh_factored: the colour parameters stepped from (dL_dRGB, camera centre) per view.  A real backward under
    factored_sh_grads() provides the factors (and leaves the colour tensors without a .grad); the dense dL_dshs of the SAME factors comes from the
    expansion kernel (bit-identical to what the backward writes, tests/test_factored_gpu.py).  Two optimizers, two steps each (the second on
    non-zero moments): parameters and both moments must be EQUAL -- the same expressions in the same order -- for the one-tensor layout with its
    two learning rates and for the reference's f_dc / f_rest pair, with coefficients above the active degree and with several views."""
    import torch
    import helpers
    import synthetic
    from diff_recon_hip import FusedAdam, ShFactors
    from diff_triangle_rasterization_2D import TriangleRasterizer, _C
    from diff_triangle_rasterization_2D.parallel import ShGradSink, factored_sh_grads
    dev = "cuda"
    P, M = 5003, (max_deg + 1) ** 2
    s = synthetic.scene(P, 128, 96, max_deg, seed=31 + max_deg)
    s["sh_degree"] = deg
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def make():
        p = {"vertex": t(s["vertex"]).requires_grad_(), "opacity": t(s["opacity"]).requires_grad_()}
        if layout == "one":
            p["shs"] = t(s["shs"]).requires_grad_()
            groups = [{"params": [p["shs"]], "lr": 1e-2, "lr_tail": 5e-4, "tail_period": 3 * M, "tail_split": 3}]
        else:
            p["f_dc"] = t(s["shs"][:, :1]).requires_grad_()
            p["f_rest"] = t(s["shs"][:, 1:]).requires_grad_()
            groups = [{"params": [p["f_dc"]], "lr": 1e-2}, {"params": [p["f_rest"]], "lr": 5e-4}]
        groups += [{"params": [p["vertex"]], "lr": 3e-2}, {"params": [p["opacity"]], "lr": 5e-2}]
        return p, FusedAdam(groups, lr=0.0, eps=1e-15)

    (a, opt_a), (b, opt_b) = make(), make()
    g = torch.Generator(device=dev).manual_seed(7)
    cams = [s["campos"] + np.float32(0.37 * v) * np.array([1.0, -0.5, 0.25], np.float32) for v in range(views)]
    with factored_sh_grads() as sink:
        for v in range(views):
            sv = dict(s, campos=cams[v])
            c2d = torch.zeros((P, 2), device=dev, requires_grad=True)
            shs = b["shs"] if layout == "one" else torch.cat((b["f_dc"], b["f_rest"]), dim=1)
            out = TriangleRasterizer(helpers.hip_settings(sv, True))(b["vertex"], c2d, b["opacity"], shs=shs)
            (out[0] * torch.rand(out[0].shape, device=dev, generator=g)).sum().backward()
    assert len(sink.colors) == views
    assert all(b[k].grad is None for k in b if k in ("shs", "f_dc", "f_rest"))  # the dense gradient was never formed
    assert b["vertex"].grad is not None
    colors = [c.clone() for c in sink.colors]
    campos = torch.stack([c.clone() for c in sink.campos])
    dense = _C.sh_grad_expand(b["vertex"].detach(), campos, torch.stack([c.reshape(P, 3) for c in colors]), deg, M)
    assert float(dense.abs().max()) > 0 and (deg == max_deg or float(dense[:, (deg + 1) ** 2:].abs().max()) == 0.0)
    gv, go = b["vertex"].grad.clone(), b["opacity"].grad.clone()
    for step in range(2):
        # dense side
        a["vertex"].grad, a["opacity"].grad = gv.clone(), go.clone()
        if layout == "one":
            a["shs"].grad = dense.clone()
        else:
            a["f_dc"].grad, a["f_rest"].grad = dense[:, :1].contiguous(), dense[:, 1:].contiguous()
        opt_a.step()
        # factored side: the same factors, the same vertex / opacity gradients
        b["vertex"].grad, b["opacity"].grad = gv.clone(), go.clone()
        sk = ShGradSink()
        for c, cp in zip(colors, campos):
            sk.append(c.clone(), cp)
        # NOTE the direction comes from the vertices of the backward: the dense array was formed from b's vertices BEFORE step 0, so step 1 reads
        # those too (`vertex=` takes any tensor with the backward's values)
        vert0 = t(s["vertex"])
        f = ShFactors(sk, vert0, deg, shs=b["shs"]) if layout == "one" else ShFactors(sk, vert0, deg, f_dc=b["f_dc"], f_rest=b["f_rest"])
        opt_b.step(sh_factors=f)
        assert not sk.colors  # consumed
    torch.cuda.synchronize()
    for k in a:
        assert torch.equal(a[k].detach(), b[k].detach()), k
        assert not torch.equal(a[k].detach(), t(s["shs"] if k == "shs" else s["shs"][:, :1] if k == "f_dc" else s["shs"][:, 1:] if k == "f_rest" else s[k])) or a[k].numel() == 0, k
        sa, sb = opt_a.state[a[k]], opt_b.state[b[k]]
        assert sa["step"] == sb["step"] == 2
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]), k
    # a dense .grad beside the factors is refused instead of dropped
    if layout == "one":
        b["shs"].grad = dense.clone()
        sk = ShGradSink()
        sk.append(colors[0].clone(), campos[0])
        with pytest.raises(RuntimeError, match="also holds a dense .grad"):
            opt_b.step(sh_factors=ShFactors(sk, t(s["vertex"]), deg, shs=b["shs"]))

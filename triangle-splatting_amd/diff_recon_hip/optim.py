"""The optimizer step of the training loop on the fused HIP kernel of libts2d.so (include/ts_optim.h).

    FusedAdam     drop-in for the reference's `torch.optim.Adam(l, lr=0.0, eps=1e-15)` (src/diff_recon/models/VanillaTS_model.py:108-124):
                  same constructor, `param_groups` (named groups whose "lr" the model rewrites every iteration, :583) and `state`
                  ({"step", "exp_avg", "exp_avg_sq"} per parameter -- what the model's pruning / densification surgery edits, :214-345),
                  so `optimizer.step()` / `zero_grad()` of VanillaTS_trainer.py:119-122 work unchanged.  ONE launch for all parameters of
                  all groups instead of torch's ~10 multi-tensor launches.
    ShardedAdam   SURVEY.md 8e for image-parallel training: the parameters are views of ONE flat fp32 buffer with the layout of the
                  gradient bucket (parallel.GradBucket) the rasterizer's backward writes; per step the bucket is reduce-scattered, every
                  rank runs the fused kernel on the 1 / world slice it owns (its Adam moments exist for that slice only) and the updated
                  parameters -- not the gradients -- are all-gathered.  Same wire volume as reduce-scatter + all-gather of the gradients,
                  1 / world of the optimizer's HBM traffic and state.

Arithmetic = torch.optim.Adam's (single-tensor form), operation by operation in fp32; the bias corrections are formed in Python doubles like
torch does.  No CPU / eager fallback: `step()` on CPU tensors raises (the gloo tests of the sharding PROTOCOL inject `step_fn`).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from diff_triangle_rasterization_2D import _C as _native

_lib = _native._lib
_fp = C.c_void_p
MAX_SLICES = 16  # TSO_MAX_SLICES


class _Slice(C.Structure):  # tso_adam_slice, include/ts_optim.h
    _fields_ = [("param", _fp), ("grad", _fp), ("exp_avg", _fp), ("exp_avg_sq", _fp), ("count", C.c_int64), ("step_size", C.c_float),
                ("bias2_sqrt", C.c_float), ("grad_scale", C.c_float), ("step_size_tail", C.c_float), ("index0", C.c_int64),
                ("period", C.c_int32), ("split", C.c_int32)]


_lib.tso_adam_step.restype = C.c_int
_lib.tso_adam_step.argtypes = [C.POINTER(_Slice), C.c_int32, C.c_double, C.c_double, C.c_double, _fp]


class _RowSlice(C.Structure):  # tso_row_slice, include/ts_optim.h
    _fields_ = [("param", _fp), ("grad", _fp), ("exp_avg", _fp), ("exp_avg_sq", _fp), ("floats_per_row", C.c_int32), ("step_size", C.c_float),
                ("bias2_sqrt", C.c_float), ("grad_scale", C.c_float)]


SH_ROW_SLICES = 2  # TSO_SH_ROW_SLICES


class _ShFactoredStep(C.Structure):  # tso_sh_factored_step, include/ts_optim.h
    _fields_ = [("P", C.c_int32), ("M", C.c_int32), ("sh_degree", C.c_int32), ("V", C.c_int32), ("vertex", _fp), ("campos", _fp), ("dL_dcolor", _fp),
                ("param_dc", _fp), ("exp_avg_dc", _fp), ("exp_avg_sq_dc", _fp), ("param_rest", _fp), ("exp_avg_rest", _fp), ("exp_avg_sq_rest", _fp),
                ("dc_stride", C.c_int64), ("rest_stride", C.c_int64), ("step_size_dc", C.c_float), ("bias2_sqrt_dc", C.c_float),
                ("step_size_rest", C.c_float), ("bias2_sqrt_rest", C.c_float), ("grad_scale", C.c_float), ("num_rows", C.c_int32),
                ("rows", _RowSlice * SH_ROW_SLICES)]


_lib.tso_adam_step_sh_factored.restype = C.c_int
_lib.tso_adam_step_sh_factored.argtypes = [C.POINTER(_ShFactoredStep), C.c_double, C.c_double, C.c_double, _fp]


class ShFactors:
    """The SH gradient of one training iteration in factored form, for `FusedAdam.step(sh_factors=...)`.

        with factored_sh_grads() as sink:        # diff_triangle_rasterization_2D.parallel: the backward passes leave dL_dshs unwritten and
            loss.backward()                      #   hand the sink (dL_dRGB (P, 3), camera centre) per view
        opt.step(sh_factors=ShFactors(sink, vertex, sh_degree, shs=shs))          # or f_dc=..., f_rest=... (the reference's two tensors)

    `vertex` is the tensor the rasterizer saw (the SH direction is centroid - camera): the step reads it before it updates it.  The colour
    parameters must reach the rasterizer WITHOUT an operation that changes their gradient (the one-tensor `shs` itself, or cat(f_dc, f_rest)):
    the factored gradient is the gradient with respect to the rasterizer's `shs` input."""

    def __init__(self, sink, vertex: torch.Tensor, sh_degree: int, shs: Optional[torch.Tensor] = None, f_dc: Optional[torch.Tensor] = None,
                 f_rest: Optional[torch.Tensor] = None, grad_scale: float = 1.0):
        if (shs is None) == (f_dc is None) or (f_dc is None) != (f_rest is None):
            raise ValueError("ShFactors: pass either shs (P, M, 3) or f_dc (P, 1, 3) + f_rest (P, M - 1, 3)")
        self.sink, self.vertex, self.sh_degree, self.shs, self.f_dc, self.f_rest, self.grad_scale = sink, vertex, int(sh_degree), shs, f_dc, f_rest, grad_scale


def _corrections(lr: float, step: int, beta1: float, beta2: float) -> Tuple[float, float]:
    """(step_size, bias2_sqrt) exactly as torch/optim/adam.py forms them (Python doubles)."""
    bias_correction1 = 1 - beta1 ** step
    bias_correction2 = 1 - beta2 ** step
    return lr / bias_correction1, math.sqrt(bias_correction2)


def adam_step_slices(slices: Sequence[dict], beta1: float, beta2: float, eps: float, device) -> None:
    """One fused launch per MAX_SLICES slices.  A slice: dict(param, grad, exp_avg, exp_avg_sq = contiguous float32 device tensors of equal
    numel, step_size, bias2_sqrt, and optionally grad_scale, step_size_tail, index0, period, split)."""
    rows = []
    for s in slices:
        p, g, m, v = s["param"], s["grad"], s["exp_avg"], s["exp_avg_sq"]
        for t in (p, g, m, v):
            if not t.is_cuda:
                raise RuntimeError("the fused Adam step needs tensors on a HIP device; there is no CPU fallback")
            if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != p.numel():
                raise RuntimeError("the fused Adam step takes contiguous float32 tensors of equal size")
        if p.numel() == 0:
            continue
        rows.append(_Slice(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), s["step_size"], s["bias2_sqrt"],
                           s.get("grad_scale", 1.0), s.get("step_size_tail", 0.0), s.get("index0", 0), s.get("period", 0), s.get("split", 0)))
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream().cuda_stream
        for i in range(0, len(rows), MAX_SLICES):
            chunk = rows[i:i + MAX_SLICES]
            arr = (_Slice * len(chunk))(*chunk)
            _native._check(_lib.tso_adam_step(arr, len(chunk), beta1, beta2, eps, stream), "adam_step")


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps) -- weight_decay = 0, amsgrad = False, maximize = False, what the reference uses -- with
    step() as one fused HIP launch.  `state[p]` = {"step": int, "exp_avg", "exp_avg_sq"}; parameters may be replaced between steps (the
    model update swaps in pruned / grown tensors together with their moments).

    A group may carry `lr_tail` + `tail_period` + `tail_split`: inside each of its tensors the elements whose flat index modulo
    `tail_period` is >= `tail_split` use `lr_tail` -- one (P, M, 3) SH tensor with the learning rates of the reference's f_dc / f_rest
    groups (tail_period = 3 M, tail_split = 3)."""

    def __init__(self, params, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8):
        if lr < 0.0:
            raise ValueError(f"Invalid learning rate: {lr}")
        if eps < 0.0:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameters: {betas}")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    def _group_of(self, p):
        for group in self.param_groups:
            if any(q is p for q in group["params"]):
                return group
        raise ValueError("FusedAdam.step(sh_factors=...): the colour tensor is not one of this optimizer's parameters")

    def _moments(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        st["step"] = int(st["step"]) + 1
        return st

    def _step_sh_factored(self, f: ShFactors):
        """include/ts_optim.h, tso_adam_step_sh_factored: the colour parameters stepped from (dL_dRGB, camera centre) per view -- the same numbers
        as step() on the dense dL_dshs, which then is neither written by the backward nor read here."""
        sink = f.sink
        if not sink.colors:
            raise RuntimeError("FusedAdam.step(sh_factors=...): no SH-mode backward pass ran under factored_sh_grads()")
        one = f.shs is not None
        tensors = (f.shs,) if one else (f.f_dc, f.f_rest)
        for t in tensors + (f.vertex,):
            if not t.is_cuda:
                raise RuntimeError("the fused Adam step needs tensors on a HIP device; there is no CPU fallback")
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError("the fused Adam step takes contiguous float32 tensors")
        if any(t.grad is not None for t in tensors):
            raise RuntimeError("FusedAdam.step(sh_factors=...): the colour tensor also holds a dense .grad -- a backward pass ran outside "
                               "factored_sh_grads(); stepping it from the factors would drop that gradient")
        P = f.vertex.shape[0]
        M = f.shs.shape[1] if one else 1 + f.f_rest.shape[1]
        if tensors[0].shape[0] != P or (not one and (f.f_dc.shape[1] != 1 or f.f_rest.shape[0] != P)):
            raise ValueError("ShFactors: vertex and colour tensors disagree on the number of triangles")
        V = len(sink.colors)
        dev = f.vertex.device
        colors = sink.colors[0].reshape(P, 3).contiguous() if V == 1 else torch.stack([c.reshape(P, 3) for c in sink.colors])
        campos = torch.stack(list(sink.campos)).to(dev, torch.float32).contiguous()
        groups = [self._group_of(t) for t in tensors]
        betas, eps = groups[0]["betas"], groups[0]["eps"]
        if any(g["betas"] != betas or g["eps"] != eps for g in groups):
            raise ValueError("ShFactors: f_dc and f_rest must share betas and eps")
        states = [self._moments(t) for t in tensors]
        if one:
            g, st = groups[0], states[0]
            lr_rest = g["lr_tail"] if g.get("tail_period") else g["lr"]
            if g.get("tail_period") and (int(g["tail_period"]) != 3 * M or int(g["tail_split"]) != 3):
                raise ValueError("ShFactors: the tail of a one-tensor colour group must be (tail_period, tail_split) = (3 M, 3)")
            (s_dc, b_dc), (s_rest, b_rest) = _corrections(g["lr"], st["step"], *betas), _corrections(lr_rest, st["step"], *betas)
            ptr = lambda t, off: t.data_ptr() + 4 * off
            row = _ShFactoredStep(P, M, f.sh_degree, V, f.vertex.data_ptr(), campos.data_ptr(), colors.data_ptr(),
                                  ptr(f.shs, 0), ptr(st["exp_avg"], 0), ptr(st["exp_avg_sq"], 0),
                                  ptr(f.shs, 3) if M > 1 else None, ptr(st["exp_avg"], 3) if M > 1 else None, ptr(st["exp_avg_sq"], 3) if M > 1 else None,
                                  3 * M, 3 * M, s_dc, b_dc, s_rest, b_rest, f.grad_scale)
        else:
            (s_dc, b_dc) = _corrections(groups[0]["lr"], states[0]["step"], *betas)
            (s_rest, b_rest) = _corrections(groups[1]["lr"], states[1]["step"], *betas)
            row = _ShFactoredStep(P, M, f.sh_degree, V, f.vertex.data_ptr(), campos.data_ptr(), colors.data_ptr(),
                                  f.f_dc.data_ptr(), states[0]["exp_avg"].data_ptr(), states[0]["exp_avg_sq"].data_ptr(),
                                  f.f_rest.data_ptr() if M > 1 else None, states[1]["exp_avg"].data_ptr() if M > 1 else None,
                                  states[1]["exp_avg_sq"].data_ptr() if M > 1 else None, 3, 3 * (M - 1), s_dc, b_dc, s_rest, b_rest, f.grad_scale)
        # the other per-triangle parameters of this optimizer (the vertices, the opacities) ride along: same launch, from their dense gradients
        # (include/ts_optim.h: tso_row_slice); step() then has nothing left for them
        fused = []
        for group in self.param_groups:
            if group["betas"] != betas or group["eps"] != eps or group.get("tail_period"):
                continue
            for q in group["params"]:
                if (len(fused) < SH_ROW_SLICES and all(q is not t for t in tensors) and q.grad is not None and not q.grad.is_sparse and q.dim() >= 1
                        and q.shape[0] == P and q.numel() > 0 and q.is_cuda and q.dtype == torch.float32 and q.is_contiguous() and q.grad.is_contiguous()):
                    st = self._moments(q)
                    ss, bs = _corrections(group["lr"], st["step"], *betas)
                    fused.append((q, _RowSlice(q.data_ptr(), q.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), q.numel() // P, ss, bs, 1.0)))
        row.num_rows = len(fused)
        for k, (_, rs) in enumerate(fused):
            row.rows[k] = rs
        with torch.cuda.device(dev):
            _native._check(_lib.tso_adam_step_sh_factored(C.byref(row), betas[0], betas[1], eps, torch.cuda.current_stream().cuda_stream),
                           "adam_step_sh_factored")
        sink.clear()
        return [q for q, _ in fused]

    @torch.no_grad()
    def step(self, closure=None, sh_factors: Optional[ShFactors] = None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        done = ()
        if sh_factors is not None:
            done = self._step_sh_factored(sh_factors)  # FIRST: it reads the vertices the backward ran on (and steps them itself, see there)
        by_hyper: Dict[Tuple[float, float, float, torch.device], List[dict]] = {}
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None or any(p is q for q in done):
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients")
                st = self._moments(p)
                step_size, bias2_sqrt = _corrections(group["lr"], st["step"], beta1, beta2)
                row = dict(param=p, grad=p.grad if p.grad.is_contiguous() else p.grad.contiguous(), exp_avg=st["exp_avg"], exp_avg_sq=st["exp_avg_sq"],
                           step_size=step_size, bias2_sqrt=bias2_sqrt)
                if group.get("tail_period"):
                    row.update(step_size_tail=_corrections(group["lr_tail"], st["step"], beta1, beta2)[0], period=int(group["tail_period"]),
                               split=int(group["tail_split"]))
                by_hyper.setdefault((beta1, beta2, group["eps"], p.device), []).append(row)
        for (beta1, beta2, eps, device), rows in by_hyper.items():
            adam_step_slices(rows, beta1, beta2, eps, device)
        return loss


def _torch_step_fn(slices, beta1, beta2, eps, device):
    """The same arithmetic in eager torch -- ONLY for the gloo / CPU tests of ShardedAdam's protocol (tests/test_parallel_cpu.py inject it)."""
    for s in slices:
        p, m, v = s["param"], s["exp_avg"], s["exp_avg_sq"]
        g = s["grad"] * s.get("grad_scale", 1.0)
        m.lerp_(g, 1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        denom = (v.sqrt() / s["bias2_sqrt"]).add_(eps)
        step = torch.full_like(p, s["step_size"])
        if s.get("period", 0):
            idx = (torch.arange(p.numel(), device=p.device) + s.get("index0", 0)) % s["period"]
            step = torch.where(idx >= s["split"], torch.full_like(p, s["step_size_tail"]), step)
        p.sub_(step * (m / denom))


class ShardedAdam:
    """Adam over ONE flat fp32 parameter buffer, sharded over the ranks of `group`.

        opt = ShardedAdam({"vertex": vertex0, "opacity": opacity0}, lrs={"vertex": 1e-3, "opacity": 5e-2}, group=g)
        vertex, opacity = opt.params["vertex"], opt.params["opacity"]      # leaf tensors, views of opt.flat_param
        with opt.bucket.capture():                                            # the backward kernels write straight into opt.bucket.flat
            ... forward / backward of this rank's view(s) ...
        opt.step()      # reduce-scatter(grads) -> fused Adam on this rank's slice -> all-gather(params), on the bucket's side stream
        opt.wait()      # the compute stream waits; every rank now holds identical, updated parameters

    Names are the rasterizer's gradient slots (GradBucket.SLOTS): "vertex", "opacity", "center2D", and ONE colour tensor under "shs",
    "feature" or "color" (all three mean the bucket's `color` slot, which the backward fills with dL_dshs / dL_dfeature).  Any other name
    raises: a capture would never write it, and step() reads nothing but the bucket.  For the same reason the tensors handed
    to the rasterizer under `bucket.capture()` must BE `opt.params[...]`: the bucket holds the gradient with respect to the rasterizer's
    input, and step() applies it to the flat parameters -- an activation in between (sigmoid(raw_opacity), cat(f_dc, f_rest)) would make
    that the wrong gradient, so the backward refuses it (diff_triangle_rasterization_2D.__init__, `expected_inputs`).

    Layout of `flat_param` == layout of `bucket.flat` (tensor after tensor, padded to a multiple of 4 * world floats).  Rank r owns
    [r * n, (r + 1) * n), n = padded / world; its moments are n floats each.  `mean=True` averages the gradients over the ranks.
    `lr_tail` / `tail_period` / `tail_split` per name as in FusedAdam.  `step_fn` replaces the HIP kernel (CPU tests only)."""

    SLOT_OF = {"vertex": "vertex", "opacity": "opacity", "center2D": "center2D", "shs": "color", "feature": "color", "color": "color"}

    def __init__(self, tensors: Dict[str, torch.Tensor], lrs: Dict[str, float], group=None, betas=(0.9, 0.999), eps: float = 1e-15,
                 mean: bool = False, tails: Optional[Dict[str, Tuple[float, int, int]]] = None, step_fn=None, force_collectives: bool = False):
        from diff_triangle_rasterization_2D.parallel import GradBucket
        names = list(tensors)
        slots = [self.SLOT_OF.get(n) for n in names]
        unknown = [n for n, sl in zip(names, slots) if sl is None]
        if unknown:
            raise ValueError(f"ShardedAdam: {unknown} are not gradient slots of the rasterizer's backward; use {sorted(self.SLOT_OF)} "
                             "(a tensor under another name would never be written by bucket.capture() and would be stepped on stale data)")
        if len(set(slots)) != len(slots):
            raise ValueError(f"ShardedAdam: {names} name the same gradient slot twice (shs / feature / color are ONE slot)")
        first = tensors[names[0]]
        self.device, self.group, self.betas, self.eps, self.mean = first.device, group, betas, eps, mean
        self.lrs = dict(lrs)
        self.tails = dict(tails or {})
        self.step_count = 0
        self._step_fn = step_fn
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.force_collectives = force_collectives
        self.bucket = GradBucket([t.shape for t in tensors.values()], self.device, group=group, names=slots, force_collectives=force_collectives)
        self.flat_param = torch.zeros(self.bucket.padded, device=self.device, dtype=torch.float32)
        self.segments: List[Tuple[str, int, int]] = []  # (name, offset, count) inside the flat buffers
        self.params: Dict[str, torch.Tensor] = {}
        off = 0
        for name, t in tensors.items():
            n = t.numel()
            self.flat_param[off:off + n].copy_(t.detach().reshape(-1))
            self.params[name] = self.flat_param[off:off + n].view(t.shape).requires_grad_(True)  # a leaf that shares the flat storage
            self.segments.append((name, off, n))
            off += n
        # what a capture of this bucket accepts as rasterizer inputs: these very leaves (see the class doc)
        self.bucket.expected_inputs = {sl: self.params[n] for n, sl in zip(names, slots)}
        self.slice_len = self.bucket.padded // self.world
        self.lo, self.hi = self.rank * self.slice_len, (self.rank + 1) * self.slice_len
        self.exp_avg = torch.zeros(self.slice_len, device=self.device, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(self.slice_len, device=self.device, dtype=torch.float32)
        self._work = None

    def set_lr(self, name: str, lr: float, lr_tail: Optional[float] = None):
        self.lrs[name] = lr
        if lr_tail is not None:
            _, period, split = self.tails[name]
            self.tails[name] = (lr_tail, period, split)

    def _my_slices(self) -> List[dict]:
        beta1, beta2 = self.betas
        rows = []
        for name, off, n in self.segments:  # the part of each tensor that falls into this rank's slice
            a, b = max(off, self.lo), min(off + n, self.hi)
            if a >= b:
                continue
            step_size, bias2_sqrt = _corrections(self.lrs[name], self.step_count, beta1, beta2)
            row = dict(param=self.flat_param[a:b], grad=self.bucket.flat[a:b], exp_avg=self.exp_avg[a - self.lo:b - self.lo],
                       exp_avg_sq=self.exp_avg_sq[a - self.lo:b - self.lo], step_size=step_size, bias2_sqrt=bias2_sqrt,
                       grad_scale=(1.0 / self.world) if self.mean else 1.0)
            if name in self.tails:
                lr_tail, period, split = self.tails[name]
                row.update(step_size_tail=_corrections(lr_tail, self.step_count, beta1, beta2)[0], period=period, split=split, index0=a - off)
            rows.append(row)
        return rows

    @torch.no_grad()
    def step(self):
        """Starts reduce-scatter -> slice update -> all-gather on the bucket's side stream (ordered behind whatever filled the bucket)."""
        self.step_count += 1
        self.bucket._filled = False
        run = self._step_fn or adam_step_slices
        collective = self.world > 1 or self.force_collectives
        mine_g = self.bucket.flat[self.lo:self.hi]
        mine_p = self.flat_param[self.lo:self.hi]

        def issue():
            if collective:
                if dist.get_backend(self.group) != "gloo":
                    dist.reduce_scatter_tensor(mine_g, self.bucket.flat, op=dist.ReduceOp.SUM, group=self.group)
                else:  # gloo has no reduce_scatter_tensor: all-reduce, then use the own slice
                    dist.all_reduce(self.bucket.flat, op=dist.ReduceOp.SUM, group=self.group)
            run(self._my_slices(), self.betas[0], self.betas[1], self.eps, self.device)
            if collective:
                return dist.all_gather_into_tensor(self.flat_param, mine_p, group=self.group, async_op=True)
            return None

        side = self.bucket._stream
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                self._work = issue()
        else:
            self._work = issue()

    def wait(self) -> Dict[str, torch.Tensor]:
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self.bucket._stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.bucket._stream)
        return self.params

"""Densification statistics of the reference model, kept by one fused HIP kernel per training iteration
(include/ts_model.h; src/diff_recon/models/VanillaTS_model.py:194-201 state, :347-363 `_training_statistic`,
:228-235 / :309-315 pruning and growth of the state arrays).

`DensificationStats` owns the six per-triangle arrays under the reference's attribute names, so the periodic
densification / pruning rules of the reference (:365-532, eager torch, every few hundred iterations) can read them
unchanged.  `update(render_pkg)` is the per-iteration part; with torch.distributed initialised and `all_views=True` the
per-view inputs are all-gathered first, so every rank applies the statistics of ALL views of the step and the replicas stay
identical.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch
import torch.distributed as dist
import torch.nn.functional as F

from diff_triangle_rasterization_2D import _C as _native

_lib = _native._lib
_fp = C.c_void_p
_lib.tsm_training_statistic.restype = C.c_int
_lib.tsm_training_statistic.argtypes = [C.c_int32, C.c_int32] + [_fp] * 11

_lib.tsm_select_scratch_bytes.restype = C.c_size_t
_lib.tsm_select_scratch_bytes.argtypes = [C.c_int32]
_lib.tsm_select_rows.restype = C.c_int
_lib.tsm_select_rows.argtypes = [C.c_int32, _fp, C.c_int32, _fp, _fp, C.c_size_t, C.POINTER(C.c_uint32), _fp]
for _name in ("tsm_scatter_rows", "tsm_gather_rows"):
    getattr(_lib, _name).restype = C.c_int
    getattr(_lib, _name).argtypes = [C.c_int64, C.c_int32, _fp, _fp, _fp, C.c_int64, _fp]
_lib.tsm_grow_classify.restype = C.c_int
_lib.tsm_grow_classify.argtypes = [C.c_int32, _fp, _fp, _fp, C.c_float, C.c_float, C.c_float, _fp, _fp]
_lib.tsm_split_vertex.restype = C.c_int
_lib.tsm_split_vertex.argtypes = [C.c_int32, _fp, _fp, _fp, _fp, _fp]
_lib.tsm_update_mask.restype = C.c_int
_lib.tsm_update_mask.argtypes = [C.c_int32, C.c_int32, _fp, _fp, _fp, C.c_float, C.c_float, _fp, _fp]
_lib.tsm_clip.restype = C.c_int
_lib.tsm_clip.argtypes = [C.c_int32, C.c_int32, _fp, C.c_float, _fp, _fp, _fp, _fp]
_lib.tsm_opacity_reset.restype = C.c_int
_lib.tsm_opacity_reset.argtypes = [C.c_int32, C.c_float, _fp, _fp, _fp, _fp]

_STATE = ("gradient_accum", "gradient_denom", "max_radii2D", "contrib_sum", "contrib_max", "contrib_denom")


class DensificationStats:
    def __init__(self, num_triangles: int, device):
        for name in _STATE:  # VanillaTS_model.py:196-201
            setattr(self, name, torch.zeros((num_triangles,), device=device, dtype=torch.float32))

    def __len__(self):
        return self.gradient_accum.shape[0]

    @torch.no_grad()
    def update(self, render_pkg: Dict[str, torch.Tensor], all_views: bool = False, group=None):
        """VanillaTS_model.py:347-363 for the view(s) in `render_pkg` ("radii", "center2D" with .grad populated, and --
        when rendered with rich_info -- "contrib_sum", "contrib_max")."""
        radii = render_pkg["radii"]
        if not radii.is_cuda:
            raise RuntimeError("DensificationStats (MI355X build) needs tensors on a HIP device; there is no CPU fallback")
        grad = render_pkg["center2D"].grad
        if grad is None:
            raise RuntimeError("center2D.grad is not populated: call update() after loss.backward()")
        P = len(self)
        rich = "contrib_sum" in render_pkg
        radii = radii.to(torch.int32).contiguous().view(1, P)
        grad = grad[:, :2].to(torch.float32).contiguous().view(1, P, 2)
        csum = render_pkg["contrib_sum"].contiguous().view(1, P) if rich else None
        cmax = render_pkg["contrib_max"].contiguous().view(1, P) if rich else None
        if all_views and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            world = dist.get_world_size(group)

            def gather(t):
                out = torch.empty((world,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
                dist.all_gather_into_tensor(out, t.contiguous(), group=group)
                return out

            radii, grad = gather(radii), gather(grad)
            if rich:
                csum, cmax = gather(csum), gather(cmax)
        V = radii.shape[0]
        with torch.cuda.device(radii.device):
            _native._check(_lib.tsm_training_statistic(
                P, V, radii.data_ptr(), grad.data_ptr(), csum.data_ptr() if rich else None, cmax.data_ptr() if rich else None,
                self.gradient_accum.data_ptr(), self.gradient_denom.data_ptr(), self.max_radii2D.data_ptr(),
                self.contrib_sum.data_ptr(), self.contrib_max.data_ptr(), self.contrib_denom.data_ptr(),
                torch.cuda.current_stream().cuda_stream), "training_statistic")

    def prune(self, prune_mask: torch.Tensor):
        """VanillaTS_model.py:228-234: drop the rows of pruned triangles."""
        keep = ~prune_mask
        for name in _STATE:
            setattr(self, name, getattr(self, name)[keep])

    def grow(self, new_count: int):
        """VanillaTS_model.py:309-315: zero state for appended triangles."""
        for name in _STATE:
            setattr(self, name, F.pad(getattr(self, name), (0, new_count), value=0))


# ---- periodic structural updates: the reference's VanillaTSModel methods on top of the native row operators -----------------------
# Every function takes the MODEL OBJECT `m` (the reference's VanillaTSModel instance, or anything with the same attributes:
# `_vertex / _opacity / _f_dc / _f_rest` parameters, `optimizer` (Adam, named param groups), the six statistics arrays,
# `config.model_update`, the schedulers of `_setup_model_update_utils`).  A maintainer replaces the body of the method of the same
# name by a one-line call.  Index logic that the reference itself expresses with torch (argsort /
# unique in `_contribution_pruning`) stays torch; every per-row pass goes through include/ts_model.h.

# the groups that carry one row per triangle (:217-218 skips the affine ones).  "shs" = the MI355X-first layout of the colour parameters: ONE
# (P, M, 3) tensor `_shs` in place of `_f_dc` + `_f_rest` (no torch.cat per forward, VanillaTS_model.py:79-80; the two learning rates live in
# the FusedAdam group as lr / lr_tail, diff_recon_hip/optim.py) -- a model carries either the reference's two groups or this one
_PARAM_GROUPS = ("vertex", "opacity", "f_dc", "f_rest", "shs")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_device(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("model-update operators (MI355X build) need tensors on a HIP device; there is no CPU fallback")


def _row_bytes(t: torch.Tensor) -> int:
    return (t.numel() // max(t.shape[0], 1)) * t.element_size() if t.shape[0] > 0 else (int(torch.Size(t.shape[1:]).numel()) * t.element_size())


def select_rows(mask: torch.Tensor, match: int = 1):
    """(pos, count): stable compaction plan of the rows with mask == match (tsm_select_rows)."""
    _need_device(mask)
    m8 = mask.contiguous().view(torch.uint8) if mask.dtype == torch.bool else mask.contiguous()
    P = m8.shape[0]
    pos = torch.empty((P,), device=mask.device, dtype=torch.int32)
    with torch.cuda.device(mask.device):
        nbytes = _lib.tsm_select_scratch_bytes(P)
        scratch = torch.empty((nbytes,), device=mask.device, dtype=torch.uint8)
        count = C.c_uint32(0)
        _native._check(_lib.tsm_select_rows(P, m8.data_ptr(), int(match), pos.data_ptr(), scratch.data_ptr(), nbytes, C.byref(count), _stream()),
                       "select_rows")
    return pos, int(count.value)


def scatter_rows(src: torch.Tensor, pos: torch.Tensor, dst: torch.Tensor, dst_row0: int = 0):
    """dst[dst_row0 + pos[i]] = src[i] for the selected rows."""
    if src.shape[0] == 0 or dst.numel() == 0:  # nothing selected
        return
    src = src.contiguous()
    with torch.cuda.device(src.device):
        _native._check(_lib.tsm_scatter_rows(src.shape[0], _row_bytes(src), pos.data_ptr(), src.data_ptr(), dst.data_ptr(), int(dst_row0),
                                             _stream()), "scatter_rows")


def gather_rows(src: torch.Tensor, idx: torch.Tensor, dst: torch.Tensor, dst_row0: int = 0):
    """dst[dst_row0 + j] = src[idx[j]]."""
    if idx.shape[0] == 0 or src.numel() == 0 or _row_bytes(src) == 0:  # zero-width rows: f_rest is (P, 0, 3) when max_sh_degree = 0
        return
    src = src.contiguous()
    with torch.cuda.device(src.device):
        _native._check(_lib.tsm_gather_rows(idx.shape[0], _row_bytes(src), idx.data_ptr(), src.data_ptr(), dst.data_ptr(), int(dst_row0),
                                            _stream()), "gather_rows")


def _selected_indices(pos: torch.Tensor, count: int) -> torch.Tensor:
    idx = torch.empty((count,), device=pos.device, dtype=torch.int32)
    scatter_rows(torch.arange(pos.shape[0], device=pos.device, dtype=torch.int32), pos, idx)
    return idx


def _replace_rows(m, build):
    """Rebuilds every per-triangle parameter with its Adam moments (`build(tensor, is_state) -> new tensor`) and re-registers them
    exactly like _prune_points_update_states / _grow_points_update_states (:214-227, 236-254)."""
    for group in m.optimizer.param_groups:
        if group["name"] not in _PARAM_GROUPS:
            continue
        param = group["params"][0]
        new_param = torch.nn.Parameter(build(param.data, False), requires_grad=True)
        stored = m.optimizer.state.pop(param, None)
        if stored:
            stored["exp_avg"] = build(stored["exp_avg"], True)
            stored["exp_avg_sq"] = build(stored["exp_avg_sq"], True)
            m.optimizer.state[new_param] = stored
        group["params"][0] = new_param
        setattr(m, f"_{group['name']}", new_param)


@torch.no_grad()
def prune_points(m, prune_mask: torch.Tensor):
    """VanillaTSModel._prune_points (:228-235): drops the masked triangles from the statistics, the parameters and the Adam moments."""
    keep = ~prune_mask
    pos, n = select_rows(keep)

    def compact(t, _is_state):
        out = torch.empty((n,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
        scatter_rows(t, pos, out)
        return out

    for name in _STATE:
        setattr(m, name, compact(getattr(m, name), True))
    _replace_rows(m, compact)
    return n


@torch.no_grad()
def densification(m, iteration: int):
    """VanillaTSModel._densification + _grow_points (:365-383, 256-315): clone the small, split the large of the triangles whose
    accumulated screen-space gradient exceeds the threshold.  Returns (grown, cloned, split) or None when the rule is inactive."""
    args = m.config.model_update.densification
    if args is None or not (args.start_iter < iteration <= args.end_iter and iteration % args.interval_iter == 0):
        return None
    grad_threshold = float(m.grad_threshold_scheduler(iteration - args.start_iter))
    vertex = m._vertex.data
    _need_device(vertex)
    P = vertex.shape[0]
    code = torch.empty((P,), device=vertex.device, dtype=torch.uint8)
    with torch.cuda.device(vertex.device):
        _native._check(_lib.tsm_grow_classify(P, vertex.contiguous().data_ptr(), m.gradient_accum.data_ptr(), m.gradient_denom.data_ptr(),
                                              float(args.min_view_count), grad_threshold, float(args.split_scale_threshold), code.data_ptr(),
                                              _stream()), "grow_classify")
    pos_keep, n_keep = select_rows(code != 2)  # `_prune_points(split_mask)`: the split parents go
    pos_c, n_c = select_rows(code, 1)
    pos_s, n_s = select_rows(code, 2)
    idx_c, idx_s = _selected_indices(pos_c, n_c), _selected_indices(pos_s, n_s)
    total = n_keep + n_c + 2 * n_s
    old_vertex = vertex.contiguous()

    def rebuild(t, is_state):
        # kept rows first, then [clones, first children, second children] (:285-288); new rows start with zero Adam moments (:245-247)
        out = (torch.zeros if is_state else torch.empty)((total,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
        scatter_rows(t, pos_keep, out)
        if not is_state:
            gather_rows(t, idx_c, out, n_keep)
            if t.data_ptr() == old_vertex.data_ptr():  # the vertex parameter: the split children get their own geometry
                with torch.cuda.device(t.device):
                    c1 = out[n_keep + n_c:n_keep + n_c + n_s]
                    c2 = out[n_keep + n_c + n_s:]
                    _native._check(_lib.tsm_split_vertex(n_s, idx_s.data_ptr(), old_vertex.data_ptr(), c1.data_ptr() if n_s else None,
                                                         c2.data_ptr() if n_s else None, _stream()), "split_vertex")
            else:
                gather_rows(t, idx_s, out, n_keep + n_c)
                gather_rows(t, idx_s, out, n_keep + n_c + n_s)
        return out

    for name in _STATE:  # :229-234 on the split parents, then zero rows for the new triangles (:309-315)
        setattr(m, name, rebuild(getattr(m, name), True))
    _replace_rows(m, rebuild)
    return n_c + n_s, n_c, n_s


def _mask(m, mode: int, a: float, b: float = 0.0) -> torch.Tensor:
    vertex, opacity = m._vertex.data.contiguous(), m._opacity.data.contiguous()
    _need_device(vertex)
    P = vertex.shape[0]
    out = torch.empty((P,), device=vertex.device, dtype=torch.uint8)
    with torch.cuda.device(vertex.device):
        _native._check(_lib.tsm_update_mask(P, mode, opacity.data_ptr(), vertex.data_ptr(), m.max_radii2D.data_ptr(), float(a), float(b),
                                            out.data_ptr(), _stream()), "update_mask")
    return out.view(torch.bool)


def _group(m, name):
    for group in m.optimizer.param_groups:
        if group["name"] == name:
            return group
    raise KeyError(name)


def _clip(m, name: str, mode: int, mask: torch.Tensor, value: float):
    """_clipping_update_states (:330-345): new Parameter object over the same storage, masked rows overwritten, their moments zeroed."""
    group = _group(m, name)
    param = group["params"][0]
    new_param = torch.nn.Parameter(param.data, requires_grad=True)
    stored = m.optimizer.state.pop(param, None)
    ea = stored["exp_avg"] if stored else None
    es = stored["exp_avg_sq"] if stored else None
    with torch.cuda.device(new_param.device):
        _native._check(_lib.tsm_clip(new_param.shape[0], mode, mask.view(torch.uint8).data_ptr(), float(value), new_param.data.data_ptr(),
                                     ea.data_ptr() if ea is not None else None, es.data_ptr() if es is not None else None, _stream()), "clip")
    if stored:
        m.optimizer.state[new_param] = stored
    group["params"][0] = new_param
    setattr(m, f"_{name}", new_param)


@torch.no_grad()
def opacity_pruning(m, iteration: int):
    """VanillaTSModel._opacity_pruning (:384-396)."""
    args = m.config.model_update.opacity_pruning
    if args is None or not (args.start_iter < iteration <= args.hold_iter and iteration % args.interval_iter == 0):
        return None
    mask = _mask(m, 0, m.opacity_pruning_scheduler(iteration - args.start_iter))
    before = mask.shape[0]
    return before - prune_points(m, mask)


@torch.no_grad()
def opacity_clipping(m, iteration: int):
    """VanillaTSModel._opacity_clipping (:397-409): opacities above the scheduled threshold are set to the logit 10."""
    args = m.config.model_update.opacity_clipping
    if args is None or not (args.start_iter < iteration <= args.hold_iter and iteration % args.interval_iter == 0):
        return None
    mask = _mask(m, 1, m.opacity_clipping_scheduler(iteration - args.start_iter))
    count = int(mask.sum().item())
    if count > 0:
        _clip(m, "opacity", 0, mask, 10.0)
    return count


@torch.no_grad()
def scale_pruning(m, iteration: int):
    """VanillaTSModel._scale_pruning (:411-427)."""
    args = m.config.model_update.scale_pruning
    if args is None or not (args.start_iter < iteration <= args.end_iter and iteration % args.interval_iter == 0):
        return None
    mask = _mask(m, 2, args.radii_threshold, args.scale_threshold)
    before = mask.shape[0]
    return before - prune_points(m, mask)


@torch.no_grad()
def scale_clipping(m, iteration: int):
    """VanillaTSModel._scale_clipping (:446-463): triangles larger than the scheduled maximum are shrunk about their centre."""
    args = m.config.model_update.scale_clipping
    if args is None or not (args.start_iter < iteration <= args.hold_iter and iteration % args.interval_iter == 0):
        return None
    scale_max = float(m.scale_max_scheduler(iteration - args.start_iter))
    mask = _mask(m, 3, scale_max)
    count = int(mask.sum().item())
    if count > 0:
        _clip(m, "vertex", 1, mask, scale_max)
    return count


@torch.no_grad()
def opacity_reset(m, iteration: int):
    """VanillaTSModel._opacity_reset + _reset_opacity_update_states (:524-537, 316-328)."""
    args = m.config.model_update.opacity_reset
    if args is None or not (args.start_iter < iteration <= args.end_iter and iteration % args.interval_iter == 0):
        return None
    group = _group(m, "opacity")
    param = group["params"][0]
    new_param = torch.nn.Parameter(param.data.clone(), requires_grad=True)
    stored = m.optimizer.state.pop(param, None)
    ea = stored["exp_avg"] if stored else None
    es = stored["exp_avg_sq"] if stored else None
    with torch.cuda.device(new_param.device):
        _native._check(_lib.tsm_opacity_reset(new_param.shape[0], float(args.reset_value), new_param.data.data_ptr(),
                                              ea.data_ptr() if ea is not None else None, es.data_ptr() if es is not None else None, _stream()),
                       "opacity_reset")
    if stored:
        m.optimizer.state[new_param] = stored
    group["params"][0] = new_param
    m._opacity = new_param
    return new_param.shape[0]


@torch.no_grad()
def contribution_pruning(m, iteration: int, inter_point_distance=None, get_inside_mask=None):
    """VanillaTSModel._contribution_pruning (:465-522): the triangles with the smallest running contrib_max / contrib_sum among those
    seen often enough are pruned, the sparsest of them retained.  The ranking is the reference's own torch index logic; the row
    surgery is native.  `inter_point_distance` / `get_inside_mask` default to the drop-in simple_knn.distCUDA2 form and the
    reference's bounding-box test (model_utils.py:34-58)."""
    args = m.config.model_update.contribution_pruning
    if args is None or not (args.start_iter < iteration <= args.end_iter and iteration % args.interval_iter == 0):
        return None
    if inter_point_distance is None:
        from simple_knn import distCUDA2
        inter_point_distance = lambda pc: distCUDA2(pc).clamp_(min=1e-10).sqrt()  # model_utils.py:34-36
    target_point_num, prune_ratio, contrib_max_ratio = args.target_point_num, args.prune_ratio, args.contrib_max_ratio
    sparsity_retain_ratio = args.sparsity_retain_ratio
    for it, point_num in zip(args.downsample_iteration, args.downsample_point_num):  # :479-485
        if iteration > it:
            target_point_num = point_num
            contrib_max_ratio *= 0.5
            new_ratio = sparsity_retain_ratio + (0.8 - sparsity_retain_ratio) * 0.5
            prune_ratio *= (1 - sparsity_retain_ratio) / (1 - new_ratio)
            sparsity_retain_ratio = new_ratio
    xyz = m._vertex.data.mean(dim=1)
    total = m._vertex.shape[0]
    if get_inside_mask is not None:
        inside = get_inside_mask(xyz, m.scene_bbox)
    elif getattr(m, "scene_bbox", None) is None:
        inside = torch.ones((total,), device=xyz.device, dtype=torch.bool)
    else:
        bb = m.scene_bbox
        half = len(bb) // 2
        inside = torch.ones((total,), device=xyz.device, dtype=torch.bool)
        for k in range(half):
            inside &= (xyz[:, k] >= bb[k]) & (xyz[:, k] <= bb[half + k])
    opac = torch.sigmoid(m._opacity.data)
    ste = (opac > m.ste_threshold if getattr(m, "ste_threshold", None) is not None else torch.ones_like(opac, dtype=torch.bool)).squeeze()
    valid = int((inside & ste).sum().item())
    select_mask = m.contrib_denom >= args.min_view_count
    select_count = int(select_mask.sum().item())
    diff = max(0, valid - target_point_num * 0.99) * total / valid
    prune_count = min(diff * prune_ratio, select_count * args.max_prune_ratio)
    n_max, n_sum = int(prune_count * contrib_max_ratio), int(prune_count * (1 - contrib_max_ratio))
    select_idx = torch.argwhere(select_mask).squeeze(1)
    idx_max = select_idx[torch.argsort(m.contrib_max[select_mask])[:n_max]]
    idx_sum = select_idx[torch.argsort(m.contrib_sum[select_mask])[:n_sum]]
    prune_idx = torch.cat((idx_max, idx_sum)).unique()
    if sparsity_retain_ratio > 0:
        dist_ = inter_point_distance(xyz.contiguous())
        retain = int(sparsity_retain_ratio * len(prune_idx))
        prune_idx = prune_idx[torch.argsort(dist_[prune_idx], descending=True)[retain:]]
    prune_mask = torch.zeros_like(select_mask)
    prune_mask[prune_idx] = 1
    m.contrib_sum[select_mask] = 0
    m.contrib_max[select_mask] = 0
    m.contrib_denom[select_mask] = 0
    prune_points(m, prune_mask)
    return int(prune_idx.shape[0])


def set_gamma(m, iteration: int):
    """VanillaTSModel._set_gamma (:548-553)."""
    from .schedulers import gamma_at
    args = m.config.model_update.gamma_schedule
    if args is not None:
        m.gamma = gamma_at(iteration, m.gamma, args.start_iter, args.end_iter, m.gamma_scheduler)


def set_sh_degree(m, iteration: int):
    """VanillaTSModel._set_sh_degree (:555-565)."""
    from .schedulers import sh_degree_at
    args = m.config.model_update.sh_schedule
    if args is not None:
        m.active_sh_degree = sh_degree_at(iteration, args.one_up_iters, m.max_sh_degree)


def run_model_update(m, iteration: int, render_pkgs=()):
    """VanillaTSModel.model_update (:567-581), same order: statistics of the step's views, densification, the pruning / clipping
    rules, opacity reset, then the gamma and SH-degree schedules.  `m` carries the reference's attribute names and inherits
    DensificationStats (its `update` is `_training_statistic`).  Returns [(rule, result)] of the rules that fired."""
    if m.config.model_update is None:
        return []
    # _training_statistic (:347-350) returns early when config.model_update.statistic is None, outside (start_iter, end_iter], or
    # without a render package: outside that window the accumulators must not move (densification, scale and contribution pruning
    # select on them)
    args = m.config.model_update.statistic
    if args is not None and args.start_iter < iteration <= args.end_iter:
        for pkg in render_pkgs:
            if pkg is not None:
                m.update(pkg)
    fired = []
    for rule in (densification, opacity_pruning, opacity_clipping, scale_pruning, scale_clipping, contribution_pruning, opacity_reset):
        res = rule(m, iteration)
        if res is not None:
            fired.append((rule.__name__, res))
    set_gamma(m, iteration)
    set_sh_degree(m, iteration)
    return fired

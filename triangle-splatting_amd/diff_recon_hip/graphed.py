"""A whole training step as ONE HIP graph (round 5).

What bounds a small scene is the host: a 10 k-triangle step at 256 x 256 is ~20 kernel launches, ~15 allocations and a walk through the autograd
engine -- 0.34-0.44 ms of host work in front of 0.18 ms of device work (profiles/r05_notes.md).  The reference cannot avoid that: its forward reads
`num_rendered` back (rasterizer.cu:189-191), and a host read cannot sit inside a graph.  The sync-free forward (include/ts2d.h: ts2d_forward) has no
host read, so the step -- render, loss, backward, optimizer -- is captured once with torch.cuda.CUDAGraph and replayed with one launch:

    step = GraphedStep(fwd_bwd, instance_capacity=cap)       # fwd_bwd(): zero_grad(set_to_none=True) + forward + loss + backward on FIXED tensors
    for it in range(n):
        new_camera_into(static_camera_tensors)               # inputs change IN PLACE (parameters by the optimizer, cameras / targets by copy_)
        step.replay()                                        # overwrites the static .grad tensors the capture left behind
        optimizer.step()                                     # eager: FusedAdam's learning rates and bias corrections are HOST scalars rewritten
        if it % 100 == 0 and step.overflowed()[0]:           #        every iteration (VanillaTS_model.py:583); inside the graph they would freeze
            step = GraphedStep(fwd_bwd, instance_capacity=2 * cap)  # one host read, whenever the caller likes

Gradients: capture with `.grad = None` (fn starts with zero_grad(set_to_none=True)); autograd then ASSIGNS the gradient tensors, they live in the
graph's pool, and every replay overwrites them in place -- do not set them to None between replays.
Constraints are CUDA-graph constraints: fixed shapes (re-capture after a densification changes the triangle count), no host synchronisation inside
`fn`, tensors created inside `fn` live in the graph's private pool and are overwritten by the next replay.  `instance_capacity` sizes the binning
state: an int, or a callable (P, width, height) -> int; a step that renders more instances than that emits nothing (background image, zero
gradients) and `overflowed()` says so.  The raster settings' tensors (view / projection matrix, camera position, background) are read through
their device pointers on every replay, so a new camera is a `copy_` into them; `tanfovx` / `tanfovy`, the image size, gamma and the flags are
baked into the captured launches."""
from __future__ import annotations

from typing import Callable, Union

import torch

import diff_triangle_rasterization_2D as _pkg


class GraphedStep:
    def __init__(self, fn: Callable[[], object], instance_capacity: Union[int, Callable[[int, int, int], int]], warmup: int = 3, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedStep needs a HIP device (there is no CPU path)")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        prev = _pkg._instance_capacity
        _pkg.set_instance_capacity(instance_capacity)
        try:
            with torch.cuda.device(self.device):
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):  # torch's capture recipe: eager iterations on a side stream first (allocator warm-up, autograd nodes
                    for _ in range(max(warmup, 1)):  # created on a non-default stream)
                        fn()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):
                    self.result = fn()
                torch.cuda.synchronize()
            self._forward = _pkg._last_sync_free_forward  # (P, W, H, geometry state, image state) of the LAST forward inside fn
        finally:
            _pkg.set_instance_capacity(prev)

    def replay(self):
        self.graph.replay()
        return self.result

    __call__ = replay

    def overflowed(self):
        """(overflowed, true instance count) of the last forward of the most recent replay -- one blocking read."""
        from diff_triangle_rasterization_2D import _C
        return _C.forward_status(*self._forward)

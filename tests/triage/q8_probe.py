"""Profiling aid: lane-group statistics of the queue blend kernels (render_q8.hip) on the headline scene; needs a library built with
-DTS2D_STATS (tools/bin/libts2d_stats.so, loaded instead of the product library when TS2D_LIB points at it)."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import synthetic, helpers
from diff_triangle_rasterization_2D import _C
P, W, H, D = 1_000_000, 1920, 1080, 3
s = synthetic.scene(P, W, H, D, seed=42)
names = ["list_entries_culled", "entry_group_pairs_queued", "wave_steps", "chunks", "pixel_entry_pairs_blended", "quadrant_waves", "batches", "rows",
         "conflict_steps", "partial_batches"]
out = {}
if hasattr(_C._lib, "ts2d_stats_read_q8"):
    buf = (ctypes.c_ulonglong * 12)()
    for what, bw in (("forward", False), ("forward+backward", True)):
        _C._lib.ts2d_stats_read_q8(buf, 1)
        hf = helpers.hip_forward_backward(s, backward=bw)
        torch.cuda.synchronize()
        _C._lib.ts2d_stats_read_q8(buf, 1)
        out[what] = dict(zip(names, list(buf)[:10]))
    f = out["forward"]
    b = {k: out["forward+backward"][k] - f[k] for k in names}
    for nm, v in (("forward", f), ("backward", b)):
        st = max(v["wave_steps"], 1)
        v.update(lane_occupancy=v["pixel_entry_pairs_blended"] / st / 64, steps_per_chunk=st / max(v["chunks"], 1), rows_per_batch=v["rows"] / max(v["batches"], 1),
                 groups_busy_per_step=v["entry_group_pairs_queued"] / st)
    res = {"scene": f"S(P={P}, {W}x{H}, D={D}, seed=42)", "num_rendered": hf["num_rendered"], "forward": f, "backward": b}
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "q8_stats.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))
else:
    print("library has no statistics (build with TS2D_EXTRA_FLAGS=-DTS2D_STATS)")

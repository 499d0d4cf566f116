"""Profiling aid: times render_fwd / render_bwd of the headline scene (HIP events of libts2d's profile hook), rich and plain,
and prints the lane-group statistics when the library was built with -DTS2D_STATS."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import synthetic, helpers
from diff_triangle_rasterization_2D import _C
P, W, H, D = 1_000_000, 1920, 1080, 3
s = synthetic.scene(P, W, H, D, seed=42)
for rich in (True, False):
    helpers.hip_forward_backward(s, rich_info=rich)  # warm
    torch.cuda.synchronize()
    _C.profile_reset(); _C.profile_only(""); _C.profile_enable(True)
    for _ in range(5):
        helpers.hip_forward_backward(s, rich_info=rich)
    torch.cuda.synchronize()
    rows = {n: ms / max(k, 1) for n, ms, k in _C.profile_read()}
    _C.profile_enable(False)
    print(f"rich={rich}: " + "  ".join(f"{n} {v:.3f}" for n, v in rows.items() if n.startswith("render")))
if hasattr(_C._lib, "ts2d_stats_read_group"):
    buf = (ctypes.c_ulonglong * 12)()
    _C._lib.ts2d_stats_read_group(buf, 1)
    hf = helpers.hip_forward_backward(s, backward=False)
    torch.cuda.synchronize()
    _C._lib.ts2d_stats_read_group(buf, 1)
    v = list(buf); N = hf["num_rendered"]
    print(f"N {N}; visits {v[0]}; (entry,block) survivors {v[1]}; (entry,quadrant) survivors {v[7]}; steps {v[2]}; windows {v[3]}; "
          f"pairs {v[4]}; waves {v[5]}; batches {v[6]}")
    print(f"steps with a row shared by two groups {v[8]} ({v[8]/max(v[2],1):.3f} of steps); passes {v[9]}, second passes {v[10]}")
    import json
    json.dump({"scene": f"S(P={P}, {W}x{H}, D={D}, seed=42)", "num_rendered": N, "list_entries_visited_per_quadrant_wave_total": v[0],
               "entry_block_survivors": v[1], "entry_quadrant_survivors": v[7], "wave_steps": v[2], "windows": v[3],
               "pixel_entry_pairs_blended": v[4], "quadrant_waves": v[5], "batches_with_work": v[6],
               "lanes_hit_per_step": v[4] / max(v[2], 1), "lane_occupancy": v[4] / max(v[2], 1) / 64,
               "steps_per_entry_quadrant_survivor": v[2] / max(v[7], 1)},
              open(os.path.join(ROOT, "gpurun_out", "blend_stats.json"), "w"), indent=1)
    print(f"steps per (entry,quadrant) survivor {v[2]/max(v[7],1):.3f}; lanes per step {v[4]/max(v[2],1):.1f} ({v[4]/max(v[2],1)/64:.3f}); "
          f"steps per batch {v[2]/max(v[6],1):.1f}; windows per batch {v[3]/max(v[6],1):.2f}; block-survivors per step {v[1]/max(v[2],1):.2f}")

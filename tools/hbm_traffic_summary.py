"""Per-kernel HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB units).
Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE under-reports wide coalesced reads by exactly 2x on gfx950 -> doubled;
WRITE_SIZE is uncalibrated there -> calibrated HERE on a store of known size in the same run: the hipMemsetAsync of the
gradient records (64 bytes per triangle, `--known-fill-bytes`), whose fill kernel appears in the same trace."""
import collections, csv, glob, json, re, sys

NAMES = ["render_fwd_group", "render_bwd_group", "render_fwd", "render_bwd", "render3d_fwd", "render3d_bwd", "preprocess_fwd", "preprocess_bwd",
         "scan_emit", "gather_blocksum", "rs_hist", "rs_prefix", "rs_scatter", "tile_ranges", "fillBuffer", "zero_words4"]


def short(k):
    for n in NAMES:
        if n in k:
            return n
    return None


def load(d, counter):
    f = sorted(glob.glob(d + "/*/*_counter_collection.csv"))[-1]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = short(r["Kernel_Name"])
        if n and r["Counter_Name"] == counter:
            acc[n].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


known = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
# optional 4th argument: a --pmc FETCH_SIZE pass of tools/bin/gather_calib (64-byte record gathers / a streaming read of KNOWN size)
gather_factor = stream_factor = None
if len(sys.argv) > 4:
    f = sorted(glob.glob(sys.argv[4] + "/*/*_counter_collection.csv"))[-1]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            acc["gather64" if "gather64" in r["Kernel_Name"] else "stream64" if "stream64" in r["Kernel_Name"] else "other"].append(float(r["Counter_Value"]))
    n_rec = 4 * 1000 * 1000 + 1
    if acc.get("gather64"):
        gather_factor = (n_rec * 68.0) / (sum(acc["gather64"]) / len(acc["gather64"]) * 1024.0)  # 64 B record + 4 B index per lane
    if acc.get("stream64"):
        stream_factor = (n_rec * 64.0) / (sum(acc["stream64"]) / len(acc["stream64"]) * 1024.0)
# optional 5th argument: a --pmc WRITE_SIZE pass of tools/bin/gather_calib (round 6: a streaming store and a 64-byte record scatter of KNOWN size)
write_stream_factor = write_scatter_factor = None
if len(sys.argv) > 5:
    f = sorted(glob.glob(sys.argv[5] + "/*/*_counter_collection.csv"))[-1]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "WRITE_SIZE":
            acc["fill64" if "fill64" in r["Kernel_Name"] else "scatter64" if "scatter64" in r["Kernel_Name"] else "other"].append(float(r["Counter_Value"]))
    n_rec = 4 * 1000 * 1000 + 1
    if acc.get("fill64") and sum(acc["fill64"]) > 0:
        write_stream_factor = (n_rec * 64.0) / (sum(acc["fill64"]) / len(acc["fill64"]) * 1024.0)
    if acc.get("scatter64") and sum(acc["scatter64"]) > 0:
        write_scatter_factor = (n_rec * 64.0) / (sum(acc["scatter64"]) / len(acc["scatter64"]) * 1024.0)
(fetch, nf), (write, nw) = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
# the step's own store of known size: the clear of the gradient records (a kernel since round 5: zero_words4_kernel; hipMemsetAsync's fill kernel before)
fill_name = "zero_words4" if write.get("zero_words4", 0) > 0 else "fillBuffer"
if known > 0 and not write.get(fill_name, 0) > 0:
    sys.exit(f"WRITE_SIZE calibration failed: the fill of {int(known)} known bytes reported {write.get(fill_name, 0)} KiB (kernels seen: {sorted(write)})")
wcal = 1.0
if known > 0:
    wcal = known / (write[fill_name] * 1024.0)
out = {"_calibration": {"fetch_factor": 2.0, "write_factor": round(wcal, 4),
                        "write_calibrated_on": f"{fill_name} kernel of {int(known)} known bytes: WRITE_SIZE reported {write.get(fill_name, 0):.1f} KiB",
                        "write_factor_measured_streaming_dwordx4_store": None if write_stream_factor is None else round(write_stream_factor, 4),
                        "write_factor_measured_64B_record_scatter": None if write_scatter_factor is None else round(write_scatter_factor, 4),
                        "fetch_factor_measured_64B_record_gather": None if gather_factor is None else round(gather_factor, 4),
                        "fetch_factor_measured_streaming_dwordx4": None if stream_factor is None else round(stream_factor, 4),
                        "note": "FETCH_SIZE x2 per MI355X_MICROARCH.md (wide reads); WRITE_SIZE x write_factor (own calibration); the blend kernels read "
                                "64-byte records by gather: `hbm_bytes_per_launch_gather_calibrated` prices their FETCH_SIZE with the factor measured "
                                "on a gather of known size (tools/gather_calib.hip)"}}
for k in sorted(set(fetch) | set(write)):
    f_kib, w_kib = fetch.get(k, 0.0), write.get(k, 0.0)
    out[k] = {"FETCH_SIZE_KiB": round(f_kib, 1), "WRITE_SIZE_KiB": round(w_kib, 1), "launches": nf.get(k, nw.get(k, 0)),
              "hbm_bytes_per_launch": int((2.0 * f_kib + wcal * w_kib) * 1024)}
    if gather_factor is not None and k.startswith("render"):
        out[k]["hbm_bytes_per_launch_gather_calibrated"] = int((gather_factor * f_kib + wcal * w_kib) * 1024)
print(json.dumps(out, indent=1))

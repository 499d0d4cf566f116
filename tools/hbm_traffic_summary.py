"""Per-kernel HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB units)."""
import collections, csv, glob, json, re, sys

NAMES = ["render_fwd", "render_bwd", "preprocess_fwd", "preprocess_bwd", "emit_instances", "tile_ranges", "gather_tiles"]


def short(k):
    for n in NAMES:
        if n in k:
            return n.replace("emit_instances", "emit_keys")
    return None


def load(d, counter):
    f = glob.glob(d + "/*/*_counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = short(r["Kernel_Name"])
        if n and r["Counter_Name"] == counter:
            acc[n].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    f_kib, w_kib = fetch.get(k, 0.0), write.get(k, 0.0)
    out[k] = {"FETCH_SIZE_KiB": round(f_kib, 1), "WRITE_SIZE_KiB": round(w_kib, 1),
              "hbm_bytes_per_launch": int((2.0 * f_kib + w_kib) * 1024),
              "correction": "FETCH_SIZE x2 (gfx950 wide-read under-count), WRITE_SIZE as reported"}
print(json.dumps(out, indent=1))

"""Condenses a rocprofv3 kernel_stats.csv (--kernel-trace --stats --output-format csv) into a short table."""
import csv, re, sys

def short(name):
    m = re.search(r"(render\w*_kernel|preprocess\w*_kernel|scan_emit_kernel|gather_blocksum_kernel|rs_\w+_kernel|tile_ranges_kernel|zero_words_kernel)(<[^>]*>)?", name)
    if m:
        return m.group(0)
    m = re.search(r"rocprim::[A-Za-z0-9_]+::detail::(\w+)<rocprim::[A-Za-z0-9_]+::detail::(\w+)", name)
    if m:
        return "rocprim::" + m.group(2)[:60]
    return name[:70]

rows = list(csv.DictReader(open(sys.argv[1])))
print("# " + (sys.argv[2] if len(sys.argv) > 2 else ""))
print("kernel,calls,avg_us,total_us,pct")
for r in rows:
    print(f"{short(r['Name'])},{r['Calls']},{float(r['AverageNs'])/1e3:.1f},{float(r['TotalDurationNs'])/1e3:.1f},{r['Percentage']}")

#!/usr/bin/env python
"""Idle time in front of each kernel of the bench step, from a rocprofv3 --kernel-trace CSV: for every kernel name the average gap between the
end of the previous kernel on the device and its own start, over the launches of the trace (the first few = warm-up, dropped).

    python tools/gap_summary.py <..._kernel_trace.csv> [skip_launches_per_kernel]
"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gap, dur, cnt, seen = defaultdict(float), defaultdict(float), defaultdict(int), defaultdict(int)
prev_end = None
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    seen[name] += 1
    if prev_end is not None and seen[name] > skip:
        gap[name] += max(0, st - prev_end) / 1e3
        dur[name] += (en - st) / 1e3
        cnt[name] += 1
    prev_end = max(prev_end or 0, en)
print(f"{'kernel':62s} {'n':>4s} {'gap_us':>8s} {'dur_us':>8s}")
tg = 0.0
for name in sorted(cnt, key=lambda n: -gap[n] / cnt[n]):
    print(f"{name:62s} {cnt[name]:4d} {gap[name] / cnt[name]:8.2f} {dur[name] / cnt[name]:8.2f}")

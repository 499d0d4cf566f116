#!/usr/bin/env python
"""Which kernels of the bench step run BESIDE each other, from a rocprofv3 --kernel-trace CSV (round 6: the forward forks a side stream).

For every kernel name: launches, average duration, the average time it shared the device with a kernel of ANOTHER stream / queue (overlap_us), and
which kernels those were.  Also the average step length seen on the device (first start to last end of one step's launches, from the per-step count
of the named anchor kernel) against the plain sum of the kernels' durations.

    python tools/overlap_summary.py <..._kernel_trace.csv> [skip_launches_per_kernel]
"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 45
short = lambda n: n.replace("(anonymous namespace)::", "").replace("ts::", "").replace("void ", "").split("(")[0][:58]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in rows))
seen = defaultdict(int)
dur, ovl, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
with_whom = defaultdict(lambda: defaultdict(float))
active = []  # kernels that started earlier and may still run
for st, en, name, q in ev:
    seen[name] += 1
    active = [a for a in active if a[1] > st]
    if seen[name] > skip:
        cnt[name] += 1
        dur[name] += (en - st) / 1e3
    for a in active:
        o = (min(en, a[1]) - st) / 1e3
        if o > 0:
            if seen[name] > skip:
                ovl[name] += o
                with_whom[name][a[2]] += o
            if seen[a[2]] > skip:
                ovl[a[2]] += o
                with_whom[a[2]][name] += o
    active.append((st, en, name, q))
print(f"{'kernel':60s} {'n':>5s} {'dur_us':>8s} {'overlap_us':>10s}  beside")
tot = 0.0
for name in sorted(cnt, key=lambda n: -dur[n]):
    n = cnt[name]
    who = ", ".join(f"{w[:28]} {v / n:.1f}" for w, v in sorted(with_whom[name].items(), key=lambda kv: -kv[1])[:3])
    print(f"{name:60s} {n:5d} {dur[name] / n:8.2f} {ovl[name] / n:10.2f}  {who}")
    tot += dur[name] / n * (n / max(cnt.values()))
print(f"sum of average durations per step (weighted by launches per step): {tot:.1f} us")
# device time covered by at least one kernel, per step of the densest kernel
m = max(cnt.values())
span_busy, last_end, first = 0.0, None, None
tail = [e for e in ev]
# union length of all intervals behind the skipped launches
started = defaultdict(int)
union, cur_s, cur_e = 0.0, None, None
for st, en, name, q in ev:
    started[name] += 1
    if started[name] <= skip:
        continue
    if cur_e is None or st > cur_e:
        if cur_e is not None:
            union += (cur_e - cur_s) / 1e3
        cur_s, cur_e = st, en
    else:
        cur_e = max(cur_e, en)
if cur_e is not None:
    union += (cur_e - cur_s) / 1e3
print(f"device time covered by at least one kernel, per step: {union / m:.1f} us  (steps counted: {m})")

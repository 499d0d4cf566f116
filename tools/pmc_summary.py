#!/usr/bin/env python
"""Per-kernel averages of a rocprofv3 --pmc counter_collection CSV: counter value per launch, summed over the dispatch's dimensions.

    python tools/pmc_summary.py <..._counter_collection.csv> [skip_launches_per_kernel]
"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 1
short = lambda n: n.replace("(anonymous namespace)::", "").replace("ts::", "").replace("void ", "").split("(")[0][:58]
per = defaultdict(lambda: defaultdict(float))   # (kernel, dispatch) -> counter -> value
for r in rows:
    per[(short(r["Kernel_Name"]), r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
by_kernel = defaultdict(list)
for (k, d), c in sorted(per.items(), key=lambda kv: int(kv[0][1])):
    by_kernel[k].append(c)
names = sorted({c for v in per.values() for c in v})
print(f"{'kernel':60s} {'n':>4s} " + " ".join(f"{c:>16s}" for c in names))
for k, lst in sorted(by_kernel.items()):
    use = lst[skip:] if len(lst) > skip else lst
    print(f"{k:60s} {len(use):4d} " + " ".join(f"{sum(c.get(nm, 0.0) for c in use) / len(use):16.1f}" for nm in names))

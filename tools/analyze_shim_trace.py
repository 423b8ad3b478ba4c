#!/usr/bin/env python3
"""Per-frame device timeline of the shim loop from a rocprofv3 kernel trace (VSLAM_SHIM_ROCPROF=dir python bench.py --only-shim):
average duration of every kernel in its position of the frame, and the idle gap in front of it (host time + launch latency).
usage: python tools/analyze_shim_trace.py gpurun_out/prof_shim/shim_kernel_trace.csv"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a frame starts at k_fast_box; keep the frames of the shim loop only (they contain k_stage)
frames, cur = [], []
for r in rows:
    name = r["Kernel_Name"].split("(")[0]
    if name == "k_fast_box" and cur:
        frames.append(cur)
        cur = []
    cur.append((name, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
frames.append(cur)
shim = [f for f in frames if any(n == "k_stage" for n, _, _ in f)][20:]
sig = defaultdict(list)
for f in shim:
    key = tuple(n for n, _, _ in f)
    sig[key].append(f)
key, fs = max(sig.items(), key=lambda kv: len(kv[1]))
print("frames with the most common launch sequence: %d of %d" % (len(fs), len(shim)))
tot_busy = tot_gap = 0.0
prev_end = None
for i, name in enumerate(key):
    dur = sum(f[i][2] - f[i][1] for f in fs) / len(fs) / 1e3
    gap = sum((f[i][1] - f[i - 1][2]) for f in fs) / len(fs) / 1e3 if i else 0.0
    tot_busy += dur
    tot_gap += gap
    print("%-22s busy %8.2f us   gap before %8.2f us" % (name, dur, gap))
span = sum(f[-1][2] - f[0][1] for f in fs) / len(fs) / 1e3
print("busy %.1f us, gaps %.1f us, first kernel start -> last kernel end %.1f us" % (tot_busy, tot_gap, span))

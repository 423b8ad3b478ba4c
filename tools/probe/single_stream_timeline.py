"""Per-frame device timeline of ONE stream through the fused path (launch sequence 4) from a rocprofv3 kernel trace:
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_single -o single -- python3 tools/probe/time_single_stream.py 1 300
   python tools/probe/single_stream_timeline.py gpurun_out/prof_single/single_kernel_trace.csv
average duration of every kernel of the frame queue in its position of the frame and the idle gap in front of it."""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
image = {"k_fast_box", "k_emit", "k_brief", "k_stereo_dist", "k_synth", "k_xcc_probe"}
frames, cur = [], []
for r in rows:
    name = r["Kernel_Name"].split("(")[0]
    if name in image or name.startswith("__amd") or "at::" in name:
        continue
    if name == "k_track_candidates" and cur:
        frames.append(cur); cur = []
    cur.append((name, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
frames = [f for f in frames[60:] if f and f[0][0] == "k_track_candidates"]
sig = defaultdict(list)
for f in frames:
    sig[tuple(n for n, _, _ in f)].append(f)
key, fs = max(sig.items(), key=lambda kv: len(kv[1]))
print("frames with the most common launch sequence: %d of %d" % (len(fs), len(frames)))
busy = gaps = 0.0
for i, name in enumerate(key):
    dur = sum(f[i][2] - f[i][1] for f in fs) / len(fs) / 1e3
    gap = 0.0 if i == 0 else sum(f[i][1] - max(e for _, _, e in f[:i]) for f in fs) / len(fs) / 1e3
    print("%-28s busy %8.2f us   start after the latest earlier end %8.2f us" % (name, dur, gap))
    busy += dur
per = sum(b[0][1] - a[0][1] for a, b in zip(fs, fs[1:]) ) / max(len(fs) - 1, 1) / 1e3
print("frame period %.1f us" % per)

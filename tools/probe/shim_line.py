import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
d=d.get('shim_path', d)
print(d["ms_per_frame"], d.get("median_ms"), d["pinned_images"]["ms_per_frame"], d["host_breakdown_ms"]["begin_call"])

"""print the key numbers of bench.py JSON lines (development aid):  python tools/bench_print.py file.json ..."""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:      # noqa: BLE001
        print(f, "unreadable:", e)
        continue
    r, ri = d.get("roofline") or {}, d.get("roofline_iteration") or {}
    rep = d.get("repeats") or {}
    print(f"{f}: {d['value']} {d['unit']} (median {rep.get('value_median')}), {d['ms_per_step']} ms/step; "
          f"{r.get('kernel')} {r.get('avg_launch_us')} us frac {r.get('frac')}; iteration frac {ri.get('frac')} "
          f"moved/it {ri.get('moved_bytes_per_iteration')}")
    for k in ("kernels_timed", "other_kernels"):
        if d.get(k):
            print("   ", k, d[k])

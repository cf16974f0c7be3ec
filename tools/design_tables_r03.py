"""Writes the round-3 measurement block of DESIGN.md section 5 from the committed bench lines profiles/r03_bench_*.json, with
the round-2 lines beside them (development aid: python tools/design_tables_r03.py).  The block sits between the markers
`<!-- r03 tables -->` and `<!-- /r03 tables -->`."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def L(name):
    p = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(p):
        return None
    return json.loads([ln for ln in open(p).read().splitlines() if ln.startswith("{")][-1])


def kern(d):
    r = d["roofline"]
    s = f"`{r['kernel'].replace('bz::', '')}` {r['avg_launch_us']:.1f} µs, {r['moved_bytes_per_launch'] / 1e9:.3f} GB → **{r['frac']:.2f}**"
    if r.get("traffic"):
        s += f"; PMC {r['traffic']:,} B per launch (ratio {r['wasted_traffic_ratio']:.4f})"
    return s


def med(d):
    return f" ({d['repeats']['value_median']:.0f})" if d.get("repeats") else ""


def main():
    rows = ("| workload (`profiles/r03_bench_*.json`, one box) | r03 it/s (median of repeats) | r02 | ms/step | dominant kernel: µs, moved GB, frac of 8 TB/s | iteration frac | CPU port |\n"
            "|---|---|---|---|---|---|---|\n")
    names = [("cfg2", "**cfg 2** n=10⁷ fp64 (headline)"), ("cfg2_driver_window", "cfg 2, the round driver's window (`--steps 20 --warmup 5`)"),
             ("cfg3", "cfg 3 2048² stencil fp64"), ("cfg3_two_loop", "cfg 3, two-loop kernels"),
             ("cfg4", "**cfg 4** 8192×65536 fp32 (one pass over A; short kernels as two launches)"),
             ("als", "**ALS** inner solve, [x; s] of 2·10⁷ (iterate-history form)"), ("cfg5_1gpu", "cfg 5 n=10⁸ on ONE GPU")]
    for key, title in names:
        d, o = L(f"r03_bench_{key}.json"), L(f"r02_bench_{key}.json")
        if d is None:
            continue
        cb = d.get("cpu_baseline") or {}
        cpu = f"{cb['value']:.2f} it/s ({cb.get('cores', 1)} thread{'s' if cb.get('cores', 1) > 1 else ''})" if cb.get("value") else ""
        if cb.get("all_cores"):
            cpu += f", {cb['all_cores']['value']:.1f} ({cb['all_cores']['cores']})"
        rows += (f"| {title} | **{d['value']:.0f}**{med(d)} | {o['value']:.0f} | {d['ms_per_step']:.4f} | {kern(d)} | "
                 f"{d['roofline_iteration']['frac']:.2f} | {cpu} |\n" if o else
                 f"| {title} | **{d['value']:.0f}**{med(d)} | — | {d['ms_per_step']:.4f} | {kern(d)} | {d['roofline_iteration']['frac']:.2f} | {cpu} |\n")
    c2 = L("r03_bench_cfg2.json")
    if c2 and c2.get("whole_alps"):
        wa = c2["whole_alps"]
        rows += (f"| cfg 2, whole `alps` | {wa['device_pointers']['value']:.0f} inner it/s with device pointers ({wa['device_pointers']['ms']:.1f} ms for "
                 f"{wa['device_pointers']['outer']} outer / {wa['device_pointers']['inner']} inner); {wa['host_pageable']['value']:.0f} from pageable host arrays "
                 f"({wa['host_pageable']['ms']:.1f} ms) | 5530 / 2564 | | | | |\n")
    if c2 and c2.get("two_loop"):
        rows += f"| cfg 2, two-loop recursion / inside outer iteration 3 | {c2['two_loop']['value']:.0f} / {c2['outer3']['value']:.0f} | 1989 / 6336 | | | | |\n"
    fam = ("\n| family (f-g-D), n = 10⁷ | r03 it/s | r02 it/s | kernel µs | kernel frac (r02) | iteration frac (r02) |\n|---|---|---|---|---|---|\n")
    for f in ("diag-l1box-box", "diag-nonneg-box", "diag-indbox-box", "diag-indboxvec-box", "diag-l1-boxvec", "diag-zero-boxveclo",
              "diag-nonneg-eitheror", "diag-zero-vc", "diag-l1-cc", "diag-l1-xor", "diag-l1-free", "diag-l1-zero"):
        d, o = L(f"r03_bench_family_{f}.json"), L(f"r02_bench_family_{f}.json")
        if d is None:
            continue
        rr = d["roofline"]
        extra = f", PMC ratio {rr['wasted_traffic_ratio']:.4f}" if rr.get("wasted_traffic_ratio") else ""
        fam += (f"| {f} | {d['value']:.0f} | {o['value']:.0f} | {rr['avg_launch_us']:.1f} | {rr['frac']:.2f} ({o['roofline']['frac']:.2f}){extra} | "
                f"{d['roofline_iteration']['frac']:.2f} ({o['roofline_iteration']['frac']:.2f}) |\n")
    p = os.path.join(ROOT, "DESIGN.md")
    s = open(p).read()
    a, b = s.index("<!-- r03 tables -->"), s.index("<!-- /r03 tables -->")
    s = s[:a] + "<!-- r03 tables -->\n" + rows + fam + s[b:]
    open(p, "w").write(s)


if __name__ == "__main__":
    main()

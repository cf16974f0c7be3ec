"""Rewrites the two measurement tables of DESIGN.md section 5 from the committed bench lines profiles/r02_bench_*.json
(development aid: python tools/design_tables.py)."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def L(name):
    return json.loads(open(os.path.join(ROOT, "profiles", name)).read().strip().splitlines()[-1])


def main():
    c2, c3, c3t = L("r02_bench_cfg2.json"), L("r02_bench_cfg3.json"), L("r02_bench_cfg3_two_loop.json")
    c4, c5, ca = L("r02_bench_cfg4.json"), L("r02_bench_cfg5_1gpu.json"), L("r02_bench_als.json")
    r = lambda d: d["roofline"]
    wa, k3 = c2["whole_alps"], c3["kernels"]
    rows = f'''| workload (`profiles/r02_bench_*.json`: the committed lines, all from one box; other boxes of the round in brackets) | it/s (median of repeats) | ms/step | dominant kernel: µs, moved GB, frac of 8 TB/s | iteration frac | CPU port |
|---|---|---|---|---|---|
| **cfg 2** n=10⁷ fp64 (headline) | **{c2['value']:.0f}** ({c2['repeats']['value_median']:.0f}) [6700–7420 box to box] | {c2['ms_per_step']:.4f} | `k_fused_compact<XR=2,UNI=2,NT=1,TRIAL=0>` {r(c2)['avg_launch_us']:.1f} µs [122–141], 0.720 GB → **{r(c2)['frac']:.2f}** [0.64–0.72]; PMC {r(c2)['traffic']:,} B per launch (ratio {r(c2)['wasted_traffic_ratio']:.4f}); the stream mix's own ceiling (below) is 0.75 | **{c2['roofline_iteration']['frac']:.2f}** [0.63–0.68] | {c2['cpu_baseline']['value']:.1f} it/s (1 thread), {c2['cpu_baseline']['all_cores']['value']:.1f} (16) |
| cfg 2, two-loop recursion (`two_loop`) | {c2['two_loop']['value']:.0f} [1940–2050] | {c2['two_loop']['ms_per_step']:.3f} | `k_twoloop_persist` + `k_fused_sep` (11 streams with the penalties as numbers) | — | |
| cfg 2 inside outer iteration 3 (`outer3`, y ≠ 0: 10 passes) | {c2['outer3']['value']:.0f} [6340–6700] | {c2['outer3']['ms_per_step']:.3f} | same kernel, UNI=1 | — | |
| cfg 2, whole `alps` (`whole_alps`) | {wa['device_pointers']['value']:.0f} inner it/s with device pointers ({wa['device_pointers']['ms']:.1f} ms for 13 outer / 180 inner) [5400–5860]; {wa['host_pageable']['value']:.0f} from pageable host arrays ({wa['host_pageable']['ms']:.1f} ms); `warm_start`: {wa['device_pointers_warm_start']['inner']} inner iterations in {wa['device_pointers_warm_start']['ms']:.1f} ms | | | | |
| cfg 3 2048² stencil fp64, compact form (default) | **{c3['value']:.0f}** ({c3['repeats']['value_median']:.0f}) [4100–4360] — start of the round: 3190–3270 | {c3['ms_per_step']:.4f} | `k_stencil_update_c<FULL=1,NT=1>` {r(c3)['avg_launch_us']:.1f} µs, 0.671 GB → {r(c3)['frac']:.2f} (PMC {r(c3)['traffic'] / 1e9:.4f} GB, ratio {r(c3)['wasted_traffic_ratio']:.3f}); `k_compact_xd<FULL=1,NT=1>` {k3['x_d']['avg_us']:.1f} µs → {k3['x_d']['moved_GBps'] / 8000:.2f} (partly out of the Infinity Cache); `k_stencil_fb<NT=1>` {k3['k_stencil_fb']['avg_us']:.1f} µs → {k3['k_stencil_fb']['moved_GBps'] / 8000:.2f} | **{c3['roofline_iteration']['frac']:.2f}** [0.68–0.71] | {c3['cpu_baseline']['value']:.1f} it/s (numpy) |
| cfg 3, two-loop kernels (`--two-loop`) | {c3t['value']:.0f} ({c3t['repeats']['value_median']:.0f}) [3620–3730] | {c3t['ms_per_step']:.3f} | `k_twoloop_persist<KR=16>` {r(c3t)['avg_launch_us']:.1f} µs → {r(c3t)['frac']:.2f} | {c3t['roofline_iteration']['frac']:.2f} | |
| cfg 4 8192×65536 fp32, affine images (default, `affine_refresh` = 16) | **{c4['value']:.0f}** [1098–1110; 979–1011 at refresh 8] (r01: 585–628) | {c4['ms_per_step']:.3f} | `k_gemv_n` {r(c4)['avg_launch_us']:.1f} µs per 2 GiB pass → {r(c4)['frac']:.2f} (PMC {r(c4)['traffic'] / 1e9:.4f} GB, ratio {r(c4)['wasted_traffic_ratio']:.4f}); `k_gemv_t_mfma` {r(c4)['other_gemv']['k_gemv_t_mfma']['avg_launch_us']:.1f} µs → {r(c4)['other_gemv']['k_gemv_t_mfma']['achieved'] / 8000:.2f} | {c4['roofline_iteration']['frac']:.2f} | {c4['cpu_baseline']['value']:.1f} it/s (numpy, BLAS 64 threads) |
| ALS inner solve (`--workload als`, SURVEY §8(f-3)): cfg 2 in slack form, inner vector [x; s] of 2·10⁷ | **{ca['value']:.0f}** (766 with the compact form as a kernel chain, 519 with the two-loop chain: the stages of this round) | {ca['ms_per_step']:.3f} | `k_fused_slack<NT=1>` {r(ca)['avg_launch_us']:.0f} µs, {r(ca)['moved_bytes_per_launch'] / 1e9:.2f} GB → {r(ca)['frac']:.2f}{(" (PMC ratio %.4f)" % r(ca)['wasted_traffic_ratio']) if r(ca).get('wasted_traffic_ratio') else ""}: the whole iteration in one pass, 17 passes over the lifted vector + 5 over n | {ca['roofline_iteration']['frac']:.2f} | {ca['cpu_baseline']['value']:.2f} it/s (numpy) |
| cfg 5 n=10⁸ on ONE GPU | {c5['value']:.0f} [693–787: the boxes that are slow at 10⁷ are the fast ones here] | {c5['ms_per_step']:.3f} | same kernel as cfg 2, {r(c5)['avg_launch_us']:.0f} µs, 7.2 GB → {r(c5)['frac']:.2f} [0.61–0.71]; the bare 8R + 1W pass at this size measured 0.69 on one of the slow ones | {c5['roofline_iteration']['frac']:.2f} | {c5['cpu_baseline']['value']:.2f} it/s (1 thread) |

'''
    prev = {'diag-l1box-box': '6760–7170', 'diag-nonneg-box': '7190–7700', 'diag-indbox-box': '6700–7240',
            'diag-indboxvec-box': '6060–6480', 'diag-l1-boxvec': '6120–6450', 'diag-zero-boxveclo': '6630–7140',
            'diag-nonneg-eitheror': '7010–7410', 'diag-zero-vc': '6590–6920', 'diag-l1-cc': '6630–6910',
            'diag-l1-xor': '5890–6260', 'diag-l1-free': '6940–7590', 'diag-l1-zero': '5690–6030'}
    first = {'diag-l1box-box': '6150–6260', 'diag-nonneg-box': '6180–6580', 'diag-indbox-box': '5770–6320',
             'diag-indboxvec-box': '5430–5720', 'diag-l1-boxvec': '5470–5760', 'diag-zero-boxveclo': '5940–6180',
             'diag-nonneg-eitheror': '5940–6710', 'diag-zero-vc': '5650–6340', 'diag-l1-cc': '5250–5920',
             'diag-l1-xor': '4780–5530', 'diag-l1-free': '6410–6730', 'diag-l1-zero': '5140'}
    fam = ('| family (f-g-D) | it/s [other boxes] | kernel µs | streams | kernel frac | iteration frac | first r02 collection |\n'
           '|---|---|---|---|---|---|---|\n')
    for f in prev:
        d = L(f"r02_bench_family_{f}.json")
        rr = d["roofline"]
        extra = f" (PMC ratio {rr['wasted_traffic_ratio']:.4f})" if rr.get("wasted_traffic_ratio") else ""
        fam += (f"| {f} | {d['value']:.0f} [{prev[f]}] | {rr['avg_launch_us']:.1f} | {rr['moved_bytes_per_launch'] // 80000000} | "
                f"{rr['frac']:.2f}{extra} | {d['roofline_iteration']['frac']:.2f} | {first[f]} |\n")
    fam += "\n"
    p = os.path.join(ROOT, "DESIGN.md")
    s = open(p).read()
    a, b = s.index("| workload (`profiles/r02_bench_*.json`"), s.index("What changed these numbers after the first r02 collection")
    s = s[:a] + rows + s[b:]
    a, b = s.index("| family (f-g-D) | it/s"), s.index("Every family but XOR is within 15 %")
    s = s[:a] + fam + s[b:]
    s = re.sub(r"headline kernel ran at [0-9.]+ µs = [0-9.]+ on the box the committed lines were taken on[^;]*; in brackets the other boxes\):",
               f"headline kernel ran at {r(c2)['avg_launch_us']:.1f} µs = {r(c2)['frac']:.2f} on the box the committed lines were taken on; in brackets the other boxes):", s)
    open(p, "w").write(s)


if __name__ == "__main__":
    main()

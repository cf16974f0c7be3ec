"""Turn rocprofv3 outputs under gpurun_out/ into the small summaries committed under profiles/.

    python tools/summarize_profiles.py --tag r02 --workload cfg2 --n 10000000 \
        --stats gpurun_out/prof_stats --fetch gpurun_out/prof_fetch --write gpurun_out/prof_write

HBM traffic per launch follows MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are collected in
SEPARATE --pmc passes (TCC slots), both are in KiB, and on gfx950 FETCH_SIZE reports exactly half
of the bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled:
    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
(k_muy is the in-run calibration on the element-wise workloads: it reads exactly two n-vectors and writes one.)

Besides the per-run files profiles/<tag>_kernel_stats_<workload>.csv and profiles/<tag>_pmc_<workload>.json, the
per-instantiation traffic goes into profiles/pmc_traffic.json, keyed the way bench.py looks it up: workload, n,
the template form the library reports for the launch (bz_profile_get2) and the hash of the kernel sources the
profile was collected on.
"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def counter_medians(d, name):
    files = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    # median: the first launches of a solve run with fewer L-BFGS pairs (m < 5) and move fewer bytes
    return {k: (sorted(v)[len(v) // 2], len(v)) for k, v in agg.items()}


def short(k):
    return k.replace("void ", "").split("(")[0]


def form_of(kernel):
    """rocprof kernel name -> the form string the library reports (bz_solver.hip: form_[cat])."""
    k = short(kernel).replace("bz::", "")
    m = re.match(r"k_fused_compact<(.*)>$", k)
    if m:
        a = [x.strip() for x in m.group(1).split(",")]
        a += ["false", "0", "0", "0", "35"][len(a) - 4:] if len(a) < 9 else []
        # <T, MM, NT, SPEC, OFF32, XR, UNI, TRIAL, FAM>
        nt, spec, xr, uni = a[2] == "true", a[3] == "true", a[5], a[6]
        trial = {"true": "1", "false": "0"}.get(a[7], a[7])
        fam = a[8] if len(a) > 8 else "35"
        if xr == "2":
            s = f"k_fused_compact<XR=2,UNI={uni},NT={int(nt)},TRIAL={trial}"
            # (the headline family's own instantiations carry no FAM tag: they are the ones with the default FAM = 35 and a
            # compile-time UNI that the host launches outside the family table)
            table = uni == "-1" or (len(a) > 8 and fam != "35")
            return s + (f",FAM={fam}>" if table else ">")
        if xr == "1":
            return f"k_fused_compact<XR=1,NT={int(nt)}>"
        return f"k_fused_compact<XR=0,SPEC={int(spec)},NT={int(nt)}>"
    m = re.match(r"(k_compact_xd|k_stencil_update_c)<\w+, \d+(?:, (true|false), (true|false))?(?:, (\d))?>$", k)
    if m:
        return f"{m.group(1)}<FULL={int(m.group(2) == 'true')},NT={int(m.group(3) == 'true')}" + (f",REGX={m.group(4)}>" if m.group(4) not in (None, "0") else ">")
    m = re.match(r"k_dense_fused<\w+, (\d+)>$", k)
    if m:
        return f"k_dense_fused<KP={m.group(1)}>"
    m = re.match(r"k_fused_slack_xr<\w+, \d+, (true|false), (true|false), (-?\d+)(?:, (\d+), (\d+))?>$", k)
    if m:      # (the fast instantiations — compile-time UNI >= 0 — carry a "(fast)" suffix, "(fast,l1-box)" with the kinds fixed too: bz_solver.hip)
        return f"k_fused_slack_xr<NT={int(m.group(1) == 'true')}>" + (("(fast,l1-box)" if m.group(4) == "1" else "(fast)") if int(m.group(3)) >= 0 else "")
    m = re.match(r"k_fused_slack<\w+, \d+, (true|false)>$", k)
    if m:
        return f"k_fused_slack<NT={int(m.group(1) == 'true')}>"
    m = re.match(r"k_stencil_fb<\w+(?:, (true|false))?>$", k)
    if m:
        return f"k_stencil_fb<NT={int(m.group(1) == 'true')}>"
    m = re.match(r"k_twoloop_persist<\w+, (\d+)>$", k)
    if m:
        return f"k_twoloop_persist<KR={m.group(1)}>"
    return re.sub(r"<.*>$", "", k)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r02")
    ap.add_argument("--workload", required=True)
    ap.add_argument("--n", type=int, required=True)
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--command", default="")
    a = ap.parse_args()
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    import bench
    sha = bench.lib_sources_sha()
    try:
        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:      # noqa: BLE001
        head = "?"
    if a.stats:
        stats = glob.glob(os.path.join(a.stats, "**", "*_kernel_stats.csv"), recursive=True)
        if stats:
            rows = list(csv.DictReader(open(stats[0])))
            with open(os.path.join(out_dir, f"{a.tag}_kernel_stats_{a.workload}.csv"), "w") as fh:
                fh.write(f"# rocprofv3 --kernel-trace --stats -- {a.command}   (lib sources {sha}, after {head})\n")
                fh.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
                for r in rows:
                    fh.write(",".join(['"%s"' % short(r["Name"])] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")]) + "\n")
    if not (a.fetch and a.write):
        return
    fetch = counter_medians(a.fetch, "FETCH_SIZE")
    write = counter_medians(a.write, "WRITE_SIZE")
    summary = {}
    for k in fetch:
        f, nf = fetch[k]
        w, nw = write.get(k, (0.0, 0))
        summary[short(k)] = {"form": form_of(k), "FETCH_SIZE_KiB_median": round(f, 1), "WRITE_SIZE_KiB_median": round(w, 1),
                             "launches_fetch_pass": nf, "launches_write_pass": nw,
                             "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
    with open(os.path.join(out_dir, f"{a.tag}_pmc_{a.workload}.json"), "w") as fh:
        json.dump({"command": f"rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (two separate passes) --kernel-trace -- {a.command}; "
                              "per kernel the MEDIAN over its launches",
                   "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request)",
                   "workload": a.workload, "n": a.n, "lib_sources_sha": sha, "collected": f"{a.tag}, after {head}",
                   "kernels": summary}, fh, indent=1)
    pmc = os.path.join(out_dir, "pmc_traffic.json")
    entries = json.load(open(pmc))["entries"] if os.path.exists(pmc) else []
    entries = [e for e in entries if not (e["workload"] == a.workload and e["n"] == a.n)]
    for k, v in summary.items():
        if "bz::" not in k:
            continue
        entries.append({"workload": a.workload, "n": a.n, "kernel": k, "form": v["form"],
                        "hbm_bytes_per_launch": v["hbm_bytes_per_launch"], "launches": v["launches_fetch_pass"],
                        "lib_sources_sha": sha, "collected": f"{a.tag}, after {head}"})
    with open(pmc, "w") as fh:
        json.dump({"how": "tools/summarize_profiles.py; bench.py attaches an entry to its roofline object only when workload, n, "
                          "form and lib_sources_sha all match the running build",
                   "entries": entries}, fh, indent=1)


if __name__ == "__main__":
    main()

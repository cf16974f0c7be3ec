"""Turn rocprofv3 outputs under gpurun_out/ into the small summaries committed under profiles/.

HBM traffic per launch follows MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are collected in
SEPARATE --pmc passes (TCC slots), both are in KiB, and on gfx950 FETCH_SIZE reports exactly half
of the bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled:
    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
(k_muy is the in-run calibration: it reads exactly two n-vectors and writes one.)
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter_avgs(d, name):
    files = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    # median: the first launches of a solve run with fewer L-BFGS pairs (m < 5) and move fewer bytes
    return {k: (sorted(v)[len(v) // 2], len(v)) for k, v in agg.items()}


def short(k):
    k = k.replace("void ", "")
    return k.split("(")[0]


def main(tag="r01"):
    go = os.path.join(ROOT, "gpurun_out")
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    stats = glob.glob(os.path.join(go, "prof_stats", "**", "*_kernel_stats.csv"), recursive=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(os.path.join(out_dir, f"{tag}_kernel_stats_bench_n1e7.csv"), "w") as fh:
            fh.write("# BZ_BENCH_PERIOD=1 rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline\n")
            fh.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
            for r in rows:
                fh.write(",".join(['"%s"' % short(r["Name"])] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")]) + "\n")
    fetch = counter_avgs(os.path.join(go, "prof_fetch"), "FETCH_SIZE")
    write = counter_avgs(os.path.join(go, "prof_write"), "WRITE_SIZE")
    summary = {}
    for k in fetch:
        f, nf = fetch[k]
        w, nw = write.get(k, (0.0, 0))
        summary[short(k)] = {"FETCH_SIZE_KiB_median": round(f, 1), "WRITE_SIZE_KiB_median": round(w, 1), "launches_fetch_pass": nf,
                             "launches_write_pass": nw, "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
    with open(os.path.join(out_dir, f"{tag}_pmc_hbm_traffic_n1e7.json"), "w") as fh:
        json.dump({"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 100 --warmup 20 "
                              "--no-cpu-baseline (two separate passes); per kernel the MEDIAN over its launches (the first 20 "
                              "iterations of a solve also store z: one more write pass)",
                   "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request)",
                   "n": 10_000_000, "kernels": summary}, fh, indent=1)
    # what bench.py looks up for roofline.traffic: measured HBM bytes per launch, by kernel
    if summary:
        with open(os.path.join(out_dir, "pmc_dominant_kernel.json"), "w") as fh:
            json.dump({"n": 10_000_000, "source": f"profiles/{tag}_pmc_hbm_traffic_n1e7.json",
                       # several instantiations of one kernel: the one that served most launches (the steady state)
                       # (ties: the instantiation that ran first — the headline problem's, before the side runs)
                       "kernels": {k.replace("bz::", "").split("<")[0]: v["hbm_bytes_per_launch"]
                                   for _, (k, v) in sorted(enumerate(summary.items()),
                                                           key=lambda t: (t[1][1]["launches_fetch_pass"], -t[0]))}}, fh, indent=1)
    # same-run agreement of the two clocks on the dominant kernel: bench.py's dispatch-bound HIP events
    # (its JSON line in prof_stats.log) against rocprofv3's kernel trace over the SAME launches (the timed
    # region = launches [warmup, warmup + steps) of the first problem the bench creates)
    log = os.path.join(go, "prof_stats.log")
    traces = glob.glob(os.path.join(go, "prof_stats", "**", "*_kernel_trace.csv"), recursive=True)
    if os.path.exists(log) and traces:
        line = [ln for ln in open(log) if ln.startswith("{")]
        if line:
            bj = json.loads(line[-1])
            kern = bj["roofline"]["kernel"].replace("bz::", "").split("<")[0]
            rows = [r for r in csv.DictReader(open(traces[0])) if kern in r["Kernel_Name"]]
            if bj["roofline"].get("kernel_form"):      # the timed category is the XR = 2 instantiations only
                import re
                rows = [r for r in rows if re.search(r"true, true, 2, \d(, (true|false))?>", r["Kernel_Name"])]
            rows.sort(key=lambda r: int(r["Start_Timestamp"]))
            w, k = bj["warmup"], bj["steps"]
            unprof = None
            plain = os.path.join(go, "bench_plain.json")      # bench.py run without the profiler in the same call
            if os.path.exists(plain):
                pl = [ln for ln in open(plain) if ln.startswith("{")]
                if pl:
                    unprof = json.loads(pl[-1])["roofline"]["avg_launch_us"]
            # the first launches of a solve (empty L-BFGS memory) go to other kernels, so the window is
            # taken from the end of the first problem's run: its last `steps` launches of this kernel
            first = rows[:w + k]
            if bj["roofline"].get("kernel_form"):
                # launches of this form inside the first problem's timed region: those that start after the
                # (w+1)-th iteration's launch and before the next problem's first kernel
                allk = [r for r in csv.DictReader(open(traces[0])) if kern in r["Kernel_Name"]]
                allk.sort(key=lambda r: int(r["Start_Timestamp"]))
                t_lo, t_hi = int(allk[w]["Start_Timestamp"]), int(allk[w + k - 1]["End_Timestamp"])
                first = [r for r in rows if t_lo <= int(r["Start_Timestamp"]) <= t_hi]
                k = len(first)
            dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in first[-k:]]
            with open(os.path.join(out_dir, f"{tag}_kernel_timing_agreement.json"), "w") as fh:
                json.dump({"kernel": bj["roofline"]["kernel"], "command": "BZ_BENCH_PERIOD=1 rocprofv3 --kernel-trace --stats -- python3 bench.py "
                           "--steps 100 --warmup 20 --no-cpu-baseline",
                           "bench_hip_events_avg_us": bj["roofline"]["avg_launch_us"],
                           "rocprofv3_trace_avg_us_same_launches": round(sum(dur) / len(dur) / 1e3, 3),
                           "launches": len(dur),
                           "bench_hip_events_avg_us_unprofiled_run_same_box": unprof,
                           "note": "the stats csv averages over every launch of the process (warm-up, the two-loop "
                                   "and outer-iteration-3 side runs included).  With the profiler attached the HIP "
                                   "events around a dispatch of this kernel read ~10 us longer than the trace of the "
                                   "same launches (not so without it: the plain bench run of the same gpurun call, same "
                                   "box, default arguments, is the figure to hold against the trace)"}, fh, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:])

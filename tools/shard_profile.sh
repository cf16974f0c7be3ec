#!/bin/bash
# The N = 8 shard of the headline problem on one GPU (n = 1.25e6: 90 MB, Infinity-Cache resident): kernel statistics and
# hardware counters of the one-pass kernel.  Separate rocprofv3 runs per counter group.
#   gpurun -- 'bash tools/shard_profile.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/prof_shard
rm -rf "$out"; mkdir -p "$out"
args="--size 1.25e6 --steps 400 --warmup 40 --no-cpu-baseline --no-extras"
export BZ_GATE=0
python3 bench.py $args > "$out/bench_plain.json" 2> "$out/bench_plain.err"; echo "plain rc=$?"
BZ_GATE=1 python3 bench.py $args > "$out/bench_gated.json" 2> "$out/bench_gated.err"; echo "gated rc=$?"
rocprofv3 --list-avail > "$out/list_avail.txt" 2>&1 || rocprofv3 -L > "$out/list_avail.txt" 2>&1
BZ_BENCH_PERIOD=1 rocprofv3 --kernel-trace --stats -d "$out/stats" -o p --output-format csv -- python3 bench.py $args > "$out/stats.log" 2>&1; echo "stats rc=$?"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d "$out/pmc$i" -o c --output-format csv -- python3 bench.py $args > "$out/pmc$i.log" 2>&1; echo "pmc$i ($grp) rc=$?"
done
find "$out" -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv, glob, collections, json, os
out = "gpurun_out/prof_shard"
res = {}
for d in sorted(glob.glob(out + "/pmc*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_fused_compact" in r["Kernel_Name"] and "2, 2" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            v.sort()
            res[k] = {"median": v[len(v) // 2], "launches": len(v)}
json.dump(res, open(out + "/counters_k_fused_compact.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
du -sh "$out"

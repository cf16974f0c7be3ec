#!/bin/bash
# Collect rocprofv3 kernel statistics and PMC traffic of one bench.py workload on the GPU box (three separate runs:
# --kernel-trace --stats ; --pmc FETCH_SIZE ; --pmc WRITE_SIZE — the TCC slots do not fit both counters in one pass).
#   usage: tools/collect_profiles.sh <tag> <bench args...>      e.g.  tools/collect_profiles.sh cfg3 --workload cfg3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
common="--steps 100 --warmup 20 --no-cpu-baseline --no-extras"
# (BZ_GATE=0: a pass launched early spends its gate wait inside the kernel, which a kernel trace would count as kernel time;
# bench.py's own HIP events are taken on the launches that are not gated)
export BZ_GATE=0
BZ_BENCH_PERIOD=1 rocprofv3 --kernel-trace --stats -d "$out/stats" -o p --output-format csv -- python3 bench.py $common "$@" > "$out/stats.log" 2>&1; echo "$tag stats rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$out/fetch" -o f --output-format csv -- python3 bench.py $common "$@" > "$out/fetch.log" 2>&1; echo "$tag fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$out/write" -o w --output-format csv -- python3 bench.py $common "$@" > "$out/write.log" 2>&1; echo "$tag write rc=$?"
# keep what the summariser needs, drop the bulky traces (gpurun_out is merged back, <= 64 MiB)
find "$out" -name "*kernel_trace.csv" -path "*stats*" -delete
du -sh "$out"

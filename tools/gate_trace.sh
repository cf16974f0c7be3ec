#!/bin/bash
# kernel trace of the headline run with and without the gated pre-launch (development aid)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for g in 0 1 2; do
  rm -rf gpurun_out/gate_trace$g
  BZ_GATE=$g BZ_BENCH_PERIOD=1000000 rocprofv3 --kernel-trace -d gpurun_out/gate_trace$g -o t --output-format csv -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras > gpurun_out/gate_trace$g.log 2>&1; echo "gate $g rc=$?"
done
python3 - <<'PY'
import csv, glob
for g in (0, 1, 2):
    f = glob.glob(f"gpurun_out/gate_trace{g}/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    fused = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_fused_compact" in r["Kernel_Name"] and "2, 2, 0" in r["Kernel_Name"]]
    col = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_collect_w" in r["Kernel_Name"]]
    fused = fused[len(fused) // 3: 2 * len(fused) // 3]
    dur = sorted(e - s for s, e in fused)
    per = sorted(fused[i + 1][0] - fused[i][0] for i in range(len(fused) - 1))
    endstart = sorted(fused[i + 1][0] - fused[i][1] for i in range(len(fused) - 1))
    endend = sorted(fused[i + 1][1] - fused[i][1] for i in range(len(fused) - 1))
    m = lambda v: v[len(v) // 2] / 1000.0
    print(f"gate {g}: fused duration median {m(dur):.1f} us, start-to-start {m(per):.1f}, end-to-next-start {m(endstart):.1f}, end-to-end {m(endend):.1f}, collect dur {m(sorted(e - s for s, e in col)):.1f}")
PY

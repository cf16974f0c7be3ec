# full GPU suite + the headline / side benches (development aid)
python -m pytest tests -x -q -m gpu > gpurun_out/r2b_pytest_full.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2b_pytest_full.log
python bench.py --no-cpu-baseline > gpurun_out/r2b_cfg2.json 2> gpurun_out/r2b_cfg2.err; echo "cfg2 rc=$?"
python bench.py --workload cfg3 --no-cpu-baseline > gpurun_out/r2b_cfg3.json 2>/dev/null; echo "cfg3 rc=$?"
python bench.py --workload cfg3 --two-loop --no-cpu-baseline > gpurun_out/r2b_cfg3tl.json 2>/dev/null; echo "cfg3 two-loop rc=$?"
python bench.py --workload cfg4 --no-cpu-baseline > gpurun_out/r2b_cfg4.json 2>/dev/null; echo "cfg4 rc=$?"
python tools/bench_print.py gpurun_out/r2b_cfg2.json gpurun_out/r2b_cfg3.json gpurun_out/r2b_cfg3tl.json gpurun_out/r2b_cfg4.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2b_cfg2.json").read().strip().splitlines()[-1])
for k in ("two_loop", "outer3", "whole_alps"):
    print(k, json.dumps(d.get(k) or d.get("extras", {}).get(k)))
PY

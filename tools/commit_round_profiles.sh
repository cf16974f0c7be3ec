#!/bin/bash
# after `gpurun -- 'bash tools/run_round_benches.sh r03'` (and `... r03 prof`): summaries + bench lines into profiles/
tag=${1:-r03}
cmd="python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras"
python tools/summarize_profiles.py --tag $tag --workload cfg2 --n 10000000 --stats gpurun_out/prof_cfg2/stats --fetch gpurun_out/prof_cfg2/fetch --write gpurun_out/prof_cfg2/write --command "$cmd"
python tools/summarize_profiles.py --tag $tag --workload cfg3 --n 4194304 --stats gpurun_out/prof_cfg3/stats --fetch gpurun_out/prof_cfg3/fetch --write gpurun_out/prof_cfg3/write --command "$cmd --workload cfg3"
python tools/summarize_profiles.py --tag $tag --workload cfg4 --n 65536 --stats gpurun_out/prof_cfg4/stats --fetch gpurun_out/prof_cfg4/fetch --write gpurun_out/prof_cfg4/write --command "$cmd --workload cfg4"
python tools/summarize_profiles.py --tag $tag --workload cfg2:diag-l1box-box --n 10000000 --stats gpurun_out/prof_fam_l1box/stats --fetch gpurun_out/prof_fam_l1box/fetch --write gpurun_out/prof_fam_l1box/write --command "$cmd --family diag-l1box-box"
python tools/summarize_profiles.py --tag $tag --workload cfg2:diag-l1-xor --n 10000000 --stats gpurun_out/prof_fam_xor/stats --fetch gpurun_out/prof_fam_xor/fetch --write gpurun_out/prof_fam_xor/write --command "$cmd --family diag-l1-xor"
python tools/summarize_profiles.py --tag $tag --workload als --n 10000000 --stats gpurun_out/prof_als/stats --fetch gpurun_out/prof_als/fetch --write gpurun_out/prof_als/write --command "$cmd --workload als"
for f in gpurun_out/$tag/bench_*.json; do cp $f profiles/${tag}_$(basename $f); done

#!/bin/bash
# All bench lines (+ profiles with "prof") of a round on the GPU box:
#   gpurun -- 'bash tools/run_round_benches.sh r03'          the bench lines
#   gpurun -- 'bash tools/run_round_benches.sh r03 prof'     the rocprofv3 kernel statistics + PMC traffic
tag=${1:-r03}
o=gpurun_out/$tag
mkdir -p $o
if [ "$2" = "prof" ]; then
  bash tools/collect_profiles.sh cfg2
  bash tools/collect_profiles.sh cfg3 --workload cfg3
  bash tools/collect_profiles.sh cfg4 --workload cfg4
  bash tools/collect_profiles.sh fam_l1box --family diag-l1box-box
  bash tools/collect_profiles.sh fam_xor --family diag-l1-xor
  bash tools/collect_profiles.sh als --workload als
  exit 0
fi
python bench.py > $o/bench_cfg2.json 2> $o/bench_cfg2.err; echo "cfg2 rc=$?"
# (the window the round driver runs)
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $o/bench_cfg2_driver_window.json 2> /dev/null; echo "cfg2 driver window rc=$?"
python bench.py --workload cfg3 > $o/bench_cfg3.json 2> $o/bench_cfg3.err; echo "cfg3 rc=$?"
python bench.py --workload cfg3 --two-loop --no-cpu-baseline > $o/bench_cfg3_two_loop.json 2> /dev/null; echo "cfg3 two-loop rc=$?"
python bench.py --workload cfg4 > $o/bench_cfg4.json 2> $o/bench_cfg4.err; echo "cfg4 rc=$?"
python bench.py --workload cfg5 --no-extras --steps 100 --warmup 20 > $o/bench_cfg5_1gpu.json 2> $o/bench_cfg5.err; echo "cfg5 rc=$?"
python bench.py --workload als > $o/bench_als.json 2> $o/bench_als.err; echo "als rc=$?"
for f in diag-l1box-box diag-nonneg-box diag-indbox-box diag-indboxvec-box diag-l1-boxvec diag-zero-boxveclo diag-l1-free diag-l1-zero diag-zero-vc diag-l1-cc diag-nonneg-eitheror diag-l1-xor; do
  python bench.py --family $f --no-extras > $o/bench_family_$f.json 2> /dev/null; echo "$f rc=$?"
done
python tools/bench_print.py $o/bench_*.json

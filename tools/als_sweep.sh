# ALS bench line under different grids / stream policies (development aid)
mkdir -p gpurun_out/r3c
for v in "BZ_GFC=1" "BZ_GFC=2" "BZ_GFC=4" "BZ_GFC=8" "BZ_NT=0"; do
  env $v timeout -k 10 200 python bench.py --workload als --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r3c/als_$v.json 2>/dev/null
  echo "$v: $(python tools/bench_print.py gpurun_out/r3c/als_$v.json)"
done

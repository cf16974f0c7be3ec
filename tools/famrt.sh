for i in 1 2 3; do
 python bench.py --no-cpu-baseline --no-extras > gpurun_out/famrt0_$i.json 2>/dev/null
 BZ_FAMRT=1 python bench.py --no-cpu-baseline --no-extras > gpurun_out/famrt1_$i.json 2>/dev/null
done
python tools/bench_print.py gpurun_out/famrt0_*.json gpurun_out/famrt1_*.json

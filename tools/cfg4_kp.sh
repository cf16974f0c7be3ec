# cfg4 with the one-pass dense kernel, and the same with the exchange compiled out (timing experiment) (development aid)
mkdir -p gpurun_out/r3b
timeout -k 10 200 python bench.py --workload cfg4 --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r3b/bench_cfg4.json 2> gpurun_out/r3b/bench_cfg4.err || exit 1
python tools/bench_print.py gpurun_out/r3b/bench_cfg4.json
BZ_TEST_DENSE_TIMEOUT=-2 timeout -k 10 200 python bench.py --workload cfg4 --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/r3b/noxchg.json 2> gpurun_out/r3b/noxchg.err; python tools/bench_print.py gpurun_out/r3b/noxchg.json

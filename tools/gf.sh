# gate fold / late gate on/off at several shard sizes (development aid)
for n in 10000000 1250000; do
for rep in 1 2; do
for k in 00 01 10 11; do
  BZ_GATEFOLD=${k:0:1} BZ_GATELATE=${k:1:1} python bench.py --n $n --no-cpu-baseline --no-extras > gpurun_out/gf_${n}_$k.json 2>gpurun_out/gf_err.log
  python -c "
import json
d=json.loads(open('gpurun_out/gf_${n}_$k.json').read().strip().splitlines()[-1])
print('n $n fold/late $k', d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['repeats']['value_median'])
"
done; done; done

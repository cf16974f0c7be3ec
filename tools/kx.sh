for k in 0 1 2 3 0 1; do
  BZ_KEEPX=$k python bench.py --no-cpu-baseline --no-extras > gpurun_out/kx_$k.json 2>/dev/null
  python -c "
import json
d=json.loads(open('gpurun_out/kx_$k.json').read().strip().splitlines()[-1])
print('keepx $k', d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['repeats']['value_median'])
"
done

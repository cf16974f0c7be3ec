# N ranks on ONE GPU through the driver's launch line (development aid; RCCL refuses two ranks per device: --no-rccl)
N=${1:-4}; SIZE=${2:-4e6}; shift 2 2>/dev/null      # further arguments go to bench.py (e.g. --workload cfg5)
port=$((20000 + RANDOM % 20000))
env BZ_BENCH_SAME_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $port bench.py --gpus $N --steps 40 --warmup 10 --size $SIZE --no-rccl "$@" 2> gpurun_out/rehearse_$N.err | grep '^{' > gpurun_out/rehearse_$N.json
python tools/bench_print.py gpurun_out/rehearse_$N.json
python -c "
import json; d=json.loads(open('gpurun_out/rehearse_$N.json').read()); print(d['n_gpus'], d['config']['scalar_transport'], d['config']['p2p_note'], d['solver'])"

mkdir -p gpurun_out/b
for cfg in "4 2" "8 2" "8 3" "8 4" "16 4" "8 6"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$1 MH_DECODE_CHAINS=$2 MH_DECODE_SELF_ROWS=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-extras --no-dit --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/b/bench_q$1_c$2.json
  python -c "import json;d=json.loads(open('gpurun_out/b/bench_q$1_c$2.json').read());print('queues=$1 chains=$2',d['value'],d['ms_per_step'])"
done

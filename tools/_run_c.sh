mkdir -p gpurun_out/c
timeout 900 python -m pytest tests/test_gpu_t5.py -x -q 2>&1 | tail -5
for cfg in "1" "2"; do
  MH_DECODE_SELF_ROWS=$cfg timeout 300 python bench.py --steps 3 --warmup 1 --no-extras --no-dit --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/c/bench_R$cfg.json
  python -c "import json;d=json.loads(open('gpurun_out/c/bench_R$cfg.json').read());print('R=$cfg',d['value'],d['ms_per_step'],d['aux'].get('stage_ms'))"
done
MH_DECODE_CHAINS=3 GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --steps 3 --warmup 1 --no-extras --no-dit --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/c/bench_c3.json
python -c "import json;d=json.loads(open('gpurun_out/c/bench_c3.json').read());print('3 chains',d['value'],d['ms_per_step'])"
MH_DECODE_CHAINS=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-extras --no-dit --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/c/bench_c1.json
python -c "import json;d=json.loads(open('gpurun_out/c/bench_c1.json').read());print('1 chain',d['value'],d['ms_per_step'])"

mkdir -p gpurun_out/g
timeout 60 tools/micro/gemv_probe_0 | tee gpurun_out/g/gemv_probe_lines2.txt
timeout 900 python -m pytest tests/test_gpu_t5.py -x -q 2>&1 | tail -5
for i in 1; do
  timeout 300 python bench.py --steps 3 --warmup 1 --no-extras --no-dit --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/g/bench.$i.json
  python -c "import json;d=json.loads(open('gpurun_out/g/bench.$i.json').read());print('bench',d['value'],d['ms_per_step'],d['aux'].get('stage_ms'))"
done
MH_DECODE_CHAINS=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-extras --no-dit --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/g/bench_c1.json
python -c "import json;d=json.loads(open('gpurun_out/g/bench_c1.json').read());print('1 chain',d['value'],d['ms_per_step'])"
MH_DECODE_CHAINS=3 GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --steps 3 --warmup 1 --no-extras --no-dit --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/g/bench_c3.json
python -c "import json;d=json.loads(open('gpurun_out/g/bench_c3.json').read());print('3 chains',d['value'],d['ms_per_step'])"

mkdir -p gpurun_out/d
timeout 60 tools/micro/gemv_probe_0 | tee gpurun_out/d/gemv_probe_staged.txt
timeout 900 python -m pytest tests/test_gpu_t5.py -x -q 2>&1 | tail -5
timeout 300 python bench.py --steps 3 --warmup 1 --no-extras --no-dit --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/d/bench.json
python -c "import json;d=json.loads(open('gpurun_out/d/bench.json').read());print('bench',d['value'],d['ms_per_step'],d['aux'].get('stage_ms'))"
MH_DECODE_CHAINS=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-extras --no-dit --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/d/bench_c1.json
python -c "import json;d=json.loads(open('gpurun_out/d/bench_c1.json').read());print('1 chain',d['value'],d['ms_per_step'])"

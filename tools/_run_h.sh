mkdir -p gpurun_out/h
timeout 900 python -m pytest tests/test_gpu_t5.py -x -q 2>&1 | tail -5
for pf in 1 0; do
  MH_DECODE_KV_PREFETCH=$pf timeout 300 python bench.py --steps 3 --warmup 1 --no-extras --no-dit --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/h/bench_pf$pf.json
  python -c "import json;d=json.loads(open('gpurun_out/h/bench_pf$pf.json').read());print('cross prefetch=$pf',d['value'],d['ms_per_step'],d['aux'].get('stage_ms'),d['roofline']['us_per_launch'])"
done

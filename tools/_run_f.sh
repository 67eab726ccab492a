mkdir -p gpurun_out/f
timeout 60 tools/micro/gemv_probe_0 | tee gpurun_out/f/gemv_probe_lines.txt
timeout 900 python -m pytest tests/test_gpu_t5.py -x -q 2>&1 | tail -5
for i in 1 2; do
for lib in libmapperhip.so libmapperhip_ab.so; do
  MAPPERHIP_LIB=$PWD/mapperatorinator_amd/lib/$lib timeout 300 python bench.py --steps 3 --warmup 1 --no-extras --no-dit --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/f/bench_$lib.$i.json
  python -c "import json;d=json.loads(open('gpurun_out/f/bench_$lib.$i.json').read());print('$lib',d['value'],d['ms_per_step'],d['aux'].get('stage_ms'))"
done
done

set -x
mkdir -p gpurun_out/a
timeout 600 python -m pytest tests/test_gpu_t5.py -x -q -k "variants or golden or teacher or invariant" 2>&1 | tail -15 > gpurun_out/a/tests.txt
cat gpurun_out/a/tests.txt
for R in 1 2 4; do
  MH_DECODE_SELF_ROWS=$R timeout 300 python bench.py --steps 3 --warmup 1 --no-extras --no-dit --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/a/bench_R$R.json
  python -c "import json;d=json.loads(open('gpurun_out/a/bench_R$R.json').read());print('R=$R',d['value'],d['ms_per_step'],d['aux'].get('stage_ms'))"
done
HIP_FORCE_DEV_KERNARG=1 MH_DECODE_SELF_ROWS=2 timeout 300 python bench.py --steps 3 --warmup 1 --no-extras --no-dit --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/a/bench_devkernarg.json
python -c "import json;d=json.loads(open('gpurun_out/a/bench_devkernarg.json').read());print('devkernarg R=2',d['value'],d['ms_per_step'])"
MH_DECODE_CHAINS=1 MH_DECODE_SELF_ROWS=4 timeout 300 python bench.py --steps 3 --warmup 1 --no-extras --no-dit --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/a/bench_1chain_R4.json
python -c "import json;d=json.loads(open('gpurun_out/a/bench_1chain_R4.json').read());print('1chain R=4',d['value'],d['ms_per_step'])"
MH_DECODE_CHAINS=3 MH_DECODE_SELF_ROWS=4 timeout 300 python bench.py --steps 3 --warmup 1 --no-extras --no-dit --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/a/bench_3chain_R4.json
python -c "import json;d=json.loads(open('gpurun_out/a/bench_3chain_R4.json').read());print('3chain R=4',d['value'],d['ms_per_step'])"

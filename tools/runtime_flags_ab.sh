#!/bin/bash
# A/B of HIP runtime flags against the decode step's graph replay (profiles/r05_graph_packet_capture.txt).
# Every run sits under its own `timeout`: a flag can hang the host's completion wait (ROC_SYSTEM_SCOPE_SIGNAL=0 does).
# usage (GPU box, repo root):  bash tools/runtime_flags_ab.sh ["FLAG=V FLAG2=V" ...]     (no argument: the recorded matrix)
cd "$(dirname "$0")/.."
Q="--steps 3 --warmup 1 --no-cpu-baseline --no-dit --no-extras --no-config5"
run() {
  echo -n "$1 :: "
  env $1 timeout 120 python bench.py $Q 2>/dev/null | python -c "
import json, sys
lines = sys.stdin.readlines()
if not lines: print('no result (timed out or failed)'); raise SystemExit
d = json.loads(lines[-1]); print(d['value'], d['ms_per_step'], d['aux']['stage_ms']['decode_ms'])"
}
if [ $# -gt 0 ]; then for f in "$@"; do run "$f"; done; exit 0; fi
C="DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"
run "$C"
run "$C MH_DECODE_CHAINS=3 GPU_MAX_HW_QUEUES=8"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 MH_DECODE_CHAINS=4 GPU_MAX_HW_QUEUES=8"
run "$C MH_DECODE_CHAINS=4 GPU_MAX_HW_QUEUES=8"
run "$C MH_DECODE_LAUNCH_THREADS=0"
run "$C HIP_FORCE_DEV_KERNARG=0"
run "$C HSA_ENABLE_INTERRUPT=0"

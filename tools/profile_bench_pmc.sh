# PMC passes (FETCH_SIZE / WRITE_SIZE, separate, --kernel-trace only) around the HEADLINE region of bench.py: one launcher thread
# (the TCC passes do not survive concurrent launcher threads), the timed loop only (--headline-only).  Each pass is bounded: a full
# default bench under a counter pass did not finish in 42 minutes (round 6).
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
export MH_DECODE_LAUNCH_THREADS=0
for attempt in 1 2 3 4; do     # (the TCC FETCH pass dumps core now and then: retry)
  rm -rf /tmp/p_f
  timeout 800 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_f -o f -- python $R/bench.py --headline-only --steps 1 --warmup 1 > $O/pmc_fetch.json 2> $O/pmc_f.err
  if [ -s $O/pmc_fetch.json ] && [ -f /tmp/p_f/f_results.db ]; then echo "FETCH pass ok on attempt $attempt"; break; fi
done
timeout 800 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_w -o w -- python $R/bench.py --headline-only --steps 1 --warmup 1 > $O/pmc_write.json 2> $O/pmc_w.err
ls -la /tmp/p_f /tmp/p_w
python $R/tools/rocpd_pmc.py /tmp/p_f/f_results.db /tmp/p_w/w_results.db $O/pmc_hbm_traffic.txt dec_cross_attn_q_kernel $O/pmc_cross_attn.json "bench.py --headline-only (config 2), two 16-row decode chains fed by one launcher thread (MH_DECODE_LAUNCH_THREADS=0)" 2 1903842816 > $O/pmc_summary.out 2>&1
tail -5 $O/pmc_summary.out

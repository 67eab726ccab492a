set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
# (1) kernel trace of the default bench command
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o bench -- python $R/bench.py > $O/bench_under_rocprof.json 2> $O/kt.err
python $R/tools/rocpd_stats.py /tmp/p_kt/bench_results.db $O/bench_kernel_stats.txt > /dev/null
# (2)(3) PMC passes: one launcher thread, headline + extras without the config-5 / CPU legs
export MH_DECODE_LAUNCH_THREADS=0
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_f -o f -- python $R/bench.py --no-cpu-baseline --no-config5 --no-runtime-ab > $O/pmc_fetch.json 2> $O/pmc_f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_w -o w -- python $R/bench.py --no-cpu-baseline --no-config5 --no-runtime-ab > $O/pmc_write.json 2> $O/pmc_w.err
ls -la /tmp/p_f /tmp/p_w
python $R/tools/rocpd_pmc.py /tmp/p_f/f_results.db /tmp/p_w/w_results.db $O/pmc_hbm_traffic.txt dec_cross_attn_q_kernel $O/pmc_cross_attn.json "bench.py config 2, two 16-row decode chains fed by one launcher thread (MH_DECODE_LAUNCH_THREADS=0)" 2 1903842816 > $O/pmc_summary.out 2>&1
tail -5 $O/pmc_summary.out

# rocprofv3 kernel trace of the DEFAULT bench command (what the driver runs), summarised on the box (the trace database is ~1 GB:
# 10 M dispatches, most of them from the config-5 long-song legs) -> gpurun_out/r06/bench_kernel_stats.txt + the bench line as the
# tracer saw it.  The PMC counter passes are tools/profile_bench_pmc.sh (the headline loop only: a counter pass around this whole
# command did not finish in 42 minutes, round 6).
#   gpurun --timeout 1500 -- 'bash tools/profile_bench.sh'
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o bench -- python $R/bench.py > $O/bench_under_rocprof.json 2> $O/kt.err
python $R/tools/rocpd_stats.py /tmp/p_kt/bench_results.db $O/bench_kernel_stats.txt > /dev/null
head -12 $O/bench_kernel_stats.txt

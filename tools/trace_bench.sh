# rocprofv3 kernel statistics of a short bench run: tools/trace_bench.sh <tag> [bench arguments]  -> gpurun_out/r3/trace_<tag>.csv
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/tb_$tag -o k -- python3 bench.py --steps 8 --warmup 2 --also none --no-cpu-baseline --no-roofline "$@" > /tmp/tb_$tag.log 2>&1
python3 tools/rocpd_stats.py /tmp/tb_$tag/k_results.db > gpurun_out/r3/trace_$tag.csv

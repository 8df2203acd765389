set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( while sleep 45; do echo "[heartbeat] $(date +%T)"; done ) &   # the profiler dumps its records for minutes without a word
HEART=$!
trap 'kill $HEART 2>/dev/null' EXIT
for c in ${CONFIGS:-c2 c3}; do
  A="--config $c --steps 4 --warmup 1 --also none --no-cpu-baseline --no-roofline"
  timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/kt_$c -o k -- python3 bench.py --config $c --steps 16 --warmup 2 --also none --no-cpu-baseline --no-roofline > gpurun_out/${TAG:-r3}_bench_${c}_under_rocprof.json 2>/tmp/kt_$c.log
  python3 tools/rocpd_stats.py /tmp/kt_$c/k_results.db > gpurun_out/${TAG:-r3}_bench_${c}_kernel_stats.csv
  export SG_BENCH_SYNC_STEPS=1   # counter passes serialise every dispatch: keep the queue short
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf_$c -o p -- python3 bench.py $A > /tmp/pf_$c.log 2>&1
  timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw_$c -o p -- python3 bench.py $A > /tmp/pw_$c.log 2>&1
  unset SG_BENCH_SYNC_STEPS
  python3 tools/summarize_traffic.py /tmp/pf_$c /tmp/pw_$c gpurun_out/${TAG:-r3}_traffic_$c.json > /tmp/st_$c.log
  head -5 /tmp/st_$c.log
done

# Evidence for the layer benchmark: its table, rocprofv3 kernel statistics, HBM counters and SQ counters.
# Run on the GPU box from the repository root; writes gpurun_out/${TAG:-r3}_layer1024*.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( while sleep 45; do echo "[heartbeat] $(date +%T)"; done ) &   # the profiler dumps its records for minutes without a word
HEART=$!
trap 'kill $HEART 2>/dev/null' EXIT
timeout -k 10 200 python3 -u tools/layer_bench_1024.py > gpurun_out/${TAG:-r3}_layer1024.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/lk -o k -- python3 tools/layer_bench_1024.py > /tmp/lk.log 2>&1
python3 tools/rocpd_stats.py /tmp/lk/k_results.db > gpurun_out/${TAG:-r3}_layer1024_kernel_stats.csv
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/lf -o p -- python3 tools/layer_bench_1024.py > /tmp/lf.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/lw -o p -- python3 tools/layer_bench_1024.py > /tmp/lw.log 2>&1
python3 tools/layer_traffic.py /tmp/lf /tmp/lw sg:: > gpurun_out/${TAG:-r3}_layer1024_hbm.json
timeout -s KILL 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d /tmp/ls -o p -- python3 tools/layer_bench_1024.py > /tmp/ls.log 2>&1
python3 tools/summarize_sq.py $(ls /tmp/ls/*/*counter_collection.csv /tmp/ls/*counter_collection.csv 2>/dev/null | head -1) sg:: > gpurun_out/${TAG:-r3}_layer1024_sq_counters.json
tail -9 gpurun_out/${TAG:-r3}_layer1024.txt
timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/lc -o p -- python3 tools/layer_bench_1024.py > /tmp/lc.log 2>&1
python3 tools/summarize_clock.py /tmp/lc sg:: > gpurun_out/${TAG:-r3}_layer1024_clock.json

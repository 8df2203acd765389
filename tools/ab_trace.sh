# kernel statistics of one conv_bench selection under two builds of the library (tools/_variants/lib_head.so and the tree's)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
for L in head new; do
  if [ $L = head ]; then export SG_OPS_LIBRARY=$PWD/tools/_variants/lib_head.so; else unset SG_OPS_LIBRARY; fi
  timeout -k 10 120 rocprofv3 --kernel-trace -d /tmp/ab_$L -o k -- python3 tools/conv_bench.py > /tmp/ab_$L.log 2>&1
  python3 tools/rocpd_stats.py /tmp/ab_$L/k_results.db | grep -i "sg::" > gpurun_out/r3/ab_$L.csv
done

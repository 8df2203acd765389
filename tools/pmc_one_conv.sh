set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export GEO=16,512,512,64 DTYPE=f32
for L in head new; do
  if [ $L = head ]; then export SG_OPS_LIBRARY=$PWD/tools/_variants/lib_head.so; else unset SG_OPS_LIBRARY; fi
  timeout -s KILL 90 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d /tmp/sq_$L -o p -- python3 tools/one_conv.py > /tmp/sq_$L.log 2>&1
  f=$(ls /tmp/sq_$L/*counter_collection.csv /tmp/sq_$L/*/*counter_collection.csv 2>/dev/null | head -1)
  cp $f gpurun_out/r3/sq_f32_$L.csv
  timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/sq2_$L -o p -- python3 tools/one_conv.py > /tmp/sq2_$L.log 2>&1
  f=$(ls /tmp/sq2_$L/*counter_collection.csv /tmp/sq2_$L/*/*counter_collection.csv 2>/dev/null | head -1)
  cp $f gpurun_out/r3/sq2_f32_$L.csv
done
echo done

#!/bin/bash
# Ablation builds of one kernel source: tools/build_variants.sh sg_band3x3 NOSTORE NOMFMA ...  -> tools/_variants/lib_<NAME>.so
set -e
cd "$(dirname "$0")/.."
src=$1; shift
objs=$(ls ga-gan_amd/build/*.o | grep -v "/$src.o")
for v in base "$@"; do
  def=""; [ "$v" != base ] && def=$(echo "$v" | tr '+' '\n' | sed 's/^/-DBAND_ABL_/' | tr '\n' ' ')
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize -I include -I ga-gan_amd/csrc $def -c ga-gan_amd/csrc/$src.hip -o tools/_variants/$src.$v.o &
done
wait
for v in base "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/_variants/$src.$v.o -o tools/_variants/lib_$v.so
done
ls -la tools/_variants/*.so

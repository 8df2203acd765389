#!/bin/bash
# Ablation builds of one kernel source: tools/build_variants.sh sg_band3x3 NOSTORE NOMFMA ...  -> tools/_variants/lib_<NAME>.so
# The switches (-DBAND_ABL_<NAME>: leave out loads, LDS commits, matrix instructions, stores; cycle stamps) are not part
# of the product sources: tools/ablation/switches.patch adds them to a scratch copy of ga-gan_amd/csrc first.
set -e
cd "$(dirname "$0")/.."
src=$1; shift
rm -rf tools/_variants/csrc && mkdir -p tools/_variants && cp -r ga-gan_amd/csrc tools/_variants/csrc
patch -p3 -d tools/_variants/csrc < tools/ablation/switches.patch > /dev/null
objs=$(ls ga-gan_amd/build/*.o | grep -v "/$src.o")
for v in base "$@"; do
  def=""; [ "$v" != base ] && def=$(echo "$v" | tr '+' '\n' | sed 's/^/-DBAND_ABL_/' | tr '\n' ' ')
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize -I include -I tools/_variants/csrc $def -c tools/_variants/csrc/$src.hip -o tools/_variants/$src.$v.o &
done
wait
for v in base "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/_variants/$src.$v.o -o tools/_variants/lib_$v.so
done
ls -la tools/_variants/*.so

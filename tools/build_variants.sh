#!/bin/bash
# Variant builds of libtbk.so for A/B timing (tools/time_variant.py): tools/exp/libtbk_<name>.so = the library with one
# translation unit rebuilt under extra -D flags.   bash tools/build_variants.sh <name> <file.hip> "<flags>" [...triples]
cd "$(dirname "$0")/.."
mkdir -p tools/exp
C=tbmodels_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I/opt/rocm/include -Wno-unused-function"
LD="-shared -L/opt/rocm/lib -lrocsolver -lrocblas -lrccl -Wl,-rpath,/opt/rocm/lib"
make -s -C $C -j8 > /dev/null
while [ $# -ge 3 ]; do
  name=$1; file=$2; defs=$3; shift 3
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c $C/$file -o tools/exp/${name}_${file%.hip}.o 2>/dev/null &&
    objs=$(ls $C/*.o | grep -v "\.exp\.o" | grep -v "/${file%.hip}.o") &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 $objs tools/exp/${name}_${file%.hip}.o $LD -o tools/exp/libtbk_$name.so 2>/dev/null && echo built $name ) &
done
wait

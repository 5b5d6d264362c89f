#!/bin/bash
# What the levers of the two-stage reduction's tile pass are worth, measured before anything is built on them
# (DESIGN.md 5.5, round 4).  Builds tools/band_time with the TIMING-ONLY ablation switches of csrc/tbk_eig_band.hip --
# results of those binaries are wrong by construction -- and runs them at 256 / 512 orbitals.
#   here:        bash tools/band_ablate.sh build
#   GPU box:     bash tools/band_ablate.sh run > gpurun_out/ablate.txt
cd "$(dirname "$0")/.."
mkdir -p tools/exp
VARIANTS=("" "-DTBK_ABLATE_OPERANDS" "-DTBK_ABLATE_BARRIER" "-DTBK_ABLATE_STORES" "-DTBK_ABLATE_STORES_ALT" "-DTBK_ABLATE_OPERANDS -DTBK_ABLATE_BARRIER"
          "-DTBK_ABLATE_OPERANDS -DTBK_ABLATE_BARRIER -DTBK_ABLATE_STORES")
NAMES=(base operands barrier stores stores_alt operands_barrier all_three)
if [ "$1" = build ]; then
  for i in "${!VARIANTS[@]}"; do
    hipcc -O3 -std=c++17 --offload-arch=gfx950 ${VARIANTS[$i]} -Iinclude -Itbmodels_amd/csrc tools/band_phase_clock.hip \
          -o tools/exp/band_${NAMES[$i]} -lrocblas 2>/dev/null &
  done
  wait
  ls tools/exp
else
  for name in "${NAMES[@]}"; do
    echo "== $name"
    TBK_BAND_FUSE=0 tools/exp/band_$name 256 8192 | tail -1
    tools/exp/band_$name 512 4096 | tail -1
  done
fi

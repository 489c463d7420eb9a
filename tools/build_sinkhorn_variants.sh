#!/bin/bash
# Diagnostic variants of the experiments build that differ in ONE object, sinkhorn_resident.o (fault hunt of DESIGN.md section 12):
#   ur-mvo_amd/liburf_front_v_<name>.so  for  name in  lgkm strongbar readback wcnt0 o1
# Usage: tools/build_sinkhorn_variants.sh      (after `make -C ur-mvo_amd/csrc experiments`)
set -e
cd "$(dirname "$0")/../ur-mvo_amd/csrc"
make experiments -j8 >/dev/null
BASE="-std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall -Wno-unused-function -Wno-pass-failed -DURF_EXPERIMENTS"
OTHERS=$(ls build_exp/*.o | grep -v sinkhorn_resident.o)
mkdir -p build_var
build() {  # name, flags
  /opt/rocm/bin/hipcc $BASE $2 -c sinkhorn_resident.hip -o build_var/sinkhorn_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../liburf_front_v_$1.so $OTHERS build_var/sinkhorn_$1.o -ldl
  echo "built liburf_front_v_$1.so ($2)"
}
build lgkm "-O3 -DURF_RS_LGKM_BARRIER" &
build strongbar "-O3 -DURF_RS_STRONG_BARRIER" &
build readback "-O3 -DURF_RS_READBACK" &
build wcnt0 "-O3 -mllvm -amdgpu-waitcnt-forcezero=1" &
build o1 "-O1" &
wait

#!/bin/bash
# serial (isolated) stage timings of the fast matcher with different kernel variants of the experiments build
#   tools/gpu_ab_serial.sh OUTDIR "name ENV=.. ENV=.." ...
out=$1; shift
mkdir -p "$out"
export URF_LIB=${AB_LIB:-$PWD/ur-mvo_amd/liburf_front_exp.so}
for spec in "$@"; do
  set -- $spec
  name=$1; shift
  env URF_PRECISION=1 "$@" timeout 300 python tools/gpu_perf.py > "$out/$name.txt" 2> "$out/$name.err"
  echo "== $name: $(grep '^PM\|TFLOP' "$out/$name.txt" | tr '\n' ' ' | cut -c1-400)"
done

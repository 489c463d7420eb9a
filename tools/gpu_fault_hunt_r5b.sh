#!/bin/bash
# Round 5, second pass: the root cause confirmed (the explicit lgkmcnt wait before the barriers of the shared form), the product soaked
# with and without its LDS padding, the column-marginal residual of clean runs.     tools/gpu_fault_hunt_r5b.sh OUTDIR [steps]
out=$1; steps=${2:-3000}
mkdir -p "$out"
D=$PWD/ur-mvo_amd
soak() {  # name lib size env...
  name=$1; lib=$2; size=$3; shift 3
  env URF_LIB=$D/$lib "$@" timeout 900 python tools/gpu_determinism.py $steps 3 $size > "$out/$name.txt" 2> "$out/$name.err"
  echo "== $name: $(tail -2 "$out/$name.txt" | tr '\n' ' ')"
}
soak shared_plain liburf_front_exp.so 1241x376 URF_SINKHORN_REGS=3
soak shared_lgkm  liburf_front_v_lgkm.so 1241x376 URF_SINKHORN_REGS=3
soak shared_lgkm2 liburf_front_v_lgkm.so 1241x376 URF_SINKHORN_REGS=1
soak product_1241 liburf_front.so 1241x376
soak product_640  liburf_front.so 640x480
soak wide_nopad_1241 liburf_front_exp.so 1241x376 URF_SINKHORN_WIDE_PAD=0
soak wide_nopad_640  liburf_front_exp.so 640x480 URF_SINKHORN_WIDE_PAD=0

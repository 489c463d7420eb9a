#!/bin/bash
# Round 5 fault hunt: soaks of the shared register-resident Sinkhorn (URF_SINKHORN_REGS=3) in the strict mode at
# 1241x376 with one thing changed at a time, the row-marginal residual of every handed-out pair logged, and the LDS hand-off probe.
#   tools/gpu_fault_hunt_r5.sh OUTDIR [steps]
out=$1; steps=${2:-3000}
mkdir -p "$out"
D=$PWD/ur-mvo_amd
soak() {  # name lib env...
  name=$1; lib=$2; shift 2
  env URF_LIB=$D/$lib "$@" timeout 600 python tools/gpu_determinism.py $steps 3 1241x376 > "$out/$name.txt" 2> "$out/$name.err"
  echo "== $name: $(grep -c 'pair' "$out/$name.txt") lines; $(tail -2 "$out/$name.txt" | tr '\n' ' ')"
}
soak base_off   liburf_front_exp.so URF_SINKHORN_REGS=3 URF_SOAK_RESID_BOUND=-1
soak base_check liburf_front_exp.so URF_SINKHORN_REGS=3
soak wide_nopad liburf_front_exp.so URF_SINKHORN_REGS=4 URF_SINKHORN_WIDE_PAD=0 URF_SOAK_RESID_BOUND=-1
soak gid        liburf_front_v_gid.so URF_SINKHORN_REGS=3 URF_SOAK_RESID_BOUND=-1
soak strongbar  liburf_front_v_strongbar.so URF_SINKHORN_REGS=3 URF_SOAK_RESID_BOUND=-1
soak readback   liburf_front_v_readback.so URF_SINKHORN_REGS=3 URF_SOAK_RESID_BOUND=-1
soak wcnt0      liburf_front_v_wcnt0.so URF_SINKHORN_REGS=3 URF_SOAK_RESID_BOUND=-1
soak o1         liburf_front_v_o1.so URF_SINKHORN_REGS=3 URF_SOAK_RESID_BOUND=-1
env URF_LIB=$D/liburf_front_exp.so timeout 300 python tools/gpu_lds_handoff.py 200 20 > "$out/handoff.txt" 2> "$out/handoff.err"; tail -1 "$out/handoff.txt"
env URF_LIB=$D/liburf_front_exp.so timeout 300 python tools/gpu_lds_handoff.py 200 5 1 > "$out/handoff_alone.txt" 2> "$out/handoff_alone.err"; tail -1 "$out/handoff_alone.txt"

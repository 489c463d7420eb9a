#!/bin/bash
# the software-pipelined attention kernel of the experiments build (URF_ATTN_IL=1: eight waves, 2: four waves) against the
# product's attn_h2_kernel (URF_ATTN_IL=0): same bits, then timings
#   tools/gpu_attn_il_check.sh OUTDIR "1 2 3"
out=${1:-gpurun_out/attn_il}; mkdir -p "$out"
variants=${2:-"1 2"}
export URF_LIB=$PWD/ur-mvo_amd/liburf_front_exp.so
URF_ATTN_IL=0 timeout 300 python tools/gpu_attn_il_check.py > "$out/check_0.txt" 2> "$out/check_0.err"
for v in $variants; do
  URF_ATTN_IL=$v timeout 300 python tools/gpu_attn_il_check.py > "$out/check_$v.txt" 2> "$out/check_$v.err"
  if diff "$out/check_0.txt" "$out/check_$v.txt" > "$out/check_diff_$v.txt"; then echo "variant $v: BIT-IDENTICAL ($(wc -l < "$out/check_$v.txt") lines)"; else echo "variant $v: DIFFERENT"; head -20 "$out/check_diff_$v.txt"; tail -3 "$out/check_$v.err"; fi
done
for v in 0 $variants 0 $variants; do
  URF_PRECISION=1 URF_ATTN_IL=$v timeout 300 python tools/gpu_perf.py > "$out/perf_il$v.txt" 2> "$out/perf_il$v.err"
  echo "il=$v: $(grep '^PM\|TFLOP' "$out/perf_il$v.txt" | tr '\n' ' ' | cut -c1-60,170-330)"
done

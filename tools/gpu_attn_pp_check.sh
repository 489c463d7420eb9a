#!/bin/bash
# the attention kernel's forms of the experiments build against the one-rhythm form (URF_ATTN_PP=0): same bits, then timings
#   tools/gpu_attn_pp_check.sh OUTDIR "1 2 3"
out=${1:-gpurun_out/attn_pp}; mkdir -p "$out"
variants=${2:-"2 3"}
export URF_LIB=$PWD/ur-mvo_amd/liburf_front_exp.so
URF_ATTN_PP=0 timeout 300 python tools/gpu_attn_pp_check.py > "$out/check_0.txt" 2> "$out/check_0.err"
for v in $variants; do
  URF_ATTN_PP=$v timeout 300 python tools/gpu_attn_pp_check.py > "$out/check_$v.txt" 2> "$out/check_$v.err"
  if diff "$out/check_0.txt" "$out/check_$v.txt" > "$out/check_diff_$v.txt"; then echo "variant $v: BIT-IDENTICAL ($(wc -l < "$out/check_$v.txt") lines)"; else echo "variant $v: DIFFERENT"; head -20 "$out/check_diff_$v.txt"; tail -3 "$out/check_$v.err"; fi
done
for v in 0 $variants 0 $variants; do
  URF_PRECISION=1 URF_ATTN_PP=$v timeout 300 python tools/gpu_perf.py > "$out/perf_pp$v.txt" 2> "$out/perf_pp$v.err"
  echo "pp=$v: $(grep '^PM\|TFLOP' "$out/perf_pp$v.txt" | tr '\n' ' ' | cut -c1-60,170-330)"
done

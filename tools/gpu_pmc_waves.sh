#!/bin/bash
# The wave-state PMC pass of tools/gpu_profile_round.sh on its own (strict, guarded, exact at 640x480):
#   gpurun --timeout 900 -- 'bash tools/gpu_pmc_waves.sh r05w r05'      -> gpurun_out/r05w/r05_pmc_waves*.json|txt
R=$PWD
OUT=$R/gpurun_out/${1:-pmc_waves}; TAG=${2:-rXX}
mkdir -p "$OUT"
export TMPDIR=/tmp
QUICK="--no-cpu-baseline --no-exact-check --no-secondary --no-guard-calibration"
run() {   # $1 = suffix, rest = bench arguments
  SUF=$1; shift
  D=$(mktemp -d /tmp/pmc_waves.XXXXXX)
  (cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d "$D" -o pmc -- python3 "$R/bench.py" --steps 5 --warmup 1 --repeats 1 $QUICK "$@" > "$OUT/pmc_waves$SUF.log" 2>&1)
  (cd "$R/tools" && python pmc_waves_summary.py "$(find "$D" -name '*.db' | head -1)" "$OUT/${TAG}_pmc_waves$SUF.json" "python3 bench.py --steps 5 --warmup 1 --repeats 1 $QUICK $*" "$R/profiles/${TAG}_pmc_mfma$SUF.json" > "$OUT/${TAG}_pmc_waves$SUF.txt")
  cat "$OUT/${TAG}_pmc_waves$SUF.txt"
}
run ""
run _guarded --precision 2
run _exact --precision 0

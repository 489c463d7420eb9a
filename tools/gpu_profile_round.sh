#!/bin/bash
# Everything profiles/rNN_* is made of, in one call on the GPU box (results under gpurun_out/$1/, copy them to profiles/):
#   gpurun --timeout 3000 -- 'bash tools/gpu_profile_round.sh r04p r04'
# PMC passes (FETCH_SIZE, WRITE_SIZE; two separate passes each, never combined with other traces; round 5: a third pass for the
# matrix pipe's busy cycles, tools/pmc_mfma_summary.py) + tools/pmc_summary.py for the
# strict-parity mode (the default of bench.py) on the 640x480 and the 1241x376 stream, the guarded fast and the exact mode; the
# kernel-trace statistics of the strict mode on both streams and of the exact mode; the bench lines.  Run it after the LAST edit
# of ur-mvo_amd/csrc: bench.py refuses a PMC summary whose source_sha differs from the kernel sources.
OUT=gpurun_out/${1:-prof}; TAG=${2:-rXX}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp
QUICK="--no-cpu-baseline --no-exact-check --no-secondary --no-guard-calibration"
pmc() {   # $1 = suffix of the summary file, rest = extra bench arguments
  SUF=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $c --kernel-trace -d $R/$OUT/pmc_$c -o pmc -- python3 $R/bench.py --steps 5 --warmup 1 --repeats 1 $QUICK "$@" > $R/$OUT/pmc_$c$SUF.log 2>&1
  done
  (cd $R && python tools/pmc_summary.py $(find $OUT/pmc_FETCH_SIZE -name "*.db" | head -1) $(find $OUT/pmc_WRITE_SIZE -name "*.db" | head -1) $OUT/${TAG}_pmc_hbm$SUF.json "python3 bench.py --steps 5 --warmup 1 --repeats 1 $QUICK $*")
  rm -rf $R/$OUT/pmc_FETCH_SIZE $R/$OUT/pmc_WRITE_SIZE
  cp $R/$OUT/${TAG}_pmc_hbm$SUF.json $R/profiles/ 2>/dev/null    # so that the bench lines below carry roofline.traffic
}
pmc_mfma() {   # the matrix pipe's busy cycles per kernel (one pass: SQ and GRBM counters share no slots); $1 = suffix
  SUF=$1; shift
  timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_F32 --kernel-trace -d $R/$OUT/pmc_mfma -o pmc -- python3 $R/bench.py --steps 5 --warmup 1 --repeats 1 $QUICK "$@" > $R/$OUT/pmc_mfma$SUF.log 2>&1
  (cd $R/tools && python pmc_mfma_summary.py $(find $R/$OUT/pmc_mfma -name "*.db" | head -1) $R/$OUT/${TAG}_pmc_mfma$SUF.json "python3 bench.py --steps 5 --warmup 1 --repeats 1 $QUICK $*")
  rm -rf $R/$OUT/pmc_mfma
  cp $R/$OUT/${TAG}_pmc_mfma$SUF.json $R/profiles/ 2>/dev/null
}
stats() {  # $1 = name, rest = bench arguments
  NAME=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_$NAME -o st -- python3 $R/bench.py --steps 20 --warmup 2 --repeats 1 $QUICK "$@" > $R/$OUT/stats_$NAME.log 2>&1
  (cd $R && cp $(find $OUT/stats_$NAME -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_${NAME}_kernel_stats.csv; rm -rf $OUT/stats_$NAME)
}
pmc ""
pmc _1241x376 --resolution 1241x376
pmc _guarded --precision 2
pmc _exact --precision 0
pmc_mfma ""
pmc_mfma _1241x376 --resolution 1241x376
pmc_mfma _guarded --precision 2
pmc_mfma _exact --precision 0
(cd $R && bash tools/gpu_pmc_waves.sh ${1:-prof} $TAG > $OUT/pmc_waves_all.log 2>&1)   # wave states, VALU beside MFMA, LDS conflicts (needs the ${TAG}_pmc_mfma*.json above in profiles/)
stats strict
# the same command with everything on ONE in-order stream: the serialised kernel durations that bench.py's roofline pass
# measures with HIP events (with three streams the tracer records every kernel's begin -> end while the others co-run)
URF_BENCH_OVERLAP=0 stats strict_serial
stats strict_1241x376 --resolution 1241x376
stats guarded --precision 2
stats exact --precision 0
cd $R
timeout 400 python bench.py > $OUT/${TAG}_bench_strict_640x480.json 2> $OUT/bench.err
timeout 300 python bench.py --resolution 1241x376 --no-cpu-baseline > $OUT/${TAG}_bench_strict_1241x376.json 2>> $OUT/bench.err
timeout 300 python bench.py --precision 2 --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_guarded_640x480.json 2>> $OUT/bench.err
timeout 300 python bench.py --precision 0 --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_exact_640x480.json 2>> $OUT/bench.err
python tools/gpu_timeline.py 3 60 2 2 > $OUT/${TAG}_timeline_strict.txt 2>&1
python tools/gpu_redo_bench.py > $OUT/${TAG}_redo_chain_alone.txt 2>&1
python tools/gpu_sp_layers.py > $OUT/${TAG}_superpoint_exact_layers.txt 2>&1
python tools/gpu_strict_margins.py 40 > $OUT/${TAG}_strict_margins.txt 2>&1
URF_LIB=$R/ur-mvo_amd/liburf_front_exp.so python tools/gpu_mfma_roof.py > $OUT/${TAG}_mfma_roof.txt 2>&1   # (experiments build: urf_probe_mfma_roof)
ls -la $OUT

#!/bin/bash
# Everything profiles/rNN_* is made of, in one call on the GPU box (results under gpurun_out/$1/, copy them to profiles/):
#   gpurun --timeout 2400 -- 'bash tools/gpu_profile_round.sh r02g r02'
# two PMC passes (FETCH_SIZE, WRITE_SIZE; never combined with other traces) + tools/pmc_summary.py, the kernel-trace statistics of
# the fast and the exact mode, and the three bench lines.  Run it after the LAST edit of ur-mvo_amd/csrc: bench.py refuses a PMC
# summary whose source_sha differs from the kernel sources.
OUT=gpurun_out/${1:-prof}; TAG=${2:-rXX}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp
PMCCMD="python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-exact-check"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace -d $R/$OUT/pmc_$c -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-exact-check > $R/$OUT/pmc_$c.log 2>&1
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_fast -o st -- python3 $R/bench.py --steps 15 --warmup 2 --repeats 1 --no-cpu-baseline --no-exact-check > $R/$OUT/stats_fast.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_exact -o st -- python3 $R/bench.py --precision 0 --steps 15 --warmup 2 --repeats 1 --no-cpu-baseline > $R/$OUT/stats_exact.log 2>&1
cd $R
python tools/pmc_summary.py $(find $OUT/pmc_FETCH_SIZE -name "*.db" | head -1) $(find $OUT/pmc_WRITE_SIZE -name "*.db" | head -1) $OUT/${TAG}_pmc_hbm.json "$PMCCMD"
cp $(find $OUT/stats_fast -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_fast_kernel_stats.csv
cp $(find $OUT/stats_exact -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_exact_kernel_stats.csv
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/stats_fast $OUT/stats_exact
cp $OUT/${TAG}_pmc_hbm.json profiles/ 2>/dev/null    # so that the bench lines below carry roofline.traffic
timeout 300 python bench.py > $OUT/${TAG}_bench_fast_640x480.json 2> $OUT/bench.err
timeout 300 python bench.py --resolution 1241x376 --no-cpu-baseline > $OUT/${TAG}_bench_fast_1241x376.json 2>> $OUT/bench.err
timeout 300 python bench.py --precision 0 --no-cpu-baseline > $OUT/${TAG}_bench_exact_640x480.json 2>> $OUT/bench.err
ls -la $OUT

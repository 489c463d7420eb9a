#!/bin/bash
# A/B runs of bench.py on the experiments build (URF_* knobs are only read by liburf_front_exp.so).
#   tools/gpu_ab.sh OUTDIR "name1 ENV=.. ENV=.." "name2 ..." ...      (extra bench flags: BENCH_FLAGS="--precision 2")
out=$1; shift
mkdir -p "$out"
export URF_LIB=${AB_LIB:-$PWD/ur-mvo_amd/liburf_front_exp.so}
for spec in "$@"; do
  set -- $spec
  name=$1; shift
  env "$@" python bench.py --steps ${STEPS:-50} --warmup 5 --repeats 3 --no-cpu-baseline --no-exact-check $BENCH_FLAGS > "$out/$name.json" 2> "$out/$name.err"
done
python - "$out" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["repeats"]["frames_per_s"], d["roofline"]["in_timed_region_ms_per_step"],
              d["near_tie_reruns"]["pairs"], d["near_tie_reruns"]["of_pairs"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-400:])
PY

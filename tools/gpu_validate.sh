#!/bin/bash
# The round's validation on ONE GPU box, as sub-commands (one script instead of a round-stamped copy per session):
#   gpurun --timeout 3000 -- 'bash tools/gpu_validate.sh TAG suite profile driver soak:4000 sweeps ab:_prev'
# TAG names the outputs (gpurun_out/TAG_*; copy what is to be judged into profiles/).  Steps, in the order given:
#   suite          the GPU test suite (pytest -m gpu)
#   profile        tools/gpu_profile_round.sh (PMC passes, kernel-trace statistics, bench lines of every mode)
#   driver         the driver's own bench command (python bench.py --gpus 1 --steps 20 --warmup 5)
#   soak:N         N-step determinism soaks: strict at both sizes, then guarded / fast (tools/gpu_determinism.py)
#   sweeps         randomised strict-vs-exact sweeps: bench weights (600 cases), 3x and 5x the residual gain (tools/gpu_sweep_strict.py)
#   ab:DIR         same-box A/B of bench.py against a second tree in the worktree DIR (built in the container:
#                  git worktree add -f DIR <commit> && make -C DIR/ur-mvo_amd/csrc -j8 && make -C DIR/oracle; DIR travels with the
#                  snapshot unless .gpurunignore lists it); optional ab:DIR:"3 2 1":2 = precisions and repeats
TAG=${1:-rXX}; shift
O=gpurun_out; mkdir -p $O
for step in "$@"; do
  case $step in
    suite)
      python -m pytest tests -q -m gpu > $O/${TAG}_pytest.txt 2>&1; grep -E "passed|failed|error" $O/${TAG}_pytest.txt | tail -1 ;;
    profile)
      bash tools/gpu_profile_round.sh ${TAG}p $TAG > $O/${TAG}_profile.log 2>&1; tail -1 $O/${TAG}_profile.log ;;
    driver)
      python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_driver_style.json 2> $O/${TAG}_bench_driver_style.err
      python - $O/${TAG}_bench_driver_style.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
s = j.get("secondary", {})
print("driver-style:", j["value"], j["repeats"]["frames_per_s"], "frac", j["roofline"]["frac"], "traffic", j["roofline"].get("traffic"),
      "native", s.get("native_frame_stream_strict_640x480", {}).get("frames_per_s"),
      "per-call", s.get("configs1_and_per_call_path_strict_parity_640x480", {}).get("matching_points_ms_per_pair"))
PY
      ;;
    soak:*)
      N=${step#soak:}
      python tools/gpu_determinism.py $N 3 > $O/${TAG}_soak_strict_$N.txt 2>&1; grep "precision 3:" $O/${TAG}_soak_strict_$N.txt | cut -c1-120
      python tools/gpu_determinism.py $N 3 1241x376 > $O/${TAG}_soak_strict_1241x376_$N.txt 2>&1; grep "precision 3:" $O/${TAG}_soak_strict_1241x376_$N.txt | cut -c1-120
      python tools/gpu_determinism.py $N 2,1 > $O/${TAG}_soak_guarded_fast_$N.txt 2>&1; grep "precision [12]:" $O/${TAG}_soak_guarded_fast_$N.txt | cut -c1-120 ;;
    sweeps)
      python tools/gpu_sweep_strict.py 600 > $O/${TAG}_sweep_strict_vs_exact_600.txt 2>&1; tail -1 $O/${TAG}_sweep_strict_vs_exact_600.txt
      python tools/gpu_sweep_strict.py 400 1 1 1.5 > $O/${TAG}_sweep_strict_vs_exact_gain3x_400.txt 2>&1; tail -2 $O/${TAG}_sweep_strict_vs_exact_gain3x_400.txt
      python tools/gpu_sweep_strict.py 400 2 2 2.5 > $O/${TAG}_sweep_strict_vs_exact_gain5x_400.txt 2>&1; tail -2 $O/${TAG}_sweep_strict_vs_exact_gain5x_400.txt ;;
    ab:*)
      IFS=: read -r _ DIR PRECS REPS <<< "$step"
      Q="--steps 60 --warmup 5 --repeats 3 --no-cpu-baseline --no-exact-check --no-secondary --no-guard-calibration"
      for rep in $(seq 1 ${REPS:-2}); do for prec in ${PRECS:-3 2 1}; do for tree in $DIR .; do
        (cd $tree && python bench.py $Q --precision $prec 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$tree', $prec, j['value'], j['repeats']['frames_per_s'])")
      done; done; done | tee $O/${TAG}_same_box_vs_$(basename $DIR).txt ;;
    *) echo "unknown step $step" ;;
  esac
done

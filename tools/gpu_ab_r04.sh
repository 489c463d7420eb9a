#!/bin/bash
# same-box A/B: round 4's tree (_r04/) against the current one, bench.py in three modes
Q="--steps 60 --warmup 5 --repeats 3 --no-cpu-baseline --no-exact-check --no-secondary --no-guard-calibration"
for rep in 1 2; do
for prec in 3 2 1; do
  for tree in _r04 .; do
    (cd $tree && python bench.py $Q --precision $prec 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$tree', $prec, j['value'], j['repeats']['frames_per_s'])")
  done
done
done

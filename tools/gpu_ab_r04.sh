#!/bin/bash
# Same-box A/B of the round against the previous round: round 4's tree in a worktree beside the current one, bench.py in the strict,
# guarded and fast modes, alternating (the boxes of the pool differ by +-2.5 %: only a same-box comparison shows a 1 - 3 % change).
#   git worktree add -f _r04 0a6345a && make -C _r04/ur-mvo_amd/csrc -j8 && make -C _r04/oracle    (once, in the build container;
#   _r04/ travels to the GPU box with the snapshot and is excluded from the history by .git/info/exclude)
#   gpurun --timeout 1200 -- "bash tools/gpu_ab_r04.sh"
Q="--steps 60 --warmup 5 --repeats 3 --no-cpu-baseline --no-exact-check --no-secondary --no-guard-calibration"
for rep in 1 2; do
for prec in 3 2 1; do
  for tree in _r04 .; do
    (cd $tree && python bench.py $Q --precision $prec 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$tree', $prec, j['value'], j['repeats']['frames_per_s'])")
  done
done
done

#!/bin/bash
# Same-box A/B of the current tree against a second tree in the worktree _pre/ (built in the container, travels with the snapshot,
# excluded from the history by .git/info/exclude):   git worktree add -f _pre <commit> && make -C _pre/ur-mvo_amd/csrc -j8
#   gpurun --timeout 1500 -- 'bash tools/gpu_ab_pre.sh "3 2 1" 2'        (precisions, repeats; bench.py alternating between the trees)
Q="--steps 60 --warmup 5 --repeats 3 --no-cpu-baseline --no-exact-check --no-secondary --no-guard-calibration"
for rep in $(seq 1 ${2:-2}); do
for prec in ${1:-3 2 1}; do
  for tree in _pre .; do
    (cd $tree && python bench.py $Q --precision $prec 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$tree', $prec, j['value'], j['repeats']['frames_per_s'])")
  done
done
done

X=$PWD/ur-mvo_amd/liburf_front_exp.so
Q="--steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-exact-check --no-secondary --no-guard-calibration"
run() { # name, env..., -- bench args
  name=$1; shift
  env URF_LIB=$X "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$name', j['value'], j['repeats']['frames_per_s'], j['near_tie_reruns']['pairs'], j['near_tie_reruns']['of_pairs'])"
}
for rep in 1 2; do
run "gain1.0 old-redo" URF_LINEAR_DMA=0 URF_ATTN_EXACT_NQT=4 python bench.py $Q --matcher-gain 1.0
run "gain1.0 policy" python bench.py $Q --matcher-gain 1.0
run "gain1.0 dma-always" URF_LINEAR_DMA=2 python bench.py $Q --matcher-gain 1.0
run "gain1.0 dma-always nqt2-always" URF_LINEAR_DMA=2 URF_ATTN_EXACT_NQT=2 python bench.py $Q --matcher-gain 1.0
run "exact old" URF_LINEAR_DMA=0 URF_ATTN_EXACT_NQT=4 python bench.py $Q --precision 0
run "exact dma-always" URF_LINEAR_DMA=2 python bench.py $Q --precision 0
run "exact dma3-always" URF_LINEAR_DMA=3 python bench.py $Q --precision 0
run "exact nqt2-always" URF_LINEAR_DMA=0 URF_ATTN_EXACT_NQT=2 python bench.py $Q --precision 0
done

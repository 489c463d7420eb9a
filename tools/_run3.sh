O=gpurun_out; mkdir -p $O
bash tools/gpu_validate.sh r06c driver
Q="--steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-exact-check --no-secondary --no-guard-calibration"
for m in 2 3 4 2 3 4; do
  URF_BENCH_MATCHERS=$m python bench.py $Q --resolution 1241x376 --batch-per-gpu 4 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('b4 1241 matchers=$m', j['value'], j['repeats']['frames_per_s'])"
done
for m in 2 3; do
  URF_BENCH_MATCHERS=$m python bench.py $Q 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('b8 640 matchers=$m', j['value'], j['repeats']['frames_per_s'])"
done
python tools/gpu_percall.py 3 40 > $O/r06c_percall_product.txt 2>&1; grep -E "precision|stage ms" $O/r06c_percall_product.txt
python tools/gpu_percall.py 2 40 > $O/r06c_percall_guarded.txt 2>&1; grep -E "precision|stage ms" $O/r06c_percall_guarded.txt

O=gpurun_out; mkdir -p $O
bash tools/gpu_validate.sh r06d suite
grep -E "FAILED|ERROR" $O/r06d_pytest.txt | head
Q="--steps 60 --warmup 5 --repeats 3 --no-cpu-baseline --no-exact-check --no-secondary --no-guard-calibration"
for rep in 1 2 3; do for m in 2 3; do
  URF_BENCH_MATCHERS=$m python bench.py $Q 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('b8 640 matchers=$m', j['value'], j['repeats']['frames_per_s'])"
done; done
URF_SOAK_AUDIT=4 python tools/gpu_determinism.py 2000 3 > $O/r06d_soak_strict_audit4_2000.txt 2>&1; grep -E "online guard|precision 3:" $O/r06d_soak_strict_audit4_2000.txt | cut -c1-400

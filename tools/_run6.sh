O=gpurun_out; mkdir -p $O
python -m pytest tests -q -m gpu -x -k "bit_exact_vs_oracle or dma_staged or superglue_n1000 or test_device_resident_pipeline_vs_oracle or small_grid or exact_mode or runs_in_the_exact_mode or split_f16" > $O/r06e_pytest.txt 2>&1; tail -3 $O/r06e_pytest.txt
Q="--steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-exact-check --no-secondary --no-guard-calibration"
for rep in 1 2; do
python bench.py $Q --precision 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('exact', j['value'], j['repeats']['frames_per_s'])"
python bench.py $Q --matcher-gain 1.0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('strict gain 1.0', j['value'], j['repeats']['frames_per_s'])"
python bench.py $Q 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('strict', j['value'], j['repeats']['frames_per_s'])"
done
python tools/gpu_redo_bench.py 2>&1 | grep -E "exact handle|strict handle"

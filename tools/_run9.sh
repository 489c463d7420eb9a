O=gpurun_out; mkdir -p $O
Q="--steps 60 --warmup 5 --repeats 3 --no-cpu-baseline --no-exact-check --no-secondary"
python bench.py $Q 2>$O/r06g_strict.err | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('strict 640', j['value'], j['repeats']['frames_per_s'], j['near_tie_reruns']['pairs'], j['near_tie_reruns']['of_pairs'])"
python bench.py $Q --resolution 1241x376 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('strict 1241', j['value'], j['repeats']['frames_per_s'], j['near_tie_reruns']['pairs'], j['near_tie_reruns']['of_pairs'])"
python bench.py $Q --matcher-gain 1.0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('strict 640 gain 1.0', j['value'], j['repeats']['frames_per_s'], j['near_tie_reruns']['pairs'], j['near_tie_reruns']['of_pairs'])"
grep "calibrated" $O/r06g_strict.err | head -3
python tools/gpu_strict_margins.py 40 > $O/r06g_strict_margins.txt 2>&1; tail -12 $O/r06g_strict_margins.txt | cut -c1-250
URF_SWEEP_AUDIT=16 python tools/gpu_sweep_strict.py 400 1 1 1.0 > $O/r06_sweep_strict_vs_exact_gain2x_400.txt 2>&1; tail -3 $O/r06_sweep_strict_vs_exact_gain2x_400.txt
URF_SWEEP_AUDIT=16 python tools/gpu_sweep_strict.py 400 1 1 1.5 > $O/r06_sweep_strict_vs_exact_gain3x_400.txt 2>&1; tail -3 $O/r06_sweep_strict_vs_exact_gain3x_400.txt

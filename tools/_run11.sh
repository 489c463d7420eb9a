O=gpurun_out; mkdir -p $O
python tools/gpu_determinism.py 12000 3 > $O/r06_soak_strict_12000.txt 2>&1; grep -E "online guard|precision 3:" $O/r06_soak_strict_12000.txt | cut -c1-330
URF_SWEEP_AUDIT=8 python tools/gpu_sweep_strict.py 1500 11 > $O/r06_sweep_strict_vs_exact_1500.txt 2>&1; tail -3 $O/r06_sweep_strict_vs_exact_1500.txt | cut -c1-330
URF_SWEEP_AUDIT=8 python tools/gpu_sweep_strict.py 1000 3 3 1.0 > $O/r06_sweep_strict_vs_exact_gain2x_1000.txt 2>&1; tail -3 $O/r06_sweep_strict_vs_exact_gain2x_1000.txt | cut -c1-330
timeout 900 python tools/gpu_sweep.py 60 21 > $O/r06_sweep_exact_vs_oracle_60.txt 2>&1; tail -2 $O/r06_sweep_exact_vs_oracle_60.txt | cut -c1-300
timeout 600 python tools/gpu_sweep_fast.py 60 22 > $O/r06_sweep_fast_vs_exact_60.txt 2>&1; tail -3 $O/r06_sweep_fast_vs_exact_60.txt | cut -c1-300

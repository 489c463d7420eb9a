O=gpurun_out; mkdir -p $O
python -m pytest tests -q -m gpu -x -k "runs_in_the_exact_mode or calibrates_itself or share_a_redo or two_begun or guarded_mode_redoes" > $O/r06f_pytest.txt 2>&1; tail -2 $O/r06f_pytest.txt
URF_SWEEP_AUDIT=16 python tools/gpu_sweep_strict.py 600 > $O/r06_sweep_strict_vs_exact_600.txt 2>&1; tail -2 $O/r06_sweep_strict_vs_exact_600.txt
URF_SWEEP_AUDIT=16 python tools/gpu_sweep_strict.py 400 1 1 1.0 > $O/r06_sweep_strict_vs_exact_gain2x_400.txt 2>&1; tail -3 $O/r06_sweep_strict_vs_exact_gain2x_400.txt
URF_SWEEP_AUDIT=16 python tools/gpu_sweep_strict.py 400 1 1 1.5 > $O/r06_sweep_strict_vs_exact_gain3x_400.txt 2>&1; tail -3 $O/r06_sweep_strict_vs_exact_gain3x_400.txt
URF_SWEEP_AUDIT=16 python tools/gpu_sweep_strict.py 400 2 2 2.5 > $O/r06_sweep_strict_vs_exact_gain5x_400.txt 2>&1; tail -3 $O/r06_sweep_strict_vs_exact_gain5x_400.txt

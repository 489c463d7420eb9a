#!/bin/bash
# Round 5, second session, the long checks at the final sources (after tools/gpu_final_validation_r5b.sh):
#   gpurun --timeout 3000 -- 'bash tools/gpu_final_validation_r5c.sh'
# 12000-step soak of the strict mode and 6000 steps of the guarded / fast modes (both sizes), the strict-vs-exact sweeps on other
# weights (three and five times the residual gain), and the same-box comparison with round 4's tree (tools/gpu_ab_r04.sh; needs _r04/).
python tools/gpu_determinism.py 12000 3 > gpurun_out/r05_soak_strict_12000.txt 2>&1; grep "precision 3:" gpurun_out/r05_soak_strict_12000.txt | cut -c1-120
python tools/gpu_determinism.py 6000 2,1 > gpurun_out/r05_soak_guarded_fast_6000.txt 2>&1; grep "precision [12]:" gpurun_out/r05_soak_guarded_fast_6000.txt | cut -c1-120
python tools/gpu_sweep_strict.py 400 1 1 1.5 > gpurun_out/r05_sweep_strict_vs_exact_other_weights_400.txt 2>&1; tail -2 gpurun_out/r05_sweep_strict_vs_exact_other_weights_400.txt
python tools/gpu_sweep_strict.py 600 2 2 2.5 > gpurun_out/r05_sweep_strict_vs_exact_gain5x_600.txt 2>&1; tail -2 gpurun_out/r05_sweep_strict_vs_exact_gain5x_600.txt
bash tools/gpu_ab_r04.sh > gpurun_out/r05_same_box_vs_round4.txt 2>&1; cat gpurun_out/r05_same_box_vs_round4.txt

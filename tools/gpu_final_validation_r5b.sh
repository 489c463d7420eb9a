#!/bin/bash
# Round 5, second session: the final sources once more on one box --   gpurun --timeout 3000 -- 'bash tools/gpu_final_validation_r5b.sh'
# the GPU suite, tools/gpu_profile_round.sh (HBM and matrix-pipe PMC passes, kernel-trace statistics, bench lines of every mode),
# the driver's own bench command, 4000-step soaks (strict at both sizes, guarded / fast) and a 600-case strict-vs-exact sweep.
# (The 12000-step soaks, the other-weights sweeps and the same-box comparison with round 4's tree are tools/gpu_final_validation_r5.sh,
# run on the sources of the first session: the kernels of the product path have not changed since.)
python -m pytest tests -q -m gpu > gpurun_out/r5z_pytest.txt 2>&1; grep -E "passed|failed" gpurun_out/r5z_pytest.txt | tail -1
bash tools/gpu_profile_round.sh r05z r05 > gpurun_out/r05z.log 2>&1; tail -1 gpurun_out/r05z.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_style.json 2> gpurun_out/r05_bench_driver_style.err
python -c "import json; j=json.load(open(\"gpurun_out/r05_bench_driver_style.json\")); print(j[\"value\"], j[\"repeats\"][\"frames_per_s\"], j[\"roofline\"][\"traffic\"], j[\"roofline\"][\"mfma_busy_counter\"], j[\"secondary\"][\"native_frame_stream_strict_640x480\"][\"frames_per_s\"])"
python tools/gpu_determinism.py 4000 3 > gpurun_out/r05_soak_strict_4000.txt 2>&1; grep "precision 3:" gpurun_out/r05_soak_strict_4000.txt | cut -c1-120
python tools/gpu_determinism.py 4000 2,1 > gpurun_out/r05_soak_guarded_fast_4000.txt 2>&1; grep "precision [12]:" gpurun_out/r05_soak_guarded_fast_4000.txt | cut -c1-120
python tools/gpu_sweep_strict.py 600 > gpurun_out/r05_sweep_strict_vs_exact_600.txt 2>&1; tail -1 gpurun_out/r05_sweep_strict_vs_exact_600.txt

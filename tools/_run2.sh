O=gpurun_out; mkdir -p $O
python -m pytest tests -q -m gpu -x -k "small_grid or dma_staged or calibrates_itself or full_depth or guarded_mode_redoes or two_begun or share_a_redo" > $O/r06b_pytest.txt 2>&1; tail -3 $O/r06b_pytest.txt
python tools/gpu_percall.py 3 40 > $O/r06b_percall_product.txt 2>&1; grep -E "precision|stage ms" $O/r06b_percall_product.txt
X=$PWD/ur-mvo_amd/liburf_front_exp.so
URF_LIB=$X URF_H2GEMM_DEEP=0 URF_ATTN_VARIANT=0 python tools/gpu_percall.py 3 40 > $O/r06b_percall_old.txt 2>&1; grep -E "precision|stage ms" $O/r06b_percall_old.txt
URF_LIB=$X URF_H2GEMM_DEEP=1 URF_ATTN_VARIANT=0 python tools/gpu_percall.py 3 40 > $O/r06b_percall_deep_only.txt 2>&1; grep -E "precision|stage ms" $O/r06b_percall_deep_only.txt
URF_LIB=$X URF_H2GEMM_DEEP=3 URF_ATTN_VARIANT=3 python tools/gpu_percall.py 3 40 > $O/r06b_percall_deep3_attn3.txt 2>&1; grep -E "precision|stage ms" $O/r06b_percall_deep3_attn3.txt
URF_LIB=$X URF_H2GEMM_DEEP=6 URF_ATTN_VARIANT=3 python tools/gpu_percall.py 3 40 > $O/r06b_percall_deep6_attn3.txt 2>&1; grep -E "precision|stage ms" $O/r06b_percall_deep6_attn3.txt
python tools/gpu_redo_bench.py > $O/r06b_redo_chain.txt 2>&1; grep -E "exact handle|strict handle" $O/r06b_redo_chain.txt
URF_LIB=$X URF_LINEAR_DMA=0 URF_ATTN_EXACT_NQT=4 python tools/gpu_redo_bench.py > $O/r06b_redo_chain_old.txt 2>&1; grep -E "exact handle|strict handle" $O/r06b_redo_chain_old.txt
URF_LIB=$X URF_LINEAR_DMA=3 URF_ATTN_EXACT_NQT=4 python tools/gpu_redo_bench.py > $O/r06b_redo_chain_dma3.txt 2>&1; grep -E "exact handle|strict handle" $O/r06b_redo_chain_dma3.txt
URF_LIB=$X URF_LINEAR_DMA=0 URF_ATTN_EXACT_NQT=2 python tools/gpu_redo_bench.py > $O/r06b_redo_chain_nqt2.txt 2>&1; grep -E "exact handle|strict handle" $O/r06b_redo_chain_nqt2.txt
URF_LIB=$X URF_SINKHORN_REGS=3 python tools/gpu_percall.py 3 40 > $O/r06b_percall_regs3.txt 2>&1; grep -E "precision|stage ms" $O/r06b_percall_regs3.txt
URF_LIB=$X URF_SINKHORN_REGS=3 URF_SINKHORN_NEAR=1 python tools/gpu_percall.py 3 40 > $O/r06b_percall_regs3_near.txt 2>&1; grep -E "precision|stage ms" $O/r06b_percall_regs3_near.txt
URF_LIB=$X URF_SINKHORN_NEAR=1 python tools/gpu_percall.py 3 40 > $O/r06b_percall_wide_near.txt 2>&1; grep -E "precision|stage ms" $O/r06b_percall_wide_near.txt
Q="--steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-exact-check --no-secondary --no-guard-calibration"
for d in 0 1 6 3 0 1; do
  URF_LIB=$X URF_H2GEMM_DEEP=$d python bench.py $Q --resolution 1241x376 --batch-per-gpu 4 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('b4 1241 deep=$d', j['value'], j['repeats']['frames_per_s'])"
done
for v in "0 4 0" "1 0 1" "0 4 0" "1 0 1" "3 0 1" "0 2 0"; do
  set -- $v
  URF_LIB=$X URF_LINEAR_DMA=$1 URF_ATTN_EXACT_NQT=$2 URF_H2GEMM_DEEP=$3 python bench.py $Q 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('b8 640 strict dma=$1 nqt=$2 deep=$3', j['value'], j['repeats']['frames_per_s'])"
done
python bench.py $Q --resolution 1241x376 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('b8 1241 product', j['value'], j['repeats']['frames_per_s'])"

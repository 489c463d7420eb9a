X=$PWD/ur-mvo_amd/liburf_front_exp.so
Q="--steps 60 --warmup 5 --repeats 3 --no-cpu-baseline --no-exact-check --no-secondary --no-guard-calibration"
for rep in 1 2; do for xf in 0 4 2; do
  URF_LIB=$X URF_REDO_OFF=1 URF_H2GEMM_XFLAGS=$xf python bench.py $Q 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['per_rank'][0]; a=j['roofline']['all_kernels']
print('no-redo xflags=$xf', j['value'], j['ms_per_step'], 'SP in situ', r.get('superpoint_ms'), 'serial ms:', {k.split('(')[0].strip()[:28]: v['ms_per_step'] for k,v in a.items()})"
done; done

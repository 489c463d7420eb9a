X=$PWD/ur-mvo_amd/liburf_front_exp.so
Q="--steps 60 --warmup 5 --repeats 3 --no-cpu-baseline --no-exact-check --no-secondary --no-guard-calibration"
for rep in 1 2; do for off in 0 1; do
  URF_LIB=$X URF_REDO_OFF=$off python bench.py $Q 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['per_rank'][0]
print('redo_off=$off', j['value'], j['ms_per_step'], j['repeats']['frames_per_s'], 'redone', j['near_tie_reruns']['pairs'], 'flagged-not-redone', j['near_tie_reruns']['pairs_flagged_not_redone'], 'SP in situ', r.get('superpoint_ms'))"
done; done

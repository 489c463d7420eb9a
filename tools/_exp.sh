URF_PRECISION=1 timeout 200 python tools/gpu_perf.py 2>&1 | grep -E "^SP" | cut -c1-330
timeout 600 python -m pytest tests -m gpu -q --timeout 300 -k "superpoint or pipeline or fast" 2>&1 | tail -3
for i in 1 2; do timeout 200 python bench.py --repeats 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('bench:', d['value'], d['ms_per_step'], d['repeats']['frames_per_s'], d['stage_ms_per_step'], d['exact_mode']['pairs_with_identical_match_list'])"; done
timeout 200 python bench.py --repeats 3 --no-cpu-baseline --resolution 1241x376 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('kitti:', d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['exact_mode']['pairs_with_identical_match_list'])"

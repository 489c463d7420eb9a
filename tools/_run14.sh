X=$PWD/ur-mvo_amd/liburf_front_exp.so
Q="--steps 60 --warmup 5 --repeats 3 --no-cpu-baseline --no-exact-check --no-secondary --no-guard-calibration"
for rep in 1 2; do for pr in 0 3 1; do
  URF_LIB=$X URF_CONV_PRIO=$pr python bench.py $Q 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('strict 640 conv prio=$pr', j['value'], j['repeats']['frames_per_s'])"
done; done
URF_LIB=$X URF_CONV_PRIO=3 python bench.py $Q --resolution 1241x376 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('strict 1241 conv prio=3', j['value'], j['repeats']['frames_per_s'])"
URF_LIB=$X URF_CONV_PRIO=0 python bench.py $Q --resolution 1241x376 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('strict 1241 conv prio=0', j['value'], j['repeats']['frames_per_s'])"

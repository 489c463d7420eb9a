bash tools/gpu_validate.sh r06a suite driver
R=$PWD; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06a_percall -o st -- python3 $R/tools/gpu_percall.py 3 40 > $R/gpurun_out/r06a_percall.log 2>&1
cd $R; cp $(find gpurun_out/r06a_percall -name "*kernel_stats.csv" | head -1) gpurun_out/r06a_percall_kernel_stats.csv; rm -rf gpurun_out/r06a_percall
tail -3 gpurun_out/r06a_percall.log

X=$PWD/ur-mvo_amd/liburf_front_exp.so
for pad in 0 1024 0 1024; do
  echo "== conv LDS pad $pad"
  URF_LIB=$X URF_CONV_LDS_PAD=$pad python tools/gpu_sp_layers.py 2>&1 | grep -E "conv1a|conv2a|conv3b|convPa|^sum"
done

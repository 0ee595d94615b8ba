# experiment: proj_out (256 -> 256, k = 1, one channel tile) through the input-stationary kernel (all four chunks' loads in flight together)
for v in 0 1; do
  TQDNE_PW_ONE_TILE=$v timeout 200 python tools/layer_table.py 64 4096 5 2>/dev/null | grep "proj_out\|inference forward" | sed "s/^/PW_ONE_TILE=$v /"
done
TQDNE_PW_ONE_TILE=1 timeout 300 python -m pytest tests/test_hip_unet.py tests/test_hip_ops.py -m gpu -q -x 2>&1 | tail -2

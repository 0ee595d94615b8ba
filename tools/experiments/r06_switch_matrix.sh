cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r06_switch_matrix.txt; : > $OUT
for sw in "TQDNE_CONV_SCHEME=bf16x3" "TQDNE_SAMPLER_LANES=1" "TQDNE_SMALL_TILE=0" "TQDNE_POLYPHASE_TRAIN=0" "TQDNE_WGRAD_H64=0" "TQDNE_BWD_STREAMS=1" "TQDNE_GN_FOLD_SMALL=0" "TQDNE_ATTN_KSPLIT=1" "TQDNE_QKV_PW=1" "TQDNE_GN_FOLD=1"; do
  echo "== $sw" >> $OUT
  env $sw python3 -m pytest tests/test_hip_unet.py tests/test_bench_config_parity.py tests/test_unet_autograd.py -q -m gpu 2>&1 | tail -n 1 >> $OUT
done
cat $OUT

for F in 12 6 20 32 48 12; do
  echo "== wgwin_fixed=$F"
  CDAE_WGRAD_STREAM=0 TUNE=wgwin_fixed=$F python3 tools/train_step_m32.py 30 1 2>&1 | grep value | cut -c1-70
  CDAE_WGRAD_STREAM=0 TUNE=wgwin_fixed=$F STEPS=30 REGIONS=2 python3 tools/exp_train.py 2>&1 | grep "train ms" | cut -c1-60
done

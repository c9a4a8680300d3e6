for T in "convwin_min_tiles=256" "convwin_min_tiles=128" "convwin_min_tiles=64" "convwin_min_tiles=32" "convwin_min_tiles=256,convwin_nj2=1" "rows16_min_m=1024" "rows16_min_m=4096" "convwin_min_tiles=256"; do
  echo "== $T"
  CDAE_WGRAD_STREAM=0 TUNE=$T python3 tools/train_step_m32.py 30 1 2>&1 | grep value | cut -c1-70
done

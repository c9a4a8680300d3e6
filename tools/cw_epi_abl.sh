cp causaldiffae_amd/libcdae.so /tmp/keep.so; cp gpurun_ab_libdev.so causaldiffae_amd/libcdae.so
for D in 0 256; do echo "== CDAE_PS_DBG=$D"; CDAE_PS_DBG=$D timeout 200 python3 tools/prof_shapes.py --time --gm --gn 2>&1 | grep conv3x3; done
cp /tmp/keep.so causaldiffae_amd/libcdae.so

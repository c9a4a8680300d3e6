cp causaldiffae_amd/libcdae.so /tmp/keep.so; cp gpurun_ab_libdev.so causaldiffae_amd/libcdae.so
for D in 0 256 4 8 20 64 284; do echo "== CDAE_PS_DBG=$D"; CDAE_PS_DBG=$D timeout 120 python3 tools/conv16_shapes.py 2>&1 | grep conv16; done
cp /tmp/keep.so causaldiffae_amd/libcdae.so

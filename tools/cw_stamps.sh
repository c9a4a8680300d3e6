cp causaldiffae_amd/libcdae.so /tmp/keep.so; cp gpurun_ab_libdev.so causaldiffae_amd/libcdae.so
CDAE_PS_DBG=32 timeout 200 python3 tools/cw_stamps.py 2>&1 | grep -v amdgpu
cp /tmp/keep.so causaldiffae_amd/libcdae.so

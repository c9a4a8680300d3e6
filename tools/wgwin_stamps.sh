cp causaldiffae_amd/libcdae.so /tmp/keep.so
for L in gpurun_ab_lib0.so gpurun_ab_lib1.so; do cp $L causaldiffae_amd/libcdae.so; echo "== $L"; timeout 120 python3 tools/wgwin_stamps.py 2>&1 | grep -v amdgpu | sort | uniq | head -40; done
cp /tmp/keep.so causaldiffae_amd/libcdae.so

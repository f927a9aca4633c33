for aux in 2 16 17 18 0; do
  echo "== VND_NT_MIN_MB=0 VND_SPEC_STORE_AUX=$aux"
  VND_NT_MIN_MB=0 VND_SPEC_STORE_AUX=$aux python tools/shard_timeline.py 128 600 2>&1 | grep -v amdgpu.ids | head -3
done
echo "== default (nt_stores=0)"
python tools/shard_timeline.py 128 600 2>&1 | grep -v amdgpu.ids | head -3
for aux in 2 16 17; do
  echo "== 1024 streams VND_SPEC_STORE_AUX=$aux"
  VND_SPEC_STORE_AUX=$aux python tools/shard_timeline.py 1024 100 2>&1 | grep -v amdgpu.ids | head -3
done

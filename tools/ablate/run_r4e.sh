# profiles/r04_shard_timeline.txt: in-kernel timeline of cfg4's N = 8 shard, uniform spans against CU chunks
echo "### (1) uniform spans (VND_WIN_CHUNKS=0), phase stamps (reads_ahead=1 build: the stamps cost the registers of one read)"
python tools/win_stamps.py 128 48000 VND_SPEC_LA=1 VND_WIN_CHUNKS=0 2>&1 | grep -v amdgpu.ids
echo; echo "### (2) CU chunks 2 + 1, no stagger, no priority (VND_WIN_STAGGER_TICKS=0 VND_WIN_CHUNK_PRIO=0)"
python tools/win_stamps.py 128 48000 VND_SPEC_LA=1 VND_WIN_STAGGER_TICKS=0 VND_WIN_CHUNK_PRIO=0 2>&1 | grep -v amdgpu.ids
echo; echo "### (3) CU chunks 2 + 1, defaults (second workgroup 3 us behind, first at raised priority, final tile without refill)"
python tools/win_stamps.py 128 48000 VND_SPEC_LA=1 2>&1 | grep -v amdgpu.ids
echo; echo "### (4) entry / exit stamps only (the product's build, reads_ahead=2): uniform, then chunks"
python tools/win_stamps.py 128 48000 VND_WIN_STAMP_PHASES=0 VND_WIN_CHUNKS=0 2>&1 | grep -v amdgpu.ids
python tools/win_stamps.py 128 48000 VND_WIN_STAMP_PHASES=0 2>&1 | grep -v amdgpu.ids
echo; echo "### (5) the same for the N = 1 pass (1024 streams, two units per workgroup)"
python tools/win_stamps.py 1024 48000 VND_WIN_STAMP_PHASES=0 2>&1 | grep -v amdgpu.ids
echo; echo "### (6) what a kernel boundary costs on this box (tools/micro/boundary.hip)"
tools/micro/boundary

#!/bin/bash
# Diagnostic builds of libvnd_amd.so (timing / power experiments of DESIGN.md §3.5; results are wrong by
# construction in the VND_ABLATE builds).  Select one with  VND_AMD_LIBRARY=$PWD/tools/ablate/libvnd_<name>.so
#   ablate1  staging + merge + stores only          ablate2  + LDS tap reads, no FMAs
#   ablate3  + FMAs, no LDS tap reads               ablate4  full kernel without the two merge barriers
#   stamps   s_memtime phase stamps (tools/stamps.py)
#   s2 / l2  non-temporal stores / loads
set -e
cd "$(dirname "$0")/../.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off"
build() { /opt/rocm/bin/hipcc $F $2 vndecorrelate_amd/csrc/vnd_amd.hip -o tools/ablate/libvnd_$1.so & }
build ablate1 -DVND_ABLATE=1
build ablate2 -DVND_ABLATE=2
build ablate3 -DVND_ABLATE=3
build ablate4 -DVND_ABLATE=4
wait
build stamps -DVND_STAMPS
build s2 -DVND_STORE_AUX=2
build l2 -DVND_LOAD_AUX=2
wait
ls -la tools/ablate/*.so
build ablate5 -DVND_ABLATE=5
wait
ls -la tools/ablate/*.so

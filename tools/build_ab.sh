#!/bin/bash
# build_ab.sh — libpartls_hip.so from the working tree + libpartls_hip_lz.so with HEAD's sweep_lazy.hip (same-box A/B: tools/ab_lazy.sh)
set -e
cd /root/repo/partitionedls.jl_amd/csrc
make 2>&1 | grep -E "rror" || true
mkdir -p _build_lz
git show HEAD:partitionedls.jl_amd/csrc/sweep_lazy.hip > sweep_lazy_head.hip
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off -c sweep_lazy_head.hip -o _build_lz/sweep_lazy_head.o 2>&1 | grep -E "rror" || true
rm -f sweep_lazy_head.hip
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libpartls_hip_lz.so _build/api.o _build/gram.o _build/misc.o _build/sweep_generic.o _build/sweep_blk.o _build/sweep_coop.o _build/solvers.o _build/multi.o _build_lz/sweep_lazy_head.o -ldl
ls -la ../*.so

#!/bin/bash
# isa.sh <file.hip> <out.s> [extra hipcc flags] — gfx950 ISA of one translation unit (device side only)
f=$1; o=$2; shift; shift
cd /root/repo/partitionedls.jl_amd/csrc
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -ffp-contract=off --cuda-device-only -S "$@" $f -o $o

#!/bin/bash
# kres.sh <file.hip> [extra hipcc flags] — per-kernel register / spill report from the compiler's resource-usage remarks
f=$1; shift
cd /root/repo/partitionedls.jl_amd/csrc
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Rpass-analysis=kernel-resource-usage "$@" -c $f -o /tmp/kres.o 2>&1 |
 awk '/Function Name/{n=$0; sub(/.*Function Name: /,"",n); sub(/ \[.*/,"",n)} / VGPRs:/{v=$(NF-1)} /ScratchSize/{s=$(NF-1)} /SGPRs Spill/{ss=$(NF-1)} /VGPRs Spill/{vs=$(NF-1); printf "%-70s vgpr=%s scratch=%s sgpr_spill=%s vgpr_spill=%s\n", n, v, s, ss, vs} /error/{print}'

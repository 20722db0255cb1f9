"""Phase stamps of the blocked sweep kernel on one problem (diagnostic build `make -C partitionedls.jl_amd/csrc stamps`; run on
the GPU box).  usage: python tools/stamps.py N D K seed [patterns]"""
import os, sys
N, D, K, seed = (int(x) for x in sys.argv[1:5])
os.environ['PARTLS_LIB'] = os.path.join(os.getcwd(), 'partitionedls.jl_amd', os.environ.get('STAMPLIB', 'libpartls_hip_stamps.so'))
os.environ['PARTLS_PRINT_STAMPS'] = '1'
os.environ['PARTLS_GRID'] = '256'
sys.path.insert(0, os.getcwd())
import numpy as np, torch, partls_amd
pk = partls_amd.package(); ctx = pk.Context()
P, wstar = pk.synth_truth(seed, D, K)
dev = torch.device('cuda:0')
dX = torch.empty(N * D, dtype=torch.float64, device=dev); dy = torch.empty(N, dtype=torch.float64, device=dev)
ctx.synth_device(seed, N, D, wstar, dX.data_ptr(), dy.data_ptr())
ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, P, 0.0, 0)
npat = min(ctx.num_patterns(), int(sys.argv[5]) if len(sys.argv) > 5 else 1 << 17)
r = ctx.opt_sweep(0, npat)
print('blk', r[0], r[1], r[3], 'sweep ms', ctx.timing(2), 'patterns per WG', npat / 256, 'pivots', ctx.pivots())
ctx.close()

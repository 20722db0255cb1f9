"""Phase stamps of the sweep kernels on one problem (diagnostic build; run on the GPU box).
usage: python tools/two_stamps.py N D K seed kernel(two|blk)"""
import os, sys
N, D, K, seed = (int(x) for x in sys.argv[1:5]); kernel = sys.argv[5]
os.environ['PARTLS_KERNEL'] = kernel
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
npat = ctx.num_patterns()
r = ctx.opt_sweep(0, min(npat, 1 << 17))
print(kernel, r[0], r[1], r[3], 'sweep ms', ctx.timing(2), 'patterns per WG', min(npat, 1 << 17) / 256)

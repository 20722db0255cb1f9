"""A/B of the two-level sweep kernel against the single-level blocked kernel on one problem (run on the GPU box).
usage: python tools/two_ab.py [N D K seed]  — compares every pattern's objective and times the sweeps."""
import os, sys, time, subprocess, json
import numpy as np

def run(kernel, N, D, K, seed):
    code = f"""
import os, sys, json, time
os.environ['PARTLS_KERNEL']={kernel!r}
sys.path.insert(0, {os.getcwd()!r})
import numpy as np, torch
import partls_amd
pk = partls_amd.package()
ctx = pk.Context()
P, wstar = pk.synth_truth({seed}, {D}, {K})
dev = torch.device('cuda:0')
dX = torch.empty({N} * {D}, dtype=torch.float64, device=dev); dy = torch.empty({N}, dtype=torch.float64, device=dev)
ctx.synth_device({seed}, {N}, {D}, wstar, dX.data_ptr(), dy.data_ptr())
# per-pattern objectives need the faithful-intercept mode (2^(K+1) patterns); the timing is taken in the default mode
ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), {N}, {D}, {N}, P, 0.0, 1)
npat = ctx.num_patterns()
r=ctx.opt_sweep(0,npat,want_all=True)
obj,pat,allo,unc=r
fa_ms=ctx.timing(2); fa_piv=int(ctx.pivots())
ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), {N}, {D}, {N}, P, 0.0, 0)
npat = ctx.num_patterns()
for rep in range(2):
    t=time.time(); r2=ctx.opt_sweep(0,npat); dt=time.time()-t
obj2,pat2,_,unc2=r2
np.save('/tmp/allopt_{kernel}.npy', allo)
print(json.dumps(dict(kernel={kernel!r}, obj=obj, pat=int(pat), unconv=int(unc), faithful_ms=fa_ms, faithful_pivots=fa_piv, obj_free=obj2, pat_free=int(pat2), unconv_free=int(unc2), wall=dt, sweep_ms=ctx.timing(2), pivots=int(ctx.pivots()))))
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    if out.returncode: print(out.stderr[-3000:]); raise SystemExit(1)
    return json.loads(out.stdout.strip().splitlines()[-1])

if __name__ == "__main__":
    N, D, K, seed = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (20000, 128, 12, 1)
    a = run("blk", N, D, K, seed); b = run("two", N, D, K, seed)
    x = np.load('/tmp/allopt_blk.npy'); y = np.load('/tmp/allopt_two.npy')
    rel = np.abs(x - y) / np.maximum(np.abs(x), 1e-300)
    print(a); print(b)
    print("max rel diff over", len(x), "patterns:", rel.max(), "argmax", int(rel.argmax()), "speedup", a['sweep_ms'] / b['sweep_ms'])

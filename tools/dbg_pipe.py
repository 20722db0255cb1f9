import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, partls_amd
from oracle import oracle as O
from conftest import load_golden
pls = partls_amd.package()
g = load_golden("toy")
model, _, rep = pls.fit(pls.Opt, g["X"], g["y"], g["P"], returnAllSolutions=True)
objs = np.array([rep.solutions._all[b] for b in range(8)])
print("toy", np.abs(objs - g["opt_all_opt"]).max())
for (N,D,K,seed) in [(300,8,3,1),(400,20,4,2),(600,40,5,3),(2000,128,6,4),(3000,256,7,5),(3000,300,6,6)]:
    X,y,P,_ = O.synth(20260100+seed, N, D, K)
    ref = O.fit_opt(X,y,P,return_all=True)
    ctx = pls.default_context()
    ctx.opt_prepare(X,y,P,0.0,pls.lowlevel.OPT_FAITHFUL_INTERCEPT)
    bo,bp,allo,unc = ctx.opt_sweep(0,-1,want_all=True)
    print(N,D,K,"unconv",unc,"max rel diff",np.max(np.abs(allo-ref["all_opt"])/np.maximum(1,ref["all_opt"])),"pivots",ctx.pivots(), "winner", bp, ref["best_index"])

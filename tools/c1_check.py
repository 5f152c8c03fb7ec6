import os, sys, time
ROOT=os.environ["GRAFT_REPO_ROOT"]; sys.path[:0]=[ROOT, os.path.join(ROOT,"delta-prox_amd")]
import torch, dprox as dp, synthetic
gt,b1,psf=synthetic.deconv_case(1,1,256,256,seed=2023)
bt=torch.from_numpy(b1).cuda(); x=dp.Variable()
s=dp.compile(dp.sum_squares(dp.conv(x,psf)-bt)+dp.norm1(dp.grad(x,dim=0))+dp.norm1(dp.grad(x,dim=1)),method="admm",device="cuda")
from dprox import _backend as be
for st in ({}, dict(iter_par_max_rows=-1), {}):
    with be.tuned(**st):
        for rep in range(3):
            s.solve(x0=bt,rhos=0.1,lams=0.005,max_iter=20); torch.cuda.synchronize()
            t0=time.perf_counter()
            for _ in range(20): s.solve(x0=bt,rhos=0.1,lams=0.005,max_iter=20)
            torch.cuda.synchronize(); print(st, f"{(time.perf_counter()-t0)/20*1e3:.3f} ms per solve")

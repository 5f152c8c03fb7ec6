import os, sys, time
sys.path[:0]=[os.environ["GRAFT_REPO_ROOT"], os.path.join(os.environ["GRAFT_REPO_ROOT"],"delta-prox_amd")]
import torch, dprox as dp, synthetic
dev=torch.device("cuda")
for shape in [(1,1,256,256),(1,3,256,256),(2,3,256,256),(1,1,512,512),(1,3,512,512),(4,3,512,512),(1,1,1024,1024),(1,3,1024,1024)]:
    B,C,H,W=shape
    gt,b,psf=synthetic.deconv_case(B,C,H,W,seed=1)
    bt=torch.from_numpy(b).to(dev); x=dp.Variable()
    s=dp.compile(dp.sum_squares(dp.conv(x,psf)-bt)+dp.norm1(dp.grad(x,dim=0))+dp.norm1(dp.grad(x,dim=1)),method="admm",device=dev)
    s.solve(x0=bt,rhos=0.1,lams=0.005,max_iter=5); torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(5): s.solve(x0=bt,rhos=0.1,lams=0.005,max_iter=50)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
    print(shape, os.environ.get("DPX_ITER_ROWS","seq"), f"{dt/50*1e6:.1f} us/it")

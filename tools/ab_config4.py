import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, ctypes as C
from pnec_amd import Batch, capi, select_best
from pnec_amd import simulation as sim
dev=torch.device("cuda:0")
Bp,N,H=64,4096,64
g=sim.generate(Bp,N,seed=9,device=dev)
gen=torch.Generator(device=dev); gen.manual_seed(5)
hyp=torch.randn(Bp*H,3,generator=gen,dtype=torch.float64,device=dev); hyp=hyp/hyp.norm(dim=1,keepdim=True); hyp[::H]=g.init_t
for conv,mi in ((0,10),(1,50)):
    opts=capi.default_options(max_num_iterations=mi,check_convergence=conv)
    with Batch.uniform(capi.MODE_TARGET,Bp,N) as b:
        b.fill(g.bvs1.reshape(-1,3),g.bvs2.reshape(-1,3),g.covs2.reshape(-1,3,3))
        for _ in range(3): r=b.solve(g.init_q,None,options=opts,hyp_t=hyp,n_hyp=H)
        torch.cuda.synchronize(); t=time.perf_counter()
        for _ in range(5): r=b.solve(g.init_q,None,options=opts,hyp_t=hyp,n_hyp=H)
        torch.cuda.synchronize(); ms=(time.perf_counter()-t)/5*1e3
        cnt=np.zeros(16,dtype=np.uint64); flag=C.c_int32(0)
        capi.check(capi.lib().pnec_hip_work_counters(0,1,cnt.ctypes.data,C.byref(flag)))
        o2=capi.default_options(max_num_iterations=mi,check_convergence=conv,flags=1)
        b.solve(g.init_q,None,options=o2,hyp_t=hyp,n_hyp=H); torch.cuda.synchronize()
        capi.check(capi.lib().pnec_hip_work_counters(0,1,cnt.ctypes.data,C.byref(flag)))
        print(json.dumps({"lib":os.environ.get("PNEC_HIP_LIB","default")[-40:],"conv":conv,"ms":ms,"its_mean":float(r.iterations.double().mean()),"full_passes_per_solve":float(cnt[13])/(Bp*H*N),"cost_passes_per_solve":float(cnt[14])/(Bp*H*N), "launch":b.describe_launch(opts)}))

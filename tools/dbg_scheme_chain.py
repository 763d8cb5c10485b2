"""debug: device vs checker eigensolver stage under scheme S on the chain test's data; dumps the outlier pairs"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pnec_oracle as po
from pnec_amd import Batch, capi, simulation as sim
from tests.test_chain_scale_gpu import _angles
S = int(sys.argv[1]); mode = sys.argv[2]
P, N = int(sys.argv[3]) if len(sys.argv) > 3 else 2000, 256
dev = torch.device("cuda:0")
g = sim.generate(P, N, seed=3, device=dev)
bad = torch.rand(P, N, device=dev, generator=torch.Generator(device=dev).manual_seed(0)) < 0.10
rnd = torch.randn(P, N, 3, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
g.bvs2 = torch.where(bad[..., None], rnd / rnd.norm(dim=-1, keepdim=True), g.bvs2)
with Batch.uniform(capi.MODE_TARGET, P, N) as b:
    b.fill(g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3))
    b.set_eigensolver_scheme(S)
    if mode == "no_ransac":
        qe, te = b.nec_eigensolver(g.init_q)
        mask = torch.ones(P * N, dtype=torch.uint8)
    else:
        qe, te, mask, cnt, its = b.ransac_eigensolver(g.init_q, seed=1)
torch.cuda.synchronize()
po.set_eigensolver_scheme(S)
f1, f2 = g.bvs1.reshape(-1, 3).cpu().numpy(), g.bvs2.reshape(-1, 3).cpu().numpy()
m = mask.cpu().numpy().reshape(P, N).astype(bool)
es_q = np.zeros((P, 4)); infos = np.zeros(P, int); nfevs = np.zeros(P, int)
R0s = g.init_R.cpu().numpy()
mism = 0
for p in range(P):
    a, bb = f1[p * N:(p + 1) * N], f2[p * N:(p + 1) * N]
    if mode == "no_ransac":
        Ro, it = po.eigensolver(a, bb, R0s[p])
    else:
        Rr, tr, mo, ito = po.ransac_eigensolver(a, bb, R0s[p], seed=1, pair_id=p)
        if not (mo == m[p]).all() or ito != int(its[p]):
            print("  pair", p, "mask differs:", int((mo != m[p]).sum()), "entries; inliers dev", int(m[p].sum()), "oracle", int(mo.sum()), "its dev", int(its[p]), "oracle", ito)
            mism += 1
        Ro = Rr
    infos[p] = po.lib().pnec_oracle_es_last_info(); nfevs[p] = po.lib().pnec_oracle_es_last_nfev()
    es_q[p] = po.quat_from_rot(Ro)
a = _angles(qe.cpu().numpy(), es_q)
idx = np.argsort(-a)[:12]
print("pairs whose mask or iteration count differs:", mism)
print("scheme", S, mode, "beyond 1e-8:", int((a > 1e-8).sum()), "beyond 1e-6:", int((a > 1e-6).sum()), "p99 %.2e" % np.percentile(a, 99), "max %.2e" % a.max())
for p in idx:
    print(" pair", p, "angle %.3e" % a[p], "oracle info", infos[p], "nfev", nfevs[p], "inliers", int(m[p].sum()))
os.makedirs("gpurun_out/r05a", exist_ok=True)
np.savez(f"gpurun_out/r05a/dbg_s{S}_{mode}.npz", idx=idx, a=a[idx], q_dev=qe.cpu().numpy()[idx], q_or=es_q[idx],
         f1=np.stack([f1[p * N:(p + 1) * N] for p in idx]), f2=np.stack([f2[p * N:(p + 1) * N] for p in idx]),
         mask=m[idx], R0=R0s[idx])

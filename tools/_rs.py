import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from oracle import pnec_oracle as po
from pnec_amd import Batch, capi
from pnec_amd import simulation as sim
dev = torch.device("cuda:0")
for n in (512, 100, 64, 65, 128, 300, 448, 449, 511, 513, 600, 700, 1100):
    m = 6
    g = sim.generate(m, n, seed=5, device=dev)
    gen = torch.Generator(device=dev).manual_seed(3)
    bad = torch.rand(m, n, device=dev, generator=gen) < 0.15
    rnd = torch.randn(m, n, 3, dtype=torch.float64, device=dev, generator=gen); rnd = rnd / rnd.norm(dim=-1, keepdim=True)
    b2 = torch.where(bad[..., None], rnd, g.bvs2)
    b = Batch.uniform(capi.MODE_TARGET, m, n)
    b.fill(g.bvs1.reshape(-1, 3), b2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3))
    qr, tr, mask, cnt, its = b.ransac_eigensolver(g.init_q, seed=1)
    torch.cuda.synchronize()
    f1 = g.bvs1.cpu().numpy(); f2 = b2.cpu().numpy(); R0 = g.init_R.cpu().numpy()
    ok = []
    for p in range(m):
        R, t, mk, it = po.ransac_eigensolver(f1[p], f2[p], R0[p], seed=1, pair_id=p)
        ok.append((int(mk.sum()), int(cnt[p]), it, int(its[p]), bool((mask[p*n:(p+1)*n].cpu().numpy().astype(bool) == mk).all())))
    print(n, ok, flush=True)

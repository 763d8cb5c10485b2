import sys, numpy as np, math
sys.path.insert(0, '/root/repo')
from oracle import pnec_oracle as po
from pnec_amd import Batch, capi, simulation as sim
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B, N = 64, 384
g = sim.generate(B, N, seed=95)
rng = np.random.default_rng(2)
f1 = g.bvs1.reshape(-1, 3).numpy().copy(); f2 = g.bvs2.reshape(-1, 3).numpy().copy()
for p in range(B):
    bad = p * N + rng.choice(N, N // 6, replace=False)
    v = rng.normal(size=(len(bad), 3)); f2[bad] = v / np.linalg.norm(v, axis=1, keepdims=True)
with Batch.uniform(capi.MODE_NEC, B, N) as b:
    b.fill(f1, f2)
    b.set_eigensolver_scheme(S)
    q, t, mask, cnt, its = b.ransac_eigensolver(g.init_q.numpy(), seed=11)
po.set_eigensolver_scheme(S)
nbad = 0
for p in range(B):
    sl = slice(p * N, (p + 1) * N)
    Ro, to, mo, ito = po.ransac_eigensolver(f1[sl], f2[sl], g.init_R[p].numpy(), seed=11, pair_id=p)
    same = (mask[sl].astype(bool) == mo).all()
    if its[p] != ito or not same:
        nbad += 1
        print("pair", p, "its dev", its[p], "oracle", ito, "mask same", same, "cnt", cnt[p], mo.sum())
print("mismatching pairs:", nbad, "of", B)

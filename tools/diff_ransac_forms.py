import numpy as np, sys
a=np.load(sys.argv[1]); b=np.load(sys.argv[2])
off=a["off"]
for k in ("its","cn","cnt"):
    d=np.flatnonzero(a[k]!=b[k]); print(k,"differs in",len(d),"pairs",d[:10], a[k][d[:10]], b[k][d[:10]])
for k in ("qr","tr","q","t"):
    d=np.flatnonzero((a[k]!=b[k]).any(axis=1)); print(k,"differs in",len(d),"pairs",d[:10], np.abs(a[k]-b[k]).max())
for k in ("mk","mask"):
    d=np.flatnonzero(a[k]!=b[k]); pr=np.unique(np.searchsorted(off,d,side="right")-1); print(k,"differs in",len(d),"entries, pairs",pr[:10])
n=np.diff(off)
d=np.flatnonzero(a["its"]!=b["its"]); print("sizes of its-differing pairs", n[d[:20]])
d=np.flatnonzero((a["qr"]!=b["qr"]).any(axis=1)); print("sizes of qr-differing pairs", n[d[:20]], "its", a["its"][d[:20]], b["its"][d[:20]])

"""reg vs lds form of the folded sweep on the 'long' shapes (M = 1M, PSIGNN_JGROUPS=1): first stored pair / iterate that differs."""
import os, sys, itertools
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import pkg
dev = torch.device("cuda:0")
eng = pkg("engine")
gen = torch.Generator().manual_seed(0)
N = 100000
c = (0.05 + (0.995 - 0.05) * torch.rand(N, 10, generator=gen)).to(dev); b = torch.randn(N, 10, generator=gen).to(dev); x0 = torch.randn(N, 10, generator=gen).to(dev)
f = lambda x: c * x + b
os.environ["PSIGNN_JGROUPS"] = sys.argv[1] if len(sys.argv) > 1 else "1"
res = {}
for name in ("reg", "lds"):
    os.environ["PSIGNN_U2D_FORM"] = name
    sv = eng.DeviceBroyden(threshold=48, keep_trace=True, n_elems=x0.numel(), seq_len=10, device=dev)
    DX = []; PA = []
    def frec(x, sv=sv, DX=DX, PA=PA):
        DX.append(sv.pair(0, x, "update"))
        if len(DX) == 21:
            PA.append(sv.pair(0, x, "parta").reshape(-1)[:977 * 32].reshape(977, 32).clone())
        return f(x)
    out = sv.solve_callable(frec, x0, 0.0)
    res[name] = dict(PA=PA, DX=DX[1:], tr=out["rel_trace"][:48], U=[sv.pair(j, x0, "U") for j in range(48)], V=[sv.pair(j, x0, "V") for j in range(48)],
                     X=[sv.iterate(i, x0) for i in range(49)])
    sv.close()
a, l = res["reg"], res["lds"]
for key in ("DX", "V", "U", "X"):
    d = [j for j in range(len(a[key])) if not torch.equal(a[key][j], l[key][j])]
    print(key, "first differing index", d[:3], "max rel diff there", float((a[key][d[0]] - l[key][d[0]]).abs().max() / a[key][d[0]].abs().max()) if d else None)
    if d:
        j = d[0]; diff = (a[key][j] != l[key][j]).reshape(-1).nonzero().reshape(-1)
        print("   differing elements:", diff.numel(), "of", a[key][j].numel(), "first", diff[:5].tolist(), "last", diff[-5:].tolist())
d = [k for k in range(48) if a["tr"][k] != l["tr"][k]]
print("rel_trace first difference", d[:1])

pa, pl = a["PA"][0], l["PA"][0]
dd = (pa[:, :20] != pl[:, :20]).nonzero()
print("parta after it=19: differing entries", dd.shape[0], dd[:10].tolist())
for r, q in dd[:5].tolist():
    print("  row", r, "q", q, float(pa[r, q]), float(pl[r, q]))

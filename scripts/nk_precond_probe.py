"""Probe (torch vector algebra, not product code): does a two-level preconditioner -- aggregates of A consecutive plan-order nodes x 10
latent channels, Galerkin coarse operator assembled from coloured JVP probes -- make the Newton systems of a large mesh solvable?

    python scripts/nk_precond_probe.py nodes=100000 agg=64 warm=300 m=200 outer=12 eta=0.01
"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
kw = dict(a.split("=") for a in sys.argv[1:])
nodes = int(kw.get("nodes", 100000)); A = int(kw.get("agg", 64)); m = int(kw.get("m", 200)); eta = float(kw.get("eta", 0.01))
outer = int(kw.get("outer", 12)); warm = int(kw.get("warm", 300)); mode = kw.get("mode", "mult")
K = int(kw.get("k", 0)); nu = int(kw.get("nu", 30))     # k > 0: adaptive coarse space from k relaxed random vectors (else 10 constants)
pkg = importlib.import_module("psi-gnn_amd"); eng = importlib.import_module("psi-gnn_amd.engine")
solver = importlib.import_module("psi-gnn_amd.utilities.solver")
dev = torch.device("cuda:0")
mesh = pkg.data.make_hex_problem(pkg.data.hex_n_for_nodes(nodes), seed=0, compute_sol=False).to(dev)
w = np.load(os.path.join(ROOT, "tests", "golden", "weights_dirichlet.npz")); sd = {k: torch.from_numpy(w[k]) for k in w.files}
Pk = "autoencoder.encoder.mlp.mlp."
h0 = eng.mlp2(mesh.x, sd[Pk + "0.weight"].to(dev), sd[Pk + "0.bias"].to(dev), sd[Pk + "2.weight"].to(dev), sd[Pk + "2.bias"].to(dev))
fm = eng.FixedPointMap(eng.MeshPlan(mesh), eng.PackedWeights(sd, dev), h0, mesh.prb_data)
N, D = fm.plan.N, 10
ob = solver.broyden(fm, h0, threshold=warm, eps=1e-9, keep_trace=False)
x = fm.to_plan(ob["result"].float().contiguous())
print(f"N={N} Broyden({warm}) lowest {ob['lowest']:.3e}", flush=True)

# ---- aggregates (plan order) and their adjacency
ids = torch.arange(N, device=dev, dtype=torch.float32)[:, None].expand(N, D).contiguous()
orig_of_plan = fm.to_plan(ids)[:, 0].long()                       # plan row r holds original node orig_of_plan[r]
plan_of_orig = torch.empty(N, dtype=torch.long, device=dev); plan_of_orig[orig_of_plan] = torch.arange(N, device=dev)
nagg = (N + A - 1) // A
ei = mesh.edge_index
ea, eb = plan_of_orig[ei[0]] // A, plan_of_orig[ei[1]] // A
pairs = torch.unique(torch.stack([torch.cat([ea, eb]), torch.cat([eb, ea])]), dim=1).cpu().numpy()
adj = [set() for _ in range(nagg)]
for a, b in pairs.T:
    if a != b: adj[a].add(int(b))
# distance-2 greedy colouring: same colour => no common neighbour and not adjacent
colour = -np.ones(nagg, dtype=np.int64)
for a in range(nagg):
    used = set()
    for b in adj[a]:
        if colour[b] >= 0: used.add(colour[b])
        for c in adj[b]:
            if colour[c] >= 0: used.add(colour[c])
    c = 0
    while c in used: c += 1
    colour[a] = c
ncol = int(colour.max()) + 1
print(f"aggregates {nagg} of {A} nodes, max degree {max(len(s) for s in adj)}, colours {ncol}, coarse dofs {nagg * D}", flush=True)
agg_of_row = (torch.arange(N, device=dev) // A)
colour_t = torch.from_numpy(colour).to(dev)
# src[c][b] = the aggregate of colour c within distance <= 1 of b (unique), or -1
src = -np.ones((ncol, nagg), dtype=np.int64)
for a in range(nagg):
    src[colour[a], a] = a
    for b in adj[a]: src[colour[a], b] = a
src_t = torch.from_numpy(src).to(dev)
Npad = nagg * A
def Aop(v):           # (I - J) v
    return v - fm.jvp_p(x, v)
def pad(v):
    vp = torch.zeros(Npad, D, device=dev, dtype=v.dtype); vp[:N] = v
    return vp
Q = None              # (nagg, A*D, KC) local bases, orthonormal columns
def make_basis():
    global Q, KC
    if K == 0:
        KC = D
        Q = torch.zeros(nagg, A, D, D, device=dev)
        for ch in range(D): Q[:, :, ch, ch] = 1.0 / A ** 0.5
        Q = Q.view(nagg, A * D, D)
        last = N - (nagg - 1) * A                                   # rows of the ragged last aggregate
        Q[-1].view(A, D, D)[last:] = 0
        return
    KC = K
    gen = torch.Generator(device=dev).manual_seed(0)
    vs = []
    for i in range(K):
        v = torch.randn(N, D, device=dev, generator=gen)
        for _ in range(nu):
            v = fm.jvp_p(x, v); v = v / v.norm()
        vs.append(pad(v).view(nagg, A * D))
    B = torch.stack(vs, dim=2)                                      # (nagg, A*D, K)
    Q, _ = torch.linalg.qr(B)
def restrict(v):      # (N, D) -> (nagg, KC)
    return torch.bmm(Q.transpose(1, 2), pad(v).view(nagg, A * D, 1))[:, :, 0]
def prolong(c):       # (nagg, KC) -> (N, D)
    return torch.bmm(Q, c.view(nagg, KC, 1)).view(Npad, D)[:N].contiguous()

def build_coarse():
    Ac = torch.zeros(nagg * KC, nagg * KC, device=dev, dtype=torch.float64)
    for c in range(ncol):
        sel = (colour_t == c).float()[:, None]                       # (nagg, 1)
        s = src_t[c]
        ok = (s >= 0).nonzero()[:, 0]
        for ch in range(KC):
            cv = torch.zeros(nagg, KC, device=dev); cv[:, ch] = sel[:, 0]
            rw = restrict(Aop(prolong(cv))).double()                 # (nagg, KC)
            rows = (ok[:, None] * KC + torch.arange(KC, device=dev)[None, :]).reshape(-1)
            cols = (s[ok] * KC + ch)[:, None].expand(-1, KC).reshape(-1)
            Ac[rows, cols] = rw[ok].reshape(-1)
    return Ac

def gmres(op, b, m, tol, prec=None):
    beta = float(b.norm())
    V = torch.empty(m + 1, b.numel(), device=dev); V[0] = b.reshape(-1) / beta
    H = torch.zeros(m + 1, m, dtype=torch.float64)
    hist = []
    k = 0
    for j in range(m):
        z = V[j].view_as(b)
        if prec is not None: z = prec(z)
        wv = op(z).reshape(-1)
        for _ in range(2):
            h = V[:j + 1] @ wv
            wv = wv - V[:j + 1].T @ h
            H[:j + 1, j] += h.double().cpu()
        hn = float(wv.norm()); H[j + 1, j] = hn
        V[j + 1] = wv / hn
        k = j + 1
        e1 = torch.zeros(k + 1, dtype=torch.float64); e1[0] = beta
        y = torch.linalg.lstsq(H[:k + 1, :k], e1[:, None]).solution[:, 0]
        res = float((H[:k + 1, :k] @ y - e1).norm()) / beta
        hist.append(res)
        if res < tol: break
    dx = (V[:k].T @ y.float().to(dev)).view_as(b)
    if prec is not None: dx = prec(dx)
    gmres.ritz = torch.linalg.eigvals(H[:k, :k])           # Ritz values of the (preconditioned) operator on this Krylov space
    return dx, hist

def rel_of(xx):
    fx = fm.fp(xx); g = fx - xx
    return float(g.norm() / (fx.norm() + 1e-9)), g

rel, g = rel_of(x)
print(f"start rel {rel:.3e}")
t0 = time.time(); make_basis(); Ac = build_coarse(); Aci = torch.linalg.inv(Ac).float(); torch.cuda.synchronize()
print(f"coarse operator: {nagg * KC} dofs, {ncol * KC} JVP probes, build + inverse {time.time() - t0:.2f}s", flush=True)
def coarse(v): return prolong((Aci @ restrict(v).reshape(-1)).view(nagg, KC))
def prec_add(v): return v + coarse(v)
def prec_mult(v):
    z = coarse(v)
    return z + (v - Aop(z))          # coarse correction, then one Richardson sweep on the remainder
prec = prec_mult if mode == "mult" else prec_add
_, h_un = gmres(Aop, g, m, 1e-6)
rz = gmres.ritz
print(f"Ritz values of I - J ({len(rz)}-step Arnoldi from g): min real {float(rz.real.min()):.3e}, {int((rz.real < 0).sum())} with negative real "
      f"part, {int((rz.abs() < 0.05).sum())} with |.| < 0.05, max |imag| {float(rz.imag.abs().max()):.2e}, max real {float(rz.real.max()):.3f}", flush=True)
if kw.get("ritz_only"): sys.exit(0)
_, h_pr = gmres(Aop, g, m, 1e-6, prec)
pick = lambda h: ["%.1e" % h[i] for i in (9, 24, 49, 99, 149, 199) if i < len(h)]
print("GMRES relative residual after 10/25/50/100/150/200 steps: plain", pick(h_un), "| two-level", pick(h_pr), flush=True)

# ---- Newton with the preconditioned solves
nfe = 0
for it in range(outer):
    if it > 0:
        make_basis(); Ac = build_coarse(); Aci = torch.linalg.inv(Ac).float()
    dx, hist = gmres(Aop, g, m, eta, prec)
    s, ok = 1.0, False
    for _ in range(10):
        r2, g2 = rel_of(x + s * dx)
        if r2 < (1 - 1e-4 * s) * rel: ok = True; break
        s *= 0.5
    print(f"outer {it}: krylov {len(hist)} (res {hist[-1]:.1e}) step {s if ok else 0:.4f} rel {rel:.3e} -> {r2 if ok else rel:.3e}", flush=True)
    if not ok: break
    x = x + s * dx; rel, g = r2, g2
    if rel < 1e-6: break

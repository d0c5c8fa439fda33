"""Why was the driver's first timed repeat 3.3x slower (BENCH_r02: 1.071 vs 0.326 ms/step)?  Candidates: (a) GPU clocks ramping up after the
CPU-only setup phase, (b) first touch of the freshly hipMalloc'ed solver state (2.35 GB of U / V at K = 20).  Each variant runs in THIS process
after an idle pause, K = 20 iterations on the 1M-node mesh, three solves with one solver object; prints ms/step per solve.
  A  idle 3 s -> new solver -> 3 timed solves                      (the bench as it was)
  B  idle 3 s -> 150 ms of f evaluations -> new solver -> 3 solves  (clock ramp without touching the solver state)
  C  idle 3 s -> new solver -> hipMemset of its state (via torch: a same-size scratch is not the same memory, so: one throw-away solve
     with eps huge = 1 iteration touching nothing of U / V beyond row 0) -> 3 solves   [control for B: tiny GPU work, no ramp]
  D  new solver -> one full throw-away solve (touches every row) -> idle 3 s -> 3 solves   (state touched, clocks idle again)
Also: per-launch times of 40 back-to-back f evaluations right after an idle pause (first vs median = the ramp itself)."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_weights, pkg
dev = torch.device("cuda:0")
data, eng = pkg("data"), pkg("engine")
sd = load_weights("dirichlet")
net = pkg("model_psignn").ModelPSIGNN(dict(latent_dim=10, n_layers=1)).eval(); net.load_state_dict(sd); net = net.to(dev)
mesh = data.make_hex_problem(577, seed=0, compute_sol=False); md = mesh.to(dev)
with torch.no_grad():
    fm = net.deqdss.f.bind(net.autoencoder.encoder(md.x), md)
K = 20
def solves(sv, n=3):
    out = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); sv.solve(fm, eps=0.0, poll_every=K); torch.cuda.synchronize()
        out.append(round(1e3 * (time.perf_counter() - t0) / K, 4))
    return out
def ramp(ms):
    xp = fm.to_plan(fm.h0); t0 = time.perf_counter()
    while 1e3 * (time.perf_counter() - t0) < ms:
        fm.picard_p(xp, 20)
    torch.cuda.synchronize()
ws = eng.DeviceBroyden(plan=fm.plan, threshold=5); ws.solve(fm, eps=0.0); ws.close(); torch.cuda.synchronize()   # kernels loaded
# the ramp itself
time.sleep(3.0)
xp = fm.to_plan(fm.h0); ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
ev[0].record()
for i in range(40):
    fm.picard_p(xp, 1); ev[i + 1].record()
torch.cuda.synchronize()
t = [1e3 * ev[i].elapsed_time(ev[i + 1]) for i in range(40)]
print("f evaluations after 3 s idle, us each: first 6", [round(x, 1) for x in t[:6]], "median", round(float(np.median(t)), 1), "sum of the first 10 / 10 x median", round(sum(t[:10]) / (10 * float(np.median(t))), 2))
for rep in range(2):
    time.sleep(3.0); sv = eng.DeviceBroyden(plan=fm.plan, threshold=K); print("A idle -> new solver -> solves:", solves(sv)); sv.close()
    time.sleep(3.0); ramp(150); sv = eng.DeviceBroyden(plan=fm.plan, threshold=K); print("B idle -> ramp 150 ms -> new solver -> solves:", solves(sv)); sv.close()
    sv = eng.DeviceBroyden(plan=fm.plan, threshold=K); sv.solve(fm, eps=0.0, poll_every=K); torch.cuda.synchronize(); time.sleep(3.0); print("D new solver, touched -> idle -> solves:", solves(sv)); sv.close()
    time.sleep(3.0); ramp(150); time.sleep(0.0); sv = eng.DeviceBroyden(plan=fm.plan, threshold=100); print("E idle -> ramp -> new solver thr=100 (11.7 GB) -> K=100 solves:", end=" ")
    out = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); sv.solve(fm, eps=0.0, poll_every=100); torch.cuda.synchronize(); out.append(round(1e3 * (time.perf_counter() - t0) / 100, 4))
    print(out); sv.close()

"""Per-phase time budget of the f tile kernel from in-kernel clock stamps (psignn_prof_tile_stamps; the library must be
built with the stamps compiled in:  make -C psi-gnn_amd/csrc clean && make -C psi-gnn_amd/csrc EXTRA=-DTILE_STAMPS=1).

    python3 scripts/tile_phases.py [nodes=1000000]
"""
import ctypes as C, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
pkg = importlib.import_module("psi-gnn_amd"); eng = importlib.import_module("psi-gnn_amd.engine"); nat = importlib.import_module("psi-gnn_amd._native")
dev = torch.device("cuda:0")
mesh = pkg.data.make_hex_problem(pkg.data.hex_n_for_nodes(nodes), seed=0, compute_sol=False).to(dev)
w = np.load(os.path.join(ROOT, "tests", "golden", "weights_dirichlet.npz")); sd = {k: torch.from_numpy(w[k]) for k in w.files}
P = "autoencoder.encoder.mlp.mlp."
h0 = eng.mlp2(mesh.x, sd[P + "0.weight"].to(dev), sd[P + "0.bias"].to(dev), sd[P + "2.weight"].to(dev), sd[P + "2.bias"].to(dev))
plan = eng.MeshPlan(mesh)
fm = eng.FixedPointMap(plan, eng.PackedWeights(sd, dev), h0, mesh.prb_data)
x = fm.to_plan(h0)
for _ in range(5):
    x = fm.fp(x)
torch.cuda.synchronize()
buf = torch.zeros(plan.n_tiles * 4 * 8, dtype=torch.int64, device=dev)
nat.lib().psignn_prof_tile_stamps(C.c_void_p(buf.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); y = fm.fp(x); e1.record(); torch.cuda.synchronize()
nat.lib().psignn_prof_tile_stamps(C.c_void_p(0))
t = buf.cpu().numpy().reshape(plan.n_tiles, 4, 8).astype(np.float64)
us = e0.elapsed_time(e1) * 1e3
span = t[..., :6][t[..., :6] > 0]
total_ticks = span.max() - span.min()
tick_ns = 10.0   # wall_clock64(): constant 100 MHz
print(f"N={plan.N} tiles={plan.n_tiles}: first .. last stamp {total_ticks * tick_ns * 1e-3:.1f} us (kernel built with -DTILE_STAMPS=1; "
      f"event time around the launch incl. launch overhead {us:.1f} us)")
names = ["stage 1 (loads + projections)", "barrier wait", "neighbour sums (slot walk)", "node update", "store"]
ok = (t[..., 5] > 0)
for i, nme in enumerate(names):
    d = (t[..., i + 1] - t[..., i])[ok] * tick_ns * 1e-3
    print(f"  {nme:32s} mean {d.mean():6.2f} us  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f}")
life = (t[..., 5] - t[..., 0])[ok] * tick_ns * 1e-3
print(f"  wave lifetime per tile          mean {life.mean():6.2f} us  p10 {np.percentile(life, 10):6.2f}  p90 {np.percentile(life, 90):6.2f}")
for wv in range(4):
    d = (t[:, wv, 1] - t[:, wv, 0])[ok[:, wv]] * tick_ns * 1e-3
    print(f"  wave {wv}: stage 1 {d.mean():.2f} us; barrier wait {((t[:, wv, 2] - t[:, wv, 1])[ok[:, wv]] * tick_ns * 1e-3).mean():.2f} us")
# concurrency: how many tiles are in flight over time
st_, en_ = t[:, :, 0].min(axis=1), t[:, :, 5].max(axis=1)
grid = np.linspace(span.min(), span.max(), 200)
inflight = [(np.sum((st_ <= g) & (en_ >= g))) for g in grid]
print("  tiles in flight (of %d slots at 5 per CU): mean %.0f, by decile of the kernel's duration: %s" %
      (5 * 256, np.mean(inflight), [int(np.mean(inflight[i * 20:(i + 1) * 20])) for i in range(10)]))

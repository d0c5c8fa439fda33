"""Every form of the on-device Broyden update, in every regime, against the reference's recurrences.

csrc/solver.hip runs the rank-one update of ``broyden`` (utilities/solver.py:185-192) in several forms chosen by vector
length and by the number k of stored pairs: two passes over U and V (short vectors), three single-array sweeps U, V, U, the
third sweep folded with the next iteration's first one while k <= 24 (kept values in registers, ``k_sweep_u2r<KB>``, or in
per-thread LDS slots, ``k_sweep_u2d``), and beyond that a WINDOW of the 16 most recent pairs kept plus a partial first sweep
over the older ones -- the regime every real solve (``fw_thres`` 500 - 1 500) spends almost all its iterations in.

A Broyden trajectory cannot pin these forms against the CPU oracle over 48 iterations: measured on the oracle itself
(f(x) = c * x + b, fp32), a relative perturbation of 1e-7 of the start moves ``rel_trace`` by O(1) from iteration ~18 on -- every
update divides by vT . dg, a difference of nearly equal numbers.  What CAN be checked exactly, at any k, is each iteration by
itself: the solver's own state (iterates, update vectors, stored pairs U_j, V_j, read back through
``psignn_broyden_get_iterate / _get_pair``) must satisfy

    vT     = -dx + sum_j (U_j . dx) V_j                                  (rmatvec, solver.py:96-104, 186)
    u      = (dx + dg - sum_j (V_j . dg) U_j) / (vT . dg)                (matvec, solver.py:106-114, 187)
    update = g - sum_{j<=k} (V_j . g) U_j                                (solver.py:192)

evaluated in float64 ON THE CPU from the device's previous state.  The tolerance is not fitted: a device quantity may be at
most 16 x as far from the float64 value as a plain float32 evaluation of the same formula (torch CPU, other summation order) is.
A missed pair, a wrong coefficient, a stale `a`, an uncovered element range are off by a factor 1e4 or more on this scale.
Besides: the first iterations against ``oracle.broyden`` (before the chaos), the converged iterate against the closed form
b / (1 - c), bit-equality of the register and LDS forms of the folded sweep, and bit-equality of batched, fused single-mesh
and host-driven solves, which carries the check over to ``psignn_broyden_solve`` and ``psignn_broyden_solve_batch``."""
import numpy as np
import pytest
import torch

from conftest import load_weights, pkg, rel_l2
from oracle import psignn_oracle as orc

pytestmark = pytest.mark.gpu

EPS32 = float(np.finfo(np.float32).eps)
K = 48          # iterations per solve: the window regime starts at k = 25 with the default limits
ENVS = {
    "default": {},                                                  # fold in registers on long vectors / big shards, in LDS below; all pairs kept to k = 24, then a window of 16
    "reg": {"PSIGNN_U2D_FORM": "reg"},                              # k_sweep_u2r<KB>
    "lds": {"PSIGNN_U2D_FORM": "lds"},                              # k_sweep_u2d: round-2 form of the same fold
    "keep8": {"PSIGNN_U2D_FORM": "reg", "PSIGNN_U2D_KEEP": "8"},    # window of 8: k_sweep_u2r<8> in the window regime
    "window_from_5": {"PSIGNN_U2D_FORM": "reg", "PSIGNN_U2D_KMAX": "4", "PSIGNN_U2D_KEEP": "2"},
    "no_fold": {"PSIGNN_U2D_KMAX": "0"},                            # plain three-sweep form
    "two_pass": {"PSIGNN_UVU": "0"},                                # dots + axpy + final
}
_KNOBS = ("PSIGNN_U2D_FORM", "PSIGNN_U2D_KEEP", "PSIGNN_U2D_KMAX", "PSIGNN_UVU", "PSIGNN_JGROUPS")


def _setenv(monkeypatch, env):
    for k in _KNOBS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)


class _Recorder:
    """Wraps f for ``DeviceBroyden.solve_callable``: keeps every evaluation point and, from the second call on, the solver's
    update vector at the time of the call (= dx of that iteration: x_new = x + update has just been formed)."""

    def __init__(self, f, back=lambda t: t):
        self.f, self.back, self.solver = f, back, None
        self.X, self.DX = [], []

    def __call__(self, x):
        if self.X:
            self.DX.append(self.solver.pair(0, self.back(x), "update"))
        self.X.append(self.back(x).clone())
        return self.f(x)


def _scale(ref64, plain32):
    """Rounding scale of a formula: distance of its plain float32 evaluation from the float64 one (plus a floor of 4 ulp)."""
    return float((plain32.double() - ref64).norm()) + 4 * EPS32 * float(ref64.norm())


def _check_iteration(k, U32, V32, U, V, dx, dg, g, Vk, Uk, upd_next, where):
    """Iteration with k stored pairs (all CPU tensors; U32, V32: (k, M) float32, U, V the same in float64): the device's V_k, U_k
    and next update against the float64 recurrences on the device's own previous state."""
    dxd, dgd, gd = dx.double(), dg.double(), g.double()
    # vT = -dx + (U dx) V
    vT = -dxd + (U @ dxd) @ V
    vT32 = -dx + (U32 @ dx) @ V32
    e = float((Vk.double() - vT).norm())
    assert e <= 16 * _scale(vT, vT32), (where, "vT", e, _scale(vT, vT32))
    # u = D1 / s,  D1 = dx + dg - (V dg) U,  s = vT . dg  (the device's own vT: the stages are checked one by one)
    D1 = dxd + dgd - (V @ dgd) @ U
    D1_32 = dx + dg - (V32 @ dg) @ U32
    Ukd = Uk.double()
    proj = float(D1 @ Ukd) / float(D1 @ D1)                   # U_k = proj * D1 + remainder, proj = 1 / s
    rem = float((Ukd - proj * D1).norm())
    tol_dir = 16 * _scale(D1, D1_32) / float(D1.norm())
    assert rem <= tol_dir * float(Ukd.norm()), (where, "U_k direction", rem / float(Ukd.norm()), tol_dir)
    s64 = float(Vk.double() @ dgd)
    s_abs = float((Vk.double() * dgd).abs().sum())
    assert abs(1.0 / proj - s64) <= 16 * EPS32 * s_abs + 4 * tol_dir * abs(s64), (where, "s = vT.dg", 1.0 / proj, s64, s_abs)
    # update = D2 - u beta,  D2 = g - (V g) U,  beta = vT . g
    D2 = gd - (V @ gd) @ U
    D2_32 = g - (V32 @ g) @ U32
    r = D2 - upd_next.double()
    beta = float(r @ Ukd) / float(Ukd @ Ukd)
    rem2 = float((r - beta * Ukd).norm())
    tol2 = 16 * _scale(D2, D2_32) + 8 * EPS32 * (float(D2.norm()) + abs(beta) * float(Ukd.norm()))
    assert rem2 <= tol2, (where, "update", rem2, tol2)
    b64 = float(Vk.double() @ gd)
    b_abs = float((Vk.double() * gd).abs().sum())
    assert abs(beta - b64) <= 16 * EPS32 * b_abs + 4 * (tol2 / float(Ukd.norm())), (where, "beta = vT.g", beta, b64, b_abs)


def _run_recorded(eng, f, x0, n_elems=None, plan=None, shard_elems=0, back=lambda t: t):
    rec = _Recorder(f, back)
    if plan is not None:
        sv = eng.DeviceBroyden(plan=plan, threshold=K, keep_trace=True, shard_elems=shard_elems)
    else:
        sv = eng.DeviceBroyden(threshold=K, keep_trace=True, n_elems=n_elems, seq_len=x0.shape[1], device=x0.device)
    rec.solver = sv
    out = sv.solve_callable(rec, x0, 0.0)
    assert out["n_iter"] == K and out["stop_reason"] == 0, (out["n_iter"], out["stop_reason"])
    rec.DX.append(sv.pair(0, back(x0), "update"))        # the update formed by the last iteration
    return rec, sv, out


def _check_run(rec, sv, its, label):
    """rec.X[i] = iterate i (i = 0 .. K), rec.DX[i] = update that led from iterate i to i + 1 (and DX[K] the one after)."""
    X, DX = rec.X, rec.DX
    assert len(X) == K + 1 and len(DX) == K + 1
    for i in (1, K // 2, K):                                      # the recorded points are the solver's iterates
        assert torch.equal(sv.iterate(i, X[0]), X[i])
    kmax = max(its)
    Ucpu = torch.stack([sv.pair(j, X[0], "U").reshape(-1).cpu() for j in range(kmax + 1)])
    Vcpu = torch.stack([sv.pair(j, X[0], "V").reshape(-1).cpu() for j in range(kmax + 1)])
    U64, V64 = Ucpu.double(), Vcpu.double()
    G = {}

    def gof(i):                                                   # g_i = f(x_i) - x_i in float32, as k_resid forms it
        if i not in G:
            xi = X[i]
            G[i] = (rec.f(rec.fwd(xi)) if hasattr(rec, "fwd") else rec.f(xi))
            G[i] = (rec.back(G[i]) - xi).reshape(-1).cpu()
        return G[i]

    for it in its:       # iteration `it` (0-based): k = it stored pairs, x_it -> x_{it+1}, stores pair `it`, forms update it+1
        g_new, g_old = gof(it + 1), gof(it)
        _check_iteration(it, Ucpu[:it], Vcpu[:it], U64[:it], V64[:it], DX[it].reshape(-1).cpu(), g_new - g_old, g_new, Vcpu[it], Ucpu[it],
                         DX[it + 1].reshape(-1).cpu(), f"{label} it={it}")


def _linear(N, cmax, dev, seed=0):
    gen = torch.Generator().manual_seed(seed)
    c = 0.05 + (cmax - 0.05) * torch.rand(N, 10, generator=gen)
    b = torch.randn(N, 10, generator=gen)
    x0 = torch.randn(N, 10, generator=gen)
    return c, b, x0


ITS = (0, 1, 4, 5, 7, 9, 16, 17, 23, 24, 25, 26, 33, 40, 46)      # default form
ITS_SHORT = (1, 5, 6, 24, 25, 26, 46)                             # the other forms: window entry (k = 5 / 25), both sides of it


@pytest.mark.parametrize("size", ["mid", "long"])
def test_update_forms_satisfy_the_broyden_recurrences(size, dev, monkeypatch):
    """f(x) = c * x + b on (100 000, 10) through the generic-callable path (``psignn_broyden_ext_*``; M = 1 M elements).
    "mid": the natural shapes at this length (4 floats per lane, unsplit sweeps: the form of 100k-node meshes and of batched
    shards); "long": PSIGNN_JGROUPS=1 selects the long-vector shapes (16 floats per lane: the form of the 1M-node headline).
    Each variant of the update: recurrences at iterations spanning the all-kept regime, the transition k = 24 -> 25 and the
    window regime; first iterations vs the CPU oracle; register and LDS forms of the fold bit-identical."""
    eng = pkg("engine")
    c, b, x0 = _linear(100000, 0.995, dev)
    cd, bd, x0d = c.to(dev), b.to(dev), x0.to(dev)
    f = lambda x: cd * x + bd
    with torch.no_grad():
        ref = orc.broyden(lambda x: c * x + b, x0, threshold=10, eps=0.0)
    base = {"PSIGNN_JGROUPS": "1"} if size == "long" else {}
    outs = {}
    for name, env in ENVS.items():
        _setenv(monkeypatch, {**base, **env})
        rec, sv, out = _run_recorded(eng, f, x0d, n_elems=x0d.numel())
        np.testing.assert_allclose(out["rel_trace"][:8], ref["rel_trace"][:8], rtol=1e-3, err_msg=name)
        # (the oracle's float32 einsum over 1 M elements is the less accurate side here: its dot products carry ~1e-5)
        assert rel_l2(rec.X[3], ref["xest_trace"][3]) < 1e-5 and rel_l2(rec.X[6], ref["xest_trace"][6]) < 5e-4, name
        _check_run(rec, sv, ITS if name == "reg" else ITS_SHORT, f"{size}/{name}")
        outs[name] = (out, [rec.X[i] for i in (10, 30, K)])
        sv.close()
    _setenv(monkeypatch, {})
    # kept values in registers or in LDS: same operations in the same order, same partial-sum shapes
    (a, xa), (l, xl) = outs["reg"], outs["lds"]
    assert a["rel_trace"] == l["rel_trace"] and all(torch.equal(p, q) for p, q in zip(xa, xl))
    d, xd = outs["default"]                                          # the default is one of the two
    assert d["rel_trace"] == a["rel_trace"] and all(torch.equal(p, q) for p, q in zip(xd, xa))
    # every form is the same iteration up to rounding: the early trace agrees closely, the residual keeps falling in all of them
    for name, (o, _) in outs.items():
        np.testing.assert_allclose(o["rel_trace"][:10], a["rel_trace"][:10], rtol=2e-3, err_msg=name)
        assert o["lowest"] < 0.5 * o["rel_trace"][9], (name, o["lowest"], o["rel_trace"][9])   # (not monotone: rank-one updates spike)


def test_long_vector_natural_size_and_closed_form(dev, monkeypatch):
    """(320 000, 10) = 3.2 M elements: the long-vector form without any knob (16 floats per lane, fold in registers, window
    from k = 25).  Recurrences in the window regime; and a well-conditioned problem (c <= 0.5) converges, in every form, to
    the closed-form fixed point b / (1 - c) (48 iterations, most of them at the rounding floor with vT.dg ~ 0 -- NaN / inf
    scrubbing of solver.py:188-189 included)."""
    eng, solver = pkg("engine"), pkg("utilities.solver")
    _setenv(monkeypatch, {})
    c, b, x0 = _linear(320000, 0.995, dev, seed=1)
    cd, bd, x0d = c.to(dev), b.to(dev), x0.to(dev)
    rec, sv, out = _run_recorded(eng, lambda x: cd * x + bd, x0d, n_elems=x0d.numel())
    _check_run(rec, sv, (23, 24, 25, 31, 46), "3.2M/default")
    sv.close()
    del rec
    c, b, x0 = _linear(320000, 0.5, dev, seed=2)
    cd, bd, x0d = c.to(dev), b.to(dev), x0.to(dev)
    xstar = b / (1 - c)
    for name, env in ENVS.items():
        _setenv(monkeypatch, env)
        o = solver.broyden(lambda x: cd * x + bd, x0d, threshold=K, eps=0.0, keep_trace=False)
        assert rel_l2(o["result"], xstar) < 1e-6, (name, rel_l2(o["result"], xstar))
        assert o["lowest"] < 2e-7 and bool(torch.isfinite(o["result"]).all()), (name, o["lowest"])
    _setenv(monkeypatch, {})


def _mesh_map(n, seed, dev, sd, phase=0.0):
    data, eng = pkg("data"), pkg("engine")
    mesh = data.make_hex_problem(n, seed=seed, compute_sol=False, phase=phase)
    md = mesh.to(dev)
    with torch.no_grad():
        h0 = orc.encoder(sd, mesh.x)
    plan = eng.MeshPlan(md)
    return eng.FixedPointMap(plan, eng.PackedWeights(sd, dev), h0.to(dev), md.prb_data, None)


def test_fused_device_solve_is_the_host_driven_solve(dev, monkeypatch):
    """``psignn_broyden_solve`` (f fused with x + update, g, dg and the norm partials; everything on the device) against the
    host-driven solve of the same map through ``psignn_broyden_ext_*`` on a 99 919-node mesh, 48 iterations: the two share
    launch_update, so every iterate is bit-identical -- which carries the recurrence check of the host-driven run (GNN map,
    window regime included) over to the fused solver."""
    eng, solver = pkg("engine"), pkg("utilities.solver")
    _setenv(monkeypatch, {})
    fm = _mesh_map(182, 0, dev, load_weights("dirichlet"))
    assert fm.plan.tiled and fm.plan.N == 99919
    rec, sv, out = _run_recorded(eng, fm.fp, fm.to_plan(fm.h0), plan=fm.plan, back=fm.from_plan)
    rec.fwd = fm.to_plan
    _check_run(rec, sv, (3, 12, 24, 25, 26, 40, 46), "gnn 100k")
    fused = solver.broyden(fm, fm.h0, threshold=K, eps=0.0, keep_trace=True)
    assert fused["n_iter"] == K
    for i in (1, 9, 24, 25, 26, 40, K):
        assert torch.equal(fused["xest_trace"][i], rec.X[i]), i
    np.testing.assert_allclose(fused["rel_trace"][:K], out["rel_trace"][:K], rtol=1e-5)   # (norm partials: per tile vs per block)
    sv.close()


@pytest.mark.parametrize("envname", ["reg", "lds", "window_from_5"])
def test_batched_shard_in_the_window_regime(envname, dev, monkeypatch):
    """``psignn_broyden_solve_batch`` through 48 iterations (window regime from k = 25; from k = 5 with the small limits) on a
    shard of 8 x 10 267-node meshes -- the shapes of BASELINE configs[3] (shard-sized reductions, 4 floats per lane): every mesh
    bit-identical to its own fused solve with the same solver object, that one bit-identical to the host-driven solve, and the
    host-driven solve satisfies the recurrences."""
    eng = pkg("engine")
    _setenv(monkeypatch, ENVS[envname])
    sd = load_weights("dirichlet")
    fms = [_mesh_map(58, s, dev, sd, phase=0.37 * s) for s in range(8)]
    w0 = fms[0].weights
    for f in fms[1:]:
        f.weights = w0                                             # one packed weight buffer for the shard
    total = sum(f.plan.N for f in fms) * 10
    solvers = [eng.DeviceBroyden(plan=f.plan, threshold=K, keep_trace=True, shard_elems=total) for f in fms]
    outs = eng.broyden_solve_batch(solvers, fms, 0.0)
    its_b = [[sv.iterate(i, f.h0) for i in (1, 24, 25, 26, 40, K)] for sv, f in zip(solvers, fms)]
    for m in (0, 5):
        single = solvers[m].solve(fms[m], 0.0)
        assert single["rel_trace"] == outs[m]["rel_trace"] and torch.equal(single["result"], outs[m]["result"])
        assert all(torch.equal(solvers[m].iterate(i, fms[m].h0), x) for i, x in zip((1, 24, 25, 26, 40, K), its_b[m]))
    for sv in solvers:
        sv.close()
    fm = fms[5]
    rec, sv, out = _run_recorded(eng, fm.fp, fm.to_plan(fm.h0), plan=fm.plan, shard_elems=total, back=fm.from_plan)
    rec.fwd = fm.to_plan
    for i, x in zip((1, 24, 25, 26, 40, K), its_b[5]):
        assert torch.equal(rec.X[i], x), i
    _check_run(rec, sv, (4, 5, 6, 24, 25, 26, 46), f"shard/{envname}")
    sv.close()
    _setenv(monkeypatch, {})

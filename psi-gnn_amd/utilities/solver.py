"""Fixed-point solver library with the reference's signatures and return dictionaries.

Mirror of ``dirichlet/psignn/utilities/solver.py`` (≡ ``mixed/psignn/utilities/solver.py``):

    solver(f, x0, threshold=int, eps=float) -> dict(result, lowest, nstep, prot_break,
                                                   abs_trace, rel_trace, xest_trace, eps, threshold)

``f`` maps an (N, d) device tensor to an (N, d) device tensor.  The arithmetic runs in
libpsignn_hip.so: when ``f`` is an ``engine.FixedPointMap`` (what the models pass) the whole
Broyden root-find — f, norms, stop tests, low-rank updates — stays on the device and the host
only polls a done flag; for any other callable the low-rank machinery is still native and
``f`` is called once per iteration.  CPU tensors are rejected: there is no CPU fallback.
"""
from __future__ import annotations

import torch

from .. import _native as nat
from ..engine import D, TRACE_BUDGET_BYTES, DeviceBroyden, FixedPointMap


class _LazyTrace:
    """``xest_trace`` of a device solve: list-like, iterates are copied out on demand."""

    def __init__(self, solver, n_iter, like, x0, result, nstep):
        self._s, self._n, self._like = solver, n_iter, like
        self._x0, self._res, self._nstep = x0, result, nstep

    def __len__(self):
        return self._n + 1

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        if self._s.keep_trace:
            return self._s.iterate(i, self._like)
        if i == 0:
            return self._x0
        if i == self._nstep:
            return self._res
        raise RuntimeError("iterates were not kept (trace would exceed the memory budget); "
                           "call broyden(..., keep_trace=True)")

    def __iter__(self):
        return (self[i] for i in range(len(self)))


def broyden(f, x0, threshold, eps=1e-3, stop_mode="rel", ls=False, name="unknown", keep_trace=None,
            poll_every=8, solver_obj=None):
    """Broyden's method on g(x) = f(x) - x  (reference: utilities/solver.py:116-207).

    Same contract as the reference: ``result`` is the lowest-``stop_mode`` iterate x (not f(x)), ``nstep`` its
    index, ``rel = |f(x)-x| / (|f(x)| + 1e-9)``, stops on ``objective < eps``, the 30-step plateau rule, the
    protective break (``rel > rel_0 * 1e3 * d``; ``abs > abs_0 * 1e6 * d`` in stop_mode "abs") or ``threshold``
    iterations.  ``ls=True`` (Armijo line search, solver.py:20-94; no call site of the reference enables it) runs the
    host-driven loop -- the trial points need extra f evaluations between the device steps -- also for a
    ``FixedPointMap``.
    """
    if stop_mode not in ("rel", "abs"):
        raise ValueError(f"stop_mode {stop_mode!r}")
    nat.require_cuda(x0, "x0")
    if x0.dim() != 2:
        raise nat.NativeError(f"x0 must be (N, d), got {tuple(x0.shape)}")
    M = x0.numel()
    if keep_trace is None:
        keep_trace = (threshold + 2) * M * 4 <= TRACE_BUDGET_BYTES
    if isinstance(f, FixedPointMap) and not ls:
        # solver_obj: a DeviceBroyden of the same plan / threshold kept by the caller between solves (a training loop
        # would otherwise allocate and free 2 * threshold * N * d floats per step); its iterates are overwritten by the
        # next solve, so it is only meant for callers that do not keep xest_trace
        solver = solver_obj if solver_obj is not None else DeviceBroyden(plan=f.plan, threshold=threshold, keep_trace=keep_trace)
        solver.set_stop_mode(stop_mode)
        out = solver.solve(f, eps, poll_every=poll_every)
        x_init = f.h0
    else:
        solver = DeviceBroyden(threshold=threshold, keep_trace=keep_trace, n_elems=M, seq_len=x0.shape[1],
                               device=x0.device)
        solver.set_stop_mode(stop_mode)
        out = solver.solve_callable(f, x0, eps, ls=ls)
        x_init = x0
    res = out["result"].reshape(x0.shape)
    trace = _LazyTrace(solver, out["n_iter"], res, x_init, res, out["nstep"])
    return {"result": res, "lowest": out["lowest"], "nstep": out["nstep"], "prot_break": out["prot_break"],
            "abs_trace": out["abs_trace"], "rel_trace": out["rel_trace"], "xest_trace": trace,
            "eps": eps, "threshold": threshold, "n_iter": out["n_iter"], "stop_reason": out["stop_reason"]}


def forward_iteration(f, z0, eps=1.e-5, threshold=50):
    """Picard iteration (reference: utilities/solver.py:301-341): returns the last iterate."""
    nat.require_cuda(z0, "z0")
    z_est = [z0]
    z_prev, z = z0, f(z0)
    trace = {"abs": [], "rel": []}
    ite = 0
    a = torch.linalg.norm(z_prev - z)
    r = a / torch.linalg.norm(z)
    trace["abs"].append(a.detach())
    trace["rel"].append(r.detach())
    z_est.append(z)
    while trace["rel"][-1] > eps and ite < threshold:
        z_prev = z
        z = f(z_prev)
        ite += 1
        a = torch.linalg.norm(z_prev - z)
        r = a / torch.linalg.norm(z)
        trace["abs"].append(a.detach())
        trace["rel"].append(r.detach())
        z_est.append(z)
    return {"result": z, "lowest": trace["rel"][-1], "abs_trace": trace["abs"], "rel_trace": trace["rel"],
            "xest_trace": z_est, "nstep": ite, "eps": eps, "threshold": threshold}


def anderson(f, x0, m=2, lam=1e-4, threshold=50, eps=1e-3, stop_mode="rel", beta=1.0, **kwargs):
    """Anderson acceleration (reference: utilities/solver.py:215-293; m=2, lam=1e-4, beta=1;
    ``rel = |f(x)-x| / (1e-5 + |f(x)|)``)."""
    nat.require_cuda(x0, "x0")
    shape = x0.shape
    n = x0.numel()
    alt = "rel" if stop_mode == "abs" else "abs"
    kw = dict(dtype=x0.dtype, device=x0.device)
    X = torch.zeros(m, n, **kw)
    Fm = torch.zeros(m, n, **kw)
    X[0] = x0.reshape(-1)
    Fm[0] = f(x0).reshape(-1)
    X[1] = Fm[0]
    Fm[1] = f(Fm[0].reshape(shape)).reshape(-1)
    H = torch.zeros(m + 1, m + 1, **kw)
    H[0, 1:] = H[1:, 0] = 1
    y = torch.zeros(m + 1, 1, **kw)
    y[0] = 1
    trace = {"abs": [], "rel": []}
    lowest = {"abs": 1e8, "rel": 1e8}
    lowest_step = {"abs": 0, "rel": 0}
    xest_trace = [x0]
    lowest_x = None
    for k in range(2, threshold):
        nn = min(k, m)
        G = Fm[:nn] - X[:nn]
        H[1:nn + 1, 1:nn + 1] = G @ G.t() + lam * torch.eye(nn, **kw)
        alpha = torch.linalg.solve(H[:nn + 1, :nn + 1], y[:nn + 1])[1:nn + 1, 0]
        X[k % m] = beta * (alpha[None] @ Fm[:nn])[0] + (1 - beta) * (alpha[None] @ X[:nn])[0]
        Fm[k % m] = f(X[k % m].reshape(shape)).reshape(-1)
        gx = Fm[k % m] - X[k % m]
        abs_diff = gx.norm().item()
        rel_diff = abs_diff / (1e-5 + Fm[k % m].norm().item())
        diff = {"abs": abs_diff, "rel": rel_diff}
        trace["abs"].append(abs_diff)
        trace["rel"].append(rel_diff)
        for mode in ("rel", "abs"):
            if diff[mode] < lowest[mode]:
                if mode == stop_mode:
                    lowest_x = X[k % m].reshape(shape).clone()
                lowest[mode] = diff[mode]
                lowest_step[mode] = k
        xest_trace.append(lowest_x)
        if trace[stop_mode][-1] < eps:
            for _ in range(threshold - 1 - k):
                trace[stop_mode].append(lowest[stop_mode])
                trace[alt].append(lowest[alt])
            break
    return {"result": lowest_x, "lowest": lowest[stop_mode], "nstep": lowest_step[stop_mode], "prot_break": False,
            "abs_trace": trace["abs"], "rel_trace": trace["rel"], "xest_trace": xest_trace,
            "eps": eps, "threshold": threshold}


def newton(f, z0, eps=1.e-5, threshold=50):
    """Node-block-diagonal Newton wrapped in forward_iteration (reference: utilities/solver.py:349-366).

    The reference builds the dense (N,d,N,d) Jacobian with autograd and keeps its per-node d x d blocks
    (``einsum('bibj->bij')``).  Here the same blocks come from analytic JVPs: block column j of node n is
    (J v)[n] for v = e_j on the nodes of one colour class of the mesh graph (no two adjacent nodes share a
    colour, so the neighbours' contributions vanish) — O(N d^2) memory instead of O((N d)^2).
    """
    if not isinstance(f, FixedPointMap):
        raise nat.NativeError("newton needs the analytic JVP of a FixedPointMap")
    from .newton_blocks import block_newton_map
    g = block_newton_map(f)
    r = forward_iteration(g, z0, eps=eps, threshold=threshold)
    return {"result": r["result"], "lowest": r["lowest"], "rel_trace": r["rel_trace"], "abs_trace": r["abs_trace"],
            "xest_trace": r["xest_trace"], "nstep": r["nstep"], "eps": eps, "threshold": threshold}


def newton_krylov(f, x0, threshold=30, eps=1e-5, inner_m=30, inner_tol=1e-2, **kwargs):
    """Jacobian-free Newton-Krylov on g(x) = f(x) - x with the ANALYTIC Jacobian-vector product.

    The reference imports ``scipy.optimize.newton_krylov`` (utilities/solver.py:6) but never calls it, so
    there is no reference implementation; the dict keys follow ``broyden``.  Outer Newton step:
    solve (J_f(x) - I) dx = -g(x) approximately with ``inner_m`` GMRES steps (inexact Newton; ``inner_tol`` is
    accepted for signature compatibility, the inner loop runs a fixed number of steps so that it needs no
    host synchronisation), x <- x + dx.  ``rel = |g|/(|f(x)| + 1e-9)`` as in broyden; ``nstep`` counts outer
    iterations, ``n_feval`` f + JVP evaluations (the unit comparable to one Broyden iteration).
    With fp32 finite-difference JVPs scipy's solver does not converge on this problem; the
    analytic JVP kernel (psignn_f_jvp) is what makes the method usable in fp32.
    """
    if not isinstance(f, FixedPointMap):
        raise nat.NativeError("newton_krylov needs the analytic JVP of a FixedPointMap")
    nat.require_cuda(x0, "x0")
    # tiled single-layer dirichlet plans: the whole iteration in plan order on the tiled f / JVP kernels (norms and
    # inner products do not depend on the node numbering); results are returned in the caller's numbering
    plan_mode = f.plan.tiled and not f.weights.mixed and f.weights.n_layers == 1
    F, J = (f.fp, f.jvp_p) if plan_mode else (f, f.jvp)
    back = f.from_plan if plan_mode else (lambda t: t)
    x = f.to_plan(x0) if plan_mode else x0.clone()
    fx = F(x)
    g = fx - x
    trace = {"abs": [], "rel": []}
    n_feval = 1
    lowest, lowest_x, lowest_step = 1e8, x, 0
    xest_trace = [x]
    nstep = 0
    for nstep in range(1, threshold + 1):
        # GMRES(inner_m) on A dx = b, A v = J_f v - v, b = -g.  The Arnoldi process runs a fixed number of steps
        # with everything (Hessenberg matrix included) on the device: no host sync inside the inner loop; the
        # small least-squares problem is solved once per outer step.
        b = -g.reshape(-1)
        beta = torch.linalg.norm(b)
        m = inner_m
        V = torch.zeros(m + 1, b.numel(), device=b.device, dtype=b.dtype)
        Hm = torch.zeros(m + 1, m, device=b.device, dtype=b.dtype)
        V[0] = b / beta
        for k in range(m):
            w = (J(x, V[k].reshape(x.shape)) - V[k].reshape(x.shape)).reshape(-1)
            n_feval += 1
            h = V[:k + 1] @ w                      # classical Gram-Schmidt, twice
            w = w - h @ V[:k + 1]
            h2 = V[:k + 1] @ w
            w = w - h2 @ V[:k + 1]
            Hm[:k + 1, k] = h + h2
            hn = torch.linalg.norm(w)
            Hm[k + 1, k] = hn
            V[k + 1] = w / hn.clamp_min(1e-30)
        e1 = torch.zeros(m + 1, 1, device=b.device, dtype=torch.float64)
        e1[0, 0] = beta.double()
        y = torch.linalg.lstsq(Hm.double().cpu(), e1.cpu()).solution[:, 0].to(b.device, b.dtype)
        dx = (y @ V[:m]).reshape(x.shape)
        x = x + dx
        fx = F(x)
        n_feval += 1
        g = fx - x
        abs_diff = torch.linalg.norm(g).item()
        rel_diff = abs_diff / (torch.linalg.norm(fx).item() + 1e-9)
        trace["abs"].append(abs_diff)
        trace["rel"].append(rel_diff)
        xest_trace.append(x)
        if rel_diff < lowest:
            lowest, lowest_x, lowest_step = rel_diff, x, nstep
        if rel_diff < eps:
            break
    return {"result": back(lowest_x), "lowest": lowest, "nstep": lowest_step, "prot_break": False,
            "abs_trace": trace["abs"], "rel_trace": trace["rel"], "xest_trace": [back(t) for t in xest_trace], "eps": eps,
            "threshold": threshold, "n_feval": n_feval, "n_outer": nstep}

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

import os

import torch

from .. import _native as nat
from ..engine import D, TRACE_BUDGET_BYTES, DeviceBroyden, DeviceFixedPointIter, FixedPointMap


class _NoTrace:
    keep_trace = False


_NO_TRACE = _NoTrace()


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
        # without solver_obj and without a kept trace the map lends its idle solver (FixedPointMap.borrow_broyden): repeated solves
        # on one map do not re-allocate their state
        pooled = solver_obj is None and not keep_trace
        if pooled:
            solver = f.borrow_broyden(threshold)
        else:
            solver = solver_obj if solver_obj is not None else DeviceBroyden(plan=f.plan, threshold=threshold, keep_trace=keep_trace)
        solver.set_stop_mode(stop_mode)
        try:
            out = solver.solve(f, eps, poll_every=poll_every)
        finally:
            if pooled:
                f.return_broyden(solver)
        if pooled:
            solver = _NO_TRACE          # (the result holds no reference to the lent solver: its buffers belong to the next solve)
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


class _LazyIterates:
    """``xest_trace`` of a device Picard / Anderson run: iterates are copied out of the solver's trace buffer on demand.
    ``index[j]`` = which stored iterate entry j of the reference's list is (Anderson appends the LOWEST iterate so far at
    every step, solver.py:275)."""

    def __init__(self, it, index, first, shape, back):
        self._it, self._index, self._first, self._shape, self._back = it, index, first, shape, back

    def __len__(self):
        return len(self._index) + 1

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        if i == 0:
            return self._first
        if not self._it.keep_trace:
            raise RuntimeError("iterates were not kept (trace would exceed the memory budget); pass keep_trace=True")
        return self._back(self._it.iterate(self._index[i - 1], self._first.reshape(-1)).reshape(self._shape))

    def __iter__(self):
        return (self[i] for i in range(len(self)))


def _plan_mode(f):
    """Tiled plans: iterate in plan order on f.fp (no permutation passes per evaluation; norms, inner products and linear
    combinations do not depend on the node numbering); results go back to the caller's numbering."""
    if isinstance(f, FixedPointMap) and f.plan.tiled:
        return f.fp, f.to_plan, f.from_plan
    ident = lambda t: t
    return f, ident, ident


def forward_iteration(f, z0, eps=1.e-5, threshold=50, keep_trace=None, poll_every=8):
    """Picard iteration z <- f(z) (reference: utilities/solver.py:301-341): after every evaluation abs = |z_prev - z|,
    rel = abs / |z|; runs while rel > eps and fewer than ``threshold`` loop passes were made; returns the LAST iterate,
    ``lowest`` = the last relative residual (a 0-dim tensor, like the reference's), ``nstep`` = loop passes.

    The norms, traces and the stop test live on the device (csrc/fpiter.hip); the host evaluates f and reads the done flag
    every ``poll_every`` evaluations, so up to ``poll_every - 1`` evaluations are issued past the stop and ignored."""
    nat.require_cuda(z0, "z0")
    F, to_p, from_p = _plan_mode(f)
    shape = z0.shape
    x0 = to_p(z0.to(torch.float32).contiguous())
    M = x0.numel()
    if keep_trace is None:
        keep_trace = (threshold + 3) * M * 4 <= TRACE_BUDGET_BYTES
    it = DeviceFixedPointIter(M, x0.device, m=1, threshold=threshold, keep_trace=keep_trace)
    with torch.cuda.device(x0.device):
        it.picard_begin(x0)
        for i in range(threshold + 1):
            x = it.picard_current(x0)
            it.picard_update(F(x).to(torch.float32).contiguous(), eps)
            if (i + 1) % poll_every == 0 and it.poll():
                break
        out = it.finish(x0)
    dev = z0.device
    rel = list(torch.tensor(out["rel_trace"], dtype=torch.float32, device=dev).unbind())
    ab = list(torch.tensor(out["abs_trace"], dtype=torch.float32, device=dev).unbind())
    n = out["n_iter"]
    trace = _LazyIterates(it, list(range(1, n + 1)), z0, shape, from_p)
    return {"result": from_p(out["result"]).reshape(shape), "lowest": rel[-1], "abs_trace": ab, "rel_trace": rel,
            "xest_trace": trace, "nstep": out["nstep"], "eps": eps, "threshold": threshold}


def anderson(f, x0, m=2, lam=1e-4, threshold=50, eps=1e-3, stop_mode="rel", beta=1.0, keep_trace=None, poll_every=8,
             **kwargs):
    """Anderson acceleration (reference: utilities/solver.py:215-293; m = 2, lam = 1e-4, beta = 1 at its call sites;
    ``rel = |f(x) - x| / (1e-5 + |f(x)|)``): loop index k = 2 .. threshold - 1, the lowest iterate in ``stop_mode`` is the
    result, traces are padded with the lowest values after an early stop.

    Gram matrix of the residual history, the (m+1) x (m+1) bordered solve, the mixing step, norms and stop test run on the
    device (csrc/fpiter.hip); the host evaluates f and polls the done flag every ``poll_every`` steps."""
    if stop_mode not in ("rel", "abs"):
        raise ValueError(f"stop_mode {stop_mode!r}")
    nat.require_cuda(x0, "x0")
    F, to_p, from_p = _plan_mode(f)
    shape = x0.shape
    xp = to_p(x0.to(torch.float32).contiguous())
    M = xp.numel()
    if keep_trace is None:
        keep_trace = (threshold + 3) * M * 4 <= TRACE_BUDGET_BYTES
    it = DeviceFixedPointIter(M, xp.device, m=m, threshold=threshold, keep_trace=keep_trace)
    c = lambda t: t.to(torch.float32).contiguous()
    with torch.cuda.device(xp.device):
        f0 = c(F(xp))
        f1 = c(F(f0))
        it.anderson_begin(xp, f0, f1, lam, beta, stop_mode == "abs")
        for k in range(2, threshold):
            x = it.anderson_next(xp)
            it.anderson_update(c(F(x)), eps)
            if (k - 1) % poll_every == 0 and it.poll():
                break
        out = it.finish(xp)
    n = out["n_iter"]
    low, low_alt = (out["lowest_abs"], out["lowest"]) if stop_mode == "abs" else (out["lowest"], out["lowest_abs"])
    rel, ab = out["rel_trace"], out["abs_trace"]
    if out["stop_reason"] == 1:   # early stop: pad as solver.py:279-282 (threshold - 1 - k entries, k = last loop index)
        pad = threshold - 1 - (n + 1)
        rel = rel + [out["lowest"]] * pad
        ab = ab + [out["lowest_abs"]] * pad
    trace = _LazyIterates(it, out["low_idx"], x0, shape, from_p)
    return {"result": from_p(out["result"]).reshape(shape), "lowest": low, "nstep": out["nstep"], "prot_break": False,
            "abs_trace": ab, "rel_trace": rel, "xest_trace": trace, "eps": eps, "threshold": threshold}


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


def newton_krylov(f, x0, threshold=30, eps=1e-5, inner_m=200, inner_tol=1e-2, max_backtracks=10, poll_every=10,
                  warm_start=0, picard_fallback=20, broyden_burst=0, burst_below=0.1, **kwargs):
    """Globalised inexact Newton-Krylov on g(x) = f(x) - x with the ANALYTIC Jacobian-vector product (BASELINE configs[4]).

    The reference imports ``scipy.optimize.newton_krylov`` (utilities/solver.py:6) but never calls it, so there is no
    reference implementation; the dict keys follow ``broyden``.  Outer step: solve (J_f(x) - I) dx = -g(x) with GMRES until
    the linear residual is below ``inner_tol`` (or ``inner_m`` Krylov vectors: no restarts -- restarted GMRES throws away the
    smooth modes the problem's long-range coupling lives in, and 288 GB of HBM hold hundreds of basis vectors of a 1M-node
    state), then a backtracking line search on |g| (s = 1, 1/2, ... ; sufficient decrease 1e-4; f is piecewise smooth --
    ReLU kinks -- and full Newton steps overshoot far from the fixed point).  Everything runs in plan order on the device:
    f and the JVP are the tiled HIP kernels, Arnoldi (classical Gram-Schmidt twice), the Hessenberg / Givens least-squares
    problem and the update are csrc/krylov.hip; the host reads two norms per trial point and a stop flag every
    ``poll_every`` Krylov steps.  ``warm_start`` > 0: that many on-device Broyden iterations first (far from the fixed point
    the kinks make Newton directions poor: on the 2 107-node fixture a cold start spends most of its evaluations on damped
    steps, a start from Broyden's iterate converges quadratically).  When no step length along the Newton direction
    decreases |g|, up to ``picard_fallback`` plain iterations x <- f(x) are taken while they decrease it, and Newton resumes.
    ``broyden_burst`` > 0: whenever the accepted step length falls below ``burst_below`` (the iterate sits in a patch of
    kinks where the linear model is only valid very locally) that many Broyden iterations are run from the current iterate
    with the device solver's machinery, and their lowest iterate is taken if it is lower -- quasi-Newton steps cross such
    patches, Newton steps finish the job once the iterate is in the basin of quadratic convergence.

    ``rel = |g| / (|f(x)| + 1e-9)`` as in broyden; ``nstep`` = outer step of the lowest iterate, ``n_feval`` = f + JVP
    evaluations (the unit comparable to one Broyden iteration), ``n_krylov`` = Krylov steps per outer iteration."""
    if not isinstance(f, FixedPointMap):
        raise nat.NativeError("newton_krylov needs the analytic JVP of a FixedPointMap")
    nat.require_cuda(x0, "x0")
    if not (f.plan.tiled and (f.weights.mixed or f.weights.n_layers == 1)):
        raise nat.NativeError("newton_krylov runs on tiled plans whose block the tiled JVP kernel covers "
                              "(single-layer dirichlet; mixed of any depth)")
    from ..engine import DeviceGmres
    shape = x0.shape
    n_feval = 0
    if warm_start > 0:
        ob = broyden(f, x0, threshold=warm_start, eps=eps, keep_trace=False)
        x = f.to_plan(ob["result"])
        n_feval += ob["n_iter"] + 1
    else:
        x = f.to_plan(x0.to(torch.float32).contiguous())
    M = x.numel()
    gm = DeviceGmres(M, x.device, inner_m)
    # PSIGNN_NK_LIN=0 keeps the direct JVP kernel (A/B runs, tests)
    use_lin = os.environ.get("PSIGNN_NK_LIN", "1") != "0" and hasattr(f, "can_linearize") and f.can_linearize()
    lin = None
    neg_g = torch.empty_like(x)
    xt = torch.empty_like(x)
    trace = {"abs": [], "rel": []}
    n_krylov, steps = [], []
    xest_trace = [x]
    with torch.cuda.device(x.device):
        fx = f.fp(x)
        n_feval += 1
        gn, fn = gm.residual_norms(x, fx, neg_g=neg_g)
        rel = gn / (fn + 1e-9)
        lowest, lowest_x, lowest_step = rel, x, 0
        nstep = 0
        for nstep in range(1, threshold + 1):
            if rel < eps:
                nstep -= 1
                break
            gm.begin(neg_g)
            k = 0
            if use_lin:   # J_f(x) is applied to every Krylov vector of this step: linearise once (engine.Linearization)
                lin = f.linearize_p(x, lin)
            for j in range(inner_m):
                if use_lin:
                    lin.jvp_p(gm.row(j, x.shape), out=gm.row(j + 1, x.shape))
                else:
                    f.jvp_p(x, gm.row(j, x.shape), out=gm.row(j + 1, x.shape))
                n_feval += 1
                k = j + 1
                if gm.step(j, 1.0, inner_tol, poll=(k % poll_every == 0 or k == inner_m)):
                    break
            # x_trial = x + s dx ; accept the first s with |g(x_trial)| <= (1 - 1e-4 s) |g(x)|
            s, accepted = 1.0, False
            for _ in range(max_backtracks + 1):
                kk, _, _ = gm.solution(x, s, xt, info=True)
                fxt = f.fp(xt)
                n_feval += 1
                gt, ft = gm.residual_norms(xt, fxt, neg_g=neg_g)   # -g(x_trial): the next right-hand side if accepted
                if gt <= (1.0 - 1e-4 * s) * gn:
                    accepted = True
                    break
                s *= 0.5
            n_krylov.append(kk)
            steps.append(s if accepted else 0.0)
            if not accepted:
                # no decrease along the Newton direction: plain iterations x <- f(x) while they decrease |g|
                moved = False
                for _ in range(picard_fallback):
                    xt.copy_(f.fp(x))
                    fxt = f.fp(xt)
                    n_feval += 2
                    gt, ft = gm.residual_norms(xt, fxt, neg_g=None)
                    if not gt < gn:
                        break
                    x, xt = xt, torch.empty_like(x)
                    gn, fn = gt, ft
                    moved = True
                if not moved:   # stalled at this precision
                    break
                gm.residual_norms(x, f.fp(x), neg_g=neg_g)
                n_feval += 1
                steps[-1] = -1.0   # marks a fallback round
            else:
                x, xt = xt, torch.empty_like(x)
                gn, fn = gt, ft
            if broyden_burst > 0 and steps[-1] < burst_below:
                ob = broyden(lambda H: f.fp(H), x, threshold=broyden_burst, eps=eps, keep_trace=False)
                n_feval += ob["n_iter"] + 1
                xb = ob["result"].to(torch.float32).contiguous()
                gb, fb = gm.residual_norms(xb, f.fp(xb), neg_g=None)
                n_feval += 1
                if gb < gn:
                    x, gn, fn = xb, gb, fb
                    gm.residual_norms(x, f.fp(x), neg_g=neg_g)
                    n_feval += 1
            rel = gn / (fn + 1e-9)
            trace["abs"].append(gn)
            trace["rel"].append(rel)
            xest_trace.append(x)
            if rel < lowest:
                lowest, lowest_x, lowest_step = rel, x, nstep
    gm.close()
    if lin is not None:
        lin.close()
    back = f.from_plan
    return {"result": back(lowest_x).reshape(shape), "lowest": lowest, "nstep": lowest_step, "prot_break": False,
            "abs_trace": trace["abs"], "rel_trace": trace["rel"], "xest_trace": [back(t).reshape(shape) for t in xest_trace],
            "eps": eps, "threshold": threshold, "n_feval": n_feval, "n_outer": len(trace["rel"]), "n_krylov": n_krylov,
            "step_lengths": steps}

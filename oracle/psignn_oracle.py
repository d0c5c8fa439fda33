"""CPU oracle for the PSI-GNN fixed-point inference path.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU restatement, operation for operation, of the reference's
hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it; the product package (``psi-gnn_amd/``) never does.

What it restates (reference file:line, all under /root/reference):

* ``remove_self_loops`` + ``MessagePassing.propagate`` for ``Phi_to`` / ``Phi_from``
  (``dirichlet/psignn/model.py:334-368``; PyG semantics in SURVEY.md §8a-a3),
* ``MLP`` (``model.py:316-332``), ``Function.forward`` dirichlet (``model.py:279-300`` ≡
  ``tests/model_psignn.py:269-290``) and mixed (``mixed/psignn/model.py:216-245``),
* ``Encoder/Decoder`` (``model.py:370-392``), ``residual_loss`` (``model.py:157-167``),
* ``ModelPSIGNN.forward`` glue (``tests/model_psignn.py:58-100``),
* the solver library ``dirichlet/psignn/utilities/solver.py`` (broyden ``:116-207`` with
  ``matvec/rmatvec :96-114`` and the ``ls=False`` branch of ``line_search :61-94``;
  ``anderson :215-293``; ``forward_iteration :301-341``; ``newton :349-366``).

Pinning status
--------------
* Solvers: PINNED.  ``oracle/make_golden.py`` runs the reference's own ``solver.py`` (importable
  in the build container) and this restatement on identical inputs; traces, step counts and
  results are compared bit-for-bit there and the vectors are committed under ``tests/golden/``.
* ``Function.forward``: the arithmetic inside ``torch_geometric`` / ``torch_sparse`` (absent, unpinned
  versions) cannot be executed here and the reference's tests hold no numeric vector for it →
  **parity unpinned at bit level** for that boundary.  It is pinned semantically by (i) the PyG
  contract restated in SURVEY §8a-a3, (ii) the reference's trained checkpoints, and (iii) the
  reference's recorded statistics on in-distribution meshes, reproduced by this oracle on the real
  gmsh mesh recovered from ``tests/special_geo/mesh_files/original`` (``tests/test_oracle_golden.py``).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------
def _mlp(sd, prefix, x):
    """``MLP``: Linear, ReLU, Linear (model.py:316-332); keys ``<prefix>.mlp.{0,2}.{weight,bias}``."""
    x = F.linear(x, sd[prefix + ".mlp.0.weight"], sd[prefix + ".mlp.0.bias"])
    x = torch.relu(x)
    return F.linear(x, sd[prefix + ".mlp.2.weight"], sd[prefix + ".mlp.2.bias"])


def is_mixed_state_dict(sd) -> bool:
    return any(k.startswith("deqdss.f.phi_neumann") for k in sd)


def n_layers_of(sd) -> int:
    ks = [k for k in sd if k.startswith("deqdss.f.update_list.")]
    return 1 + max(int(k.split(".")[3]) for k in ks)


# --------------------------------------------------------------------------------------
# message passing
# --------------------------------------------------------------------------------------
def remove_self_loops(edge_index, edge_attr):
    """torch_geometric.utils.remove_self_loops: keep edges with row != col (model.py:342,360)."""
    keep = edge_index[0] != edge_index[1]
    return edge_index[:, keep], edge_attr[keep]


def phi(sd, prefix, h, edge_index, edge_attr, flow):
    """One ``Phi_to`` (flow='source_to_target') / ``Phi_from`` ('target_to_source') pass.

    PyG: (i, j) = (1, 0) for source_to_target else (0, 1); x_i = h[edge_index[i]],
    x_j = h[edge_index[j]]; message = MLP([x_i, x_j, edge_attr]); summed at edge_index[i].
    """
    ei, ea = remove_self_loops(edge_index, edge_attr)
    i, j = (1, 0) if flow == "source_to_target" else (0, 1)
    x_i = h.index_select(0, ei[i])
    x_j = h.index_select(0, ei[j])
    msg = _mlp(sd, prefix + ".mlp", torch.cat([x_i, x_j, ea], dim=1))
    out = torch.zeros_like(h)
    out.index_add_(0, ei[i], msg)
    return out


def function_forward(sd, h, h_initial, batch):
    """``Function.forward`` — dirichlet (model.py:279-300) or mixed (mixed/psignn/model.py:216-245),
    chosen by the state dict."""
    P = "deqdss.f."
    nl = n_layers_of(sd)
    if not is_mixed_state_dict(sd):
        idx_d = torch.where(batch.tags == 1)[0]
        for k in range(nl):
            mp_to = phi(sd, f"{P}phi_to_list.{k}", h, batch.edge_index, batch.edge_attr, "source_to_target")
            mp_from = phi(sd, f"{P}phi_from_list.{k}", h, batch.edge_index, batch.edge_attr, "target_to_source")
            cat = torch.cat([h, mp_to, mp_from, batch.prb_data], dim=1)
            alpha = torch.sigmoid(F.linear(cat, sd[P + "alpha.0.weight"], sd[P + "alpha.0.bias"]))
            upd = alpha * _mlp(sd, f"{P}update_list.{k}", cat)
            if k == nl - 1:
                h = F.layer_norm(h + upd, (h.shape[1],), sd[P + "laynorm.weight"], sd[P + "laynorm.bias"], 1e-5)
            else:
                h = h + upd
            h[idx_d, :] = h_initial[idx_d, :]
        return h
    # mixed: note the reference never reassigns h inside the loop
    idx_d = torch.where(batch.tags[:, 1] == 1)[0]
    idx_n = torch.where(batch.tags[:, 2] == 1)[0]
    h_next = h
    for k in range(nl):
        mp_to = phi(sd, f"{P}phi_to_list.{k}", h, batch.edge_index, batch.edge_attr, "source_to_target")
        mp_from = phi(sd, f"{P}phi_from_list.{k}", h, batch.edge_index, batch.edge_attr, "target_to_source")
        mp_neu = phi(sd, f"{P}phi_neumann", h, batch.edge_index, batch.edge_attr, "target_to_source")
        cat = torch.cat([h, mp_to, mp_from, batch.prb_data], dim=1)
        alpha = torch.sigmoid(F.linear(cat, sd[P + "alpha.0.weight"], sd[P + "alpha.0.bias"]))
        upd = alpha * _mlp(sd, f"{P}update_list.{k}", cat)
        cat_n = torch.cat([h, mp_neu, batch.prb_data, batch.unit_normal_vector], dim=1)
        upd_n = _mlp(sd, f"{P}update_neumann", cat_n)
        h_next = h + upd
        h_next[idx_n, :] = upd_n[idx_n, :]
        if k == nl - 1:
            h_next = F.layer_norm(h_next, (h.shape[1],), sd[P + "laynorm.weight"], sd[P + "laynorm.bias"], 1e-5)
        h_next[idx_d, :] = h_initial[idx_d, :]
    return h_next


def encoder(sd, x):
    return _mlp(sd, "autoencoder.encoder.mlp", x)


def decoder(sd, h):
    return _mlp(sd, "autoencoder.decoder.mlp", h)


def residual_loss(u, batch):
    """mean((A u - y)^2), A = COO(row=edge_index[0], col=edge_index[1], a_ij) incl. the diagonal
    (model.py:157-167)."""
    r, c = batch.edge_index[0], batch.edge_index[1]
    Au = torch.zeros_like(u)
    Au.index_add_(0, r, batch.a_ij.reshape(-1, 1) * u.index_select(0, c))
    return torch.mean((Au - batch.y) ** 2)


# --------------------------------------------------------------------------------------
# solvers  (dirichlet/psignn/utilities/solver.py)
# --------------------------------------------------------------------------------------
def _rmatvec(Us, VTs, x):  # solver.py:96-104
    if Us.nelement() == 0:
        return -x
    xTU = torch.einsum("bij, bijd -> bd", x, Us)
    return -x + torch.einsum("bd, bdij -> bij", xTU, VTs)


def _matvec(Us, VTs, x):  # solver.py:106-114
    if Us.nelement() == 0:
        return -x
    VTx = torch.einsum("bdij, bij -> bd", VTs, x)
    return -x + torch.einsum("bijd, bd -> bij", Us, VTx)


def _armijo(phi, phi0, slope0, c1=1e-4, s0=1, smin=0):
    """scalar_search_armijo, solver.py:20-59: first trial s0; then the minimiser of the quadratic through phi(0),
    phi'(0), phi(s0); then cubic interpolation through the last two trials until phi(s) <= phi0 + c1 s slope0, each new
    trial pulled back to s1 / 2 when it moved less than 4 % or more than 50 %; None below smin.  As in the reference the
    quadratic trial itself is never tested.  Returns (s, phi(s), cubic rounds)."""
    rounds = 0
    p0 = phi(s0)
    if p0 <= phi0 + c1 * s0 * slope0:
        return s0, p0, rounds
    s1 = -(slope0) * s0 ** 2 / 2.0 / (p0 - phi0 - slope0 * s0)
    p1 = phi(s1)
    while s1 > smin:
        scale = s0 ** 2 * s1 ** 2 * (s1 - s0)
        ca = s0 ** 2 * (p1 - phi0 - slope0 * s1) - s1 ** 2 * (p0 - phi0 - slope0 * s0)
        ca = ca / scale
        cb = -s0 ** 3 * (p1 - phi0 - slope0 * s1) + s1 ** 3 * (p0 - phi0 - slope0 * s0)
        cb = cb / scale
        s2 = (-cb + torch.sqrt(torch.abs(cb ** 2 - 3 * ca * slope0))) / (3.0 * ca)
        p2 = phi(s2)
        rounds += 1
        if p2 <= phi0 + c1 * s2 * slope0:
            return s2, p2, rounds
        if (s1 - s2) > s1 / 2.0 or (1 - s2 / s1) < 0.96:
            s2 = s1 / 2.0
        s0, s1, p0, p1 = s1, s2, p1, p2
    return None, p1, rounds


def _line_search(update, x, gx, g, on):
    """line_search, solver.py:61-94: step length by ``_armijo`` on phi(s) = |g(x + s update)|^2 with slope -phi(0) and
    smin = 1e-2 (s = 1 when ``on`` is false or the search fails); the residual at the accepted s is reused when it was the
    last one evaluated.  Returns (x_new, g_new, x_new - x, g_new - g)."""
    last = {"s": 0, "g": gx, "phi": torch.norm(gx) ** 2}

    def phi(s):
        if s == last["s"]:
            return last["phi"]
        gn = g((x + s * update)[0, :]).view_as(x)
        val = torch.norm(gn) ** 2 if torch.isfinite(gn).all() else np.inf
        last.update(s=s, g=gn, phi=val)
        return val

    s = None
    if on:
        s, _, _ = _armijo(phi, last["phi"], -last["phi"], smin=1e-2)
    if s is None:
        s = 1.0
    x_new = x + s * update
    g_new = last["g"] if s == last["s"] else g(x_new[0, :]).view_as(x)
    return x_new, g_new, x_new - x, g_new - gx


def broyden(f, x0, threshold, eps=1e-3, stop_mode="rel", ls=False, name="unknown"):
    """solver.py:116-207 (``ls=False`` is the only mode any caller of the reference uses)."""
    x0 = x0[None, :]
    bsz, total_hsize, seq_len = x0.size()
    g = lambda y: f(y) - y
    alt = "rel" if stop_mode == "abs" else "abs"
    x_est = x0
    gx = g(x_est[0, :]).view_as(x0)
    Us = torch.zeros(bsz, total_hsize, seq_len, threshold).to(x0.device)
    VTs = torch.zeros(bsz, threshold, total_hsize, seq_len).to(x0.device)
    update = -_matvec(Us[:, :, :, :0], VTs[:, :0], gx)
    prot_break = False
    protect_thres = (1e6 if stop_mode == "abs" else 1e3) * seq_len
    trace = {"abs": [], "rel": []}
    lowest = {"abs": 1e8, "rel": 1e8}
    lowest_step = {"abs": 0, "rel": 0}
    nstep, lowest_xest = 0, x_est[0, :]
    xest_trace = [x_est[0, :]]
    while nstep < threshold:
        x_est, gx, delta_x, delta_gx = _line_search(update, x_est, gx, g, ls)  # on=False: s = 1.0 (solver.py:85-94)
        xest_trace.append(x_est[0, :])
        nstep += 1
        abs_diff = torch.norm(gx).item()
        rel_diff = abs_diff / (torch.norm(gx + x_est).item() + 1e-9)
        diff = {"abs": abs_diff, "rel": rel_diff}
        trace["abs"].append(abs_diff)
        trace["rel"].append(rel_diff)
        for mode in ("rel", "abs"):
            if diff[mode] < lowest[mode]:
                if mode == stop_mode:
                    lowest_xest = x_est[0, :].clone().detach()
                lowest[mode] = diff[mode]
                lowest_step[mode] = nstep
        obj = diff[stop_mode]
        if obj < eps:
            break
        if obj < 3 * eps and nstep > 30 and np.max(trace[stop_mode][-30:]) / np.min(trace[stop_mode][-30:]) < 1.3:
            break
        if obj > trace[stop_mode][0] * protect_thres:
            prot_break = True
            break
        pU, pV = Us[:, :, :, :nstep - 1], VTs[:, :nstep - 1]
        vT = _rmatvec(pU, pV, delta_x)
        u = (delta_x - _matvec(pU, pV, delta_gx)) / torch.einsum("bij, bij -> b", vT, delta_gx)[:, None, None]
        vT[vT != vT] = 0
        u[u != u] = 0
        VTs[:, nstep - 1] = vT
        Us[:, :, :, nstep - 1] = u
        update = -_matvec(Us[:, :, :, :nstep], VTs[:, :nstep], gx)
    for _ in range(threshold + 1 - len(trace[stop_mode])):
        trace[stop_mode].append(lowest[stop_mode])
        trace[alt].append(lowest[alt])
    return {"result": lowest_xest, "lowest": lowest[stop_mode], "nstep": lowest_step[stop_mode],
            "prot_break": prot_break, "abs_trace": trace["abs"], "rel_trace": trace["rel"],
            "xest_trace": xest_trace, "eps": eps, "threshold": threshold}


def anderson(f, x0, m=2, lam=1e-4, threshold=50, eps=1e-3, stop_mode="rel", beta=1.0, **kwargs):
    """solver.py:215-293."""
    x0 = x0[None, :]
    bsz, d, L = x0.shape
    alt = "rel" if stop_mode == "abs" else "abs"
    X = torch.zeros(bsz, m, d * L, dtype=x0.dtype, device=x0.device)
    Fm = torch.zeros(bsz, m, d * L, dtype=x0.dtype, device=x0.device)
    X[:, 0] = x0.reshape(bsz, -1)
    Fm[:, 0] = f(x0[0, :]).view_as(x0).reshape(bsz, -1)
    X[:, 1] = Fm[:, 0]
    Fm[:, 1] = f(Fm[:, 0].reshape_as(x0)[0, :]).view_as(x0).reshape(bsz, -1)
    H = torch.zeros(bsz, m + 1, m + 1, dtype=x0.dtype, device=x0.device)
    H[:, 0, 1:] = H[:, 1:, 0] = 1
    y = torch.zeros(bsz, m + 1, 1, dtype=x0.dtype, device=x0.device)
    y[:, 0] = 1
    trace = {"abs": [], "rel": []}
    lowest = {"abs": 1e8, "rel": 1e8}
    lowest_step = {"abs": 0, "rel": 0}
    xest_trace = [x0[0, :]]
    lowest_xest = None
    for k in range(2, threshold):
        n = min(k, m)
        G = Fm[:, :n] - X[:, :n]
        H[:, 1:n + 1, 1:n + 1] = torch.bmm(G, G.transpose(1, 2)) + lam * torch.eye(n, dtype=x0.dtype, device=x0.device)[None]
        alpha = torch.linalg.solve(H[:, :n + 1, :n + 1], y[:, :n + 1])[:, 1:n + 1, 0]
        X[:, k % m] = beta * (alpha[:, None] @ Fm[:, :n])[:, 0] + (1 - beta) * (alpha[:, None] @ X[:, :n])[:, 0]
        Fm[:, k % m] = f(X[:, k % m].reshape_as(x0)[0, :]).view_as(x0).reshape(bsz, -1)
        gx = (Fm[:, k % m] - X[:, k % m]).view_as(x0)
        abs_diff = gx.norm().item()
        rel_diff = abs_diff / (1e-5 + Fm[:, k % m].norm().item())
        diff = {"abs": abs_diff, "rel": rel_diff}
        trace["abs"].append(abs_diff)
        trace["rel"].append(rel_diff)
        for mode in ("rel", "abs"):
            if diff[mode] < lowest[mode]:
                if mode == stop_mode:
                    lowest_xest = X[:, k % m].view_as(x0).clone().detach()
                lowest[mode] = diff[mode]
                lowest_step[mode] = k
        xest_trace.append(lowest_xest[0, :])
        if trace[stop_mode][-1] < eps:
            for _ in range(threshold - 1 - k):
                trace[stop_mode].append(lowest[stop_mode])
                trace[alt].append(lowest[alt])
            break
    return {"result": lowest_xest[0, :], "lowest": lowest[stop_mode], "nstep": lowest_step[stop_mode],
            "prot_break": False, "abs_trace": trace["abs"], "rel_trace": trace["rel"],
            "xest_trace": xest_trace, "eps": eps, "threshold": threshold}


def forward_iteration(f, z0, eps=1.e-5, threshold=50):
    """solver.py:301-341."""
    z_est = [z0]
    z_prev = z0
    z = f(z0)
    trace = {"abs": [], "rel": []}
    ite = 0
    abs_res = torch.linalg.norm(z_prev - z)
    rel_res = abs_res / torch.linalg.norm(z)
    trace["abs"].append(abs_res.detach())
    trace["rel"].append(rel_res.detach())
    z_est.append(z)
    while trace["rel"][-1] > eps and ite < threshold:
        z_prev = z
        z = f(z_prev)
        ite += 1
        abs_res = torch.linalg.norm(z_prev - z)
        rel_res = abs_res / torch.linalg.norm(z)
        trace["abs"].append(abs_res.detach())
        trace["rel"].append(rel_res.detach())
        z_est.append(z)
    return {"result": z, "lowest": trace["rel"][-1], "abs_trace": trace["abs"], "rel_trace": trace["rel"],
            "xest_trace": z_est, "nstep": ite, "eps": eps, "threshold": threshold}


def newton(f, z0, eps=1.e-5, threshold=50):
    """solver.py:349-366 — node-block-diagonal Newton wrapped in forward_iteration."""
    f_root = lambda z: f(z) - z
    g = lambda z: z - torch.linalg.solve(
        torch.einsum("bibj->bij", torch.autograd.functional.jacobian(f_root, z)), f_root(z))
    r = forward_iteration(g, z0, eps=eps, threshold=threshold)
    return {"result": r["result"], "lowest": r["lowest"], "rel_trace": r["rel_trace"],
            "abs_trace": r["abs_trace"], "xest_trace": r["xest_trace"], "nstep": r["nstep"],
            "eps": eps, "threshold": threshold}


# --------------------------------------------------------------------------------------
# model glue (tests/model_psignn.py:58-100)
# --------------------------------------------------------------------------------------
def model_forward(sd, batch, solver=broyden, fw_tol=1e-5, fw_thres=500):
    """``ModelPSIGNN.forward`` → (u_final, loss_dic, solver_out)."""
    mixed = is_mixed_state_dict(sd)
    with torch.no_grad():
        h0 = encoder(sd, batch.x)
        out = solver(lambda H: function_forward(sd, H, h0, batch), h0, threshold=fw_thres, eps=fw_tol)
        h = out["result"]
        u = decoder(sd, h)
        mse = torch.nn.functional.mse_loss
        idx_d = torch.where((batch.tags[:, 1] if mixed else batch.tags[:, 0]) == 1)[0]
        loss = {
            "residual_loss": residual_loss(u, batch),
            "encoder_loss": mse(encoder(sd, u), h),
            "autoencoder_loss": mse(decoder(sd, encoder(sd, u)), u),
            "mse_loss": mse(u, batch.sol),
            "mse_dirichlet_loss": mse(u[idx_d, :], batch.x[idx_d, :]),
            "nsteps": out["nstep"],
        }
    return u, loss, out


def function_jvp(sd, h, h_initial, batch, v):
    """Directional derivative of ``Function.forward`` w.r.t. h along v (autograd forward-mode via
    double-backward trick is avoided: torch.autograd.functional.jvp)."""
    fn = lambda H: function_forward(sd, H, h_initial, batch)
    _, jv = torch.autograd.functional.jvp(fn, (h,), (v,))
    return jv


def function_vjp(sd, h, h_initial, batch, v):
    """vᵀ J of ``Function.forward`` (what ``autograd.grad(new_H, H, v)`` computes, model.py:214,432,449)."""
    hh = h.detach().clone().requires_grad_(True)
    out = function_forward(sd, hh, h_initial, batch)
    return torch.autograd.grad(out, hh, v)[0]


# --------------------------------------------------------------------------------------
# training side (dirichlet/psignn/model.py:58-99,184-225; training_class.py:146-166)
# --------------------------------------------------------------------------------------
def function_param_vjp(sd, h, h_initial, batch, v):
    """What ``loss.backward()`` leaves behind for ``new_H = f(H, H_init)`` with cotangent v: ({deqdss.f.* name: grad},
    vᵀ∂f/∂H, vᵀ∂f/∂H_init).  Plain autograd on the restated f."""
    p = {k: (t.detach().clone().requires_grad_(True) if k.startswith("deqdss.f.") else t) for k, t in sd.items()}
    hh = h.detach().clone().requires_grad_(True)
    h0 = h_initial.detach().clone().requires_grad_(True)
    out = function_forward(p, hh, h0, batch)
    names = [k for k in p if k.startswith("deqdss.f.")]
    g = torch.autograd.grad(out, [p[k] for k in names] + [hh, h0], v, allow_unused=True)
    return {k[len("deqdss.f."):]: t for k, t in zip(names, g[:-2])}, g[-2], g[-1]


def function_vjp_backward(sd, h, h_initial, batch, v, gbar):
    """Double backward of the VJP, as ``loss.backward()`` runs it for the Jacobian regulariser (jac_loss_estimate,
    dirichlet/psignn/model.py:416-435: ``autograd.grad(f0, z0, v, create_graph=True)``): gradient of
    gbar . (J_f(h)ᵀ v), gbar constant, w.r.t. the ``deqdss.f.*`` parameters and h.  Returns ({name: grad}, d/dh, Jᵀv)."""
    p = {k: (t.detach().clone().requires_grad_(True) if k.startswith("deqdss.f.") else t) for k, t in sd.items()}
    hh = h.detach().clone().requires_grad_(True)
    out = function_forward(p, hh, h_initial.detach(), batch)
    g = torch.autograd.grad(out, hh, v, create_graph=True)[0]
    names = [k for k in p if k.startswith("deqdss.f.")]
    gr = torch.autograd.grad((g * gbar).sum(), [p[k] for k in names] + [hh], allow_unused=True)
    zero = lambda k, t: torch.zeros_like(p[k]) if t is None else t
    return {k[len("deqdss.f."):]: zero(k, t) for k, t in zip(names, gr[:-1])}, gr[-1], g.detach()


def training_step(sd, batch, solver=broyden, fw_tol=1e-5, fw_thres=500, bw_tol=1e-8, bw_thres=500, jac_weight=0.0,
                  probe=None):
    """One training forward + ``loss.backward()`` of the dirichlet or mixed model (chosen by the state dict; the two
    ``ModelDEQDSS.forward`` differ only in the tag column of the Dirichlet rows): returns (loss, loss_dic,
    {state_dict name: grad}, forward dict, backward dict).  ``jac_weight`` (utilities/utils.py:58 default 0; the launch
    scripts use 1.0) weights the Jacobian regulariser ``|probeᵀ J|² / (N d)`` built with ``create_graph=True``
    (jac_loss_estimate, model.py:416-435; ``probe`` stands for its ``torch.randn``).

    Restates ``ModelDEQDSS.forward`` (model.py:58-99), ``DeepEquilibrium.forward`` incl. the backward hook that swaps
    the incoming gradient for the solution of y = Jᵀy + grad (model.py:184-225) and the loss combination of
    ``TrainModel.train_loop`` (training_class.py:155-159)."""
    p = {k: t.detach().clone().requires_grad_(True) for k, t in sd.items()}
    mse = torch.nn.functional.mse_loss
    h_init = encoder(p, batch.x)
    with torch.no_grad():
        out_fw = solver(lambda H: function_forward(p, H, h_init, batch), h_init, threshold=fw_thres, eps=fw_tol)
    H_star = out_fw["result"].detach().clone().requires_grad_()
    new_H = function_forward(p, H_star, h_init, batch)
    state = {}
    jac_loss = torch.zeros(())
    if probe is not None:   # model.py:207, before the hook is registered
        vJ = torch.autograd.grad(new_H, H_star, probe, retain_graph=True, create_graph=True)[0]
        jac_loss = vJ.norm() ** 2 / H_star.numel()

    def backward_hook(grad):
        state["hook"].remove()   # the reference removes the hook first so that the VJPs below do not re-enter it
        out_bw = solver(lambda y: torch.autograd.grad(new_H, H_star, y, retain_graph=True)[0] + grad,
                        torch.zeros_like(grad), threshold=bw_thres, eps=bw_tol)
        state["out_bw"] = out_bw
        return out_bw["result"]

    state["hook"] = new_H.register_hook(backward_hook)
    u = decoder(p, new_H)
    loss_dic = {"residual_loss": residual_loss(u, batch)}
    u_d, h_d = u.detach(), new_H.detach()
    loss_dic["encoder_loss"] = mse(encoder(p, u_d), h_d)
    loss_dic["autoencoder_loss"] = mse(decoder(p, encoder(p, u_d).detach()), u_d)
    loss_dic["mse_loss"] = mse(u, batch.sol)
    idx = torch.where(batch.tags == 1)[0]   # model.py:87 (mixed: one-hot tags -> every row)
    loss_dic["mse_dirichlet"] = mse(u[idx, :], batch.x[idx, :])
    loss_dic["jacobian_loss"] = jac_loss
    loss = loss_dic["residual_loss"] + jac_weight * jac_loss + loss_dic["encoder_loss"] + loss_dic["autoencoder_loss"]
    loss.backward()
    grads = {k: (t.grad if t.grad is not None else torch.zeros_like(t)) for k, t in p.items()}
    return loss.detach(), {k: v.detach() for k, v in loss_dic.items()}, grads, out_fw, state.get("out_bw")


# --------------------------------------------------------------------------------------
# DS-GPS: the unrolled recurrent baseline (dirichlet/dsgps/model.py:28-163), SURVEY §8f-4
# --------------------------------------------------------------------------------------
def dsgps_step(sd, h, h0, batch):
    """One update of ``ModelDSGPS.inference`` (dirichlet/dsgps/model.py:141-155): the same two message passes as
    PSI-GNN, then a GRU-style node update with three Linear(32, 10) gates; Dirichlet rows <- H['0'] rows."""
    mess_to = phi(sd, "phi_to", h, batch.edge_index, batch.edge_attr, "source_to_target")
    mess_from = phi(sd, "phi_from", h, batch.edge_index, batch.edge_attr, "target_to_source")
    cat = torch.cat([h, mess_to, mess_from, batch.prb_data], dim=1)
    alpha = torch.sigmoid(F.linear(cat, sd["z_k.mlp.0.weight"], sd["z_k.mlp.0.bias"]))
    reset = torch.sigmoid(F.linear(cat, sd["r_k.mlp.0.weight"], sd["r_k.mlp.0.bias"]))
    cat2 = torch.cat([reset * h, mess_to, mess_from, batch.prb_data], dim=1)
    corr = torch.tanh(F.linear(cat2, sd["correction.mlp.0.weight"], sd["correction.mlp.0.bias"]))
    h_next = h + alpha * corr
    if "phi_neumann.mlp.mlp.0.weight" in sd:   # mixed/dsgps/model.py:79-93: Neumann rows replaced, then Dirichlet rows
        mp_n = phi(sd, "phi_neumann", h, batch.edge_index, batch.edge_attr, "target_to_source")
        upd_n = _mlp(sd, "update_neumann", torch.cat([h, mp_n, batch.prb_data, batch.unit_normal_vector], dim=1))
        idx_n = torch.where(batch.tags[:, 2] == 1)[0]
        h_next[idx_n, :] = upd_n[idx_n, :]
        idx_d = torch.where(batch.tags[:, 1] == 1)[0]
    else:
        idx_d = torch.where(batch.tags == 1)[0]
    h_next[idx_d, :] = h0[idx_d, :]
    return h_next


def dsgps_inference(sd, batch, k, trace=False):
    """``ModelDSGPS.inference`` (model.py:130-163): U_k = decoder(H_k) after k updates from H_0 = encoder(x).
    With ``trace`` also the per-step diagnostics ``forward`` records (model.py:64-66,93-97,113): residual / mse of
    every decoded iterate."""
    with torch.no_grad():
        h0 = encoder(sd, batch.x)
        h = h0
        res, mse = [float(residual_loss(batch.x, batch))], [float(F.mse_loss(batch.x, batch.sol))]
        for _ in range(k):
            h = dsgps_step(sd, h, h0, batch)
            if trace:
                u = decoder(sd, h)
                res.append(float(residual_loss(u, batch)))
                mse.append(float(F.mse_loss(u, batch.sol)))
        u = decoder(sd, h)
    return (u, h, res, mse) if trace else (u, h)


def dsgps_training_step(sd, batch, k, gamma):
    """``ModelDSGPS.forward`` in train mode + ``loss.backward()`` on ``train_loss`` (dirichlet/dsgps/model.py:48-128,
    training_class.py:141-146): residual of every decoded iterate weighted gamma^(k-t-1), the encoder term with the
    decoder's parameters frozen (:100-105), the autoencoder term with the encoder's frozen (:107-112); the mixed family
    (mixed/dsgps/model.py:100-107) uses detached iterates for those two terms instead.  Plain autograd
    through the k restated updates.  Returns (train_loss, {name: grad}, last residual / mse)."""
    p = {n: (t.detach().clone().requires_grad_(True) if torch.is_tensor(t) and t.is_floating_point() and t.dim() > 0 else t)
         for n, t in sd.items()}
    froz = lambda prefix: {n: (t.detach() if n.startswith(prefix) else t) for n, t in p.items()}
    p_dec_frozen, p_enc_frozen = froz("autoencoder.decoder."), froz("autoencoder.encoder.")
    mixed = "phi_neumann.mlp.mlp.0.weight" in sd
    H0 = encoder(p, batch.x)
    h, total = H0, None
    for t in range(k):
        h = dsgps_step(p, h, H0, batch)
        u = decoder(p, h)
        res = residual_loss(u, batch)
        if mixed:   # mixed/dsgps/model.py:100-107
            u_d, h_d = u.detach(), h.detach()
            enc = F.mse_loss(encoder(p, u_d), h_d)
            aenc = F.mse_loss(decoder(p, encoder(p, u_d).detach()), u_d)
        else:
            enc = F.mse_loss(encoder(p, decoder(p_dec_frozen, h)), h)
            aenc = F.mse_loss(decoder(p, encoder(p_enc_frozen, u)), u)
        term = res * gamma ** (k - t - 1) + enc + aenc
        total = term if total is None else total + term
    total.backward()
    grads = {n: (t.grad if t.grad is not None else torch.zeros_like(t)) for n, t in p.items()
             if torch.is_tensor(t) and t.requires_grad}
    return total.detach(), grads, float(res.detach()), float(F.mse_loss(u, batch.sol).detach())


# --------------------------------------------------------------------------------------
# DSS: Deep Statistical Solver baseline (dirichlet/dss/model.py:25-147), SURVEY §8f-4
# --------------------------------------------------------------------------------------
def dss_batch(mesh):
    """The DSS view of a Poisson problem (dirichlet/dataset/generate_data.py:100-128 + dss/utilities/reader.py:61-92):
    graph = off-diagonal entries of A (Dirichlet rows are identity rows -> no out-edges), scalar edge feature a_ij
    normalised with the reference constants, node input b' = (b, 0, 0) on interior rows / (0, 1, g) on Dirichlet rows."""
    ei = mesh.edge_index
    keep = ei[0] != ei[1]
    a = mesh.a_ij.reshape(-1, 1)[keep]
    diri = (mesh.tags.reshape(-1) == 1)
    b = mesh.y.reshape(-1)
    bp = torch.stack([torch.where(diri, torch.zeros_like(b), b), diri.to(b.dtype),
                      torch.where(diri, b, torch.zeros_like(b))], dim=1)
    mean = torch.tensor([0.0002, 0.1435, -0.0006], dtype=b.dtype)
    std = torch.tensor([0.0507, 0.3506, 3.2935], dtype=b.dtype)
    out = type(mesh)(x=mesh.sol, sol=mesh.sol, edge_index=ei[:, keep], a_ij=a, a_ij_norm=(a + 0.5838) / 0.0924,
                     b_prime=bp, b_prime_norm=(bp - mean) / std, pos=mesh.pos, tags=mesh.tags)
    return out


def dss_residual_loss(U, edge_index, a_ij, y):
    """dirichlet/dss/model.py:122-139."""
    B0, B1, B2 = y[:, 0:1], y[:, 1:2], y[:, 2:3]
    p1 = (1 - B1) * (-B0) + B1 * (U - B2)
    fbar = a_ij * (U.index_select(0, edge_index[1]) - U.index_select(0, edge_index[0]))
    s = torch.zeros_like(U).index_add_(0, edge_index[0], fbar)
    return torch.mean((p1 + s) ** 2)


def dss_step(sd, t, h, batch, alpha):
    """Update t of ``DeepStatisticalSolver.inference`` (model.py:107-117): per-step Phi_to / Phi_from (edge input
    [x_i | x_j | a_ij_norm], 21 wide) and Psi MLP on [h | mess_to | mess_from | b'_norm] (33 wide)."""
    mess_to = phi(sd, f"phi_to_list.{t}", h, batch.edge_index, batch.a_ij_norm, "source_to_target")
    mess_from = phi(sd, f"phi_from_list.{t}", h, batch.edge_index, batch.a_ij_norm, "target_to_source")
    cat = torch.cat([h, mess_to, mess_from, batch.b_prime_norm], dim=1)
    return h + alpha * _mlp(sd, f"psi_list.{t}.mlp", cat)


def dss_inference(sd, batch, k, alpha, trace=False):
    """``DeepStatisticalSolver.inference`` (model.py:97-120): H_0 = 0, k updates with per-step weights, U_k from the
    k-th decoder."""
    with torch.no_grad():
        h = torch.zeros(batch.x.shape[0], sd["phi_to_list.0.mlp.mlp.2.bias"].numel(), dtype=batch.x.dtype)
        res = []
        for t in range(k):
            h = dss_step(sd, t, h, batch, alpha)
            if trace:
                res.append(float(dss_residual_loss(_mlp(sd, f"decoder_list.{t}.mlp", h), batch.edge_index, batch.a_ij, batch.b_prime)))
        u = _mlp(sd, f"decoder_list.{k - 1}.mlp", h)
    return (u, h, res) if trace else (u, h)


def dss_training_step(sd, batch, k, alpha, gamma):
    """``DeepStatisticalSolver.forward`` + ``loss.backward()`` on ``train_loss`` (dirichlet/dss/model.py:59-95,
    training_class.py): sum over the updates of gamma^(k-t-1) residual(decoder_t(H_{t+1})).  Plain autograd through the
    restated updates.  Returns (train_loss, {name: grad}, last residual)."""
    p = {n: (t.detach().clone().requires_grad_(True) if torch.is_tensor(t) and t.is_floating_point() and t.dim() > 0 else t)
         for n, t in sd.items()}
    h = torch.zeros(batch.x.shape[0], p["phi_to_list.0.mlp.mlp.2.bias"].numel(), dtype=batch.x.dtype)
    total = None
    for t in range(k):
        h = dss_step(p, t, h, batch, alpha)
        res = dss_residual_loss(_mlp(p, f"decoder_list.{t}.mlp", h), batch.edge_index, batch.a_ij, batch.b_prime)
        term = res * gamma ** (k - t - 1)
        total = term if total is None else total + term
    total.backward()
    grads = {n: (t.grad if t.grad is not None else torch.zeros_like(t)) for n, t in p.items()
             if torch.is_tensor(t) and t.requires_grad}
    return total.detach(), grads, float(res.detach())

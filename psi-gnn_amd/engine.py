"""Host-side handles of the HIP hot path: mesh plan, packed weights, f / JVP, device Broyden.

Everything numerical happens in libpsignn_hip.so; this module only owns lifetimes, packs the
reference ``state_dict`` into the flat weight buffer (layout: csrc/common.h ``WLayout``) and passes
device pointers.
"""
from __future__ import annotations

import ctypes as C
import weakref

import numpy as np
import torch

from . import _native as nat

D = nat.D


# ---------------------------------------------------------------------------------------------
# weights
# ---------------------------------------------------------------------------------------------
def is_mixed_state_dict(sd) -> bool:
    return any(k.startswith("deqdss.f.phi_neumann") for k in sd)


def n_layers_of(sd) -> int:
    return 1 + max(int(k.split(".")[3]) for k in sd if k.startswith("deqdss.f.update_list."))


def pack_weights(sd, device=None) -> torch.Tensor:
    """Flatten the ``deqdss.f.*`` tensors of a reference state_dict into the kernel layout.

    Key layout of the reference: dirichlet/psignn/model.py:265-277 (mixed/psignn/model.py:198-214),
    SURVEY §8b.  Order (csrc/common.h): shared{ln_gamma, ln_beta, alpha_w, alpha_b} padded to 64;
    per layer phi_to{W1,b1,W2,b2} phi_from{..} update{U1,c1,U2,c2}; mixed tail phi_neumann, update_neumann.
    """
    mixed = is_mixed_state_dict(sd)
    nl = n_layers_of(sd)
    P = "deqdss.f."
    g = lambda k: sd[P + k].detach().to("cpu", torch.float32).reshape(-1)
    p = 3 if mixed else 2
    if sd[P + "laynorm.weight"].numel() != D:
        raise nat.NativeError(f"latent_dim {sd[P + 'laynorm.weight'].numel()} != {D}: the HIP kernels are built for d = {D}")
    if sd[P + "alpha.0.weight"].numel() != 3 * D + p:
        raise nat.NativeError("alpha gate width does not match the boundary-condition family")
    shared = torch.cat([g("laynorm.weight"), g("laynorm.bias"), g("alpha.0.weight"), g("alpha.0.bias")])
    parts = [torch.nn.functional.pad(shared, (0, 64 - shared.numel()))]
    for l in range(nl):
        for phi in ("phi_to_list", "phi_from_list"):
            parts += [g(f"{phi}.{l}.mlp.mlp.0.weight"), g(f"{phi}.{l}.mlp.mlp.0.bias"),
                      g(f"{phi}.{l}.mlp.mlp.2.weight"), g(f"{phi}.{l}.mlp.mlp.2.bias")]
        parts += [g(f"update_list.{l}.mlp.0.weight"), g(f"update_list.{l}.mlp.0.bias"),
                  g(f"update_list.{l}.mlp.2.weight"), g(f"update_list.{l}.mlp.2.bias")]
        # fold block: second Phi layer pushed through the consumers of mp_to / mp_from (float64, rounded once)
        m = lambda k: sd[P + k].detach().to("cpu", torch.float64)
        U1, wa = m(f"update_list.{l}.mlp.0.weight"), m("alpha.0.weight").reshape(-1)
        fold = []
        for phi, c0 in (("phi_to_list", D), ("phi_from_list", 2 * D)):
            W2, b2 = m(f"{phi}.{l}.mlp.mlp.2.weight"), m(f"{phi}.{l}.mlp.mlp.2.bias")
            fold += [(U1[:, c0:c0 + D] @ W2).reshape(-1), U1[:, c0:c0 + D] @ b2]
        for phi, c0 in (("phi_to_list", D), ("phi_from_list", 2 * D)):
            fold.append(wa[c0:c0 + D] @ m(f"{phi}.{l}.mlp.mlp.2.weight"))
        for phi, c0 in (("phi_to_list", D), ("phi_from_list", 2 * D)):
            fold.append((wa[c0:c0 + D] @ m(f"{phi}.{l}.mlp.mlp.2.bias")).reshape(1))
        fold = torch.cat(fold).to(torch.float32)
        parts.append(torch.nn.functional.pad(fold, (0, 244 - fold.numel())))
    if mixed:
        parts += [g("phi_neumann.mlp.mlp.0.weight"), g("phi_neumann.mlp.mlp.0.bias"),
                  g("phi_neumann.mlp.mlp.2.weight"), g("phi_neumann.mlp.mlp.2.bias"),
                  g("update_neumann.mlp.0.weight"), g("update_neumann.mlp.0.bias"),
                  g("update_neumann.mlp.2.weight"), g("update_neumann.mlp.2.bias")]
        m = lambda k: sd[P + k].detach().to("cpu", torch.float64)
        N1 = m("update_neumann.mlp.0.weight")
        nf = torch.cat([(N1[:, D:2 * D] @ m("phi_neumann.mlp.mlp.2.weight")).reshape(-1),
                        N1[:, D:2 * D] @ m("phi_neumann.mlp.mlp.2.bias")]).to(torch.float32)
        parts.append(torch.nn.functional.pad(nf, (0, 112 - nf.numel())))
    # ---- transposed section for the tile kernel: [in k][out o] blocks (WLayout::T_* / N_*)
    m = lambda k: sd[P + k].detach().to("cpu", torch.float64)
    f32 = lambda t: t.to(torch.float32).reshape(-1)

    def padto(t, n):
        t = f32(t)
        return torch.nn.functional.pad(t, (0, n - t.numel()))
    mir = torch.tensor([-1.0, -1.0, 1.0], dtype=torch.float64)[:, None]  # in-edge attr = mirror of the stored out-edge attr
    for l in range(nl):
        Wt, Wf = m(f"phi_to_list.{l}.mlp.mlp.0.weight"), m(f"phi_from_list.{l}.mlp.mlp.0.weight")
        U1, U2 = m(f"update_list.{l}.mlp.0.weight"), m(f"update_list.{l}.mlp.2.weight")
        G_to = U1[:, D:2 * D] @ m(f"phi_to_list.{l}.mlp.mlp.2.weight")
        G_fr = U1[:, 2 * D:3 * D] @ m(f"phi_from_list.{l}.mlp.mlp.2.weight")
        g_to = U1[:, D:2 * D] @ m(f"phi_to_list.{l}.mlp.mlp.2.bias")
        g_fr = U1[:, 2 * D:3 * D] @ m(f"phi_from_list.{l}.mlp.mlp.2.bias")
        tp = [f32(Wt[:, D:2 * D].t()), f32(Wf[:, D:2 * D].t()), f32(Wt[:, :D].t()), f32(Wf[:, :D].t()),
              f32(Wt[:, 2 * D:].t() * mir), f32(Wf[:, 2 * D:].t()),
              f32(m(f"phi_to_list.{l}.mlp.mlp.0.bias")), f32(m(f"phi_from_list.{l}.mlp.mlp.0.bias")),
              f32(U1[:, :D].t()), f32(G_to.t()), f32(G_fr.t()), padto(U1[:, 3 * D:].t(), 30),
              f32(m(f"update_list.{l}.mlp.0.bias")), f32(g_to), f32(g_fr), f32(U2.t()),
              f32(m(f"update_list.{l}.mlp.2.bias"))]
        tp = torch.cat(tp)
        assert tp.numel() == 950
        parts.append(tp)
    if mixed:
        Wn, N1, N2 = m("phi_neumann.mlp.mlp.0.weight"), m("update_neumann.mlp.0.weight"), m("update_neumann.mlp.2.weight")
        Gn = N1[:, D:2 * D] @ m("phi_neumann.mlp.mlp.2.weight")
        gn = N1[:, D:2 * D] @ m("phi_neumann.mlp.mlp.2.bias")
        tn = torch.cat([f32(Wn[:, D:2 * D].t()), f32(Wn[:, :D].t()), f32(Wn[:, 2 * D:].t()),
                        f32(m("phi_neumann.mlp.mlp.0.bias")), f32(N1[:, :D].t()), f32(Gn.t()),
                        padto(N1[:, 2 * D:].t(), 50), f32(m("update_neumann.mlp.0.bias")), f32(gn), f32(N2.t()),
                        f32(m("update_neumann.mlp.2.bias"))])
        assert tn.numel() == 620
        parts.append(tn)
    flat = torch.cat(parts).contiguous()
    return flat if device is None else flat.to(device)


class PackedWeights:
    def __init__(self, sd, device):
        self.mixed = is_mixed_state_dict(sd)
        self.n_layers = n_layers_of(sd)
        self.flat = pack_weights(sd, device)
        expect = nat.lib().psignn_weights_size(int(self.mixed), self.n_layers)
        if self.flat.numel() != expect:
            raise nat.NativeError(f"packed weight length {self.flat.numel()} != native layout {expect}")


# ---------------------------------------------------------------------------------------------
# mesh plan
# ---------------------------------------------------------------------------------------------
_EXPORT = {"csr_ptr": (0, np.int32), "csr_nbr": (1, np.int32), "csr_eid": (2, np.int32),
           "csc_ptr": (3, np.int32), "csc_nbr": (4, np.int32), "csc_eid": (5, np.int32),
           "node_flags": (6, np.uint8), "csr_attr": (7, np.float32), "csc_attr": (8, np.float32),
           "a_ptr": (9, np.int32), "a_col": (10, np.int32), "a_val": (11, np.float32),
           "perm": (12, np.int32), "tile_ptr": (13, np.int32), "halo_cnt": (14, np.int32), "halo": (15, np.int32),
           "slice_off": (16, np.int32), "slice_deg": (17, np.uint8), "ell": (18, np.uint32),
           "tile_slice": (20, np.int32)}

HALO_CAP = 512


class MeshPlan:
    """Iteration-invariant device data of one mesh (or a disjoint union).  See csrc/plan.hip."""

    def __init__(self, batch, tile_target=0):
        """tile_target: nodes per tile (0 = library default, < 0 = untiled global-gather kernels)."""
        ei = batch.edge_index
        nat.require_cuda(ei, "batch.edge_index")
        self.device = ei.device
        N = int(batch.x.shape[0])
        tags = batch.tags.to(torch.float32).contiguous()
        if tags.dim() == 1:
            tags = tags[:, None]
        ea = batch.edge_attr.to(torch.float32).contiguous()
        aij = getattr(batch, "a_ij", None)
        aij = None if aij is None else aij.to(torch.float32).reshape(-1).contiguous()
        eic = ei.to(torch.int64).contiguous()
        if ea.shape[0] != eic.shape[1] or ea.shape[1] != 3:
            raise nat.NativeError(f"edge_attr shape {tuple(ea.shape)} does not match edge_index {tuple(eic.shape)}")
        if tags.shape[0] != N:
            raise nat.NativeError("tags and x disagree on the node count")
        pos = getattr(batch, "pos", None)
        if pos is not None:
            pos = pos.to(torch.float32).contiguous()
            if pos.shape != (N, 2) or pos.device != self.device:
                pos = None  # coordinates only steer the tiling; without them the given numbering is kept
        gid = getattr(batch, "batch", None)
        if pos is not None and gid is not None and gid.numel() == N and int(gid.max()) > 0:
            # disjoint union of graphs (PyG Batch): the graphs overlap in space, which would mix them inside a tile.
            # Shift graph g by g bounding-box widths for the TILING only (positions never enter the arithmetic).
            span = (pos[:, 0].max() - pos[:, 0].min()) * 1.05 + 1e-6
            pos = pos.clone()
            pos[:, 0] += gid.to(pos.dtype) * span
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            nat.check(nat.lib().psignn_plan_create(C.byref(h), N, eic.shape[1], nat.ptr(eic), nat.ptr(ea),
                                                   nat.ptr(aij), nat.ptr(tags), tags.shape[1], nat.ptr(pos),
                                                   int(tile_target), nat.stream_ptr(self.device)),
                      "psignn_plan_create")
        self.handle = h
        self._fin = weakref.finalize(self, nat.lib().psignn_plan_destroy, h)
        self.N = N
        self.E = int(eic.shape[1])
        self.Ep = int(nat.lib().psignn_plan_num_nonself_edges(h))
        self.mixed = tags.shape[1] == 3
        self.tiled = bool(nat.lib().psignn_plan_is_tiled(h))
        self.n_tiles = int(nat.lib().psignn_plan_num_tiles(h))
        self.ell_rows = int(nat.lib().psignn_plan_ell_rows(h))
        self.max_tile_rows = int(nat.lib().psignn_plan_max_tile_rows(h))
        self._work = None
        self._pwork = None
        self._jwork = None

    def export(self, name):
        which, dt = _EXPORT[name]
        if which >= 12 and not self.tiled:
            raise nat.NativeError("plan has no tile structures")
        n_slices = None
        if name in ("slice_off", "slice_deg"):
            n_slices = int(self.export("tile_slice")[-1])
        n = {"csr_ptr": self.N + 1, "csc_ptr": self.N + 1, "a_ptr": self.N + 1, "node_flags": self.N,
             "csr_attr": 3 * self.Ep, "csc_attr": 3 * self.Ep, "a_col": self.E, "a_val": self.E,
             "perm": self.N, "tile_ptr": self.n_tiles + 1, "tile_slice": self.n_tiles + 1, "halo_cnt": self.n_tiles,
             "halo": self.n_tiles * HALO_CAP, "slice_off": (n_slices or 0) + 1, "slice_deg": (n_slices or 0),
             "ell": self.ell_rows * 64 * 4}.get(name, self.Ep)
        out = np.empty(n, dtype=dt)
        nat.check(nat.lib().psignn_plan_export(self.handle, which, out.ctypes.data_as(C.c_void_p), out.nbytes),
                  "psignn_plan_export")
        return out

    def workspace(self):
        if self._work is None:
            n = int(nat.lib().psignn_f_workspace_floats(self.handle))
            self._work = torch.empty(n, dtype=torch.float32, device=self.device)
        return self._work

    def pgrad_workspace(self):
        if self._pwork is None:
            n = int(nat.lib().psignn_f_param_vjp_workspace_floats(self.handle))
            self._pwork = torch.empty(n, dtype=torch.float32, device=self.device)
        return self._pwork

    def vjp_backward_workspace(self):
        if self._jwork is None:
            n = int(nat.lib().psignn_f_vjp_backward_workspace_floats(self.handle))
            self._jwork = torch.empty(n, dtype=torch.float32, device=self.device)
        return self._jwork

    def permute(self, t, to_plan=True):
        """Rows of an (N, cols) float tensor between the caller's numbering and plan order."""
        tc = _f32c(t)
        out = torch.empty_like(tc)
        with torch.cuda.device(self.device):
            nat.check(nat.lib().psignn_plan_permute(self.handle, nat.ptr(tc), tc.shape[1], nat.ptr(out), int(to_plan),
                                                    nat.stream_ptr(self.device)), "psignn_plan_permute")
        return out


def plan_for(batch, tile_target=0) -> MeshPlan:
    """Plan cached on the batch object (rebuilt if edge_index/tags storage changed)."""
    key = (batch.edge_index.data_ptr(), tuple(batch.edge_index.shape), batch.tags.data_ptr(),
           batch.edge_attr.data_ptr(), str(batch.edge_index.device), tile_target)
    cached = getattr(batch, "_psignn_plan", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    plan = MeshPlan(batch, tile_target)
    try:
        batch._psignn_plan = (key, plan)
    except Exception:
        pass
    return plan


def _f32c(t):
    return t.to(torch.float32).contiguous()


class FixedPointMap:
    """``H -> f(H, H_init, batch)`` bound to a plan and packed weights.

    Stands where the reference passes ``lambda H: self.f(H, H_init, batch)`` to the solver
    (dirichlet/psignn/model.py:189, tests/model_psignn.py:226).  Being an object instead of a lambda lets
    ``utilities.solver.broyden`` recognise it and run the whole root-find on the device.
    """

    def __init__(self, plan: MeshPlan, weights: PackedWeights, h_initial, prb_data, normals=None):
        if plan.mixed != weights.mixed:
            raise nat.NativeError("boundary-condition family of the weights and of the batch differ "
                                  f"(weights mixed={weights.mixed}, batch mixed={plan.mixed})")
        self.plan, self.weights = plan, weights
        self.h0 = _f32c(h_initial)
        self.prb = _f32c(prb_data)
        self.nrm = None if normals is None else _f32c(normals)
        if plan.mixed and self.nrm is None:
            raise nat.NativeError("mixed problems need batch.unit_normal_vector")
        exp_p = 3 if plan.mixed else 2
        if self.prb.shape != (plan.N, exp_p) or self.h0.shape != (plan.N, D):
            raise nat.NativeError(f"shape mismatch: prb_data {tuple(self.prb.shape)}, h_initial {tuple(self.h0.shape)}")
        self._p = None  # plan-order copies of h0 / prb / normals, made on first use

    # -- plan-order fast path (no permutation passes per call) -------------------------------------
    def to_plan(self, H):
        return self.plan.permute(H, True)

    def from_plan(self, Hp):
        return self.plan.permute(Hp, False)

    def fp(self, Hp):
        """f in plan order: Hp and the result are numbered like the plan's tiles (see MeshPlan.permute)."""
        if self._p is None:
            self._p = (self.to_plan(self.h0), self.to_plan(self.prb), None if self.nrm is None else self.to_plan(self.nrm))
        h0p, prbp, nrmp = self._p
        Hc = _f32c(Hp)
        out = torch.empty_like(Hc)
        with torch.cuda.device(Hc.device):
            nat.check(nat.lib().psignn_f_forward_p(self.plan.handle, nat.ptr(self.weights.flat), self.weights.n_layers,
                                                   nat.ptr(Hc), nat.ptr(h0p), nat.ptr(prbp), nat.ptr(nrmp),
                                                   nat.ptr(out), nat.ptr(self.plan.workspace()),
                                                   nat.stream_ptr(Hc.device)), "psignn_f_forward_p")
        return out

    def __call__(self, H):
        nat.require_cuda(H, "H")
        Hc = _f32c(H)
        if Hc.shape != (self.plan.N, D):
            raise nat.NativeError(f"H has shape {tuple(Hc.shape)}, expected {(self.plan.N, D)}")
        out = torch.empty_like(Hc)
        with torch.cuda.device(Hc.device):
            nat.check(nat.lib().psignn_f_forward(self.plan.handle, nat.ptr(self.weights.flat), self.weights.n_layers,
                                                 nat.ptr(Hc), nat.ptr(self.h0), nat.ptr(self.prb), nat.ptr(self.nrm),
                                                 nat.ptr(out), nat.ptr(self.plan.workspace()),
                                                 nat.stream_ptr(Hc.device)), "psignn_f_forward")
        return out

    def picard_p(self, Hp, n):
        """n applications of f in plan order, back to back on the device (no host work in between)."""
        if self._p is None:
            self.fp(Hp)
        h0p, prbp, nrmp = self._p
        x = _f32c(Hp).clone()
        tmp = torch.empty_like(x)
        with torch.cuda.device(x.device):
            nat.check(nat.lib().psignn_picard_p(self.plan.handle, nat.ptr(self.weights.flat), self.weights.n_layers,
                                                nat.ptr(x), nat.ptr(tmp), nat.ptr(h0p), nat.ptr(prbp), nat.ptr(nrmp),
                                                nat.ptr(self.plan.workspace()), int(n), nat.stream_ptr(x.device)),
                      "psignn_picard_p")
        return x

    def jvp(self, H, V):
        """Analytic J_f(H) V."""
        Hc, Vc = _f32c(H), _f32c(V)
        out = torch.empty_like(Hc)
        with torch.cuda.device(Hc.device):
            nat.check(nat.lib().psignn_f_jvp(self.plan.handle, nat.ptr(self.weights.flat), self.weights.n_layers,
                                             nat.ptr(Hc), nat.ptr(self.prb), nat.ptr(self.nrm), nat.ptr(Vc),
                                             nat.ptr(out), nat.ptr(self.plan.workspace()),
                                             nat.stream_ptr(Hc.device)), "psignn_f_jvp")
        return out

    def jvp_p(self, Hp, Vp, out=None):
        """J_f(Hp) Vp with everything in plan order (tiled plans: single-layer dirichlet, mixed of any depth).  ``out``: a
        contiguous (N, d) float32 tensor to write into (e.g. a row of a Krylov basis)."""
        if self._p is None:
            self.fp(Hp)
        _, prbp, nrmp = self._p
        Hc, Vc = _f32c(Hp), _f32c(Vp)
        if out is None:
            out = torch.empty_like(Hc)
        elif out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != Hc.numel():
            raise nat.NativeError("jvp_p: out must be a contiguous float32 tensor of the state's size")
        with torch.cuda.device(Hc.device):
            nat.check(nat.lib().psignn_f_jvp_p(self.plan.handle, nat.ptr(self.weights.flat), self.weights.n_layers,
                                               nat.ptr(Hc), nat.ptr(prbp), nat.ptr(nrmp), nat.ptr(Vc), nat.ptr(out),
                                               nat.stream_ptr(Hc.device)), "psignn_f_jvp_p")
        return out

    # ---- one idle Broyden solver kept between solves of this map (utilities.solver.broyden without keep_trace / solver_obj):
    # creating and destroying the 2 * threshold state vectors costs 2.2 ms per call at 1M nodes and threshold 20 -- a quarter of
    # such a solve (profiles/r3_cold_solve_probe.txt).  Bounded by PSIGNN_SOLVER_CACHE_GB (default 16; 0 disables).
    def borrow_broyden(self, threshold):
        sv = getattr(self, "_idle_broyden", None)
        self._idle_broyden = None
        if sv is not None and sv.threshold == int(threshold) and not sv.keep_trace:
            return sv
        if sv is not None:
            sv.close()
        return DeviceBroyden(plan=self.plan, threshold=threshold, keep_trace=False)

    def return_broyden(self, sv):
        import os
        limit = float(os.environ.get("PSIGNN_SOLVER_CACHE_GB", "16")) * 1e9
        old = getattr(self, "_idle_broyden", None)
        if old is not None and old is not sv:
            old.close()
        if sv.nbytes <= limit:
            self._idle_broyden = sv
        else:
            self._idle_broyden = None
            sv.close()

    def can_linearize(self):
        """True when ``linearize_p`` applies: tiled plan; dirichlet family: single-layer block (csrc/fgnn_tile_lin.hip)."""
        return bool(self.plan.tiled) and (bool(self.plan.mixed) or self.weights.n_layers == 1)

    def linearize_p(self, Hp, lin=None):
        """Linearisation of f at ``Hp`` (plan order) for solvers that apply J_f(Hp) to many vectors: one pass stores the relu
        masks and per-node gate / update / LayerNorm quantities, ``lin.jvp_p(Vp)`` then applies the Jacobian as a linear
        operator (about half the cost of ``jvp_p``, same product up to fp32 summation order).  ``lin``: a Linearization of this
        map to rebuild at the new state (keeps its device buffers)."""
        if self._p is None:
            self.fp(Hp)
        if lin is None:
            lin = Linearization(self)
        lin.build(Hp)
        return lin

    def vjp(self, H, Wv):
        """Wv^T J_f(H): what ``autograd.grad(f(H), H, Wv)`` returns in the reference (model.py:214,432,449)."""
        Hc, Wc = _f32c(H), _f32c(Wv)
        out = torch.empty_like(Hc)
        with torch.cuda.device(Hc.device):
            nat.check(nat.lib().psignn_f_vjp(self.plan.handle, nat.ptr(self.weights.flat), self.weights.n_layers,
                                             nat.ptr(Hc), nat.ptr(self.prb), nat.ptr(self.nrm), nat.ptr(Wc), nat.ptr(out),
                                             nat.ptr(self.plan.workspace()), nat.stream_ptr(Hc.device)),
                      "psignn_f_vjp")
        return out

    def vjp_p(self, Hp, Wp):
        """vjp with Hp, Wp and the result in plan order (tiled kernels: single-layer dirichlet plans, mixed plans)."""
        if self._p is None:
            self.fp(Hp)
        _, prbp, nrmp = self._p
        Hc, Wc = _f32c(Hp), _f32c(Wp)
        out = torch.empty_like(Hc)
        with torch.cuda.device(Hc.device):
            nat.check(nat.lib().psignn_f_vjp_p(self.plan.handle, nat.ptr(self.weights.flat), self.weights.n_layers,
                                               nat.ptr(Hc), nat.ptr(prbp), nat.ptr(nrmp), nat.ptr(Wc), nat.ptr(out),
                                               nat.ptr(self.plan.workspace()), nat.stream_ptr(Hc.device)),
                      "psignn_f_vjp_p")
        return out

    def param_vjp_p(self, Hp, Wp):
        """(flat parameter gradient, Wp^T df/dh) at Hp, everything in plan order (tiled dirichlet plans).

        The flat gradient follows the leading section of the packed weights; ``unpack_param_grads`` names it."""
        if self._p is None:
            self.fp(Hp)
        _, prbp, _ = self._p
        Hc, Wc = _f32c(Hp), _f32c(Wp)
        l = nat.lib()
        grad = torch.empty(int(l.psignn_param_grad_size(int(self.weights.mixed), self.weights.n_layers)),
                           dtype=torch.float32, device=Hc.device)
        out = torch.empty_like(Hc)
        work = self.plan.pgrad_workspace()
        with torch.cuda.device(Hc.device):
            nat.check(l.psignn_f_param_vjp_p(self.plan.handle, nat.ptr(self.weights.flat), self.weights.n_layers,
                                             nat.ptr(Hc), nat.ptr(prbp), nat.ptr(Wc), nat.ptr(grad), nat.ptr(out),
                                             nat.ptr(work), nat.stream_ptr(Hc.device)), "psignn_f_param_vjp_p")
        return grad, out

    def param_vjp(self, H, Wv):
        """What ``loss.backward()`` leaves in the ``deqdss.f`` parameters for new_H = f(H) with cotangent Wv
        (dirichlet/psignn/model.py:203-225; mixed/psignn/model.py likewise): ({name: grad}, Wv^T df/dH) in the
        caller's numbering.  Single-layer blocks of both families, tiled or not."""
        Hc, Wc = _f32c(H), _f32c(Wv)
        l = nat.lib()
        grad = torch.empty(int(l.psignn_param_grad_size(int(self.weights.mixed), self.weights.n_layers)),
                           dtype=torch.float32, device=Hc.device)
        out = torch.empty_like(Hc)
        with torch.cuda.device(Hc.device):
            nat.check(l.psignn_f_param_vjp(self.plan.handle, nat.ptr(self.weights.flat), self.weights.n_layers,
                                           nat.ptr(Hc), nat.ptr(self.prb), nat.ptr(self.nrm), nat.ptr(Wc), nat.ptr(grad),
                                           nat.ptr(out), nat.ptr(self.plan.pgrad_workspace()),
                                           nat.stream_ptr(Hc.device)), "psignn_f_param_vjp")
        return unpack_param_grads(grad, self.weights.n_layers, self.weights.mixed), out

    def vjp_backward(self, H, V, Gbar):
        """Gradient of  Gbar . (J_f(H)^T V)  with Gbar held constant: ({name: grad}, d / dH) -- what autograd's double
        backward computes for ``autograd.grad(f(H), H, V, create_graph=True)`` (jac_loss_estimate,
        dirichlet/psignn/model.py:416-435).  Single-layer blocks of both families, caller's numbering."""
        Hc, Vc, Gc = _f32c(H), _f32c(V), _f32c(Gbar)
        l = nat.lib()
        grad = torch.empty(int(l.psignn_param_grad_size(int(self.weights.mixed), self.weights.n_layers)),
                           dtype=torch.float32, device=Hc.device)
        out = torch.empty_like(Hc)
        with torch.cuda.device(Hc.device):
            nat.check(l.psignn_f_vjp_backward(self.plan.handle, nat.ptr(self.weights.flat), self.weights.n_layers,
                                              nat.ptr(Hc), nat.ptr(self.prb), nat.ptr(self.nrm), nat.ptr(Vc), nat.ptr(Gc), nat.ptr(grad),
                                              nat.ptr(out), nat.ptr(self.plan.vjp_backward_workspace()),
                                              nat.stream_ptr(Hc.device)), "psignn_f_vjp_backward")
        return unpack_param_grads(grad, self.weights.n_layers, self.weights.mixed), out

    def phi(self, H, which: int, layer: int = 0):
        """One aggregation: 0 Phi_to, 1 Phi_from, 2 Phi_neumann."""
        Hc = _f32c(H)
        out = torch.empty_like(Hc)
        with torch.cuda.device(Hc.device):
            nat.check(nat.lib().psignn_phi(self.plan.handle, nat.ptr(self.weights.flat), self.weights.n_layers,
                                           layer, which, nat.ptr(Hc), nat.ptr(out), nat.ptr(self.plan.workspace()),
                                           nat.stream_ptr(Hc.device)), "psignn_phi")
        return out


def pack_dsgps(sd, device=None) -> torch.Tensor:
    """Flat weight buffer of the DS-GPS kernels from a ``ModelDSGPS`` state_dict (dirichlet/dsgps/model.py:35-45,
    mixed/dsgps/model.py:35-48; layout in csrc/dsgps_tile.hip): every matrix transposed to [in k][out o]."""
    m = lambda k: sd[k].detach().to("cpu", torch.float32)
    mixed = "phi_neumann.mlp.mlp.0.weight" in sd
    p = 3 if mixed else 2
    if m("phi_to.mlp.mlp.0.weight").shape != (D, 2 * D + 3) or m("z_k.mlp.0.weight").shape != (D, 3 * D + p):
        raise nat.NativeError("DS-GPS kernels are built for latent_dim = 10, 3 edge features, 2 (mixed: 3) problem features")
    t = lambda a: a.t().contiguous().reshape(-1)
    Wt, Wf = m("phi_to.mlp.mlp.0.weight"), m("phi_from.mlp.mlp.0.weight")
    mir = torch.tensor([-1.0, -1.0, 1.0])[:, None]
    parts = [t(Wt[:, D:2 * D]), t(Wf[:, D:2 * D]), t(Wt[:, :D]), t(Wf[:, :D]),
             (Wt[:, 2 * D:].t() * mir).reshape(-1), t(Wf[:, 2 * D:]),
             m("phi_to.mlp.mlp.0.bias"), m("phi_from.mlp.mlp.0.bias"),
             t(m("phi_to.mlp.mlp.2.weight")), m("phi_to.mlp.mlp.2.bias"),
             t(m("phi_from.mlp.mlp.2.weight")), m("phi_from.mlp.mlp.2.bias")]
    for g in ("z_k", "r_k", "correction"):
        parts += [t(m(f"{g}.mlp.0.weight")), m(f"{g}.mlp.0.bias")]
    if mixed:   # transposed Neumann block, same layout as the mixed PSI-GNN kernels (WLayout<3>::N_*)
        d64 = lambda k: sd[k].detach().to("cpu", torch.float64)
        f32 = lambda a: a.to(torch.float32).reshape(-1)
        Wn, N1, N2 = d64("phi_neumann.mlp.mlp.0.weight"), d64("update_neumann.mlp.0.weight"), d64("update_neumann.mlp.2.weight")
        Gn = N1[:, D:2 * D] @ d64("phi_neumann.mlp.mlp.2.weight")
        gn = N1[:, D:2 * D] @ d64("phi_neumann.mlp.mlp.2.bias")
        n1p = f32(N1[:, 2 * D:].t())
        parts += [f32(Wn[:, D:2 * D].t()), f32(Wn[:, :D].t()), f32(Wn[:, 2 * D:].t()), f32(d64("phi_neumann.mlp.mlp.0.bias")),
                  f32(N1[:, :D].t()), f32(Gn.t()), torch.nn.functional.pad(n1p, (0, 50 - n1p.numel())),
                  f32(d64("update_neumann.mlp.0.bias")), f32(gn), f32(N2.t()), f32(d64("update_neumann.mlp.2.bias"))]
    flat = torch.cat([q.reshape(-1) for q in parts]).contiguous()
    if flat.numel() != int(nat.lib().psignn_dsgps_weights_size(int(mixed))):
        raise nat.NativeError(f"packed DS-GPS weight length {flat.numel()} != native layout")
    return flat if device is None else flat.to(device)


def dsgps_forward(plan: "MeshPlan", wflat, h0, prb, k: int, nrm=None):
    """H_k of ``ModelDSGPS.inference`` (model.py:141-155): k recurrent updates from the encoder state h0."""
    nat.require_cuda(h0, "h0")
    hc, pc = _f32c(h0), _f32c(prb)
    out = torch.empty_like(hc)
    work = torch.empty(4 * plan.N * D, dtype=torch.float32, device=hc.device)
    with torch.cuda.device(hc.device):
        nat.check(nat.lib().psignn_dsgps_forward(plan.handle, nat.ptr(wflat), int(k), nat.ptr(hc), nat.ptr(pc),
                                                 nat.ptr(None if nrm is None else _f32c(nrm)), nat.ptr(out),
                                                 nat.ptr(work), nat.stream_ptr(hc.device)), "psignn_dsgps_forward")
    return out


def dsgps_step_p(plan: "MeshPlan", wflat, hp, h0p, prbp, nrmp=None):
    """One DS-GPS update, node tensors in plan order."""
    out = torch.empty_like(hp)
    with torch.cuda.device(hp.device):
        nat.check(nat.lib().psignn_dsgps_step_p(plan.handle, nat.ptr(wflat), nat.ptr(_f32c(hp)), nat.ptr(h0p), nat.ptr(prbp),
                                                nat.ptr(nrmp), nat.ptr(out), nat.stream_ptr(hp.device)), "psignn_dsgps_step_p")
    return out


# f_theta layouts (csrc/common.h WLayout<P>): shared 64 | layer 0 = phi_to 350 | phi_from 350 | update 10 (30+P) + 120 | fold 244
# [| phi_neumann 350 | update_neumann 370 | nfold 112]
_FL_PHI_TO, _FL_PHI_FROM = 64, 414
_FL_MIXED_PHI_NEU, _FL_MIXED_UPD_NEU = 64 + 700 + 450 + 244, 64 + 700 + 450 + 244 + 350


def pack_dsgps_train(sd, device):
    """(Phi modules -- mixed: and update_neumann -- in the f_theta weight layout, gates [Wz|bz|Wr|br|Wc|bc]) for
    ``psignn_dsgps_step_backward``.  Assembled on the device (no host round trip: this runs once per training forward); only
    the blocks the backward kernels read are filled."""
    mixed = "phi_neumann.mlp.mlp.0.weight" in sd
    g = lambda k: sd[k].detach().to(device, torch.float32).reshape(-1)
    n = int(nat.lib().psignn_weights_size(int(mixed), 1))
    if mixed and int(nat.lib().psignn_param_grad_size(1, 1)) != _FL_MIXED_UPD_NEU + 370 + 112:
        raise nat.NativeError("f_theta weight layout changed: update engine._FL_* offsets")
    wf = torch.zeros(n, dtype=torch.float32, device=device)
    blocks = [("phi_to", _FL_PHI_TO), ("phi_from", _FL_PHI_FROM)] + ([("phi_neumann", _FL_MIXED_PHI_NEU)] if mixed else [])
    for m, o in blocks:
        blk = torch.cat([g(f"{m}.mlp.mlp.0.weight"), g(f"{m}.mlp.mlp.0.bias"), g(f"{m}.mlp.mlp.2.weight"), g(f"{m}.mlp.mlp.2.bias")])
        wf[o:o + blk.numel()] = blk
    if mixed:
        blk = torch.cat([g(f"update_neumann.mlp.{q}") for q in ("0.weight", "0.bias", "2.weight", "2.bias")])
        wf[_FL_MIXED_UPD_NEU:_FL_MIXED_UPD_NEU + blk.numel()] = blk
    wg = torch.cat([g(f"{m}.mlp.0.{q}") for m in ("z_k", "r_k", "correction") for q in ("weight", "bias")])
    return wf, wg


def unpack_dsgps_grads(flat, mixed):
    """{state_dict name: gradient} from the flat buffer of ``psignn_dsgps_step_backward``."""
    base = int(nat.lib().psignn_param_grad_size(int(mixed), 1))
    out = {}
    for k, t in unpack_param_grads(flat[:base], 1, mixed).items():
        if k.startswith(("phi_to_list.0.", "phi_from_list.0.")):
            out[k.replace("_list.0.", ".")] = t
        elif k.startswith(("phi_neumann.", "update_neumann.")):
            out[k] = t
    cat = 3 * D + (3 if mixed else 2)
    o = base
    for m in ("z_k", "r_k", "correction"):
        out[f"{m}.mlp.0.weight"] = flat[o:o + D * cat].reshape(D, cat)
        o += D * cat
        out[f"{m}.mlp.0.bias"] = flat[o:o + D]
        o += D
    return out


def dsgps_step_backward(plan: "MeshPlan", wf, wg, h, prb, w, nrm=None):
    """({name: grad}, w^T dh'/dh) of one DS-GPS update (caller's numbering; mixed plans need the unit normals)."""
    hc, wc = _f32c(h), _f32c(w)
    l = nat.lib()
    mixed = nrm is not None
    grad = torch.empty(int(l.psignn_dsgps_grad_size(int(mixed))), dtype=torch.float32, device=hc.device)
    out = torch.empty_like(hc)
    if getattr(plan, "_dswork", None) is None:
        plan._dswork = torch.empty(int(l.psignn_dsgps_step_backward_workspace_floats(plan.handle)), dtype=torch.float32,
                                   device=hc.device)
    with torch.cuda.device(hc.device):
        nat.check(l.psignn_dsgps_step_backward(plan.handle, nat.ptr(wf), nat.ptr(wg), nat.ptr(hc), nat.ptr(_f32c(prb)),
                                               nat.ptr(None if nrm is None else _f32c(nrm)), nat.ptr(wc), nat.ptr(grad),
                                               nat.ptr(out), nat.ptr(plan._dswork), nat.stream_ptr(hc.device)),
                  "psignn_dsgps_step_backward")
    return unpack_dsgps_grads(grad, mixed), out


# f_theta base layout with three node inputs (csrc/common.h WLayout<3>): shared 64 | phi_to 350 | phi_from 350 | update 450 | fold
_DSS_PHI_TO, _DSS_PHI_FROM, _DSS_PSI = 64, 414, 764


def pack_dss_train(sd, t, device):
    """Update t's modules in the f_theta weight layout for ``psignn_dss_step_backward``: the (10, 21) first Phi layers padded
    to (10, 23) (edge-feature weight in column 22, the plan carries the scalar feature in the third attr column), Psi in
    the update slots.  Assembled on the device (it runs once per update in the backward pass)."""
    n = int(nat.lib().psignn_dss_grad_size())
    wf = torch.zeros(n, dtype=torch.float32, device=device)
    g = lambda k: sd[k].detach().to(device, torch.float32)
    for name, o in ((f"phi_to_list.{t}", _DSS_PHI_TO), (f"phi_from_list.{t}", _DSS_PHI_FROM)):
        w1 = g(f"{name}.mlp.mlp.0.weight")
        w1p = torch.zeros(D, 2 * D + 3, dtype=torch.float32, device=device)
        w1p[:, :2 * D], w1p[:, 2 * D + 2] = w1[:, :2 * D], w1[:, 2 * D]
        blk = torch.cat([w1p.reshape(-1), g(f"{name}.mlp.mlp.0.bias"), g(f"{name}.mlp.mlp.2.weight").reshape(-1),
                         g(f"{name}.mlp.mlp.2.bias")])
        wf[o:o + blk.numel()] = blk
    psi = f"psi_list.{t}.mlp.mlp"
    blk = torch.cat([g(f"{psi}.0.weight").reshape(-1), g(f"{psi}.0.bias"), g(f"{psi}.2.weight").reshape(-1), g(f"{psi}.2.bias")])
    wf[_DSS_PSI:_DSS_PSI + blk.numel()] = blk
    return wf


def unpack_dss_grads(flat, t):
    out = {}
    for name, o in ((f"phi_to_list.{t}", _DSS_PHI_TO), (f"phi_from_list.{t}", _DSS_PHI_FROM)):
        w1 = flat[o:o + D * (2 * D + 3)].reshape(D, 2 * D + 3)
        out[f"{name}.mlp.mlp.0.weight"] = torch.cat([w1[:, :2 * D], w1[:, 2 * D + 2:2 * D + 3]], dim=1)
        o += D * (2 * D + 3)
        out[f"{name}.mlp.mlp.0.bias"] = flat[o:o + D]
        out[f"{name}.mlp.mlp.2.weight"] = flat[o + D:o + D + D * D].reshape(D, D)
        out[f"{name}.mlp.mlp.2.bias"] = flat[o + D + D * D:o + 2 * D + D * D]
    o, cat = _DSS_PSI, 3 * D + 3
    psi = f"psi_list.{t}.mlp.mlp"
    out[f"{psi}.0.weight"] = flat[o:o + D * cat].reshape(D, cat)
    out[f"{psi}.0.bias"] = flat[o + D * cat:o + D * cat + D]
    o += D * cat + D
    out[f"{psi}.2.weight"] = flat[o:o + D * D].reshape(D, D)
    out[f"{psi}.2.bias"] = flat[o + D * D:o + D * D + D]
    return out


def dss_step_backward(plan: "MeshPlan", wf_t, t, alpha, h, bprime_norm, w):
    """({name: grad}, w^T dh'/dh) of DSS update t (caller's numbering)."""
    hc, wc = _f32c(h), _f32c(w)
    l = nat.lib()
    grad = torch.empty(int(l.psignn_dss_grad_size()), dtype=torch.float32, device=hc.device)
    out = torch.empty_like(hc)
    if getattr(plan, "_dsswork", None) is None:
        plan._dsswork = torch.empty(int(l.psignn_dss_step_backward_workspace_floats(plan.handle)), dtype=torch.float32,
                                    device=hc.device)
    with torch.cuda.device(hc.device):
        nat.check(l.psignn_dss_step_backward(plan.handle, nat.ptr(wf_t), float(alpha), nat.ptr(hc), nat.ptr(_f32c(bprime_norm)),
                                             nat.ptr(wc), nat.ptr(grad), nat.ptr(out), nat.ptr(plan._dsswork),
                                             nat.stream_ptr(hc.device)), "psignn_dss_step_backward")
    return unpack_dss_grads(grad, t), out


def pack_dss(sd, k, device=None) -> torch.Tensor:
    """Flat per-step weight buffer of the DSS kernels from a ``DeepStatisticalSolver`` state_dict
    (dirichlet/dss/model.py:33-55; layout in csrc/dss_tile.hip)."""
    m = lambda n: sd[n].detach().to("cpu", torch.float32)
    t = lambda a: a.t().contiguous().reshape(-1)
    z20 = torch.zeros(2 * D)
    steps = []
    for s in range(k):
        Wt, Wf = m(f"phi_to_list.{s}.mlp.mlp.0.weight"), m(f"phi_from_list.{s}.mlp.mlp.0.weight")
        if Wt.shape != (D, 2 * D + 1) or m(f"psi_list.{s}.mlp.mlp.0.weight").shape != (D, 3 * D + 3):
            raise nat.NativeError("DSS kernels are built for latent_dim = 10, a scalar edge feature and a 3-wide node input")
        steps += [t(Wt[:, D:2 * D]), t(Wf[:, D:2 * D]), t(Wt[:, :D]), t(Wf[:, :D]),
                  z20, Wt[:, 2 * D], z20, Wf[:, 2 * D],
                  m(f"phi_to_list.{s}.mlp.mlp.0.bias"), m(f"phi_from_list.{s}.mlp.mlp.0.bias"),
                  t(m(f"phi_to_list.{s}.mlp.mlp.2.weight")), m(f"phi_to_list.{s}.mlp.mlp.2.bias"),
                  t(m(f"phi_from_list.{s}.mlp.mlp.2.weight")), m(f"phi_from_list.{s}.mlp.mlp.2.bias"),
                  t(m(f"psi_list.{s}.mlp.mlp.0.weight")), m(f"psi_list.{s}.mlp.mlp.0.bias"),
                  t(m(f"psi_list.{s}.mlp.mlp.2.weight")), m(f"psi_list.{s}.mlp.mlp.2.bias")]
    flat = torch.cat([p.reshape(-1) for p in steps]).contiguous()
    if flat.numel() != int(nat.lib().psignn_dss_weights_size(int(k))):
        raise nat.NativeError(f"packed DSS weight length {flat.numel()} != native layout")
    return flat if device is None else flat.to(device)


def dss_forward(plan: "MeshPlan", wflat, bprime_norm, k: int, alpha: float):
    """H_k of ``DeepStatisticalSolver.inference`` (model.py:97-120) from H_0 = 0."""
    nat.require_cuda(bprime_norm, "b_prime_norm")
    bc = _f32c(bprime_norm)
    out = torch.empty((plan.N, D), dtype=torch.float32, device=bc.device)
    work = torch.empty(23 * plan.N, dtype=torch.float32, device=bc.device)
    with torch.cuda.device(bc.device):
        nat.check(nat.lib().psignn_dss_forward(plan.handle, nat.ptr(wflat), int(k), float(alpha), nat.ptr(bc), nat.ptr(out),
                                               nat.ptr(work), nat.stream_ptr(bc.device)), "psignn_dss_forward")
    return out


def dss_step_p(plan: "MeshPlan", wflat, t: int, alpha: float, hp, bprime_p):
    """DSS update t, state and b'_norm in plan order."""
    out = torch.empty_like(hp)
    with torch.cuda.device(hp.device):
        nat.check(nat.lib().psignn_dss_step_p(plan.handle, nat.ptr(wflat), int(t), float(alpha), nat.ptr(_f32c(hp)),
                                              nat.ptr(bprime_p), nat.ptr(out), nat.stream_ptr(hp.device)), "psignn_dss_step_p")
    return out


def unpack_param_grads(flat, n_layers=1, mixed=False):
    """Name the entries of a flat parameter gradient (layout = leading section of ``pack_weights``)."""
    if n_layers != 1:
        raise nat.NativeError("parameter gradients are implemented for single-layer blocks")
    p = 3 if mixed else 2
    cat, ein = 3 * D + p, 2 * D + 3
    out, o = {}, 0

    def take(name, *shape):
        nonlocal o
        n = 1
        for k in shape:
            n *= k
        out[name] = flat[o:o + n].reshape(shape)
        o += n
    take("laynorm.weight", D)
    take("laynorm.bias", D)
    take("alpha.0.weight", 1, cat)
    take("alpha.0.bias", 1)
    o = 64
    for phi in ("phi_to_list", "phi_from_list"):
        take(f"{phi}.0.mlp.mlp.0.weight", D, ein)
        take(f"{phi}.0.mlp.mlp.0.bias", D)
        take(f"{phi}.0.mlp.mlp.2.weight", D, D)
        take(f"{phi}.0.mlp.mlp.2.bias", D)
    take("update_list.0.mlp.0.weight", D, cat)
    take("update_list.0.mlp.0.bias", D)
    take("update_list.0.mlp.2.weight", D, D)
    take("update_list.0.mlp.2.bias", D)
    if mixed:
        o += 244   # fold slots of the layer
        take("phi_neumann.mlp.mlp.0.weight", D, ein)
        take("phi_neumann.mlp.mlp.0.bias", D)
        take("phi_neumann.mlp.mlp.2.weight", D, D)
        take("phi_neumann.mlp.mlp.2.bias", D)
        take("update_neumann.mlp.0.weight", D, 2 * D + p + 2)
        take("update_neumann.mlp.0.bias", D)
        take("update_neumann.mlp.2.weight", D, D)
        take("update_neumann.mlp.2.bias", D)
    return out


def mlp2_backward(x, gy, w1, b1, w2, need_gx=True):
    """Backward of ``mlp2``: (gx | None, gW1, gb1, gW2, gb2)."""
    xc, gc = _f32c(x), _f32c(gy)
    n, din = xc.shape
    hid, dout = w1.shape[0], w2.shape[0]
    l = nat.lib()
    gx = torch.empty_like(xc) if need_gx else None
    gflat = torch.empty(hid * din + hid + dout * hid + dout, dtype=torch.float32, device=xc.device)
    work = torch.empty(int(l.psignn_mlp2_backward_workspace_floats(n)), dtype=torch.float32, device=xc.device)
    with torch.cuda.device(xc.device):
        nat.check(l.psignn_mlp2_backward(nat.ptr(xc), nat.ptr(gc), n, din, hid, dout, nat.ptr(_f32c(w1)),
                                         nat.ptr(_f32c(b1)), nat.ptr(_f32c(w2)), nat.ptr(gx), nat.ptr(gflat),
                                         nat.ptr(work), nat.stream_ptr(xc.device)), "psignn_mlp2_backward")
    o1, o2, o3 = hid * din, hid * din + hid, hid * din + hid + dout * hid
    return gx, gflat[:o1].reshape(hid, din), gflat[o1:o2], gflat[o2:o3].reshape(dout, hid), gflat[o3:]


class _MLP2Fn(torch.autograd.Function):
    """mlp2 with its HIP backward, for the training path (encoder / decoder terms of the loss)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        ctx.save_for_backward(x, w1, b1, w2)
        return mlp2(x, w1, b1, w2, b2)

    @staticmethod
    def backward(ctx, gy):
        x, w1, b1, w2 = ctx.saved_tensors
        gx, g1, gb1, g2, gb2 = mlp2_backward(x, gy.contiguous(), w1, b1, w2, need_gx=ctx.needs_input_grad[0])
        return gx, g1, gb1, g2, gb2


def mlp2_autograd(x, w1, b1, w2, b2):
    if torch.is_grad_enabled() and any(t.requires_grad for t in (x, w1, b1, w2, b2)):
        return _MLP2Fn.apply(x, w1, b1, w2, b2)
    return mlp2(x, w1, b1, w2, b2)


def residual_t(plan: "MeshPlan", a_ij, r):
    """A^T r (backward of ``residual`` w.r.t. u); a_ij in the caller's edge order."""
    rc, ac = _f32c(r).reshape(-1), _f32c(a_ij).reshape(-1)
    if ac.numel() != plan.E:
        raise nat.NativeError(f"a_ij has {ac.numel()} entries, the plan {plan.E} edges")
    out = torch.empty_like(rc)
    with torch.cuda.device(rc.device):
        nat.check(nat.lib().psignn_residual_t(plan.handle, nat.ptr(ac), nat.ptr(rc), nat.ptr(out),
                                              nat.stream_ptr(rc.device)), "psignn_residual_t")
    return out.reshape(-1, 1)


class _ResidualFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, y, plan, a_ij):
        ctx.plan, ctx.a_ij = plan, a_ij
        return residual(plan, u, y)

    @staticmethod
    def backward(ctx, gr):
        return residual_t(ctx.plan, ctx.a_ij, gr.contiguous()), None, None, None


def residual_autograd(plan, u, y, a_ij):
    if torch.is_grad_enabled() and u.requires_grad:
        return _ResidualFn.apply(u, y, plan, a_ij)
    return residual(plan, u, y)


def mlp2(x, w1, b1, w2, b2):
    """relu(x W1^T + b1) W2^T + b2 for the tiny encoder/decoder MLPs (model.py:370-392)."""
    nat.require_cuda(x, "x")
    xc = _f32c(x)
    n, din = xc.shape
    hid, dout = w1.shape[0], w2.shape[0]
    out = torch.empty((n, dout), dtype=torch.float32, device=xc.device)
    with torch.cuda.device(xc.device):
        nat.check(nat.lib().psignn_mlp2(nat.ptr(xc), n, din, hid, dout, nat.ptr(_f32c(w1)), nat.ptr(_f32c(b1)),
                                        nat.ptr(_f32c(w2)), nat.ptr(_f32c(b2)), nat.ptr(out),
                                        nat.stream_ptr(xc.device)), "psignn_mlp2")
    return out


def residual(plan: MeshPlan, u, y):
    """A u - y with A = COO(edge_index, a_ij) including the diagonal (model.py:157-167)."""
    uc, yc = _f32c(u).reshape(-1), _f32c(y).reshape(-1)
    out = torch.empty_like(uc)
    with torch.cuda.device(uc.device):
        nat.check(nat.lib().psignn_residual(plan.handle, nat.ptr(uc), nat.ptr(yc), nat.ptr(out),
                                            nat.stream_ptr(uc.device)), "psignn_residual")
    return out.reshape(-1, 1)


# ---------------------------------------------------------------------------------------------
# device Broyden
# ---------------------------------------------------------------------------------------------
TRACE_BUDGET_BYTES = 8 << 30  # keep every iterate only while (thr+2)*N*d*4 stays under this


class Linearization:
    """Stored linearisation of a FixedPointMap at one state (psignn_lin_* in include/psignn_hip.h)."""

    def __init__(self, fmap):
        if not fmap.can_linearize():
            raise nat.NativeError("linearize_p: tiled plans (dirichlet family: single-layer blocks); use jvp_p otherwise")
        self.fmap = fmap
        h = C.c_void_p()
        with torch.cuda.device(fmap.weights.flat.device):
            nat.check(nat.lib().psignn_lin_create(C.byref(h), fmap.plan.handle), "psignn_lin_create")
        self.handle = h
        self.bytes = int(nat.lib().psignn_lin_bytes(h))

    def build(self, Hp):
        fm = self.fmap
        _, prbp, nrmp = fm._p
        Hc = _f32c(Hp)
        with torch.cuda.device(Hc.device):
            nat.check(nat.lib().psignn_lin_build(self.handle, nat.ptr(fm.weights.flat), fm.weights.n_layers, nat.ptr(Hc),
                                                 nat.ptr(prbp), nat.ptr(nrmp), nat.stream_ptr(Hc.device)), "psignn_lin_build")
        return self

    def jvp_p(self, Vp, out=None):
        """J_f(H) Vp for the H of the last ``build`` (plan order); ``out`` as in FixedPointMap.jvp_p."""
        fm = self.fmap
        Vc = _f32c(Vp)
        if out is None:
            out = torch.empty_like(Vc)
        elif out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != Vc.numel():
            raise nat.NativeError("jvp_p: out must be a contiguous float32 tensor of the state's size")
        with torch.cuda.device(Vc.device):
            nat.check(nat.lib().psignn_lin_jvp(self.handle, nat.ptr(fm.weights.flat), fm.weights.n_layers, nat.ptr(Vc),
                                               nat.ptr(out), nat.stream_ptr(Vc.device)), "psignn_lin_jvp")
        return out

    def close(self):
        if self.handle is not None:
            nat.lib().psignn_lin_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceBroyden:
    def __init__(self, plan=None, threshold=50, keep_trace=False, n_elems=None, seq_len=D, device=None, shard_elems=0):
        """``shard_elems`` > 0: the solver will run inside ``broyden_solve_batch`` with others; its reduction shapes are sized
        for the whole shard (sum of N * d), and its single-mesh solves give the same bits as the batched ones."""
        h = C.c_void_p()
        self.plan = plan
        self.threshold = int(threshold)
        self.keep_trace = bool(keep_trace)
        self.device = plan.device if plan is not None else device
        with torch.cuda.device(self.device):
            if plan is not None:
                nat.check(nat.lib().psignn_broyden_create_for_batch(C.byref(h), plan.handle, self.threshold, int(keep_trace),
                                                                     int(shard_elems)), "psignn_broyden_create_for_batch")
                self.M = plan.N * D
            else:
                nat.check(nat.lib().psignn_broyden_create_n(C.byref(h), int(n_elems), int(seq_len), self.threshold,
                                                            int(keep_trace)), "psignn_broyden_create_n")
                self.M = int(n_elems)
        self.handle = h
        self._fin = weakref.finalize(self, nat.lib().psignn_broyden_destroy, h)

    def close(self):
        self._fin()

    @property
    def nbytes(self):
        return int(nat.lib().psignn_broyden_bytes(self.handle))

    def set_stop_mode(self, stop_mode: str):
        """"rel" (default, every call site of the reference) or "abs" (solver.py:116,140,174)."""
        if stop_mode not in ("rel", "abs"):
            raise nat.NativeError(f"stop_mode {stop_mode!r}: 'rel' or 'abs'")
        self.stop_mode = stop_mode
        nat.check(nat.lib().psignn_broyden_set_stop_mode(self.handle, int(stop_mode == "abs")), "psignn_broyden_set_stop_mode")

    def _collect(self, info, rel, abs_, shape, dev):
        n_it = info.n_iter
        low = float(info.lowest_abs if getattr(self, "stop_mode", "rel") == "abs" else info.lowest)
        out = {"nstep": int(info.nstep), "n_iter": int(n_it), "lowest": low,
               "prot_break": bool(info.prot_break), "stop_reason": int(info.stop_reason)}
        # reference pads both traces to threshold+1 entries with the lowest values (solver.py:195-197)
        rel_l = list(rel[:n_it]) + [float(info.lowest)] * (self.threshold + 1 - n_it)
        abs_l = list(abs_[:n_it]) + [float(info.lowest_abs)] * (self.threshold + 1 - n_it)
        out["rel_trace"], out["abs_trace"] = rel_l, abs_l
        return out

    def solve(self, fmap: FixedPointMap, eps, poll_every=8):
        result = torch.empty_like(fmap.h0)
        info = nat.SolveInfo()
        rel = (C.c_double * self.threshold)()
        abs_ = (C.c_double * self.threshold)()
        with torch.cuda.device(self.device):
            nat.check(nat.lib().psignn_broyden_solve(
                self.handle, nat.ptr(fmap.weights.flat), fmap.weights.n_layers, nat.ptr(fmap.h0), nat.ptr(fmap.prb),
                nat.ptr(fmap.nrm), float(eps), int(poll_every), nat.ptr(result), C.byref(info), rel, abs_,
                nat.stream_ptr(self.device)), "psignn_broyden_solve")
        out = self._collect(info, rel, abs_, result.shape, result.device)
        out["result"] = result
        return out

    def solve_adjoint(self, fmap: FixedPointMap, h_star, grad, eps, poll_every=8):
        """y = J_f(h*)^T y + grad, y_0 = 0, entirely on the device (VJP kernel inside the Broyden loop)."""
        hs, gr = _f32c(h_star), _f32c(grad)
        result = torch.empty_like(gr)
        info = nat.SolveInfo()
        rel = (C.c_double * self.threshold)()
        abs_ = (C.c_double * self.threshold)()
        with torch.cuda.device(self.device):
            nat.check(nat.lib().psignn_broyden_solve_adjoint(
                self.handle, nat.ptr(fmap.weights.flat), fmap.weights.n_layers, nat.ptr(hs), nat.ptr(fmap.prb),
                nat.ptr(fmap.nrm), nat.ptr(gr), float(eps), int(poll_every), nat.ptr(result), C.byref(info), rel, abs_,
                nat.stream_ptr(self.device)), "psignn_broyden_solve_adjoint")
        out = self._collect(info, rel, abs_, result.shape, result.device)
        out["result"] = result
        return out

    def iterate(self, i, like):
        dst = torch.empty_like(like)
        with torch.cuda.device(self.device):
            nat.check(nat.lib().psignn_broyden_get_iterate(self.handle, int(i), nat.ptr(dst),
                                                           nat.stream_ptr(self.device)), "psignn_broyden_get_iterate")
        return dst

    def pair(self, j, like, which="U"):
        """Stored rank-one pair j of the last solve (``U_j`` / ``V_j``; the reference's ``Us[..., j]`` / ``VTs[:, j]``,
        utilities/solver.py:190-191) or, ``which="update"``, the current update vector (``j`` ignored); caller's numbering."""
        dst = torch.empty_like(like)
        with torch.cuda.device(self.device):
            nat.check(nat.lib().psignn_broyden_get_pair(self.handle, int(j), {"U": 0, "V": 1, "update": 2, "parta": 3}[which], nat.ptr(dst),
                                                        nat.stream_ptr(self.device)), "psignn_broyden_get_pair")
        return dst

    def _armijo_step(self, f, phi0, lib, sp, like):
        """Step length of the reference's line search (line_search / scalar_search_armijo, utilities/solver.py:20-94) for
        the current update direction: backtracking on phi(s) = |f(x + s u) - (x + s u)|^2 with phi'(0) taken as -phi(0),
        c1 = 1e-4, first trial s = 1, then the minimiser of the quadratic through (0, 1), then cubic interpolation through
        the last two trials, safeguarded to shrink by at least 4 % and at most 50 % per round, given up below s = 1e-2
        (-> s = 1).  Like the reference, the quadratic trial is never tested against the Armijo condition itself.
        Returns (s, f(x + s u) if it was evaluated at that s else None, phi(s) or None)."""
        c1, amin, der0 = 1e-4, 1e-2, -phi0
        cache = {}

        def phi(s):
            xt = torch.empty_like(like)
            nat.check(lib.psignn_broyden_ext_trial_x(self.handle, float(s), nat.ptr(xt), sp), "ext_trial_x")
            fx = _f32c(f(xt.clone()))
            g = fx - xt
            val = float(g.norm()) ** 2 if bool(torch.isfinite(g).all()) else float("inf")
            cache["s"], cache["fx"], cache["phi"] = s, fx, val
            return val

        def accept(s, val):
            return val <= phi0 + c1 * s * der0

        s0, p0 = 1.0, phi(1.0)
        if accept(s0, p0):
            return s0, cache["fx"], p0
        s1 = -der0 * s0 ** 2 / 2.0 / (p0 - phi0 - der0 * s0)
        p1 = phi(s1)
        while s1 > amin:
            den = s0 ** 2 * s1 ** 2 * (s1 - s0)
            r0, r1 = p0 - phi0 - der0 * s0, p1 - phi0 - der0 * s1
            ca = (s0 ** 2 * r1 - s1 ** 2 * r0) / den
            cb = (-s0 ** 3 * r1 + s1 ** 3 * r0) / den
            s2 = (-cb + abs(cb ** 2 - 3.0 * ca * der0) ** 0.5) / (3.0 * ca)
            p2 = phi(s2)
            if accept(s2, p2):
                return s2, cache["fx"], p2
            if (s1 - s2) > s1 / 2.0 or (1.0 - s2 / s1) < 0.96:
                s2 = s1 / 2.0
            s0, s1, p0, p1 = s1, s2, p1, p2
        # no admissible step: s = 1 (solver.py:85-93; f there is re-evaluated unless the last trial was at s = 1)
        return 1.0, (cache["fx"] if cache["s"] == 1.0 else None), None

    def solve_callable(self, f, x0, eps, ls=False):
        """Generic f (any Python callable on device tensors): one f call per iteration from the host (ls=True: plus the
        trial evaluations of the Armijo line search)."""
        x0c = _f32c(x0)
        lib, sp = nat.lib(), nat.stream_ptr(self.device)
        with torch.cuda.device(self.device):
            fx = _f32c(f(x0c))
            nat.check(lib.psignn_broyden_ext_begin(self.handle, nat.ptr(x0c), nat.ptr(fx), sp), "ext_begin")
            done = C.c_int(0)
            xn = torch.empty_like(x0c)
            phi_cur = float((fx - x0c).norm()) ** 2 if ls else 0.0
            self.ls_steps = []
            for _ in range(self.threshold):
                if ls:
                    s, fx_s, phi_s = self._armijo_step(f, phi_cur, lib, sp, x0c)
                    self.ls_steps.append(s)
                    if s != 1.0:
                        nat.check(lib.psignn_broyden_ext_scale_step(self.handle, float(s), sp), "ext_scale_step")
                    nat.check(lib.psignn_broyden_ext_next_x(self.handle, nat.ptr(xn), sp), "ext_next_x")
                    fx = fx_s if fx_s is not None else _f32c(f(xn.clone()))
                    phi_cur = phi_s if phi_s is not None else float((fx - xn).norm()) ** 2
                    nat.check(lib.psignn_broyden_ext_update(self.handle, nat.ptr(fx), float(eps), C.byref(done), sp),
                              "ext_update")
                    if done.value:
                        break
                    continue
                nat.check(lib.psignn_broyden_ext_next_x(self.handle, nat.ptr(xn), sp), "ext_next_x")
                fx = _f32c(f(xn.clone()))
                nat.check(lib.psignn_broyden_ext_update(self.handle, nat.ptr(fx), float(eps), C.byref(done), sp),
                          "ext_update")
                if done.value:
                    break
            result = torch.empty_like(x0c)
            info = nat.SolveInfo()
            rel = (C.c_double * self.threshold)()
            abs_ = (C.c_double * self.threshold)()
            nat.check(lib.psignn_broyden_ext_finish(self.handle, nat.ptr(result), C.byref(info), rel, abs_, sp),
                      "ext_finish")
        out = self._collect(info, rel, abs_, result.shape, result.device)
        out["result"] = result
        return out


def shard_batchable(solvers) -> bool:
    """Whether ``broyden_solve_batch`` takes these solvers together: tiled plans of one family (all dirichlet or all mixed) and one
    size class (``psignn_broyden_batchable``).  A host-side decision -- real errors of the batched solve still raise."""
    n = len(solvers)
    if n == 0:
        return False
    arr = (C.c_void_p * n)(*[s.handle.value for s in solvers])
    return bool(nat.lib().psignn_broyden_batchable(n, arr))


def broyden_solve_batch(solvers, fmaps, eps, poll_every=8):
    """One lockstep device solve of several independent meshes (``psignn_broyden_solve_batch``): ``solvers[i]`` is a
    ``DeviceBroyden`` of ``fmaps[i].plan``; all dirichlet or all mixed.  Returns the list of per-mesh result dicts of
    ``DeviceBroyden.solve`` -- each bit-identical to solving that mesh alone."""
    n = len(solvers)
    if n == 0:
        return []
    if len(fmaps) != n:
        raise nat.NativeError("one FixedPointMap per solver")
    w0 = fmaps[0].weights
    for s, f in zip(solvers, fmaps):
        if f.weights is not w0 and f.weights.flat.data_ptr() != w0.flat.data_ptr():
            raise nat.NativeError("batched solve: all meshes must share one packed weight buffer")
        if s.plan is not f.plan:
            raise nat.NativeError("batched solve: solver and map were built from different plans")
    dev = solvers[0].device
    thr = solvers[0].threshold
    results = [torch.empty_like(f.h0) for f in fmaps]
    arr = lambda ptrs: (C.c_void_p * n)(*ptrs)
    infos = (nat.SolveInfo * n)()
    rel = [(C.c_double * s.threshold)() for s in solvers]
    abs_ = [(C.c_double * s.threshold)() for s in solvers]
    dpp = lambda rows: (C.POINTER(C.c_double) * n)(*[C.cast(r, C.POINTER(C.c_double)) for r in rows])
    with torch.cuda.device(dev):
        nat.check(nat.lib().psignn_broyden_solve_batch(
            n, arr([s.handle.value for s in solvers]), nat.ptr(w0.flat), w0.n_layers, arr([nat.ptr(f.h0) for f in fmaps]),
            arr([nat.ptr(f.prb) for f in fmaps]), arr([nat.ptr(f.nrm) for f in fmaps]) if w0.mixed else None,
            float(eps), int(poll_every), arr([nat.ptr(r) for r in results]), infos,
            dpp(rel), dpp(abs_), nat.stream_ptr(dev)), "psignn_broyden_solve_batch")
    outs = []
    for i, s in enumerate(solvers):
        o = s._collect(infos[i], rel[i], abs_[i], results[i].shape, dev)
        o["result"] = results[i]
        outs.append(o)
    return outs


# ---------------------------------------------------------------------------------------------
# Picard / Anderson on the device (csrc/fpiter.hip)
# ---------------------------------------------------------------------------------------------
class DeviceFixedPointIter:
    """Vector work, norms, stop tests and the small bordered solve of ``forward_iteration`` / ``anderson``
    (utilities/solver.py:301-341, :215-293) on the device.  The caller evaluates f between the calls; nothing is read back
    per iteration unless asked (``poll``)."""

    def __init__(self, n_elems, device, m=2, threshold=50, keep_trace=False):
        h = C.c_void_p()
        self.device, self.M, self.m = device, int(n_elems), int(m)
        self.threshold, self.keep_trace = int(threshold), bool(keep_trace)
        with torch.cuda.device(device):
            nat.check(nat.lib().psignn_fpiter_create(C.byref(h), self.M, self.m, self.threshold, int(self.keep_trace)),
                      "psignn_fpiter_create")
        self.handle = h
        self._fin = weakref.finalize(self, nat.lib().psignn_fpiter_destroy, h)

    def close(self):
        self._fin()

    def _sp(self):
        return nat.stream_ptr(self.device)

    def poll(self):
        d = C.c_int(0)
        nat.check(nat.lib().psignn_fpiter_poll(self.handle, C.byref(d), self._sp()), "psignn_fpiter_poll")
        return bool(d.value)

    # Picard
    def picard_begin(self, x0):
        nat.check(nat.lib().psignn_picard_begin(self.handle, nat.ptr(x0), self._sp()), "psignn_picard_begin")

    def picard_current(self, like):
        x = torch.empty_like(like)
        nat.check(nat.lib().psignn_picard_current_x(self.handle, nat.ptr(x), self._sp()), "psignn_picard_current_x")
        return x

    def picard_update(self, fx, eps):
        nat.check(nat.lib().psignn_picard_update(self.handle, nat.ptr(fx), float(eps), None, self._sp()), "psignn_picard_update")

    # Anderson
    def anderson_begin(self, x0, f0, f1, lam, beta, stop_abs):
        nat.check(nat.lib().psignn_anderson_begin(self.handle, nat.ptr(x0), nat.ptr(f0), nat.ptr(f1), float(lam), float(beta),
                                                  int(stop_abs), self._sp()), "psignn_anderson_begin")

    def anderson_next(self, like):
        x = torch.empty_like(like)
        nat.check(nat.lib().psignn_anderson_next_x(self.handle, nat.ptr(x), self._sp()), "psignn_anderson_next_x")
        return x

    def anderson_update(self, fx, eps):
        nat.check(nat.lib().psignn_anderson_update(self.handle, nat.ptr(fx), float(eps), None, self._sp()),
                  "psignn_anderson_update")

    def finish(self, like):
        result = torch.empty_like(like)
        info = nat.SolveInfo()
        n = self.threshold + 2
        rel, abs_, low = (C.c_double * n)(), (C.c_double * n)(), (C.c_int32 * n)()
        nat.check(nat.lib().psignn_fpiter_finish(self.handle, nat.ptr(result), C.byref(info), rel, abs_, low, self._sp()),
                  "psignn_fpiter_finish")
        k = int(info.n_iter)
        return {"result": result, "n_iter": k, "nstep": int(info.nstep), "lowest": float(info.lowest),
                "lowest_abs": float(info.lowest_abs), "stop_reason": int(info.stop_reason),
                "rel_trace": list(rel[:k]), "abs_trace": list(abs_[:k]), "low_idx": list(low[:k])}

    def iterate(self, i, like):
        dst = torch.empty_like(like)
        nat.check(nat.lib().psignn_fpiter_get_iterate(self.handle, int(i), nat.ptr(dst), self._sp()), "psignn_fpiter_get_iterate")
        return dst


# ---------------------------------------------------------------------------------------------
# GMRES on the device (csrc/krylov.hip)
# ---------------------------------------------------------------------------------------------
class DeviceGmres:
    """Krylov workspace of ``newton_krylov``: the basis is a torch tensor (rows are handed to the JVP kernel as views),
    everything else -- Gram-Schmidt sweeps, Hessenberg / Givens least squares, stop flag -- lives in the library."""

    def __init__(self, n_elems, device, m_max):
        self.M, self.m, self.device = int(n_elems), int(m_max), device
        self.ld = (self.M + 63) // 64 * 64
        self.V = torch.empty((self.m + 1, self.ld), dtype=torch.float32, device=device)
        h = C.c_void_p()
        with torch.cuda.device(device):
            nat.check(nat.lib().psignn_gmres_create(C.byref(h), self.M, self.ld, self.m, nat.ptr(self.V)), "psignn_gmres_create")
        self.handle = h
        self._fin = weakref.finalize(self, nat.lib().psignn_gmres_destroy, h)

    def close(self):
        self._fin()

    def row(self, j, shape):
        return self.V[j, :self.M].view(shape)

    def _sp(self):
        return nat.stream_ptr(self.device)

    def residual_norms(self, x, fx, g=None, neg_g=None):
        """(|fx - x|, |fx|) as the fp32 norms the reference reads back with .item(); optionally stores g and -g."""
        out = (C.c_double * 2)()
        nat.check(nat.lib().psignn_residual_norms(self.handle, nat.ptr(x), nat.ptr(fx), nat.ptr(g), nat.ptr(neg_g), out,
                                                  self._sp()), "psignn_residual_norms")
        return float(out[0]), float(out[1])

    def begin(self, b):
        nat.check(nat.lib().psignn_gmres_begin(self.handle, nat.ptr(b), self._sp()), "psignn_gmres_begin")

    def step(self, j, shift, eta, poll=False):
        d = C.c_int(0)
        nat.check(nat.lib().psignn_gmres_step(self.handle, int(j), float(shift), float(eta), C.byref(d) if poll else None,
                                              self._sp()), "psignn_gmres_step")
        return bool(d.value)

    def solution(self, base, scale, dst, k=0, info=False):
        inf = (C.c_double * 3)()
        nat.check(nat.lib().psignn_gmres_solution(self.handle, int(k), nat.ptr(base), float(scale), nat.ptr(dst),
                                                  inf if info else None, self._sp()), "psignn_gmres_solution")
        return (int(inf[0]), float(inf[1]), float(inf[2])) if info else None

    def reorth_count(self):
        """Arnoldi steps of the current solve whose second Gram-Schmidt pass ran."""
        n = C.c_int(0)
        nat.check(nat.lib().psignn_gmres_reorth_count(self.handle, C.byref(n), self._sp()), "psignn_gmres_reorth_count")
        return int(n.value)

    def history(self):
        h = (C.c_double * (self.m + 1))()
        nat.check(nat.lib().psignn_gmres_history(self.handle, h, self._sp()), "psignn_gmres_history")
        return list(h)

"""PSI-GNN inference models with the reference's constructor, ``forward()`` and ``state_dict`` layout.

Drop-in for ``tests/model_psignn.py`` (``ModelPSIGNN``, ``ModelPSIGNNIterative``),
``dirichlet/psignn/model.py`` (``ModelDEQDSS``: ``forward`` diagnostics / ``inference`` /
``iterative_inference``) and their mixed (Dirichlet + Neumann) counterparts
``mixed/psignn/test/model_test.py`` / ``mixed/psignn/model.py``.

* ``Model(config)`` takes the reference's config dict (``latent_dim, n_layers, solver, fw_tol, fw_thres,
  bw_tol, bw_thres, path_logs``; ``hidden_dim`` ignored as in the reference).  An optional key
  ``"bc"`` = ``"dirichlet"`` | ``"mixed"`` selects the family (the reference uses two copies of the file).
* ``load_state_dict(ckpt["state_dict"])`` of a reference checkpoint works unchanged: parameter names
  and shapes are identical.
* ``batch`` is any object with the PyG ``Data`` attributes (see ``data/meshdata.py``), already on the GPU.

The numerical work — encoder/decoder MLPs, the GNN block f, the Broyden root-find, the residual
SpMV — runs in libpsignn_hip.so.  No torch_geometric / torch_sparse.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import _native as nat
from . import engine
from .utilities import solver as _solver

# ----------------------------------------------------------------------------------------------
# parameter containers (names = the reference's module tree; arithmetic is NOT done by these)
# ----------------------------------------------------------------------------------------------


def initialize_weights_xavier(m, gain=1.0):  # model.py:310-314
    if isinstance(m, nn.Linear):
        nn.init.xavier_uniform_(m.weight, gain=gain)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)


class MLP(nn.Module):
    """Linear, ReLU, Linear (model.py:316-332).  Two-layer only, like every instance in the reference."""

    def __init__(self, hidden_channels=None, activation=None):
        super().__init__()
        if len(hidden_channels) != 3:
            raise nat.NativeError("the HIP path implements the reference's two-layer MLPs only")
        a, b, c = hidden_channels
        self.mlp = nn.Sequential(nn.Linear(a, b), nn.ReLU(), nn.Linear(b, c)).apply(initialize_weights_xavier)

    def forward(self, x):
        l0, l2 = self.mlp[0], self.mlp[2]
        return engine.mlp2_autograd(x, l0.weight, l0.bias, l2.weight, l2.bias)


class _Phi(nn.Module):
    """Parameter holder of ``Phi_to`` / ``Phi_from`` (model.py:334-368)."""

    def __init__(self, hidden_channels=None, activation=None):
        super().__init__()
        self.mlp = MLP(hidden_channels, activation)


class Phi_to(_Phi):
    pass


class Phi_from(_Phi):
    pass


class Encoder(nn.Module):
    def __init__(self, hidden_channels=None, activation=None):
        super().__init__()
        self.mlp = MLP(hidden_channels, activation)

    def forward(self, x):
        return self.mlp(x)


class Decoder(Encoder):
    pass


class Autoencoder(nn.Module):  # model.py:394-406
    def __init__(self, hidden_channels=None, activation=None):
        super().__init__()
        self.encoder = Encoder(hidden_channels, activation)
        self.decoder = Decoder(list(reversed(hidden_channels)), activation)

    def forward(self, x, sens):
        if sens == "latent":
            return self.encoder(self.decoder(x))
        if sens == "physics":
            return self.decoder(self.encoder(x))
        print("Specify autoencoder direction")


class Function(nn.Module):
    """The GNN block f_theta (dirichlet: model.py:263-300; mixed: mixed/psignn/model.py:196-245)."""

    def __init__(self, n_layers=None, latent_dim=None, edge_features_dim=None, second_member_dim=None,
                 activation=None, mixed=False):
        super().__init__()
        if latent_dim != engine.D or edge_features_dim != 3:
            raise nat.NativeError(f"HIP kernels are built for latent_dim={engine.D}, edge_features_dim=3")
        self.n_layers, self.mixed = n_layers, mixed
        d, p = latent_dim, second_member_dim
        self.laynorm = nn.LayerNorm(d)
        self.phi_to_list = nn.ModuleList([Phi_to([2 * d + 3, d, d], activation) for _ in range(n_layers)])
        self.phi_from_list = nn.ModuleList([Phi_from([2 * d + 3, d, d], activation) for _ in range(n_layers)])
        self.alpha = nn.Sequential(nn.Linear(3 * d + p, 1), nn.Sigmoid()).apply(initialize_weights_xavier)
        self.update_list = nn.ModuleList([MLP([3 * d + p, d, d], activation) for _ in range(n_layers)])
        if mixed:
            self.phi_neumann = Phi_from([2 * d + 3, d, d], activation)
            self.update_neumann = MLP([2 * d + p + 2, d, d], activation)
        self._packed = None
        self._packed_key = None

    def packed(self, device) -> engine.PackedWeights:
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._packed_key != key:
            sd = {"deqdss.f." + k: v for k, v in self.state_dict().items()}
            self._packed = engine.PackedWeights(sd, device)
            self._packed_key = key
        return self._packed

    def bind(self, h_initial, batch) -> engine.FixedPointMap:
        """The map H -> f(H, h_initial, batch) as a device object the solvers understand."""
        plan = engine.plan_for(batch)
        nrm = getattr(batch, "unit_normal_vector", None) if self.mixed else None
        return engine.FixedPointMap(plan, self.packed(h_initial.device), h_initial, batch.prb_data, nrm)

    def forward(self, h, h_initial, batch):
        return self.bind(h_initial, batch)(h)


class _DEQFn(torch.autograd.Function):
    """new_H = f(H*) with H* = solver(f, H_init), differentiable the way the reference's training variant is
    (dirichlet/psignn/model.py:184-225): the forward solve runs without a graph; backward replaces the incoming
    gradient by the solution y of  y = J_f(H*)^T y + grad  (the reference's ``backward_hook``) and pushes y through
    one application of f: parameter gradients from the HIP parameter-VJP kernels, and y on the Dirichlet rows for
    ``H_init`` (those rows of f are copies of ``H_init``, model.py:298)."""

    @staticmethod
    def forward(ctx, H_init, deq, batch, names, *params):
        cfg = deq.config_deq
        H0 = H_init.detach()
        fmap = deq.f.bind(H0, batch)
        if cfg["solver"] is _solver.broyden:   # forward solver state kept between training steps on the same plan
            old = getattr(deq, "_fw_key", None)
            if old is None or old[0] is not fmap.plan or old[1] != cfg["fw_thres"]:
                if getattr(deq, "_fw_solver", None) is not None:
                    deq._fw_solver.close()
                deq._fw_solver = engine.DeviceBroyden(plan=fmap.plan, threshold=cfg["fw_thres"], keep_trace=False)
                deq._fw_key = (fmap.plan, cfg["fw_thres"])
            out_fw = _solver.broyden(fmap, H0, threshold=cfg["fw_thres"], eps=cfg["fw_tol"], keep_trace=False,
                                     solver_obj=deq._fw_solver)
        else:
            out_fw = cfg["solver"](fmap, H0, threshold=cfg["fw_thres"], eps=cfg["fw_tol"])
        H_star = out_fw["result"]
        deq.last_forward = out_fw
        _log(deq.path_logs, "forward_iteration.csv", "\n{} \t {}".format(out_fw["lowest"], out_fw["nstep"]))
        ctx.deq, ctx.batch, ctx.names, ctx.fmap = deq, batch, names, fmap
        ctx.save_for_backward(H_star, H0)
        return fmap(H_star)

    @staticmethod
    def backward(ctx, grad):
        H_star, H0 = ctx.saved_tensors
        deq = ctx.deq
        out_bw = deq.implicit_backward(H_star, H0, ctx.batch, grad.contiguous())
        deq.last_backward = out_bw
        _log(deq.path_logs, "backward_iteration.csv", "\n{} \t {}".format(out_bw["lowest"], out_bw["nstep"]))
        y = out_bw["result"]
        grads, _ = ctx.fmap.param_vjp(H_star, y)
        tags = ctx.batch.tags.reshape(y.shape[0], -1)
        col = 1 if deq.f.mixed else 0   # one-hot [interior, dirichlet, neumann] in the mixed family
        g_init = torch.where(tags[:, col:col + 1] == 1, y, torch.zeros_like(y))
        return (g_init, None, None, None) + tuple(grads[n] for n in ctx.names)


class _JacLossFn(torch.autograd.Function):
    """jac_loss = |v^T J_f(H*)|^2 / (N d) with its gradient w.r.t. the parameters of f: the reference builds the VJP with
    ``create_graph=True`` (jac_loss_estimate, dirichlet/psignn/model.py:416-435) and lets ``loss.backward()`` run the
    double backward; here backward is the HIP backward-of-the-VJP (csrc/gather_backward.hip).  H* is a leaf in the
    reference (model.py:204), so nothing flows back into the solve."""

    @staticmethod
    def forward(ctx, fmap, H_star, v, names, *params):
        g = fmap.vjp(H_star, v)
        ctx.fmap, ctx.names = fmap, names
        ctx.save_for_backward(H_star, v, g)
        return g.norm() ** 2 / H_star.numel()

    @staticmethod
    def backward(ctx, grad_out):
        H_star, v, g = ctx.saved_tensors
        grads, _ = ctx.fmap.vjp_backward(H_star, v, g * (2.0 * grad_out / H_star.numel()))
        return (None, None, None, None) + tuple(grads[n] for n in ctx.names)


def _log(path_logs, name, line):
    if path_logs:
        with open(os.path.join(path_logs, name), "a") as f:
            f.write(line)


class DeepEquilibrium(nn.Module):
    """Inference variant (tests/model_psignn.py:216-243): one solver call, returns the solver dict; plus the
    training variant's differentiable forward (dirichlet/psignn/model.py:184-243) as ``train_forward``."""

    def __init__(self, function=None, config_deq=None):
        super().__init__()
        self.f = function
        self.config_deq = config_deq
        self.path_logs = self.config_deq.get("path_logs")

    def forward(self, H_init, batch):
        return self.config_deq["solver"](self.f.bind(H_init, batch), H_init,
                                         threshold=self.config_deq["fw_thres"], eps=self.config_deq["fw_tol"])

    inference = forward  # dirichlet/psignn/model.py:245-253

    def train_forward(self, H_init, batch, generator=None):
        """(new_H_star, jacobian_loss) of the training variant.  With gradients enabled new_H_star carries the
        implicit-function backward; without (validation) the spectral radius is logged like the reference does.
        The Jacobian regulariser carries its gradient w.r.t. the parameters of f (``_JacLossFn``; the reference's launch
        scripts train with ``jac_weight 1.0``)."""
        if torch.is_grad_enabled():
            if self.f.n_layers != 1:
                raise nat.NativeError("the training path is implemented for single-layer blocks")
            named = list(self.f.named_parameters())
            new_H = _DEQFn.apply(H_init, self, batch, tuple(n for n, _ in named), *[p for _, p in named])
            H_star = self.last_forward["result"]
        else:
            out_fw = self.forward(H_init, batch)
            self.last_forward = out_fw
            H_star = out_fw["result"]
            new_H = self.f.bind(H_init, batch)(H_star)
        if torch.is_grad_enabled():
            # differentiable w.r.t. the parameters of f, as in the reference (vecs = 1, model.py:207)
            v = torch.randn(H_star.shape, device=H_star.device, generator=generator)
            self.last_probe = v
            jac_loss = _JacLossFn.apply(self.f.bind(H_init.detach(), batch), H_star, v, tuple(n for n, _ in named),
                                        *[p for _, p in named])
            return new_H, jac_loss
        with torch.no_grad():
            jac_loss = self.jac_loss_estimate(H_star, H_init.detach(), batch, vecs=1, generator=generator)
            if not torch.is_grad_enabled() and self.path_logs:
                _, sradius = self.power_method(H_star, H_init.detach(), batch, n_iters=150, generator=generator)
                _log(self.path_logs, "spectral_radius.csv", "\n{}".format(sradius.item()))
        return new_H, jac_loss

    # ---- adjoint side of the reference's training variant (dirichlet/psignn/model.py:204-241), on the VJP kernel
    def implicit_backward(self, H_star, H_init, batch, grad):
        """Solve y = J_f(H*)^T y + grad with the configured solver (the reference's backward hook, model.py:210-223):
        returns the solver dict; ``out["result"]`` is the gradient w.r.t. the fixed point's input."""
        fmap = self.f.bind(H_init, batch)
        g = grad.contiguous()
        if self.config_deq["solver"] is _solver.broyden:  # whole adjoint solve on the device
            # the solver state (2 * bw_thres * N * d floats) is kept between calls on the same plan: a training loop
            # would otherwise allocate and free it once per step
            key = (fmap.plan, self.config_deq["bw_thres"])
            old = getattr(self, "_bw_key", None)
            if old is None or old[0] is not key[0] or old[1] != key[1]:
                if getattr(self, "_bw_solver", None) is not None:
                    self._bw_solver.close()
                self._bw_solver = engine.DeviceBroyden(plan=fmap.plan, threshold=self.config_deq["bw_thres"], keep_trace=False)
                self._bw_key = key
            sv = self._bw_solver
            out = sv.solve_adjoint(fmap, H_star, g, self.config_deq["bw_tol"])
            out.update(eps=self.config_deq["bw_tol"], threshold=self.config_deq["bw_thres"])
            return out
        return self.config_deq["solver"](lambda y: fmap.vjp(H_star, y) + g, torch.zeros_like(g),
                                         threshold=self.config_deq["bw_thres"], eps=self.config_deq["bw_tol"])

    @staticmethod
    def _vjp_in_plan_order(fmap, H_star):
        """(vjp, to_plan, from_plan) working in plan order where the tiled VJP applies (saves the four permutation
        passes of the caller-order entry point per product; norms and inner products do not depend on the numbering)."""
        if fmap.plan.tiled and (fmap.weights.mixed or fmap.weights.n_layers == 1):
            Hp = fmap.to_plan(H_star)
            return (lambda w: fmap.vjp_p(Hp, w)), fmap.to_plan, fmap.from_plan
        ident = lambda t: t
        return (lambda w: fmap.vjp(H_star, w)), ident, ident

    def jac_loss_estimate(self, H_star, H_init, batch, vecs=1, generator=None, probes=None):
        """Hutchinson estimate of tr(J^T J) / (N d) (model.py:416-435) with the VJP kernel.  ``probes``: the Gaussian
        vectors to use instead of drawing ``vecs`` of them (so that a test can fix them)."""
        fmap = self.f.bind(H_init, batch)
        vjp, to_p, _ = self._vjp_in_plan_order(fmap, H_star)
        acc = 0.0
        n = vecs if probes is None else len(probes)
        for i in range(n):
            v = torch.randn(H_star.shape, device=H_star.device, generator=generator) if probes is None else probes[i]
            acc = acc + vjp(to_p(v)).norm() ** 2
        return acc / n / H_star.numel()

    def power_method(self, H_star, H_init, batch, n_iters=150, generator=None, v0=None):
        """Spectral-radius estimate of J by power iteration on v^T J (model.py:437-452).  ``v0``: start vector instead
        of a Gaussian draw."""
        fmap = self.f.bind(H_init, batch)
        vjp, to_p, from_p = self._vjp_in_plan_order(fmap, H_star)
        ev = to_p(torch.randn(H_star.shape, device=H_star.device, generator=generator) if v0 is None else v0)
        val = torch.zeros((), device=H_star.device)
        for _ in range(n_iters):
            vj = vjp(ev)
            val = (vj * ev).sum() / (ev * ev).sum()
            ev = vj / vj.norm()
        return from_p(ev), val.abs()


# ----------------------------------------------------------------------------------------------
# models
# ----------------------------------------------------------------------------------------------
class _Base(nn.Module):
    MIXED = False

    def __init__(self, config):
        super().__init__()
        self.config = dict(config)
        bc = self.config.get("bc")
        self.mixed = self.MIXED if bc is None else (bc == "mixed")
        self.config.setdefault("solver", _solver.broyden)
        for k, v in (("fw_tol", 1e-5), ("fw_thres", 300), ("bw_tol", 1e-8), ("bw_thres", 300), ("path_logs", None)):
            self.config.setdefault(k, v)
        d = self.config["latent_dim"]
        self.autoencoder = Autoencoder(hidden_channels=[1, d, d], activation=nn.ReLU())
        self.config_deq = {k: self.config[k] for k in ("solver", "fw_tol", "fw_thres", "bw_tol", "bw_thres", "path_logs")}
        self.deqdss = DeepEquilibrium(
            function=Function(n_layers=self.config["n_layers"], latent_dim=d, edge_features_dim=3,
                              second_member_dim=3 if self.mixed else 2, activation=nn.ReLU(), mixed=self.mixed),
            config_deq=self.config_deq)
        self.mse_loss = nn.MSELoss()

    # -- helpers -------------------------------------------------------------------------------
    def _dirichlet_index(self, batch):
        t = batch.tags[:, 1] if self.mixed else batch.tags.reshape(batch.tags.shape[0], -1)[:, 0]
        return torch.where(t == 1)[0]

    def residual_loss(self, u, batch):
        """mean((A u - y)^2), A incl. the diagonal (model.py:157-167)."""
        r = engine.residual_autograd(engine.plan_for(batch), u, batch.y, batch.a_ij)
        return torch.mean(r ** 2)

    @torch.no_grad()
    def _solve(self, batch):
        nat.require_cuda(batch.x, "batch.x")
        h_initial = self.autoencoder.encoder(batch.x)
        out = self.deqdss(h_initial, batch)
        return h_initial, out

    @torch.no_grad()
    def _diagnostics(self, u_final, h_final, batch, key_dir, idx=None):
        enc = self.autoencoder.encoder(u_final)
        if idx is None:
            idx = self._dirichlet_index(batch)
        return {
            "residual_loss": self.residual_loss(u_final, batch),
            "encoder_loss": self.mse_loss(enc, h_final),
            "autoencoder_loss": self.mse_loss(self.autoencoder.decoder(enc), u_final),
            "mse_loss": self.mse_loss(u_final, batch.sol),
            key_dir: self.mse_loss(u_final[idx, :], batch.x[idx, :]),
        }

    @torch.no_grad()
    def inference(self, batch):
        """u_final only (dirichlet/psignn/model.py:99-107)."""
        _, out = self._solve(batch)
        return self.autoencoder.decoder(out["result"])

    @torch.no_grad()
    def _iterative(self, batch):
        """tests/model_psignn.py:145-194 ≡ dirichlet/psignn/model.py:109-155."""
        out_dic = {"sol_dic": [], "res_dic": [], "mse_dic": [], "bound_mse_dic": [], "inter_mse_dic": [], "nstep": []}
        if self.mixed:
            ib = torch.where(batch.tags[:, 1] == 1)[0]
            ii = torch.where(batch.tags[:, 0] == 1)[0]
        else:
            ib = torch.where(batch.tags == 1)[0]
            ii = torch.where(batch.tags == 0)[0]

        def record(u):
            out_dic["sol_dic"].append(u.cpu())
            out_dic["res_dic"].append(self.residual_loss(u, batch).cpu().item())
            out_dic["mse_dic"].append(torch.mean((u - batch.sol) ** 2).cpu().item())
            out_dic["bound_mse_dic"].append(torch.mean((u[ib, :] - batch.sol[ib, :]) ** 2).cpu().item())
            out_dic["inter_mse_dic"].append(torch.mean((u[ii, :] - batch.sol[ii, :]) ** 2).cpu().item())

        record(batch.x)
        _, out_fw = self._solve(batch)
        for h_star in out_fw["xest_trace"]:
            record(self.autoencoder.decoder(h_star))
        out_dic["nstep"] = out_fw["nstep"]
        return out_dic


class ModelPSIGNN(_Base):
    """tests/model_psignn.py:28-112: ``forward(batch) -> (u_final, loss_dic incl. 'nsteps')``."""

    @torch.no_grad()
    def forward(self, batch):
        _, out = self._solve(batch)
        h_final = out["result"]
        u_final = self.autoencoder.decoder(h_final)
        loss_dic = self._diagnostics(u_final, h_final, batch, "mse_dirichlet_loss")
        loss_dic["nsteps"] = out["nstep"]
        return u_final, loss_dic


class ModelPSIGNNIterative(_Base):
    """tests/model_psignn.py:114-206: per-iterate diagnostics dictionary."""

    def forward(self, batch):
        return self._iterative(batch)


class ModelDEQDSS(_Base):
    """dirichlet/psignn/model.py:28-167: ``forward`` (training: differentiable, implicit backward; eval: diagnostics),
    ``inference``, ``iterative_inference``."""

    def forward(self, batch):
        if self.training and torch.is_grad_enabled():
            return self._train_forward(batch)
        with torch.no_grad():  # validation branch of DeepEquilibrium.forward (model.py:227-241): one more f on H*
            nat.require_cuda(batch.x, "batch.x")
            h_initial = self.autoencoder.encoder(batch.x)
            h_final, jacobian_loss = self.deqdss.train_forward(h_initial, batch)
            u_final = self.autoencoder.decoder(h_final)
            # ModelDEQDSS.forward is ONE function for both modes: its Dirichlet statistic uses where(tags == 1)[0] also on
            # the mixed family's one-hot (N, 3) tags, i.e. every row (mixed/psignn/model.py:87) -- as _train_forward below
            loss_dic = self._diagnostics(u_final, h_final, batch, "mse_dirichlet", idx=torch.where(batch.tags == 1)[0])
            loss_dic["jacobian_loss"] = jacobian_loss
            return u_final, loss_dic

    def _train_forward(self, batch):
        """dirichlet/psignn/model.py:58-99.  Every term carries the gradient it has in the reference: residual_loss
        through decoder, implicit DEQ backward, f parameters and encoder; encoder_loss into the encoder;
        autoencoder_loss into the decoder only (the encoder output is detached there)."""
        nat.require_cuda(batch.x, "batch.x")
        ae = self.autoencoder
        h_initial = ae.encoder(batch.x)
        h_final, jacobian_loss = self.deqdss.train_forward(h_initial, batch)
        u_final = ae.decoder(h_final)
        residual_loss = self.residual_loss(u_final, batch)
        u_d, h_d = u_final.detach(), h_final.detach()
        encoder_loss = self.mse_loss(ae.encoder(u_d), h_d)
        autoencoder_loss = self.mse_loss(ae.decoder(ae.encoder(u_d).detach()), u_d)
        # model.py:87 in BOTH families: where(tags == 1)[0]; on the mixed family's one-hot (N, 3) tags that is every row
        idx = torch.where(batch.tags == 1)[0]
        loss_dic = {"residual_loss": residual_loss, "jacobian_loss": jacobian_loss, "encoder_loss": encoder_loss,
                    "autoencoder_loss": autoencoder_loss, "mse_loss": self.mse_loss(u_final, batch.sol),
                    "mse_dirichlet": self.mse_loss(u_final[idx, :], batch.x[idx, :])}
        return u_final, loss_dic

    def iterative_inference(self, batch):
        return self._iterative(batch)

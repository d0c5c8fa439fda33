"""DSS, the Deep Statistical Solver baseline of the reference (dirichlet/dss/model.py:25-147), on the HIP tile kernels.

``DeepStatisticalSolver(config)`` takes the reference's config (``latent_dim, k, alpha, gamma, path_logs``) and has the
reference's module tree (``phi_to_list / phi_from_list / psi_list / decoder_list``, one set of weights per update), so
``load_state_dict(ckpt["state_dict"])`` of ``dirichlet/dss/results/dss_results/ckpt/best_model.pt`` works unchanged.

* ``inference(batch) -> U_k``   (model.py:97-120)
* ``forward(batch) -> (U, loss_dic)``: every decoded iterate and the ``residual_loss / mse_loss / mse_dirichlet_loss``
  traces plus ``train_loss`` (model.py:59-95, tests/model_dss.py:58-104); in train mode ``train_loss`` carries its gradients
  (back-propagation through the k updates on ``psignn_dss_step_backward``; trainer: ``training_class.TrainModelDSS``)
* ``residual_loss(U, edge_index, a_ij, b_prime)``   (model.py:122-139).

The copy under the reference's ``tests/`` names the edge fields ``edge_attr / edge_attr_norm`` instead of
``a_ij / a_ij_norm``; both spellings are accepted.

``batch`` carries the DSS schema of ``dirichlet/dss/utilities/reader.py:61-92`` (``edge_index`` without self loops,
``a_ij``, ``a_ij_norm``, ``b_prime``, ``b_prime_norm``, ``pos``); ``to_dss_batch(mesh)`` derives it from a PSI-GNN
``MeshData`` the way ``dirichlet/dataset/generate_data.py:100-128`` derives ``A_prime / b_prime`` from ``A / b``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _native as nat
from . import engine
from .data.meshdata import MeshData
from .model_psignn import MLP, Decoder, Phi_from, Phi_to

AIJ_MEAN, AIJ_STD = -0.5838, 0.0924                      # reader.py:63-64
B_MEAN, B_STD = (0.0002, 0.1435, -0.0006), (0.0507, 0.3506, 3.2935)   # reader.py:66-67


def to_dss_batch(mesh) -> MeshData:
    """A' = A without its diagonal (Dirichlet rows are identity rows: they keep no edge), b' = (b, 0, 0) on interior rows
    and (0, 1, g) on Dirichlet rows, both normalised with the reference's constants."""
    ei = mesh.edge_index
    keep = ei[0] != ei[1]
    a = mesh.a_ij.reshape(-1, 1)[keep]
    diri = mesh.tags.reshape(mesh.tags.shape[0], -1)[:, 0] == 1
    b = mesh.y.reshape(-1)
    zero = torch.zeros_like(b)
    bp = torch.stack([torch.where(diri, zero, b), diri.to(b.dtype), torch.where(diri, b, zero)], dim=1)
    mean = torch.tensor(B_MEAN, dtype=b.dtype, device=b.device)
    std = torch.tensor(B_STD, dtype=b.dtype, device=b.device)
    out = MeshData(x=mesh.sol, sol=mesh.sol, edge_index=ei[:, keep], a_ij=a, a_ij_norm=(a - AIJ_MEAN) / AIJ_STD,
                   b_prime=bp, b_prime_norm=(bp - mean) / std, pos=mesh.pos, tags=mesh.tags)
    for k in ("batch", "ptr", "num_graphs"):      # a union batch stays one (the plan tiles graph by graph)
        if getattr(mesh, k, None) is not None:
            setattr(out, k, getattr(mesh, k))
    return out


class _DssStepFn(torch.autograd.Function):
    """Update t, H_t -> H_{t+1} (dirichlet/dss/model.py:75-83), as an autograd node: forward on the tile kernel, backward on
    ``psignn_dss_step_backward`` (VJP w.r.t. H_t and the gradients of update t's phi_to / phi_from / psi)."""

    @staticmethod
    def forward(ctx, h, sc, t, names, *params):
        hp = sc["plan"].permute(h.detach(), True)
        out = sc["plan"].permute(engine.dss_step_p(sc["plan"], sc["w"], t, sc["alpha"], hp, sc["bp"]), False)
        ctx.sc, ctx.t, ctx.names = sc, t, names
        ctx.save_for_backward(h.detach())
        return out

    @staticmethod
    def backward(ctx, w):
        (h,) = ctx.saved_tensors
        sc, t = ctx.sc, ctx.t
        wf = engine.pack_dss_train(sc["sd"], t, h.device)
        grads, dh = engine.dss_step_backward(sc["plan"], wf, t, sc["alpha"], h, sc["b"], w.contiguous())
        return (dh, None, None, None) + tuple(grads[n] for n in ctx.names)


class Psi(nn.Module):
    def __init__(self, hidden_channels=None, activation=None):
        super().__init__()
        self.mlp = MLP(hidden_channels, activation)


class DeepStatisticalSolver(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = dict(config)
        d, k = self.config["latent_dim"], self.config["k"]
        if d != engine.D:
            raise nat.NativeError(f"HIP kernels are built for latent_dim = {engine.D}")
        self.phi_to_list = nn.ModuleList([Phi_to([2 * d + 1, d, d], nn.ReLU()) for _ in range(k)])
        self.phi_from_list = nn.ModuleList([Phi_from([2 * d + 1, d, d], nn.ReLU()) for _ in range(k)])
        self.psi_list = nn.ModuleList([Psi([3 * d + 3, d, d], nn.ReLU()) for _ in range(k)])
        self.decoder_list = nn.ModuleList([Decoder([d, d, 1], nn.ReLU()) for _ in range(k)])
        self.mse_loss = nn.MSELoss()
        self._packed, self._packed_key = None, None

    def packed(self, device):
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._packed_key != key:
            self._packed = engine.pack_dss(self.state_dict(), self.config["k"], device)
            self._packed_key = key
        return self._packed

    @staticmethod
    def _fields(batch):
        """(a_ij, a_ij_norm) under either spelling (dirichlet/dss/model.py vs tests/model_dss.py)."""
        a = getattr(batch, "a_ij", None)
        an = getattr(batch, "a_ij_norm", None)
        if an is None:
            a, an = batch.edge_attr, batch.edge_attr_norm
        return a, an

    @staticmethod
    def _plan(batch):
        """Mesh plan of the DSS graph: the scalar edge feature rides in the third edge_attr column."""
        plan = getattr(batch, "_dss_plan", None)
        if plan is None:
            a_ij, a_norm = DeepStatisticalSolver._fields(batch)
            z = torch.zeros((a_norm.shape[0], 2), dtype=torch.float32, device=a_norm.device)
            view = MeshData(x=batch.x, edge_index=batch.edge_index, a_ij=a_ij,
                            edge_attr=torch.cat([z, a_norm.reshape(-1, 1).float()], dim=1).contiguous(),
                            tags=torch.zeros((batch.x.shape[0], 1), dtype=torch.float32, device=batch.x.device),
                            pos=batch.pos)
            if getattr(batch, "batch", None) is not None:
                view.batch = batch.batch
            plan = engine.MeshPlan(view)
            batch._dss_plan = plan
        return plan

    @torch.no_grad()
    def latent(self, batch, k=None):
        nat.require_cuda(batch.x, "batch.x")
        k = self.config["k"] if k is None else k
        if not 1 <= k <= self.config["k"]:
            raise nat.NativeError("DSS has one set of weights per update: 1 <= k <= config['k']")
        return engine.dss_forward(self._plan(batch), self.packed(batch.x.device), batch.b_prime_norm, k, self.config["alpha"])

    @torch.no_grad()
    def inference(self, batch):
        k = self.config["k"]
        return self.decoder_list[k - 1](self.latent(batch, k))

    def forward(self, batch):
        if self.training and torch.is_grad_enabled():
            return self._train_forward(batch)
        with torch.no_grad():
            return self._eval_forward(batch)

    def _train_forward(self, batch):
        """model.py:59-95 with gradients: ``train_loss`` = sum_t gamma^(k-t-1) residual(U_t) back-propagates through decoder t
        and the updates 0..t-1 (``_DssStepFn`` nodes); the residual's sparse product runs on the SpMV kernel of the plan."""
        nat.require_cuda(batch.x, "batch.x")
        k, alpha, gamma = self.config["k"], self.config["alpha"], self.config["gamma"]
        plan, dev = self._plan(batch), batch.x.device
        a_ij, _ = self._fields(batch)
        idx = torch.where(batch.b_prime[:, 1] == 1)[0]
        zeros = torch.zeros((plan.N, 1), dtype=torch.float32, device=dev)
        rowsum = engine.residual(plan, torch.ones_like(zeros), zeros)
        B0, B1, B2 = batch.b_prime[:, 0:1], batch.b_prime[:, 1:2], batch.b_prime[:, 2:3]

        def residual(u):   # model.py:122-139: sum_j a_ij (u_j - u_i) = (A' u)_i - rowsum_i u_i
            r = (1 - B1) * (-B0) + B1 * (u - B2) + engine.residual_autograd(plan, u, zeros, a_ij) - rowsum * u
            return torch.mean(r ** 2)
        sc = {"plan": plan, "w": self.packed(dev), "alpha": alpha, "bp": plan.permute(batch.b_prime_norm, True),
              "b": batch.b_prime_norm, "sd": self.state_dict()}
        h = torch.zeros((plan.N, engine.D), dtype=torch.float32, device=dev)
        U = {"0": self.decoder_list[0](h) + batch.x * 0}
        res, mse = {"0": residual(U["0"])}, {"0": self.mse_loss(U["0"], batch.x)}
        msd = {"0": self.mse_loss(U["0"][idx, :], batch.x[idx, :])}
        total = None
        for t in range(k):
            named = [(n, p) for n, p in self.named_parameters()
                     if n.startswith((f"phi_to_list.{t}.", f"phi_from_list.{t}.", f"psi_list.{t}."))]
            h = _DssStepFn.apply(h, sc, t, tuple(n for n, _ in named), *[p for _, p in named])
            s = str(t + 1)
            U[s] = self.decoder_list[t](h)
            res[s], mse[s] = residual(U[s]), self.mse_loss(U[s], batch.x)
            msd[s] = self.mse_loss(U[s][idx, :], batch.x[idx, :])
            term = res[s] * gamma ** (k - t - 1)
            total = term if total is None else total + term
        return U, {"train_loss": total, "residual_loss": res, "mse_loss": mse, "mse_dirichlet_loss": msd}

    def _eval_forward(self, batch):
        nat.require_cuda(batch.x, "batch.x")
        k, alpha, gamma = self.config["k"], self.config["alpha"], self.config["gamma"]
        plan, w = self._plan(batch), self.packed(batch.x.device)
        a_ij, _ = self._fields(batch)
        idx = torch.where(batch.b_prime[:, 1] == 1)[0]
        hp = torch.zeros((plan.N, engine.D), dtype=torch.float32, device=batch.x.device)
        bp = plan.permute(batch.b_prime_norm, True)
        U = {"0": self.decoder_list[0](hp) + batch.x * 0}     # H_0 = 0 in any numbering
        res = {"0": self.residual_loss(U["0"], batch.edge_index, a_ij, batch.b_prime)}
        mse = {"0": self.mse_loss(U["0"], batch.x)}
        msd = {"0": self.mse_loss(U["0"][idx, :], batch.x[idx, :])}
        total = None
        for t in range(k):
            hp = engine.dss_step_p(plan, w, t, alpha, hp, bp)
            s = str(t + 1)
            U[s] = self.decoder_list[t](plan.permute(hp, False))
            res[s] = self.residual_loss(U[s], batch.edge_index, a_ij, batch.b_prime)
            mse[s] = self.mse_loss(U[s], batch.x)
            msd[s] = self.mse_loss(U[s][idx, :], batch.x[idx, :])
            term = res[s] * gamma ** (k - t - 1)
            total = term if total is None else total + term
        return U, {"train_loss": total, "residual_loss": res, "mse_loss": mse, "mse_dirichlet_loss": msd}

    @torch.no_grad()
    def residual_loss(self, U, edge_index, a_ij, y):
        """model.py:122-139: mean(((1 - B1)(-B0) + B1 (U - B2) + sum_j a_ij (U_j - U_i))^2)."""
        B0, B1, B2 = y[:, 0:1], y[:, 1:2], y[:, 2:3]
        p1 = (1 - B1) * (-B0) + B1 * (U - B2)
        fbar = a_ij * (U.index_select(0, edge_index[1]) - U.index_select(0, edge_index[0]))
        s = torch.zeros_like(U).index_add_(0, edge_index[0], fbar)
        return torch.mean((p1 + s) ** 2)

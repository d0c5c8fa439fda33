"""DS-GPS, the reference's unrolled recurrent baseline (dirichlet/dsgps/model.py:28-176), on the HIP tile kernels.

``ModelDSGPS(config)`` takes the reference's config (``latent_dim, k, alpha, gamma, path_logs``), has the reference's
module tree -- so ``load_state_dict(ckpt["state_dict"])`` of ``dirichlet/dsgps/results/**/best_model.pt`` works
unchanged -- and the reference's two entry points:

* ``inference(batch) -> U_k``                                   (model.py:130-163)
* ``forward(batch) -> (U, loss_dic)``: every decoded iterate ``U['0'..'k']`` and the per-step loss dictionaries
  (``residual_loss, encoder_loss, autoencoder_loss, mse_dirichlet, mse_loss`` keyed '0'..'k', ``train_loss``)
  (model.py:48-128).  In train mode ``train_loss`` carries its gradients (both families): back-propagation through
  the k unrolled updates runs on ``psignn_dsgps_step_backward`` (``training_class.TrainModelDSGPS`` is the trainer of
  dirichlet/dsgps/training_class.py); in eval mode / under ``no_grad`` the same dictionaries come as diagnostics.

Same gather -> edge MLP -> segment-sum kernels as PSI-GNN's f; k launches back to back, no solver.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _native as nat
from . import engine
from .model_psignn import MLP, Autoencoder, Phi_from, Phi_to, initialize_weights_xavier


class _StepCtx:
    """Per-forward constants of the unrolled training pass: plan, packed weights, plan-order inputs."""

    def __init__(self, net, batch, H0):
        dev = H0.device
        self.plan = engine.plan_for(batch)
        self.w = net.packed(dev)
        self.wf, self.wg = engine.pack_dsgps_train(net.state_dict(), dev)
        self.prb = batch.prb_data
        self.prbp = self.plan.permute(batch.prb_data, True)
        self.h0p = self.plan.permute(H0.detach(), True)
        self.nrm = batch.unit_normal_vector if net.mixed else None
        self.nrmp = self.plan.permute(self.nrm, True) if net.mixed else None
        col = 1 if net.mixed else 0   # mixed: one-hot [interior, dirichlet, neumann]
        self.dirichlet = batch.tags.reshape(batch.tags.shape[0], -1)[:, col:col + 1] == 1


class _DsgpsStepFn(torch.autograd.Function):
    """One recurrent update H_t -> H_{t+1} (dirichlet/dsgps/model.py:72-89) as an autograd node: forward on the tile kernel,
    backward on ``psignn_dsgps_step_backward`` (VJP w.r.t. H_t, parameter gradients of phi_to / phi_from and the three
    gates; the Dirichlet rows' cotangent goes to H_0).  Autograd chains the k nodes: back-propagation through time."""

    @staticmethod
    def forward(ctx, h, h0, sc, names, *params):
        hp = sc.plan.permute(h.detach(), True)
        out = sc.plan.permute(engine.dsgps_step_p(sc.plan, sc.w, hp, sc.h0p, sc.prbp, sc.nrmp), False)
        ctx.sc, ctx.names = sc, names
        ctx.save_for_backward(h.detach())
        return out

    @staticmethod
    def backward(ctx, w):
        (h,) = ctx.saved_tensors
        sc = ctx.sc
        w = w.contiguous()
        grads, dh = engine.dsgps_step_backward(sc.plan, sc.wf, sc.wg, h, sc.prb, w, sc.nrm)
        g0 = torch.where(sc.dirichlet, w, torch.zeros_like(w))
        return (dh, g0, None, None) + tuple(grads[n] for n in ctx.names)


class MLPActivation(nn.Module):
    """Linear + activation (dirichlet/dsgps/model.py:209-225); single layer, as every instance in the reference."""

    def __init__(self, hidden_channels=None, activation=None):
        super().__init__()
        if len(hidden_channels) != 2:
            raise nat.NativeError("the HIP path implements the reference's single-layer gate blocks only")
        self.mlp = nn.Sequential(nn.Linear(hidden_channels[0], hidden_channels[1]), activation).apply(initialize_weights_xavier)


class ModelDSGPS(nn.Module):
    """``config["bc"] = "mixed"`` (or ``mixed=True``) selects mixed/dsgps/model.py: 3 problem features, Phi_neumann and
    update_neumann (Neumann rows are replaced by their output), Dirichlet rows in tag column 1."""

    def __init__(self, config, mixed=None):
        super().__init__()
        self.config = dict(config)
        self.mixed = (self.config.get("bc") == "mixed") if mixed is None else bool(mixed)
        d, p = self.config["latent_dim"], 3 if self.mixed else 2
        if d != engine.D:
            raise nat.NativeError(f"HIP kernels are built for latent_dim = {engine.D}")
        self.laynorm = nn.LayerNorm(d)   # declared (and checkpointed) by the reference, unused in its forward
        self.phi_to = Phi_to([2 * d + 3, d, d], nn.ReLU())
        self.phi_from = Phi_from([2 * d + 3, d, d], nn.ReLU())
        self.z_k = MLPActivation([3 * d + p, d], nn.Sigmoid())
        self.r_k = MLPActivation([3 * d + p, d], nn.Sigmoid())
        self.correction = MLPActivation([3 * d + p, d], nn.Tanh())
        if self.mixed:
            self.phi_neumann = Phi_from([2 * d + 3, d, d], nn.ReLU())
            self.update_neumann = MLP([2 * d + 5, d, d], nn.ReLU())
        self.autoencoder = Autoencoder([1, d, d], nn.ReLU())
        self.mse_loss = nn.MSELoss()
        self._packed, self._packed_key = None, None

    def packed(self, device):
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._packed_key != key:
            self._packed = engine.pack_dsgps(self.state_dict(), device)
            self._packed_key = key
        return self._packed

    def residual_loss(self, u, batch):
        r = engine.residual(engine.plan_for(batch), u, batch.y)
        return torch.mean(r ** 2)

    @torch.no_grad()
    def inference(self, batch):
        nat.require_cuda(batch.x, "batch.x")
        h0 = self.autoencoder.encoder(batch.x)
        hk = engine.dsgps_forward(engine.plan_for(batch), self.packed(h0.device), h0, batch.prb_data, self.config["k"],
                                  batch.unit_normal_vector if self.mixed else None)
        return self.autoencoder.decoder(hk)

    def forward(self, batch):
        if self.training and torch.is_grad_enabled():
            return self._train_forward(batch)
        with torch.no_grad():
            return self._eval_forward(batch)

    def _train_forward(self, batch):
        """dirichlet/dsgps/model.py:48-128 with gradients: every term of ``train_loss`` carries the gradient it has in the
        reference -- the residual of each decoded iterate through the decoder and the unrolled updates (BPTT), the
        encoder term with the decoder's parameters frozen, the autoencoder term with the encoder's frozen (:100-112)."""
        nat.require_cuda(batch.x, "batch.x")
        ae, k, gamma = self.autoencoder, self.config["k"], self.config["gamma"]
        plan = engine.plan_for(batch)
        residual = lambda u: torch.mean(engine.residual_autograd(plan, u, batch.y, batch.a_ij) ** 2)
        idx = torch.where(batch.tags.reshape(batch.tags.shape[0], -1)[:, 1 if self.mixed else 0] == 1)[0]
        U = {"0": batch.x}
        res, mse = {"0": residual(batch.x)}, {"0": self.mse_loss(batch.x, batch.sol)}
        enc, aenc, msd = {}, {}, {}
        H0 = ae.encoder(batch.x)
        sc = _StepCtx(self, batch, H0)
        named = [(n, p) for n, p in self.named_parameters()
                 if n.startswith(("phi_to.", "phi_from.", "z_k.", "r_k.", "correction.", "phi_neumann.", "update_neumann."))]
        names, params = tuple(n for n, _ in named), [p for _, p in named]

        def frozen(module, fn):
            ps = list(module.parameters())
            for p in ps:
                p.requires_grad = False
            try:
                return fn()
            finally:
                for p in ps:
                    p.requires_grad = True
        h, total = H0, None
        for t in range(k):
            h = _DsgpsStepFn.apply(h, H0, sc, names, *params)
            u = ae.decoder(h)
            s = str(t + 1)
            U[s] = u
            res[s], mse[s] = residual(u), self.mse_loss(u, batch.sol)
            if self.mixed:   # mixed/dsgps/model.py:100-107: detached iterates instead of frozen halves
                u_d, h_d = u.detach(), h.detach()
                enc[s] = self.mse_loss(ae.encoder(u_d), h_d)
                aenc[s] = self.mse_loss(ae.decoder(ae.encoder(u_d).detach()), u_d)
            else:
                enc[s] = frozen(ae.decoder, lambda: self.mse_loss(ae(h, sens="latent"), h))
                aenc[s] = frozen(ae.encoder, lambda: self.mse_loss(ae(u, sens="physics"), u))
            msd[s] = self.mse_loss(u[idx, :], batch.sol[idx, :])
            term = res[s] * gamma ** (k - t - 1) + enc[s] + aenc[s]
            total = term if total is None else total + term
        return U, {"train_loss": total, "residual_loss": res, "encoder_loss": enc, "autoencoder_loss": aenc,
                   "mse_dirichlet": msd, "mse_loss": mse}

    def _eval_forward(self, batch):
        nat.require_cuda(batch.x, "batch.x")
        ae, k, gamma = self.autoencoder, self.config["k"], self.config["gamma"]
        plan = engine.plan_for(batch)
        w = self.packed(batch.x.device)
        idx = torch.where(batch.tags.reshape(batch.tags.shape[0], -1)[:, 1 if self.mixed else 0] == 1)[0]
        U = {"0": batch.x}
        res, mse = {"0": self.residual_loss(batch.x, batch)}, {"0": self.mse_loss(batch.x, batch.sol)}
        enc, aenc = {}, {}
        msd = {"0": self.mse_loss(batch.x[idx, :], batch.sol[idx, :])}   # present in tests/model_dsgps.py:71, harmless otherwise
        h0 = ae.encoder(batch.x)
        h0p, prbp = plan.permute(h0, True), plan.permute(batch.prb_data, True)
        nrmp = plan.permute(batch.unit_normal_vector, True) if self.mixed else None
        hp = h0p
        total = None
        for t in range(k):
            hp = engine.dsgps_step_p(plan, w, hp, h0p, prbp, nrmp)
            h = plan.permute(hp, False)
            u = ae.decoder(h)
            s = str(t + 1)
            U[s] = u
            res[s], mse[s] = self.residual_loss(u, batch), self.mse_loss(u, batch.sol)
            enc[s] = self.mse_loss(ae(h, sens="latent"), h)
            aenc[s] = self.mse_loss(ae(u, sens="physics"), u)
            msd[s] = self.mse_loss(u[idx, :], batch.sol[idx, :])
            term = res[s] * gamma ** (k - t - 1) + enc[s] + aenc[s]
            total = term if total is None else total + term
        return U, {"train_loss": total, "residual_loss": res, "encoder_loss": enc, "autoencoder_loss": aenc,
                   "mse_dirichlet": msd, "mse_loss": mse}

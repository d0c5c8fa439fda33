"""Training loop with the reference's surface (dirichlet/psignn/training_class.py): ``TrainModel(config)`` with the
same config keys, two Adam optimisers (DEQ block / autoencoder) with ReduceLROnPlateau, gradient clipping, the same
loss combination, history dictionaries and checkpoint dictionary keys.

What is different by design (MI355X-first):
  * the model step is the HIP path (device Broyden forward, on-device adjoint solve, parameter-VJP kernels); there is
    no ``torch_geometric.nn.DataParallel`` wrapper -- ``self.model`` is the module itself;
  * data parallelism is one process per GPU: each rank steps through its own shard of the loader and the gradients
    are averaged with ONE all-reduce of a flat buffer per step (1 444 floats; RCCL on GPUs, gloo in the CPU tests) --
    the only collective of the training path;
  * plotting (matplotlib figures of the reference) is not part of the path and is left out.
"""
from __future__ import annotations

import os
import time
from math import ceil

import torch
import torch.distributed as dist

from . import _native as nat

_KEYS = ("loss", "residual_loss", "jacobian_loss", "encoder_loss", "autoencoder_loss", "mse_loss")


def allreduce_mean_grads(params, group=None):
    """Average ``p.grad`` over the ranks with a single all-reduce of one flat buffer.  No-op without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    ps = [p for p in params if p.grad is not None]
    if not ps:
        return
    flat = torch.cat([p.grad.reshape(-1) for p in ps])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= dist.get_world_size(group)
    o = 0
    for p in ps:
        n = p.grad.numel()
        p.grad.copy_(flat[o:o + n].reshape(p.grad.shape))
        o += n


class TrainModel:
    KEYS = _KEYS
    _LINE = "\t Res : {:.4e} \t Jac : {:.4e} \t Enc : {:.4e} \t AEnc : {:.4e} \t MSE : {:.4e}"
    _EPOCH = "\t Res : {:.5e} \t Jac : {:.5e} \t Enc : {:.5e} \t AE : {:.5e} \t MSE : {:.5e}"

    @staticmethod
    def _value(loss_dic, key):
        return loss_dic[key].mean().item()

    def __init__(self, config):
        self.loader_train = config["loader_train"]
        self.loader_val = config["loader_val"]
        self.model = config["model"]
        self.config_model = config["config_model"]
        self.lr_deq, self.lr_ae = config["lr_deq"], config["lr_ae"]
        self.sched_step_deq, self.sched_step_ae = config["sched_step_deq"], config["sched_step_ae"]
        self.path_ckpt = config["path_ckpt"]
        self.path_logs = self.config_model.get("path_logs")
        self.min_loss_save = config["min_loss_save"]
        self.max_epochs = config["max_epochs"]
        self.gradient_clip = config["gradient_clip"]
        self.sup_weight = config.get("sup_weight", 0.0)
        self.jac_weight = config.get("jac_weight", 0.0)
        self.training_time = 0
        self.hist_train = {k: [] for k in self.KEYS}
        self.hist_val = {k: [] for k in self.KEYS}
        self.createOptimizerAndScheduler()

    @property
    def net(self):
        """The model itself, whether or not it came wrapped in ``loader.DataParallel`` (the reference always wraps)."""
        return getattr(self.model, "module", self.model)

    def createOptimizerAndScheduler(self):
        self.opt_deq = torch.optim.Adam(self.net.deqdss.parameters(), lr=self.lr_deq)
        self.sched_deq = torch.optim.lr_scheduler.ReduceLROnPlateau(self.opt_deq, mode="min", factor=self.sched_step_deq)
        self.opt_ae = torch.optim.Adam(self.net.autoencoder.parameters(), lr=self.lr_ae)
        self.sched_ae = torch.optim.lr_scheduler.ReduceLROnPlateau(self.opt_ae, mode="min", factor=self.sched_step_ae)

    # ---- checkpoints: same dictionary keys as the reference ------------------------------------------------------
    def checkpoint(self, epoch):
        return {"epoch": epoch, "hyperparameters": self.config_model, "state_dict": self.net.state_dict(),
                "hist_train": self.hist_train, "hist_val": self.hist_val, "opt_deq": self.opt_deq.state_dict(),
                "opt_ae": self.opt_ae.state_dict(), "sched_deq": self.sched_deq.state_dict(),
                "sched_ae": self.sched_ae.state_dict(), "training_time": self.training_time,
                "min_loss_save": self.min_loss_save}

    def save_model(self, state, dirName=None, model_name=None):
        with open(os.path.join(dirName, "{}.pt".format(model_name)), mode="wb") as f:
            torch.save(state, f)

    def load_model(self, path):
        """Resume from a checkpoint of this class or of the reference's (same keys).  Loaded with
        ``weights_only=True``: tensors, plain containers and the solver function by name; nothing from the file runs."""
        from .utilities import solver
        fns = [getattr(solver, n) for n in ("broyden", "anderson", "forward_iteration", "newton")]
        torch.serialization.add_safe_globals(fns + [(f, f"utilities.solver.{f.__name__}") for f in fns])
        ck = torch.load(path, map_location="cpu", weights_only=True)
        self.net.load_state_dict(ck["state_dict"])
        self.opt_deq.load_state_dict(ck["opt_deq"])
        self.opt_ae.load_state_dict(ck["opt_ae"])
        self.sched_deq.load_state_dict(ck["sched_deq"])
        self.sched_ae.load_state_dict(ck["sched_ae"])
        self.min_loss_save = ck.get("min_loss_save", self.min_loss_save)
        self.hist_train, self.hist_val = ck["hist_train"], ck["hist_val"]
        self.training_time = ck["training_time"]

    # ---- one optimisation step (training_class.py:146-166) ---------------------------------------------------------
    def total_loss(self, loss_dic):
        return (loss_dic["residual_loss"].mean() + self.jac_weight * loss_dic["jacobian_loss"].mean()
                + loss_dic["encoder_loss"].mean() + loss_dic["autoencoder_loss"].mean())

    def train_step(self, batch):
        self.opt_ae.zero_grad()
        self.opt_deq.zero_grad()
        _, loss_dic = self.model(batch)
        loss = self.total_loss(loss_dic)
        loss.backward()
        allreduce_mean_grads(self.net.parameters())
        torch.nn.utils.clip_grad_norm_(self.net.parameters(), self.gradient_clip)
        self.opt_deq.step()
        self.opt_ae.step()
        return loss, loss_dic

    def _write(self, text):
        if self.path_logs and self._rank() == 0:
            with open(os.path.join(self.path_logs, "train_metrics.csv"), "a") as f:
                f.write(text)

    @staticmethod
    def _rank():
        return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0

    @staticmethod
    def _mean_over_ranks(vals):
        """Epoch statistics: mean over the ranks' shards (one small all-reduce per epoch)."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return vals
        t = torch.tensor(vals, dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t)
        return (t / dist.get_world_size()).tolist()

    @staticmethod
    def _common_batches(n_local, what):
        """Batches every rank steps through this epoch.  Each train step ends in a gradient all-reduce, so all ranks must
        run the SAME number of steps; round-robin shards differ by up to one batch (5 meshes over 2 ranks: 3 and 2).  The
        epoch is cut to the smallest shard (drop_last semantics, one MIN/MAX all-reduce per epoch); a rank without any
        batch is an error, not a hang."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return n_local
        t = torch.tensor([float(n_local), -float(n_local)], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        n_min, n_max = int(t[0].item()), int(-t[1].item())
        if n_min <= 0:
            raise RuntimeError(f"{what}: a rank holds no batch (local {n_local}, max over ranks {n_max}); "
                               "give every rank at least one batch")
        return n_min

    def train_loop(self, current_epoch):
        self.model.train()
        n = self._common_batches(len(self.loader_train), "train_loop")
        KEYS = self.KEYS
        cumul = dict.fromkeys(KEYS, 0.0)
        run = dict.fromkeys(KEYS, 0.0)
        cnt = 0
        marks = {ceil(0.25 * n), ceil(0.5 * n), ceil(0.75 * n)}
        for i, batch in enumerate(self.loader_train):
            if i >= n:
                break   # uneven shards: the extra batch of the longer shards is dropped this epoch
            loss, loss_dic = self.train_step(batch)
            vals = {"loss": loss.item(), **{k: self._value(loss_dic, k) for k in KEYS[1:]}}
            for k in KEYS:
                cumul[k] += vals[k]
                run[k] += vals[k]
            cnt += 1
            if i in marks:
                self._write(("\nEpoch {}, {:d}% \t Loss : {:.4e} " + self._LINE).format(
                    current_epoch, int(i * 100 / n), *[run[k] / cnt for k in KEYS]))
                run = dict.fromkeys(KEYS, 0.0)
                cnt = 0
        means = self._mean_over_ranks([cumul[k] / max(n, 1) for k in KEYS])
        for k, v in zip(KEYS, means):
            self.hist_train[k].append(v)
        self._write(("\nTraining Epoch {} : \t Train : {:.5e} " + self._EPOCH).format(current_epoch, *means))

    def validation_loop(self, current_epoch):
        self.model.eval()
        n = len(self.loader_val)
        KEYS = self.KEYS
        cumul = dict.fromkeys(KEYS, 0.0)
        with torch.no_grad():
            for batch in self.loader_val:
                _, loss_dic = self.model(batch)
                cumul["loss"] += self.total_loss(loss_dic).item()
                for k in KEYS[1:]:
                    cumul[k] += self._value(loss_dic, k)
        means = self._mean_over_ranks([cumul[k] / max(n, 1) for k in KEYS])
        for k, v in zip(KEYS, means):
            self.hist_val[k].append(v)
        self._write(("\nValidation Epoch {} : \t Train : {:.5e} " + self._EPOCH).format(current_epoch, *means))

    def train_model(self):
        for epoch in range(self.max_epochs):
            t0 = time.time()
            self.train_loop(epoch)
            self.validation_loop(epoch)
            self.sched_deq.step(self.hist_val["loss"][-1])
            self.sched_ae.step(self.hist_val["loss"][-1])
            self.training_time += time.time() - t0
            if self.opt_deq.param_groups[0]["lr"] <= 1.e-7 and self.opt_ae.param_groups[0]["lr"] <= 1.e-7:
                self._write("\nTraining exit because both learning rates too low !")
                break
            saved = False
            if self._rank() == 0 and self.path_ckpt:
                ck = self.checkpoint(epoch)
                self.save_model(ck, dirName=self.path_ckpt, model_name="running_model")
                if self.hist_val["residual_loss"][-1] <= self.min_loss_save:
                    self.min_loss_save = self.hist_val["residual_loss"][-1]
                    ck["min_loss_save"] = self.min_loss_save
                    self.save_model(ck, dirName=self.path_ckpt, model_name="best_model")
                    saved = True
            self._write("\nTraining Epoch {} finished, took current epoch {:.2f}s, cumulative time {:.2f}s".format(
                epoch, time.time() - t0, self.training_time))
            self._write("\nCurrent Learning rate DEQ : {}".format(self.opt_deq.param_groups[0]["lr"]))
            self._write("\nCurrent Learning rate AUTOENC : {}".format(self.opt_ae.param_groups[0]["lr"]))
            if saved:
                self._write("\nMODEL SAVED")
        # the reference saves the last checkpoint once more as final_model and returns the model (training_class.py:332-335)
        if self._rank() == 0 and self.path_ckpt and self.max_epochs > 0 and len(self.hist_train["loss"]) > 0:
            self.save_model(self.checkpoint(len(self.hist_train["loss"]) - 1), dirName=self.path_ckpt, model_name="final_model")
        return self.model


class TrainModelDSGPS(TrainModel):
    """``dirichlet/dsgps/training_class.py``: one Adam over all parameters (``config["lr"]``), the loss is the model's
    ``train_loss`` (residual of every unrolled update weighted by gamma^(k-t-1), plus encoder / autoencoder terms,
    dsgps/model.py:116-121), statistics are those of the last update ``str(k)``; no schedulers; the optimiser state is
    checkpointed under ``opt_deq`` like the reference does."""
    KEYS = ("loss", "residual_loss", "encoder_loss", "autoencoder_loss", "mse_loss")
    _LINE = "\t Res : {:.4e} \t Enc : {:.4e} \t AEnc : {:.4e} \t MSE : {:.4e}"
    _EPOCH = "\t Res : {:.5e} \t Enc : {:.5e} \t AE : {:.5e} \t MSE : {:.5e}"

    def __init__(self, config):
        cfg = dict(config)
        self.lr = cfg["lr"]
        for k, v in (("lr_deq", cfg["lr"]), ("lr_ae", cfg["lr"]), ("sched_step_deq", 1.0), ("sched_step_ae", 1.0)):
            cfg.setdefault(k, v)
        super().__init__(cfg)

    def _value(self, loss_dic, key):
        return loss_dic[key][str(self.config_model["k"])].mean().item()

    def createOptimizerAndScheduler(self):
        self.opt = torch.optim.Adam(self.net.parameters(), lr=self.lr)

    def total_loss(self, loss_dic):
        return loss_dic["train_loss"].mean()

    def train_step(self, batch):
        self.opt.zero_grad()
        _, loss_dic = self.model(batch)
        loss = self.total_loss(loss_dic)
        loss.backward()
        allreduce_mean_grads(self.net.parameters())
        torch.nn.utils.clip_grad_norm_(self.net.parameters(), self.gradient_clip)
        self.opt.step()
        return loss, loss_dic

    def checkpoint(self, epoch):
        return {"epoch": epoch, "hyperparameters": self.config_model, "state_dict": self.net.state_dict(),
                "hist_train": self.hist_train, "hist_val": self.hist_val, "opt_deq": self.opt.state_dict(),
                "training_time": self.training_time}

    def load_model(self, path):
        ck = torch.load(path, map_location="cpu", weights_only=True)
        self.net.load_state_dict(ck["state_dict"])
        self.opt.load_state_dict(ck["opt_deq"])
        self.hist_train, self.hist_val = ck["hist_train"], ck["hist_val"]
        self.training_time = ck["training_time"]

    def train_model(self):
        for epoch in range(self.max_epochs):
            t0 = time.time()
            self.train_loop(epoch)
            self.validation_loop(epoch)
            self.training_time += time.time() - t0
            saved = False
            if self._rank() == 0 and self.path_ckpt:
                ck = self.checkpoint(epoch)
                self.save_model(ck, dirName=self.path_ckpt, model_name="running_model")
                if self.hist_val["residual_loss"][-1] <= self.min_loss_save:
                    self.save_model(ck, dirName=self.path_ckpt, model_name="best_model")
                    self.min_loss_save = self.hist_val["residual_loss"][-1]
                    saved = True
            self._write("\nTraining Epoch {} finished, took current epoch {:.2f}s, cumulative time {:.2f}s".format(
                epoch, time.time() - t0, self.training_time))
            self._write("\nCurrent Learning rate : {}".format(self.opt.param_groups[0]["lr"]))
            if saved:
                self._write("\nMODEL SAVED")
        if self._rank() == 0 and self.path_ckpt and len(self.hist_train["loss"]) > 0:
            self.save_model(self.checkpoint(len(self.hist_train["loss"]) - 1), dirName=self.path_ckpt, model_name="final_model")
        return self.model


class TrainModelDSS(TrainModelDSGPS):
    """``dirichlet/dss/training_class.py``: as the DS-GPS trainer with residual / mse statistics only and the optimiser
    state under ``opt``."""
    KEYS = ("loss", "residual_loss", "mse_loss")
    _LINE = "\t Res : {:.4e} \t MSE : {:.4e}"
    _EPOCH = "\t Res : {:.5e} \t MSE : {:.5e}"

    def checkpoint(self, epoch):
        ck = super().checkpoint(epoch)
        ck["opt"] = ck.pop("opt_deq")
        return ck

    def load_model(self, path):
        ck = torch.load(path, map_location="cpu", weights_only=True)
        self.net.load_state_dict(ck["state_dict"])
        self.opt.load_state_dict(ck["opt"])
        self.hist_train, self.hist_val = ck["hist_train"], ck["hist_val"]
        self.training_time = ck["training_time"]

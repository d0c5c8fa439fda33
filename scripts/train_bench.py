"""Training-step timing: one optimisation step of the dirichlet model on a union batch of synthetic
hexagon meshes -- forward Broyden solve, on-device adjoint solve, parameter-VJP, optimiser -- on the HIP path, with
the CPU oracle's restated training step (autograd + restated broyden) timed beside it on the same batch.

    python3 scripts/train_bench.py [graphs_per_batch=50] [hex_n=13] [steps=5] [cpu=1] [jac_weight=0] [family=dirichlet|mixed]

jac_weight = 1 is what the reference's launch scripts use (launch_local.sh:24): the step then also runs the backward of
the VJP (csrc/gather_backward.hip).

The reference trains on ~500-node meshes (hsize 0.08) in PyG batches; hex_n = 13 gives 547 nodes per graph.
Prints one JSON line."""
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = lambda n="": importlib.import_module("psi-gnn_amd" + ("." + n if n else ""))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 13
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    cpu = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    jw = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
    mixed = len(sys.argv) > 6 and sys.argv[6] == "mixed"
    data, nat = pkg("data"), pkg("_native")
    import numpy as np
    w = np.load(os.path.join(ROOT, "tests", "golden", "weights_mixed.npz" if mixed else "weights_dirichlet.npz"))
    sd = {k: torch.from_numpy(w[k]) for k in w.files}
    meshes = [data.make_hex_problem(n, seed=s, phase=0.37 * s, mixed=mixed) for s in range(B)]
    batch = data.collate(meshes)
    dev = torch.device("cuda:0")
    solver = pkg("utilities.solver")
    cfg = dict(latent_dim=10, n_layers=1, solver=solver.broyden, fw_tol=1e-5, fw_thres=500, bw_tol=1e-8, bw_thres=500)
    net = (pkg("mixed") if mixed else pkg("model_psignn")).ModelDEQDSS(cfg)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    TrainModel = pkg("training_class").TrainModel
    tr = TrainModel(dict(loader_train=[], loader_val=[], model=net, config_model=net.config, lr_deq=1e-6, lr_ae=1e-6,
                         sched_step_deq=0.5, sched_step_ae=0.5, path_ckpt=None, min_loss_save=1e9, max_epochs=0,
                         gradient_clip=1e-2, sup_weight=0.0, jac_weight=jw))
    bd = batch.to(dev)
    warm, _ = tr.train_step(bd)  # warm-up (plan build, allocations); lr = 1e-6 keeps the weights at the checkpoint
    warm = float(warm.detach())
    torch.cuda.synchronize()
    nat.prof_enable(True)
    t0 = time.perf_counter()
    fw, bw, losses = [], [], []
    for _ in range(steps):
        loss, _ = tr.train_step(bd)
        fw.append(net.deqdss.last_forward["nstep"])
        bw.append(net.deqdss.last_backward["nstep"])
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    kern = nat.prof_collect()
    nat.prof_enable(False)
    out = {"workload": f"training step ({'mixed' if mixed else 'dirichlet'} family), union batch of {B} hexagon meshes (n={n}): {batch.num_nodes} nodes, "
                       f"{batch.num_edges} edges; fw_tol 1e-5 / bw_tol 1e-8, thresholds 500 (reference defaults), jac_weight {jw}",
           "gpu_s_per_step": dt, "gpu_graphs_per_s": B / dt, "first_step_loss": warm, "fw_nstep": fw, "bw_nstep": bw, "loss": losses,
           "kernels_ms_per_step": {k: round(v[1] / steps, 3) for k, v in sorted(kern.items(), key=lambda kv: -kv[1][1])[:12]}}
    if cpu:
        from oracle import psignn_oracle as orc
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        t0 = time.perf_counter()
        probe = torch.randn(batch.num_nodes, 10, generator=torch.Generator().manual_seed(0)) if jw else None
        wl, _, _, ofw, obw = orc.training_step(sd, batch, fw_tol=1e-5, fw_thres=500, bw_tol=1e-8, bw_thres=500,
                                               jac_weight=jw, probe=probe)
        ct = time.perf_counter() - t0
        out.update(cpu_s_per_step=ct, cpu_threads=torch.get_num_threads(), cpu_fw_nstep=ofw["nstep"],
                   cpu_bw_nstep=obw["nstep"], cpu_loss=float(wl), speedup=ct / dt,
                   cpu_kind="port: oracle training_step (autograd on the restated f + restated broyden), first step only")
    print(json.dumps(out))


if __name__ == "__main__":
    main()

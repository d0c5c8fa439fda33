"""CPU, gloo, world_size 2: the N > 1 path of the batch driver (sharding of independent meshes, the loss mean
over replicas, max-over-ranks timing as bench.py does it).  No GPU work: per-mesh results are stand-ins."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import pkg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_meshes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch = pkg("batch")
    mine = batch.shard_indices(n_meshes, rank, world)
    # stand-in per-mesh "residual": a function of the mesh index only
    local = {"residual_loss": torch.tensor(sum(float(i + 1) for i in mine) / max(len(mine), 1)),
             "mse_loss": torch.tensor(float(rank))}
    mean = batch.mean_over_replicas(local)
    total = batch.gather_counts(len(mine))
    dist.barrier()
    t = batch.max_over_ranks(0.1 * (rank + 1))          # bench.py: the job's time is the slowest rank's
    seeds = batch.rank_seeds(rank, world, 3)            # bench.py: problem seeds of this rank's meshes
    q.put((rank, mine, {k: float(v) for k, v in mean.items()}, total, t, seeds))
    dist.destroy_process_group()


def test_shards_cover_all_meshes_once():
    batch = pkg("batch")
    for n, w in ((64, 8), (5, 2), (3, 4), (0, 2)):
        shards = [batch.shard_indices(n, r, w) for r in range(w)]
        flat = sorted(i for s in shards for i in s)
        assert flat == list(range(n))
        assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
    assert [len(batch.shard_indices(64, r, 8)) for r in range(8)] == [8] * 8  # BASELINE configs[3]


def test_two_ranks_gloo():
    world, n_meshes = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_meshes, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, m0, mean0, tot0, t0, s0), (r1, m1, mean1, tot1, t1, s1) = res
    assert s0 == [0, 1, 2] and s1 == [3, 4, 5]          # distinct problems per rank; 8 ranks x 8 meshes = seeds 0 .. 63
    assert sorted(sum((pkg("batch").rank_seeds(r, 8, 8) for r in range(8)), [])) == list(range(64))
    assert m0 == [0, 2, 4] and m1 == [1, 3]
    assert tot0 == tot1 == n_meshes
    # mean over replicas of the per-replica means, as DataParallel's loss.mean() (training_class.py:156-159)
    want = 0.5 * ((1 + 3 + 5) / 3 + (2 + 4) / 2)
    assert abs(mean0["residual_loss"] - want) < 1e-6 and mean0 == mean1
    assert abs(mean0["mse_loss"] - 0.5) < 1e-6
    assert t0 == t1 == 0.2  # max over ranks


def _grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tc = pkg("training_class")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Linear(4, 1))   # same init on both ranks
    for i, p in enumerate(net.parameters()):
        p.grad = torch.full_like(p, float((rank + 1) * (i + 1)))
    tc.allreduce_mean_grads(net.parameters())
    stats = tc.TrainModel._mean_over_ranks([float(rank), 10.0 * (rank + 1)])
    q.put((rank, [float(p.grad.flatten()[0]) for p in net.parameters()], stats))
    dist.destroy_process_group()


def test_gradient_allreduce_two_ranks_gloo():
    """The training path's only collective: one flat all-reduce of the gradients per step (mean over ranks)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, g0, s0), (_, g1, s1) = res
    assert g0 == g1 == [1.5 * (i + 1) for i in range(4)]
    assert s0 == s1 == [0.5, 15.0]


class _ToyModel(torch.nn.Module):
    """CPU stand-in with the trainer's surface: .deqdss / .autoencoder parameter groups and the loss dictionary of
    ModelDEQDSS.forward (dirichlet/psignn/model.py:58-99).  The trainer's control flow is what is under test."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.deqdss = torch.nn.Linear(2, 1)
        self.autoencoder = torch.nn.Linear(2, 1)
        self.steps = 0

    def forward(self, batch):
        self.steps += 1
        u = self.deqdss(batch) + self.autoencoder(batch)
        z = torch.zeros(())
        return u, {"residual_loss": (u ** 2).mean(), "jacobian_loss": z, "encoder_loss": (self.autoencoder(batch) ** 2).mean(),
                   "autoencoder_loss": z, "mse_loss": z, "mse_dirichlet": z}


def _uneven_worker(rank, world, port, tmp, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tc, batch = pkg("training_class"), pkg("batch")
    g = torch.Generator().manual_seed(3)
    data = [torch.randn(6, 2, generator=g) for _ in range(5)]
    mine = [data[i] for i in batch.shard_indices(5, rank, world)]     # rank 0: 3 batches, rank 1: 2
    net = _ToyModel()
    tr = tc.TrainModel(dict(loader_train=mine, loader_val=mine[:1], model=net, config_model={}, lr_deq=1e-2, lr_ae=1e-2,
                            sched_step_deq=0.5, sched_step_ae=0.5, path_ckpt=tmp, min_loss_save=1e9, max_epochs=2,
                            gradient_clip=1.0, sup_weight=0.0, jac_weight=0.0))
    ret = tr.train_model()
    q.put((rank, len(mine), net.steps, [p.detach().reshape(-1).tolist() for p in net.parameters()],
           len(tr.hist_train["loss"]), ret is net))
    dist.destroy_process_group()


def test_uneven_shards_do_not_hang_two_ranks_gloo(tmp_path):
    """5 batches over 2 ranks = 3 and 2: every train step ends in a gradient all-reduce, so both ranks must take the same
    number of steps (the epoch is cut to the smallest shard); the weights stay identical across ranks; rank 0 writes
    running / best / final checkpoints and train_model returns the model (reference training_class.py:296-335)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_uneven_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, n0, steps0, w0, h0, ret0), (_, n1, steps1, w1, h1, ret1) = res
    assert (n0, n1) == (3, 2)
    # per epoch: 2 train steps (the common count) + 1 validation forward
    assert steps0 == steps1 == 2 * (2 + 1)
    assert w0 == w1
    assert h0 == h1 == 2 and ret0 and ret1
    for f in ("running_model.pt", "best_model.pt", "final_model.pt"):
        assert (tmp_path / f).exists(), f
    ck = torch.load(tmp_path / "final_model.pt", weights_only=True)
    assert ck["epoch"] == 1 and len(ck["hist_train"]["loss"]) == 2


def _bench(env_extra, *argv):
    import json
    import subprocess
    env = dict(os.environ, PSIGNN_BENCH_DRYRUN="1", **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=300)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, [json.loads(l) for l in lines], p.stderr


def test_bench_launches_its_own_ranks():
    """``python bench.py --gpus N`` without a launcher (no RANK / WORLD_SIZE in the environment) starts its N ranks itself: fresh
    child processes with the rendezvous variables set, rank 0's single JSON line relayed, exit code non-zero if any rank
    fails.  Rehearsed here without a GPU (PSIGNN_BENCH_DRYRUN=1: the ranks only rendezvous over gloo, take the MAX of a
    per-rank time and report the world size); the GPU form of the same launch is tests/test_gpu_bench_contract.py."""
    rc, lines, err = _bench({}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert rc == 0 and len(lines) == 1, err[-2000:]
    d = lines[0]
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["backend"] == "gloo" and d["steps"] == 3
    assert abs(d["max_over_ranks_s"] - 0.002) < 1e-12            # MAX over ranks of 0.001 * (rank + 1)
    rc, lines, err = _bench({}, "--gpus", "1", "--steps", "3")
    assert rc == 0 and lines[0]["n_gpus"] == 1 and lines[0]["ranks"] == 1
    # a failing rank makes the launcher fail, whatever rank 0 printed
    rc, lines, err = _bench({"PSIGNN_BENCH_DRYRUN_FAIL": "1"}, "--gpus", "2", "--steps", "3")
    assert rc != 0 and "ranks failed" in err
    # under a launcher (RANK / WORLD_SIZE set by torch.distributed.run) the process is one rank and spawns nothing
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"],
                       env=dict(os.environ, PSIGNN_BENCH_DRYRUN="1"), capture_output=True, text=True, timeout=300)
    import json
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["ranks"] == 2, p.stderr[-2000:]

"""Independent meshes across the GPUs of one node: one process per GPU, no data-path collective.

The reference's only parallelism is ``torch_geometric.nn.DataParallel`` (``dirichlet/psignn/main.py:106``):
a list of graphs is split over devices, and the per-replica scalar losses are averaged
(``training_class.py:156-159``).  Meshes are independent fixed-point problems, so inference shards them
round-robin over ranks and every rank solves its own; the only collective that exists on the reference's path —
the loss mean over replicas — is one tiny all-reduce (RCCL over xGMI on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

import torch


def shard_indices(n_items: int, rank: int, world: int):
    """Round-robin ownership: item i belongs to rank i % world (BASELINE configs[3]: 64 meshes, 8 per GPU)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    return list(range(rank, n_items, world))


def rank_seeds(rank: int, world: int, meshes_per_gpu: int):
    """Problem seeds of the meshes rank `rank` solves in the weak-scaling bench: rank r owns seeds r * MPG .. r * MPG + MPG - 1,
    so that the `world * MPG` meshes of a run are distinct problems and adding GPUs adds problems (BASELINE configs[3]:
    8 ranks x 8 meshes = seeds 0 .. 63)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    return [rank * meshes_per_gpu + j for j in range(meshes_per_gpu)]


def max_over_ranks(value: float, device=None, group=None) -> float:
    """MAX all-reduce of a per-rank wall time: the job's time is the slowest rank's (bench.py contract)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def solve_shard_batched(model, meshes, device, indices=None, group=8):
    """``solve_shard`` with the device's share of the batch solved in LOCKSTEP: ``group`` meshes at a time go through one
    batched device solve (``engine.broyden_solve_batch``: every per-iteration pass is one launch over all of them, own stop
    test per mesh).  Each mesh's result is bit-identical to its own solve with a solver of the same configuration
    (``DeviceBroyden(..., shard_elems=...)``: the sweeps' reduction shapes are sized for the shard) and agrees with the plain
    ``solve_shard`` path to solver tolerance; both families (a shard is all dirichlet or all mixed, as every dataset of the
    reference is).  Meshes the batched solver cannot take together (untiled plans, different size classes, multi-layer blocks)
    are solved one by one -- decided on the host before anything is allocated or launched; errors of the batched solve raise."""
    import importlib
    eng = importlib.import_module(__package__ + ".engine")
    slv = importlib.import_module(__package__ + ".utilities.solver")
    net = getattr(model, "module", model)
    cfg = net.deqdss.config_deq
    idx = list(range(len(meshes)) if indices is None else indices)
    out = []
    with torch.no_grad():
        for g0 in range(0, len(idx), group):
            ids = idx[g0:g0 + group]
            mds = [meshes[i].to(device) for i in ids]
            h0s = [net.autoencoder.encoder(md.x) for md in mds]
            fmaps = [net.deqdss.f.bind(h0, md) for h0, md in zip(h0s, mds)]
            solved = None
            # batchable? decided on the host BEFORE any solver state is allocated: tiled plans of one family, single-layer
            # block; then (solvers exist, nothing solved yet) one size class.  Anything the batched solve raises is a real error.
            if (cfg["solver"] is slv.broyden and len(ids) > 1 and all(f.plan.tiled for f in fmaps)
                    and len({f.plan.mixed for f in fmaps}) == 1 and fmaps[0].weights.n_layers == 1):
                total = sum(f.plan.N for f in fmaps) * eng.D
                solvers = [eng.DeviceBroyden(plan=f.plan, threshold=cfg["fw_thres"], keep_trace=False, shard_elems=total)
                           for f in fmaps]
                try:
                    if eng.shard_batchable(solvers):
                        solved = eng.broyden_solve_batch(solvers, fmaps, cfg["fw_tol"])
                finally:
                    for sv in solvers:
                        sv.close()
            for k, i in enumerate(ids):
                if solved is not None:
                    h_final, nstep = solved[k]["result"], solved[k]["nstep"]
                else:
                    o = net.deqdss(h0s[k], mds[k])
                    h_final, nstep = o["result"], o["nstep"]
                u = net.autoencoder.decoder(h_final)
                loss = net._diagnostics(u, h_final, mds[k], "mse_dirichlet_loss")
                loss["nsteps"] = nstep
                out.append((i, u, loss))
    return out


def solve_shard(model, meshes, device, indices=None, streams=1):
    """Solve the given meshes on `device`; returns [(index, u_final, loss_dic)] in index order.

    ``streams`` > 1 runs that many solves concurrently, each on its own HIP stream and host thread (the library calls
    release the GIL): a 50 k-node solve leaves most of an MI355X idle, BASELINE configs[3] keeps 8 per GPU in flight.
    Every mesh is still its own fixed-point problem; results do not depend on ``streams``."""
    idx = list(range(len(meshes)) if indices is None else indices)
    if streams <= 1 or len(idx) <= 1:
        out = []
        for i in idx:
            u, loss = model(meshes[i].to(device))
            out.append((i, u, loss))
        return out
    import concurrent.futures as cf
    pool = [torch.cuda.Stream(device) for _ in range(min(streams, len(idx)))]
    ready = torch.cuda.Event()
    ready.record(torch.cuda.current_stream(device))

    def work(slot):
        res = []
        with torch.cuda.device(device), torch.cuda.stream(pool[slot]):
            pool[slot].wait_event(ready)
            for i in idx[slot::len(pool)]:
                u, loss = model(meshes[i].to(device))
                res.append((i, u, loss))
            pool[slot].synchronize()
        return res

    with cf.ThreadPoolExecutor(len(pool)) as ex:
        parts = list(ex.map(work, range(len(pool))))
    return sorted((r for p in parts for r in p), key=lambda r: r[0])


def mean_over_replicas(values: dict, group=None) -> dict:
    """All-reduce(SUM) / world of a dict of scalar tensors: ``loss.mean()`` over DataParallel replicas
    (``training_class.py:156-159``).  One flat tensor, one collective."""
    import torch.distributed as dist
    keys = sorted(values)
    flat = torch.stack([torch.as_tensor(values[k], dtype=torch.float32).reshape(()) for k in keys])
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat = flat / dist.get_world_size(group)
    return {k: flat[i] for i, k in enumerate(keys)}


def gather_counts(local_count: int, device=None, group=None):
    """Sum of a per-rank integer (meshes / iterations / edges processed) over ranks."""
    import torch.distributed as dist
    t = torch.tensor([float(local_count)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())

#!/usr/bin/env python3
"""Headline benchmark: PSI-GNN fixed-point iterations/s and edges/s on a 1M-node Dirichlet Poisson mesh.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[4], the configuration the metric is quoted on): one synthetic
hexagon-lattice Dirichlet Poisson mesh of 1 000 519 nodes / E' = 5.98 M directed non-self edges per
GPU (SURVEY §8d recipe), trained dirichlet checkpoint weights, fp32.
A *step* is one fixed-point iteration of the reference's solver — Broyden, utilities/solver.py:116-207:
one evaluation of the GNN block f plus the residual norms, stop tests and the rank-1 inverse-Jacobian
update — executed entirely on the device through libpsignn_hip.so.  The timed region is iterations
1..K of a solve started from the encoder's initial state (eps = 0, so exactly K iterations run; the
k-th iteration sweeps k stored (u, v) pairs, which is part of the solver's real cost).  Inputs are
resident in HBM when the clock starts.  edges/s = E' x iterations/s (each edge counted once although
Phi_to and Phi_from both process it).  N > 1: every rank solves its own mesh (same topology, its own
problem seed) with no data-path collective -> weak scaling; value = sum over ranks.

The timed region (exactly K iterations, barrier + synchronize on both sides, MAX over ranks) is repeated --repeats times
(default 5); `value` / `ms_per_step` are the MEDIAN repeat, the spread is reported beside them.  `value` depends on K --
iteration k sweeps k stored pairs, as in the reference -- so the k-independent figures of the GNN block are in the same
line: `f_only` (plain f, back-to-back evaluations) and `roofline_f`.

One JSON line on stdout (rank 0).  Extra objects:
  roofline      dominant kernel (by time) : algorithmic bytes per launch / average launch duration
                (HIP events on the launch stream, instrumented repeat of the same K steps); `traffic` = HBM bytes per launch
                from the committed rocprofv3 FETCH_SIZE / WRITE_SIZE passes (profiles/pmc_traffic_model.json: per kernel a
                linear model a + b k in the number of stored pairs, evaluated for this run's K)
  roofline_f    the same for the GNN block f inside the loop (k_f_tile_fused: x + update, f, g, dg, norms in one kernel) --
                the kernel north_star's 60 % target is about
  f_only        plain f (k_f_tile), 100 back-to-back evaluations: us per evaluation, edges/s, fraction of the HBM peak
  roofline_jvp  the analytic JVP kernel of the Newton-Krylov path (k_jvp_tile), 50 back-to-back products
  newton_krylov one inner solve of that path: 30 Arnoldi steps of the device GMRES around the JVP (time, bytes, fraction of the peak)
  roofline_iter whole-iteration algorithmic bytes / un-instrumented wall time
  cpu_baseline  the CPU oracle (port of the reference path) timed on this box's host cores on a bounded sample
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
HBM_COPY_GBS = 6290.0  # the guide's measured float4-copy rate: the practical ceiling of a streaming kernel
D = 10


WORKLOADS = {  # preset -> (nodes per mesh, boundary conditions, meshes per GPU, BASELINE.json config)
    "mesh1m": (1_000_000, "dirichlet", 1, "configs[4] size"),
    "dir100k": (100_000, "dirichlet", 1, "configs[1]"),
    "mixed100k": (100_000, "mixed", 1, "configs[2]"),
    "batch50k": (50_000, "dirichlet", 8, "configs[3]"),
    "f1m": (1_000_000, "dirichlet", 1, "configs[4] size, GNN block only"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="mesh1m",
                    help="mesh1m = BASELINE configs[4] size (headline); dir100k = configs[1]; mixed100k = configs[2]; "
                         "batch50k = configs[3] (8 independent 50k-node meshes per GPU, concurrent streams)")
    ap.add_argument("--nodes", type=int, default=None, help="override the preset's per-mesh node count")
    ap.add_argument("--bc", choices=["dirichlet", "mixed"], default=None)
    ap.add_argument("--meshes-per-gpu", type=int, default=None)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of exactly --steps iterations each; the median is reported")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-kernel-timing", action="store_true")
    return ap.parse_args()


def load_weights(kind):
    w = np.load(os.path.join(ROOT, "tests", "golden", f"weights_{kind}.npz"))
    return {k: torch.from_numpy(w[k]) for k in w.files}


def algorithmic_bytes(N, Ep, K, mixed):
    """Compulsory HBM bytes: each operand read once, each result written once, int32 indices."""
    M = N * D
    b_f = (102 if mixed else 89) * N + 20 * Ep                 # one f evaluation
    per_launch = {
        # f is two launches; the pair is priced with the fused-op figure B_f
        "f(k_project+k_node)": b_f,
        "k_xnext": 3 * M * 4, "k_resid": 5 * M * 4, "k_final": 4 * M * 4,
        "jvp": b_f + 40 * N,                                        # + the tangent v (B_jvp)
    }
    # k-dependent sweeps: iteration it (0-based) has k = it stored pairs
    dots = [2 * k * M * 4 + 3 * M * 4 for k in range(1, K)]     # launched only when k > 0
    axpy = [2 * k * M * 4 + 6 * M * 4 for k in range(0, K - 1)]  # (iteration K's stop test fires first: its axpy / final return at once)
    per_launch["f_fused"] = b_f + 16 * M                        # + upd, g_old reads; g, dg writes (x_next replaces f(x))
    # whole iteration, BASELINE.md: B_broyden(k) = 16 k M + 48 M + B_f  (U and V swept twice, ~12 state-vector passes)
    total_iter = sum(16 * k * M + 48 * M + b_f for k in range(K)) + b_f - (8 * (K - 1) * M + 40 * M)   # (minus the last iteration's axpy + final)
    # three-sweep form of the update (csrc/solver.hip launch_update, long vectors): U for a; V ONCE for c, b and V a; U for U c, U b.
    # While there are few stored pairs (k <= KMAX) sweep 3 is k_sweep_u2d, which also delivers the next iteration's a: that
    # iteration then launches no k_sweep_u1.
    KMAX = int(os.environ.get("PSIGNN_U2D_KMAX", "24"))

    KEEP = int(os.environ.get("PSIGNN_U2D_KEEP", "16"))

    def sweeps3_for(kmax):
        # sweep 3 of iteration k (k stored pairs): the folded kernel keeps the pairs from keep0(k) on (all of them while k <= kmax,
        # the most recent KEEP afterwards) and delivers their a_j for iteration k + 1, whose sweep 1 then covers keep0(k) pairs only
        folded = [kmax >= -1 and k + 1 < K and (k <= kmax or KEEP > 0) and kmax > 0 for k in range(K)]
        keep0 = [(0 if k <= kmax else k - KEEP) if folded[k] else k + 1 for k in range(K)]
        u1_cols = {k: (keep0[k - 1] if folded[k - 1] else k) for k in range(1, K)}
        u1_ks = [c for k, c in u1_cols.items() if c > 0]                       # regressor of sweep 1: the pairs it reads
        u2d_ks = [k for k in range(K) if folded[k]]
        u2_ks = [k for k in range(K - 1) if not folded[k]]                     # (iteration K's stop test fires in front of its sweeps 2 and 3)
        byts = {"k_sweep_u1": [(c + 1) * M * 4 for c in u1_ks],                # its columns of U + dx
                "k_sweep_v": [(k + 4) * M * 4 for k in range(0, K - 1)],       # k columns of V + dx, dg, g; writes V[k]
                "k_sweep_u2": [(k + 5) * M * 4 for k in u2_ks],                # k columns of U + dx, dg, g; writes U[k], update
                "k_sweep_u2d": [(k + 5) * M * 4 for k in u2d_ks]}
        ks = {"k_sweep_u1": u1_ks, "k_sweep_v": list(range(0, K - 1)), "k_sweep_u2": u2_ks, "k_sweep_u2d": u2d_ks}
        # sweeps + fused f's extra 16 M (no k_final: merged into sweep 3)
        return byts, ks, sum(sum(v) for v in byts.values()) + K * (16 * M + b_f) + b_f
    sweeps3, ks3, total_iter3 = sweeps3_for(-2)          # plain three-sweep form (mid-size vectors, batched shards)
    sweeps3d, ks3d, total_iter3d = sweeps3_for(KMAX)     # long vectors: folded sweep 3 while k <= KMAX
    per_launch["_sweeps3d"], per_launch["_sweeps3d_ks"], per_launch["_total_iter3d"] = sweeps3d, ks3d, total_iter3d
    per_launch["_sweeps3_ks"] = ks3
    per_launch["_sweeps3"] = sweeps3
    per_launch["_total_iter3"] = total_iter3
    return per_launch, dots, axpy, total_iter


def _time_picard(fmap, xp, n, dev):
    """ms per evaluation of n back-to-back x <- f(x) in plan order (one library call, no host work in between)."""
    fmap.picard_p(xp, 3)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    fmap.picard_p(xp, n)
    torch.cuda.synchronize(dev)
    return 1e3 * (time.perf_counter() - t0) / n


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    # one process per GPU; PSIGNN_BENCH_BACKEND=gloo lets several ranks share the cards of a small box for a
    # rehearsal of the N > 1 path (RCCL refuses two ranks on one device)
    backend = os.environ.get("PSIGNN_BENCH_BACKEND", "nccl")
    local = local % max(torch.cuda.device_count(), 1) if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            if backend == "nccl":
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()
        torch.cuda.synchronize(dev)

    pkg = importlib.import_module("psi-gnn_amd")
    eng = importlib.import_module("psi-gnn_amd.engine")
    nat = importlib.import_module("psi-gnn_amd._native")

    nodes_p, bc_p, mpg_p, cfg_name = WORKLOADS[args.workload]
    nodes = args.nodes or nodes_p
    args.bc = args.bc or bc_p
    MPG = args.meshes_per_gpu or mpg_p
    mixed = args.bc == "mixed"
    model_mod = importlib.import_module("psi-gnn_amd.mixed" if mixed else "psi-gnn_amd.model_psignn")
    n = pkg.data.hex_n_for_nodes(nodes)
    if 3 * (n - 1) ** 2 + 3 * (n - 1) + 1 >= 0.999 * nodes:
        n -= 1  # 99 919 counts as "100k"
    t0 = time.time()
    sd = load_weights(args.bc)
    net = model_mod.ModelPSIGNN(dict(latent_dim=10, n_layers=1)).eval()
    net.load_state_dict(sd)
    net = net.to(dev)
    meshes, fmaps = [], []
    batch_mod = importlib.import_module("psi-gnn_amd.batch")
    for j, seed in enumerate(batch_mod.rank_seeds(rank, world, MPG)):  # independent problems: own seed, same topology (phase-shifted warp)
        m = pkg.data.make_hex_problem(n, seed=seed, mixed=mixed, compute_sol=False, phase=0.37 * j)
        md = m.to(dev)
        with torch.no_grad():
            h0 = net.autoencoder.encoder(md.x)
            fmaps.append(net.deqdss.f.bind(h0, md))
        meshes.append(m)
    mesh, fmap = meshes[0], fmaps[0]
    N, E = mesh.num_nodes, mesh.num_edges
    Ep = fmap.plan.Ep
    Ep_rank = sum(f.plan.Ep for f in fmaps)
    t_setup = time.time() - t0
    K, W = args.steps, args.warmup

    if args.workload == "f1m":
        # f only: a step is ONE evaluation of the GNN block (Picard iteration x <- f(x), no solver bookkeeping)
        xp = fmap.to_plan(fmap.h0)
        fmap.picard_p(xp, max(W, 1))
        barrier()
        t0 = time.perf_counter()
        xp = fmap.picard_p(xp, K)
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
        barrier()
        assert bool(torch.isfinite(xp).all())
        b_f = (102 if mixed else 89) * N + 20 * Ep
        gbs = K * b_f / elapsed / 1e9
        print(json.dumps({
            "metric": "GNN-block edges/sec (E' x f evaluations/sec) on 1M-node Poisson mesh", "value": world * Ep * K / elapsed,
            "unit": "edges/s", "iters_per_sec": world * K / elapsed, "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"f1m: {K} back-to-back evaluations of the {args.bc} GNN block on a {N}-node mesh (plan order)",
                       "nodes": N, "edges_nonself": Ep},
            "roofline": {"kernel": "k_f_tile", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs / HBM_PEAK_GBS, "traffic": None, "alg_bytes_per_launch": b_f,
                         "avg_launch_us": 1e6 * elapsed / K}}))
        return

    # ---- warmup: W untimed iterations (own solver object so its memory is released)
    if W > 0:
        ws = eng.DeviceBroyden(plan=fmap.plan, threshold=W, keep_trace=False)
        ws.solve(fmap, eps=0.0)
        ws.close()
        del ws
    shard = sum(f.plan.N for f in fmaps) * D if MPG > 1 and os.environ.get("PSIGNN_BENCH_STREAMS", "0") != "1" else 0
    solvers = [eng.DeviceBroyden(plan=f.plan, threshold=K, keep_trace=False, shard_elems=shard) for f in fmaps]
    solver = solvers[0]
    streams = [torch.cuda.Stream(dev) for _ in range(MPG)] if MPG > 1 else [None]

    use_streams = os.environ.get("PSIGNN_BENCH_STREAMS", "0") == "1"   # round 1's form: one HIP stream + host thread per mesh

    def run_all():
        """Exactly K Broyden iterations on every mesh of this rank.  Several meshes per GPU (configs[3]) go through the
        batched device solver: one launch per pass over all of them (PSIGNN_BENCH_STREAMS=1: concurrent streams instead)."""
        if MPG == 1:
            return [solver.solve(fmap, eps=0.0, poll_every=max(K, 1))]
        if not use_streams:
            return eng.broyden_solve_batch(solvers, fmaps, 0.0, poll_every=max(K, 1))
        import concurrent.futures as cf

        def one(j):
            torch.cuda.set_device(local)
            with torch.cuda.stream(streams[j]):
                return solvers[j].solve(fmaps[j], eps=0.0, poll_every=max(K, 1))
        with cf.ThreadPoolExecutor(MPG) as ex:  # ctypes releases the GIL inside the library calls
            return list(ex.map(one, range(MPG)))

    # ---- timed regions: each exactly K iterations (per mesh), barrier + synchronize on both sides, MAX over ranks
    R = max(1, args.repeats)
    times = []
    for _ in range(R):
        barrier()
        t0 = time.perf_counter()
        outs = run_all()
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
        barrier()
        for o in outs:
            assert o["n_iter"] == K, f"solver stopped after {o['n_iter']} of {K} iterations (reason {o['stop_reason']})"
            assert np.all(np.isfinite(o["rel_trace"][:K]))
        times.append(batch_mod.max_over_ranks(elapsed, device=dev if backend == "nccl" else "cpu"))
    out = outs[0]
    t_max = float(np.median(times))
    elapsed = t_max
    iters_per_s = world * MPG * K / t_max
    edges_per_s = world * Ep_rank * K / t_max

    per_launch, dots_b, axpy_b, total_iter_bytes = algorithmic_bytes(N, Ep, K, mixed)
    result = {
        "metric": "fixed-point edges/sec (E' x Broyden iterations/sec) on 1M-node Poisson mesh" if args.workload == "mesh1m"
                  else f"fixed-point edges/sec (E' x Broyden iterations/sec), workload {args.workload}",
        "value": edges_per_s, "unit": "edges/s", "iters_per_sec": iters_per_s,
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * t_max / K,
        "repeats": R, "ms_per_step_all": [round(1e3 * t / K, 5) for t in times],
        "ms_per_step_spread": [1e3 * min(times) / K, 1e3 * max(times) / K],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {args.bc}/psignn, {MPG} x {N}-node hexagon Poisson mesh per GPU "
                               f"(BASELINE {cfg_name}), on-device Broyden iterations 1..{K} per mesh (iteration k sweeps k stored "
                               f"pairs: value depends on K; k-independent f figures in f_only / roofline_f), trained checkpoint weights",
                   "nodes": N, "edges_nonself": Ep, "edges_total": E, "solver": "broyden", "latent_dim": D,
                   "meshes_per_gpu": MPG, "parallelism": f"independent meshes x{world * MPG}, " + ("one mesh per GPU" if MPG == 1 else
                                   (f"{MPG} concurrent streams per GPU" if use_streams else
                                    f"{MPG} per GPU in one batched device solve (one launch per pass over all meshes)")),
                   "tiled_plan": bool(fmap.plan.tiled), "tiles": fmap.plan.n_tiles,
                   "max_tile_rows": fmap.plan.max_tile_rows, "ell_rows": fmap.plan.ell_rows},
        "rel_residual_after_K": out["rel_trace"][K - 1], "setup_s": round(t_setup, 2),
        "broyden_state_bytes": solver.nbytes,
        "roofline_iter": {"bound": "hbm", "achieved": MPG * total_iter_bytes / elapsed / 1e9, "peak": HBM_PEAK_GBS,
                          "unit": "GB/s", "frac": MPG * total_iter_bytes / elapsed / 1e9 / HBM_PEAK_GBS,
                          "note": "sum of algorithmic bytes of all kernels in the K iterations / wall time of this rank"},
    }

    # long vectors run the update as three single-array sweeps (csrc/solver.hip broyden_alloc: 16 floats per lane, no split):
    # the iteration's algorithmic bytes are then 12 k M + 56 M + B_f
    if MPG == 1 and N * D >= 768 * 4096 and os.environ.get("PSIGNN_UVU", "1") != "0":
        tb3 = per_launch["_total_iter3d"]
        result["roofline_iter"].update({
            "achieved": tb3 / elapsed / 1e9, "frac": tb3 / elapsed / 1e9 / HBM_PEAK_GBS,
            "note": "sum of algorithmic bytes of all kernels in the K iterations / wall time of this rank; the update runs as three "
                    "single-array sweeps (12 k M + 56 M + B_f bytes per iteration instead of BASELINE.md's 16 k M + 48 M + B_f), two "
                    "(8 k M + 52 M + B_f) while there are few stored pairs",
            "frac_by_the_two_pass_formula": total_iter_bytes / elapsed / 1e9 / HBM_PEAK_GBS})

    # ---- per-kernel durations: HIP events on the launch stream, instrumented repeat of the same K steps
    if rank == 0 and not args.no_kernel_timing:
        nat.prof_enable(True)
        run_all() if (MPG > 1 and not use_streams) else solver.solve(fmap, eps=0.0, poll_every=max(K, 1))
        prof = nat.prof_collect()
        nat.prof_enable(False)
        nmesh_launch = MPG if (MPG > 1 and not use_streams) else 1   # a batched launch covers every mesh of the shard
        kern = {}
        f_names = ("k_project", "k_node", "k_f_tile", "k_f_tile_fused")
        if "k_f_tile_fused" in prof:   # the solver's iterations use the fused kernel; judge f by it
            f_calls, f_ms = prof["k_f_tile_fused"]
            f_label, f_bytes = "f(k_f_tile_fused)", per_launch["f_fused"]
            if "k_f_tile" in prof:
                kern["f(k_f_tile)"] = (prof["k_f_tile"][0], prof["k_f_tile"][1],
                                       per_launch["f(k_project+k_node)"] * prof["k_f_tile"][0])
        else:
            f_ms = sum(prof.get(k, (0, 0.0))[1] for k in f_names)
            f_calls = max(prof.get("k_node", (0, 0.0))[0], prof.get("k_f_tile", (0, 0.0))[0], 1)
            f_label = "f(k_f_tile)" if "k_f_tile" in prof else "f(k_project+k_node)"
            f_bytes = per_launch["f(k_project+k_node)"]
        kern[f_label] = (f_calls, f_ms, f_bytes * f_calls * nmesh_launch)
        for name in ("k_xnext", "k_resid", "k_final"):
            if name in prof:
                kern[name] = (prof[name][0], prof[name][1], per_launch[name] * prof[name][0] * nmesh_launch)
        if "k_sweep_v" in prof:   # three-sweep update: its own algorithmic bytes, and the iteration's total with them
            folded = "k_sweep_u2d" in prof
            for name, lst in per_launch["_sweeps3d" if folded else "_sweeps3"].items():
                if name in prof:
                    kern[name] = (prof[name][0], prof[name][1], sum(lst) * nmesh_launch)
            tb3 = per_launch["_total_iter3d" if folded else "_total_iter3"]
            per_launch["_ks_used"] = per_launch["_sweeps3d_ks" if folded else "_sweeps3_ks"]
            result["roofline_iter"].update({
                "achieved": MPG * tb3 / elapsed / 1e9, "frac": MPG * tb3 / elapsed / 1e9 / HBM_PEAK_GBS,
                "note": "sum of algorithmic bytes of all kernels in the K iterations / wall time of this rank; the update runs as three "
                        "single-array sweeps (12 k M + 56 M + B_f bytes per iteration instead of BASELINE.md's 16 k M + 48 M + B_f)",
                "frac_by_the_two_pass_formula": MPG * total_iter_bytes / elapsed / 1e9 / HBM_PEAK_GBS})
        if "k_dots" in prof:
            kern["k_dots"] = (prof["k_dots"][0], prof["k_dots"][1], sum(dots_b) * nmesh_launch)
        if "k_axpy" in prof:
            kern["k_axpy"] = (prof["k_axpy"][0], prof["k_axpy"][1], sum(axpy_b) * nmesh_launch)
        table = []
        for name, (calls, ms, byts) in kern.items():
            gbs = byts / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            table.append({"kernel": name, "launches": calls, "avg_us": 1e3 * ms / max(calls, 1),
                          "time_share": ms, "alg_bytes_per_launch": byts / max(calls, 1), "GBps": gbs,
                          "frac_of_8TBps": gbs / HBM_PEAK_GBS})
        tot_ms = sum(v[1] for v in prof.values())
        for row in table:
            row["time_share"] = row["time_share"] / tot_ms
        for name, (calls, ms) in prof.items():
            if name not in f_names and name not in kern:
                table.append({"kernel": name, "launches": calls, "avg_us": 1e3 * ms / max(calls, 1),
                              "time_share": ms / tot_ms})
        table.sort(key=lambda r: -r["time_share"])
        dom = next(r for r in table if "GBps" in r)
        result["roofline"] = {"kernel": dom["kernel"], "bound": "hbm", "achieved": dom["GBps"], "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": dom["frac_of_8TBps"], "traffic": None,
                              "avg_launch_us": dom["avg_us"], "alg_bytes_per_launch": dom["alg_bytes_per_launch"],
                              "timing": "hipEvent pairs on the launch stream, instrumented repeat of the K timed steps",
                              "frac_of_measured_copy_6290GBps": dom["GBps"] / HBM_COPY_GBS}
        fr = next(r for r in table if r["kernel"] == f_label)
        result["roofline_f"] = {"kernel": fr["kernel"], "bound": "hbm", "achieved": fr["GBps"], "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": fr["frac_of_8TBps"], "traffic": None,
                                "avg_launch_us": fr["avg_us"], "alg_bytes_per_launch": fr["alg_bytes_per_launch"],
                                "f_evals_per_sec": 1e6 / fr["avg_us"], "edges_per_sec_f_only": Ep * 1e6 / fr["avg_us"]}
        result["kernels"] = table
        # HBM traffic per launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate passes, gfx950
        # read-side correction as MI355X_MICROARCH.md prescribes; scripts/collect_profiles.sh + summarise_pmc.py).  The sweeps'
        # traffic grows with the number of stored pairs k, so the summary is a per-kernel linear model  a + b k  fitted over
        # the launches of a K = 50 solve; it is evaluated here for THIS run's launches (k = 0 .. K-1), whatever K is.
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic_model.json")
        if os.path.exists(pmc) and args.workload == "mesh1m":
            try:
                t = json.load(open(pmc))
                if t.get("_nodes") == N:
                    ks = {"k_dots": range(1, K), "k_axpy": range(0, K), **per_launch.get("_ks_used", per_launch["_sweeps3_ks"])}
                    for key, row in (("roofline", dom), ("roofline_f", fr)):
                        e = t["kernels"].get(row["kernel"])
                        if e:
                            kk = list(ks.get(row["kernel"], [0]))
                            result[key]["traffic"] = float(np.mean([e["a"] + e["b"] * k for k in kk]))
                            result[key]["traffic_note"] = e["note"]
                            result[key]["traffic_over_algorithmic"] = result[key]["traffic"] / row["alg_bytes_per_launch"]
            except Exception as ex:   # a malformed summary must not take the bench line down
                result["traffic_error"] = repr(ex)

    # ---- k-independent figures of the GNN block: plain f and the analytic JVP, back-to-back launches in plan order
    if rank == 0 and MPG == 1 and not args.no_kernel_timing:
        def timed(fn, reps):
            fn()
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize(dev)
            return 1e3 * e0.elapsed_time(e1) / reps   # us per call (torch's current stream = the launch stream here)
        xp = fmap.to_plan(out["result"] if "result" in out else fmap.h0)
        us_f = 1e3 * _time_picard(fmap, xp, 100, dev)
        b_f = per_launch["f(k_project+k_node)"]
        result["f_only"] = {"kernel": "k_f_tile", "us_per_eval": us_f, "f_evals_per_sec": 1e6 / us_f,
                            "edges_per_sec": Ep * 1e6 / us_f, "alg_bytes_per_launch": b_f,
                            "achieved": b_f / us_f * 1e-3, "unit": "GB/s", "peak": HBM_PEAK_GBS,
                            "frac": b_f / us_f * 1e-3 / HBM_PEAK_GBS, "bound": "hbm",
                            "note": "100 back-to-back evaluations x <- f(x) on the device (psignn_picard_p), wall time / 100"}
        if fmap.plan.tiled and not mixed:
            vp = torch.randn_like(xp)
            us_j = timed(lambda: fmap.jvp_p(xp, vp), 50)
            result["roofline_jvp"] = {"kernel": "k_jvp_tile", "bound": "hbm", "achieved": per_launch["jvp"] / us_j * 1e-3,
                                      "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": per_launch["jvp"] / us_j * 1e-3 / HBM_PEAK_GBS,
                                      "traffic": None, "avg_launch_us": us_j, "alg_bytes_per_launch": per_launch["jvp"],
                                      "note": "analytic J_f(x) v of the Newton-Krylov path (BASELINE configs[4]); 50 back-to-back "
                                              "products, HIP events on the launch stream (includes the host's launch overhead)"}
            # one inner solve of that path: m Arnoldi steps of the device GMRES (csrc/krylov.hip) on (J - I) dx = -g at this
            # iterate -- step j = one JVP + classical Gram-Schmidt twice against j + 1 basis vectors (4 (j + 1) sweeps of M floats)
            m_nk = 30
            gm = eng.DeviceGmres(xp.numel(), dev, m_nk)
            neg_g = torch.empty_like(xp)
            gm.residual_norms(xp, fmap.fp(xp), neg_g=neg_g)

            def arnoldi():
                gm.begin(neg_g)
                for j in range(m_nk):
                    fmap.jvp_p(xp, gm.row(j, xp.shape), out=gm.row(j + 1, xp.shape))
                    gm.step(j, 1.0, 0.0, poll=False)
            us_nk = timed(arnoldi, 3)
            hist = gm.history()
            Mb = xp.numel() * 4
            nk_bytes = sum(per_launch["jvp"] + (4 * (j + 1) + 3) * Mb for j in range(m_nk))
            result["newton_krylov"] = {"arnoldi_steps": m_nk, "us_per_inner_solve": us_nk, "us_per_step_avg": us_nk / m_nk,
                                       "alg_bytes": nk_bytes, "achieved": nk_bytes / us_nk * 1e-3, "unit": "GB/s",
                                       "frac": nk_bytes / us_nk * 1e-3 / HBM_PEAK_GBS,
                                       "linear_residual_after_m": float(hist[-1]) if len(hist) else None,
                                       "note": "device GMRES without restarts around the analytic JVP, Hessenberg / Givens on the "
                                               "device, no host synchronisation inside the m steps"}
            gm.close()
    for sv in solvers:
        sv.close()

    # ---- CPU baseline: the oracle (port of the reference path) on this box's host cores, bounded sample
    if rank == 0 and world == 1 and MPG == 1 and not args.no_cpu_baseline:
        from oracle import psignn_oracle as orc
        ncpu = os.cpu_count() or 1
        with torch.no_grad():
            h0c = orc.encoder(sd, mesh.x)
            f = lambda H: orc.function_forward(sd, H, h0c, mesh)
            # thread count: torch's CPU ops get SLOWER beyond a few dozen threads on this path (measured on the
            # 256-thread MI355X host: one 1M-node f call 0.99 s at 16 threads, 1.10 s at 32, 8.0 s at 256), so the
            # baseline is timed at the fastest of a short calibration, not at os.cpu_count().
            best = None
            for nt in [t for t in (8, 16, 32) if t <= ncpu] or [ncpu]:
                torch.set_num_threads(nt)
                f(h0c)
                t1 = time.perf_counter()
                f(h0c)
                dt = time.perf_counter() - t1
                if best is None or dt < best[1]:
                    best = (nt, dt)
            cores, t_f = best
            torch.set_num_threads(cores)
            S = int(max(2, min(K, args.cpu_seconds / (1.6 * t_f))))
            t1 = time.perf_counter()
            o = orc.broyden(f, h0c, threshold=S, eps=0.0)
            t_cpu = time.perf_counter() - t1
        result["cpu_baseline"] = {
            "value": Ep * S / t_cpu, "unit": "edges/s", "iters_per_sec": S / t_cpu, "cores": cores, "kind": "port",
            "sample": f"oracle broyden iterations 1..{S} on the same {N}-node mesh and weights "
                      f"(torch CPU ops, {cores} of {ncpu} host threads = fastest of a calibration over 8/16/32; "
                      f"one f call {t_f:.2f} s; total {t_cpu:.1f} s); "
                      f"CPU cost per iteration grows with k like the GPU's, the sample covers k < {S} only",
            "rel_residual_after_S": o["rel_trace"][S - 1],
            "gpu_rel_residual_at_S": out["rel_trace"][S - 1]}

    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark: PSI-GNN fixed-point iterations/s and edges/s on a 1M-node Dirichlet Poisson mesh.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Under ``torch.distributed.run`` (RANK / WORLD_SIZE in the environment) this process IS one
rank; started bare, ``bench.py --gpus N`` launches its N ranks itself -- the parent spawns N fresh child processes before it
touches the GPU, relays rank 0's JSON line and exits non-zero if any child does.

Workload (BASELINE.json configs[4], the configuration the metric is quoted on): one synthetic hexagon-lattice Dirichlet Poisson
mesh of 1 000 519 nodes / E' = 5.98 M directed non-self edges per GPU (SURVEY section 8d recipe), trained dirichlet checkpoint
weights, fp32 -- for every N, so that the per-N values the driver divides are values of ONE workload (weak scaling: every rank
solves its own mesh, own problem seed, no data-path collective; value = sum over ranks).  ``--workload batch50k`` runs BASELINE
configs[3] instead (8 independent 50k-node meshes per GPU, seeds rank * 8 + 0..7, one batched device solve per rank).
A *step* is one fixed-point iteration of the reference's solver -- Broyden, utilities/solver.py:116-207: one evaluation of the
GNN block f plus the residual norms, stop tests and the rank-1 inverse-Jacobian update -- executed entirely on the device
through libpsignn_hip.so.  The timed region is iterations 1..K of a solve started from the encoder's initial state (eps = 0, so
exactly K iterations run; the k-th iteration sweeps k stored (u, v) pairs, which is part of the solver's real cost).  Inputs are
resident in HBM when the clock starts.  edges/s = E' x iterations/s (each edge counted once although Phi_to and Phi_from both
process it).

Warm-up: the W untimed iterations the command line asks for, then ONE untimed solve of K iterations on the timed solver object
(its state has then been touched and every kernel variant the K steps need has been launched once: BENCH_r02's first timed
repeat ran 3.3x slower than the others).  The timed region (exactly K iterations, barrier + synchronize on both sides, MAX over
ranks) is repeated --repeats times (default 5); `value` / `ms_per_step` are the MEDIAN repeat, `ms_per_step_first` and the
spread are reported beside them.  `value` depends on K -- iteration k sweeps k stored pairs, as in the reference -- so the
k-independent figures of the GNN block are in the same line: `f_only` (plain f, back-to-back evaluations) and `roofline_f`.

One JSON line on stdout (rank 0).  Extra objects:
  roofline      dominant kernel (by time): algorithmic bytes per launch / average launch duration.  Both come from the
                library's own launch records of an instrumented repeat of the same K steps (HIP events on the launch stream; the
                bytes are stated at the launch sites from what was launched -- stored pairs swept, kept window, meshes of the
                shard -- csrc/common.h PROF_BYTES); `traffic` = HBM bytes per launch from the committed rocprofv3 FETCH_SIZE /
                WRITE_SIZE passes (profiles/pmc_traffic_model.json: per kernel a linear model in the launch's algorithmic bytes)
  roofline_f    the same for the GNN block f inside the loop (k_f_tile_fused: x + update, f, g, dg, norms in one kernel) --
                the kernel north_star's 60 % target is about
  f_only        plain f (k_f_tile), 100 back-to-back evaluations: us per evaluation, edges/s, fraction of the HBM peak
  roofline_jvp  the JVP of the Newton-Krylov path (k_jvp_lin: stored linearisation), 50 back-to-back products; roofline_jvp_direct: k_jvp_tile
  newton_krylov one inner solve of that path: 30 Arnoldi steps of the device GMRES around the JVP
  roofline_iter whole-iteration algorithmic bytes (sum over every launch of the instrumented repeat) / un-instrumented wall time
  cpu_baseline  the CPU oracle (port of the reference path) timed on this box's host cores on a bounded sample
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import socket
import subprocess
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
    "mixedbatch50k": (50_000, "mixed", 8, "configs[3] shape, mixed family"),
    "f1m": (1_000_000, "dirichlet", 1, "configs[4] size, GNN block only"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="mesh1m",
                    help="mesh1m = BASELINE configs[4] size (headline, every N); dir100k = configs[1]; mixed100k = configs[2]; "
                         "batch50k = configs[3] (8 independent 50k-node meshes per GPU, one batched device solve)")
    ap.add_argument("--nodes", type=int, default=None, help="override the preset's per-mesh node count")
    ap.add_argument("--bc", choices=["dirichlet", "mixed"], default=None)
    ap.add_argument("--meshes-per-gpu", type=int, default=None)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of exactly --steps iterations each; the median is reported")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--dump-launch-log", default=None, help="write the instrumented repeat's launches (name, ms, algorithmic bytes) as JSON")
    return ap.parse_args()


def launch_ranks(args) -> int:
    """``bench.py --gpus N`` started without a launcher: spawn the N ranks (fresh processes; this parent never touches the GPU),
    relay rank 0's JSON line, return non-zero if any rank failed."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    lines = [l for l in (out0 or "").splitlines() if l.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad or not lines:
        print(f"bench.py: ranks failed (rank, exit code): {bad}; JSON line from rank 0: {bool(lines)}", file=sys.stderr)
        return next((rc for _, rc in bad), 1) or 1
    return 0


def load_weights(kind):
    w = np.load(os.path.join(ROOT, "tests", "golden", f"weights_{kind}.npz"))
    return {k: torch.from_numpy(w[k]) for k in w.files}


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
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))          # (before any GPU call or torch.cuda query in this process)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    # one process per GPU; PSIGNN_BENCH_BACKEND=gloo lets several ranks share the cards of a small box for a
    # rehearsal of the N > 1 path (RCCL refuses two ranks on one device)
    backend = os.environ.get("PSIGNN_BENCH_BACKEND", "nccl")
    dry = os.environ.get("PSIGNN_BENCH_DRYRUN", "0") == "1"   # launcher / rendezvous rehearsal without a GPU (tests/test_multi_rank.py)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if dry:
        if dist is not None:
            dist.init_process_group("gloo")
            t = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            n_ranks, be = dist.get_world_size(), dist.get_backend()
            dist.barrier()
            dist.destroy_process_group()
        else:
            t, n_ranks, be = torch.tensor([0.001]), 1, None
        if os.environ.get("PSIGNN_BENCH_DRYRUN_FAIL") == str(rank):
            sys.exit(3)               # (test hook: a rank that fails after the collectives -> the launcher must report it)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks": n_ranks, "backend": be, "max_over_ranks_s": float(t[0]),
                              "steps": args.steps, "warmup": args.warmup, "workload": args.workload}))
        return
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    local = local % max(torch.cuda.device_count(), 1) if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if dist is not None:
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
    b_f = (102 if mixed else 89) * N + 20 * Ep                 # one f evaluation

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
    use_streams = os.environ.get("PSIGNN_BENCH_STREAMS", "0") == "1"   # round 1's form: one HIP stream + host thread per mesh
    shard = sum(f.plan.N for f in fmaps) * D if MPG > 1 and not use_streams else 0
    solvers = [eng.DeviceBroyden(plan=f.plan, threshold=K, keep_trace=False, shard_elems=shard) for f in fmaps]
    solver = solvers[0]
    batched = MPG > 1 and not use_streams
    if batched:
        assert eng.shard_batchable(solvers), "the shard's meshes are not of one size class"
    streams = [torch.cuda.Stream(dev) for _ in range(MPG)] if MPG > 1 else [None]

    def run_all():
        """Exactly K Broyden iterations on every mesh of this rank.  Several meshes per GPU (configs[3]) go through the
        batched device solver: one launch per pass over all of them (PSIGNN_BENCH_STREAMS=1: concurrent streams instead)."""
        if MPG == 1:
            return [solver.solve(fmap, eps=0.0, poll_every=max(K, 1))]
        if batched:
            return eng.broyden_solve_batch(solvers, fmaps, 0.0, poll_every=max(K, 1))
        import concurrent.futures as cf

        def one(j):
            torch.cuda.set_device(local)
            with torch.cuda.stream(streams[j]):
                return solvers[j].solve(fmaps[j], eps=0.0, poll_every=max(K, 1))
        with cf.ThreadPoolExecutor(MPG) as ex:  # ctypes releases the GIL inside the library calls
            return list(ex.map(one, range(MPG)))

    run_all()                                   # untimed: the timed solver objects' state touched, every kernel variant of K steps launched
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

    result = {
        "metric": "fixed-point edges/sec (E' x Broyden iterations/sec) on 1M-node Poisson mesh" if args.workload == "mesh1m"
                  else f"fixed-point edges/sec (E' x Broyden iterations/sec), workload {args.workload}",
        "value": edges_per_s, "unit": "edges/s", "iters_per_sec": iters_per_s,
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * t_max / K,
        "ms_per_step_first": 1e3 * times[0] / K,
        "repeats": R, "ms_per_step_all": [round(1e3 * t / K, 5) for t in times],
        "ms_per_step_spread": [1e3 * min(times) / K, 1e3 * max(times) / K],
        "warmup_extra": "one untimed solve of K steps on the timed solver objects after the W warm-up iterations",
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "ranks": {"world_size": dist.get_world_size() if dist is not None else 1,
                  "backend": (dist.get_backend() if dist is not None else None),
                  "collective": "none on the data path; barrier + MAX of the elapsed time over ranks around every timed region"},
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
    }

    # ---- per-kernel durations and algorithmic bytes: the library's launch records of an instrumented repeat of the same K steps
    # (HIP events on the launch stream around every launch; bytes stated at the launch sites, csrc/common.h PROF_BYTES)
    if rank == 0 and not args.no_kernel_timing:
        nat.prof_enable(True)
        run_all() if batched else solver.solve(fmap, eps=0.0, poll_every=max(K, 1))
        prof = nat.prof_collect(with_bytes=True)
        log = nat.prof_launch_log()
        nat.prof_enable(False)
        if args.dump_launch_log:
            json.dump(log, open(args.dump_launch_log, "w"))
        # consistency of the records: one fused f per iteration (+ the plain f of g0 per mesh), one V sweep / reduce per iteration
        nf = prof.get("k_f_tile_fused", (0,))[0]
        assert nf in (0, K, 2 * K), f"{nf} fused f launches for {K} iterations"       # (2 K: mixed two-group launch)
        assert sum(1 for nm, _, _ in log if nm == "k_reduce_check") == K
        tot_ms = sum(v[1] for v in prof.values())
        table = []
        for name, (calls, ms, byts) in prof.items():
            row = {"kernel": name, "launches": calls, "avg_us": 1e3 * ms / max(calls, 1), "time_share": ms / tot_ms}
            if byts > 0:
                # launches that state no bytes (the stop-test no-ops of the last iteration return at once) do not count as launches of work
                work = sum(1 for nm, _, b in log if nm == name and b > 0)
                ms_work = sum(m for nm, m, b in log if nm == name and b > 0)
                gbs = byts / (ms_work * 1e-3) / 1e9 if ms_work > 0 else 0.0
                row.update({"work_launches": work, "avg_us": 1e3 * ms_work / max(work, 1), "alg_bytes_per_launch": byts / max(work, 1),
                            "GBps": gbs, "frac_of_8TBps": gbs / HBM_PEAK_GBS})
            table.append(row)
        table.sort(key=lambda r: -r["time_share"])
        total_bytes = sum(v[2] for v in prof.values())
        result["roofline_iter"] = {"bound": "hbm", "achieved": total_bytes / elapsed / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": total_bytes / elapsed / 1e9 / HBM_PEAK_GBS, "alg_bytes": total_bytes,
                                   "note": "sum of the algorithmic bytes of every launch of the K iterations (library records) / "
                                           "median un-instrumented wall time of this rank; BASELINE.md's two-pass formula "
                                           "16 k M + 48 M + B_f per iteration is undercut by the three-sweep / folded update",
                                   "frac_by_the_two_pass_formula": MPG * (sum(16 * k * N * D + 48 * N * D + b_f for k in range(K)) + b_f) / elapsed / 1e9 / HBM_PEAK_GBS}
        dom = next(r for r in table if "GBps" in r)
        result["roofline"] = {"kernel": dom["kernel"], "bound": "hbm", "achieved": dom["GBps"], "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": dom["frac_of_8TBps"], "traffic": None,
                              "avg_launch_us": dom["avg_us"], "alg_bytes_per_launch": dom["alg_bytes_per_launch"],
                              "timing": "hipEvent pairs on the launch stream around every launch, instrumented repeat of the K timed steps",
                              "frac_of_measured_copy_6290GBps": dom["GBps"] / HBM_COPY_GBS}
        fr = next((r for r in table if r["kernel"] == "k_f_tile_fused" and "GBps" in r), None) or \
            next(r for r in table if r["kernel"].startswith("k_f_tile") or r["kernel"] in ("k_node",))
        if "GBps" in fr:
            result["roofline_f"] = {"kernel": fr["kernel"], "bound": "hbm", "achieved": fr["GBps"], "peak": HBM_PEAK_GBS,
                                    "unit": "GB/s", "frac": fr["frac_of_8TBps"], "traffic": None,
                                    "avg_launch_us": fr["avg_us"], "alg_bytes_per_launch": fr["alg_bytes_per_launch"],
                                    "f_evals_per_sec": MPG * 1e6 / fr["avg_us"], "edges_per_sec_f_only": Ep_rank * 1e6 / fr["avg_us"]}
        result["kernels"] = table
        # HBM traffic per launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate passes, gfx950
        # read-side correction as MI355X_MICROARCH.md prescribes; scripts/collect_profiles.sh + summarise_pmc.py): per kernel a
        # linear model  traffic = alpha + beta x (algorithmic bytes of the launch), fitted over the launches of a K = 50 solve
        # lined up with the library's launch log; evaluated here at THIS run's mean algorithmic bytes per launch.
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic_model.json")
        if os.path.exists(pmc) and args.workload == "mesh1m":
            try:
                t = json.load(open(pmc))
                if t.get("_nodes") == N and t.get("_model") == "alpha + beta * alg_bytes":
                    for key, row in (("roofline", dom), ("roofline_f", fr)):
                        e = t["kernels"].get(row["kernel"])
                        if e and key in result:
                            result[key]["traffic"] = float(e["alpha"] + e["beta"] * row["alg_bytes_per_launch"])
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
        result["f_only"] = {"kernel": "k_f_tile", "us_per_eval": us_f, "f_evals_per_sec": 1e6 / us_f,
                            "edges_per_sec": Ep * 1e6 / us_f, "alg_bytes_per_launch": b_f,
                            "achieved": b_f / us_f * 1e-3, "unit": "GB/s", "peak": HBM_PEAK_GBS,
                            "frac": b_f / us_f * 1e-3 / HBM_PEAK_GBS, "bound": "hbm",
                            "note": "100 back-to-back evaluations x <- f(x) on the device (psignn_picard_p), wall time / 100"}
        if fmap.plan.tiled and not mixed:
            b_jvp = b_f + 40 * N                                                    # + the tangent v (B_jvp)
            vp = torch.randn_like(xp)
            us_j = timed(lambda: fmap.jvp_p(xp, vp), 50)
            result["roofline_jvp_direct"] = {"kernel": "k_jvp_tile", "bound": "hbm", "achieved": b_jvp / us_j * 1e-3,
                                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b_jvp / us_j * 1e-3 / HBM_PEAK_GBS,
                                             "traffic": None, "avg_launch_us": us_j, "alg_bytes_per_launch": b_jvp,
                                             "note": "J_f(x) v with the value path recomputed per product (any state, any plan); "
                                                     "50 back-to-back products, HIP events on the launch stream"}
            # what Newton-Krylov applies (csrc/fgnn_tile_lin.hip): masks and per-node quantities stored once per Newton step
            # (psignn_lin_build), the product reads v, the 96-byte node record and one dword per slot (psignn_lin_jvp)
            lin = fmap.linearize_p(xp)
            b_lin = N * (81 + 96) + int(fmap.plan.ell_rows) * 64 * 4
            b_build = 49 * N + 20 * Ep + N * 96 + int(fmap.plan.ell_rows) * 64 * 4
            us_b = timed(lambda: fmap.linearize_p(xp, lin), 20)
            us_l = timed(lambda: lin.jvp_p(vp), 50)
            result["roofline_jvp"] = {"kernel": "k_jvp_lin", "bound": "hbm", "achieved": b_lin / us_l * 1e-3,
                                      "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b_lin / us_l * 1e-3 / HBM_PEAK_GBS,
                                      "traffic": None, "avg_launch_us": us_l, "alg_bytes_per_launch": b_lin,
                                      "build_us": us_b, "build_alg_bytes": b_build, "build_frac": b_build / us_b * 1e-3 / HBM_PEAK_GBS,
                                      "frac_by_survey_B_jvp": b_jvp / us_l * 1e-3 / HBM_PEAK_GBS,
                                      "note": "J_f(x) v of the Newton-Krylov path (BASELINE configs[4]) from the stored "
                                              "linearisation: one build per Newton step (build_us), then this product per "
                                              "Krylov vector; 50 back-to-back products, HIP events on the launch stream"}
            # one inner solve of that path: m Arnoldi steps of the device GMRES (csrc/krylov.hip) on (J - I) dx = -g at this
            # iterate -- step j = one JVP + classical Gram-Schmidt against j + 1 basis vectors: one dots + one axpy sweep
            # (2 (j + 1) vectors), and the same again when the device's DGKS test asks for re-orthogonalisation
            m_nk = 30
            gm = eng.DeviceGmres(xp.numel(), dev, m_nk)
            neg_g = torch.empty_like(xp)
            gm.residual_norms(xp, fmap.fp(xp), neg_g=neg_g)

            def arnoldi():
                gm.begin(neg_g)
                fmap.linearize_p(xp, lin)
                for j in range(m_nk):
                    lin.jvp_p(gm.row(j, xp.shape), out=gm.row(j + 1, xp.shape))
                    gm.step(j, 1.0, 0.0, poll=False)
            us_nk = timed(arnoldi, 3)
            hist = gm.history()
            n_reorth = int(gm.reorth_count())      # steps of the LAST inner solve whose second Gram-Schmidt pass ran
            Mb = xp.numel() * 4
            # every step: JVP + first pass 2 (j + 1) + 3 vectors; the n_reorth steps that re-orthogonalised: 2 (j + 1) more.  Which
            # steps they were is not recorded: the bound below prices them as the LAST n_reorth steps (most bytes) and as the FIRST
            lo = b_build + sum(b_lin + (2 * (j + 1) + 3) * Mb for j in range(m_nk))
            extra = sorted(2 * (j + 1) * Mb for j in range(m_nk))
            nk_lo, nk_hi = lo + sum(extra[:n_reorth]), lo + sum(extra[m_nk - n_reorth:])
            result["newton_krylov"] = {"arnoldi_steps": m_nk, "reorthogonalised_steps": n_reorth, "us_per_inner_solve": us_nk,
                                       "us_per_step_avg": us_nk / m_nk, "alg_bytes": nk_hi if n_reorth == m_nk else [nk_lo, nk_hi],
                                       "achieved": nk_hi / us_nk * 1e-3, "unit": "GB/s", "frac": nk_hi / us_nk * 1e-3 / HBM_PEAK_GBS,
                                       "frac_lower": nk_lo / us_nk * 1e-3 / HBM_PEAK_GBS,
                                       "linear_residual_after_m": float(hist[-1]) if len(hist) else None,
                                       "linear_residual_first": float(hist[0]) if len(hist) else None,
                                       "note": "one linearisation + device GMRES without restarts around its JVP, Hessenberg / Givens on the "
                                               "device, no host synchronisation inside the m steps; bytes priced with the number of "
                                               "steps whose second Gram-Schmidt pass ran (psignn_gmres_reorth_count)"}
            gm.close()
            lin.close()
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

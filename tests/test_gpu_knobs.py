"""Every run-time knob of the library that no other test selects gets a test here: the non-default branch runs, and what it
computes is the default's result -- bit for bit where the arithmetic is the same (launch order, occupancy cap, gather forms of the
mixed adjoint products share nothing but the result), within rounding where partial sums are shaped differently (vector widths,
split of the sweeps over the stored pairs, unconditional second Gram-Schmidt pass)."""
import numpy as np
import pytest
import torch

from conftest import load_weights, pkg, rel_l2
from oracle import psignn_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _fmap(mesh, sd, dev, **plan_kw):
    eng = pkg("engine")
    md = mesh.to(dev)
    with torch.no_grad():
        h0 = orc.encoder(sd, mesh.x)
    plan = eng.MeshPlan(md, **plan_kw)
    fm = eng.FixedPointMap(plan, eng.PackedWeights(sd, dev), h0.to(dev), md.prb_data, getattr(md, "unit_normal_vector", None))
    return md, h0, plan, fm


def test_tile_launch_order_and_occupancy_cap_change_nothing(dev, knobs):
    """PSIGNN_TILE_ORDER=cost (costliest tiles first inside every XCD's run) and PSIGNN_TILE_LDS_MIN (larger LDS request = fewer
    workgroups per CU; also honoured by the JVP kernels): experiments of DESIGN section 4 -- same bits out of f, of the fused Broyden
    step and of the linearised JVP."""
    data, solver = pkg("data"), pkg("utilities.solver")
    sd = load_weights("dirichlet")
    mesh = data.make_hex_problem(data.hex_n_for_nodes(60000), seed=6, compute_sol=False)
    md, h0, plan, fm = _fmap(mesh, sd, dev)
    assert plan.tiled and plan.n_tiles > 200
    x = fm.to_plan(fm.h0)
    v = torch.randn_like(x)

    def run():
        y = fm.fp(fm.fp(x))
        out = solver.broyden(fm, fm.h0, threshold=8, eps=0.0, keep_trace=False)
        lin = fm.linearize_p(y)
        jv = lin.jvp_p(v).clone()
        lin.close()
        return y.clone(), out["result"].clone(), list(out["rel_trace"]), jv, fm.jvp_p(y, v).clone()
    base = run()
    for env in ({"PSIGNN_TILE_ORDER": "cost"}, {"PSIGNN_TILE_LDS_MIN": "40000"}, {"PSIGNN_TILE_ORDER": "cost", "PSIGNN_TILE_LDS_MIN": "60000"}):
        knobs(**env)
        got = run()
        knobs(**{k: None for k in env})
        assert torch.equal(got[0], base[0]) and torch.equal(got[1], base[1]) and got[2] == base[2], env
        assert torch.equal(got[3], base[3]) and torch.equal(got[4], base[4]), env


def test_mixed_adjoint_products_on_the_gather_kernels(dev, knobs):
    """PSIGNN_MIXED_JVP=gather / PSIGNN_MIXED_VJP=gather: the mixed family's J v and w^T J through the global-gather kernels instead of the
    tile kernels (A/B timing knobs) -- the same products."""
    data = pkg("data")
    sd = load_weights("mixed")
    mesh = data.make_hex_problem(data.hex_n_for_nodes(30000), seed=7, mixed=True, compute_sol=False)
    md, h0, plan, fm = _fmap(mesh, sd, dev)
    assert plan.tiled and plan.mixed
    gen = torch.Generator().manual_seed(8)
    h = (fm.h0 + 0.1 * torch.randn(h0.shape, generator=gen).to(dev)).contiguous()
    v = torch.randn(h0.shape, generator=gen).to(dev)
    nat = pkg("_native")
    nat.prof_enable(True)
    nat.prof_collect()
    jt, vt = fm.jvp(h, v), fm.vjp(h, v)
    ran_t = nat.prof_collect()
    knobs(PSIGNN_MIXED_JVP="gather", PSIGNN_MIXED_VJP="gather")
    jg, vg = fm.jvp(h, v), fm.vjp(h, v)
    ran_g = nat.prof_collect()
    nat.prof_enable(False)
    knobs(PSIGNN_MIXED_JVP=None, PSIGNN_MIXED_VJP=None)
    assert "k_jvp_tile" in ran_t and "k_vjp_tile_a" in ran_t and "k_jvp_tile" not in ran_g and "k_vjp_tile_a" not in ran_g, (ran_t, ran_g)
    assert rel_l2(jt, jg) < 2e-6 and rel_l2(vt, vg) < 2e-6, (rel_l2(jt, jg), rel_l2(vt, vg))


@pytest.mark.parametrize("env", [{"PSIGNN_VEC16_MIN": "1"}, {"PSIGNN_VEC16_MIN": "1000000000"}, {"PSIGNN_VEC_AX4": "1", "PSIGNN_UVU": "0"},
                                 {"PSIGNN_VEC_AX4": "0", "PSIGNN_UVU": "0"}])
def test_solver_shape_knobs_follow_the_default_solve(env, dev, monkeypatch):
    """Vector width of the solver kernels (PSIGNN_VEC16_MIN), the 4-float axpy pass of the two-pass form (PSIGNN_VEC_AX4): other
    partial-sum shapes, the same iteration -- a linear, well-conditioned problem
    through the generic-callable path, first iterations against the default run, final iterate against the closed form."""
    eng = pkg("engine")
    gen = torch.Generator().manual_seed(12)
    n = 40000
    c = (0.05 + 0.85 * torch.rand((n, 10), generator=gen)).to(dev)
    b = torch.randn((n, 10), generator=gen).to(dev)
    x0 = torch.zeros((n, 10), device=dev)
    f = lambda x: c * x + b
    solver = pkg("utilities.solver")

    def run():
        return solver.broyden(f, x0, threshold=40, eps=0.0)
    base = run()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    got = run()
    for k in env:
        monkeypatch.delenv(k)
    np.testing.assert_allclose(got["rel_trace"][:10], base["rel_trace"][:10], rtol=1e-3)
    star = b / (1.0 - c)
    assert rel_l2(got["result"], star) < 1e-4 and rel_l2(base["result"], star) < 1e-4
    assert got["lowest"] < 1e-5


def test_gmres_unconditional_reorthogonalisation(dev, knobs):
    """PSIGNN_GMRES_REORTH=always (second Gram-Schmidt pass in every Arnoldi step instead of on the device's DGKS test): the
    Newton-Krylov iterates of a small mesh agree with the default's to the accuracy of the inner solves."""
    data, solver = pkg("data"), pkg("utilities.solver")
    sd = load_weights("dirichlet")
    mesh = data.make_hex_problem(data.hex_n_for_nodes(3000), seed=9, compute_sol=False)
    md, h0, plan, fm = _fmap(mesh, sd, dev)
    a = solver.newton_krylov(fm, fm.h0, threshold=5, eps=1e-7, inner_m=40)
    knobs(PSIGNN_GMRES_REORTH="always")
    b = solver.newton_krylov(fm, fm.h0, threshold=5, eps=1e-7, inner_m=40)
    knobs(PSIGNN_GMRES_REORTH=None)
    np.testing.assert_allclose(a["rel_trace"][:2], b["rel_trace"][:2], rtol=5e-2)
    assert 0.2 < a["lowest"] / b["lowest"] < 5.0


def test_idle_solver_is_lent_between_solves_of_one_map(dev, monkeypatch):
    """utilities.solver.broyden without keep_trace / solver_obj borrows the map's idle DeviceBroyden and returns it: the second solve
    runs on the same state buffers (no 2 * threshold vectors re-allocated) and gives the same bits; another threshold replaces the
    idle solver; PSIGNN_SOLVER_CACHE_GB=0 keeps nothing; a kept trace never shares a solver."""
    data, solver = pkg("data"), pkg("utilities.solver")
    sd = load_weights("dirichlet")
    mesh = data.make_hex_problem(data.hex_n_for_nodes(20000), seed=10, compute_sol=False)
    md, h0, plan, fm = _fmap(mesh, sd, dev)
    a = solver.broyden(fm, fm.h0, threshold=12, eps=0.0, keep_trace=False)
    sv = fm._idle_broyden
    assert sv is not None and sv.threshold == 12
    b = solver.broyden(fm, fm.h0, threshold=12, eps=0.0, keep_trace=False)
    assert fm._idle_broyden is sv
    assert torch.equal(a["result"], b["result"]) and a["rel_trace"] == b["rel_trace"]
    assert torch.equal(a["xest_trace"][0], fm.h0) and torch.equal(a["xest_trace"][a["nstep"]], a["result"])
    with pytest.raises(RuntimeError):
        a["xest_trace"][3]
    c = solver.broyden(fm, fm.h0, threshold=7, eps=0.0, keep_trace=False)
    assert fm._idle_broyden is not sv and fm._idle_broyden.threshold == 7
    np.testing.assert_allclose(c["rel_trace"][:7], a["rel_trace"][:7], rtol=1e-4)
    kept = solver.broyden(fm, fm.h0, threshold=7, eps=0.0, keep_trace=True)
    assert fm._idle_broyden.threshold == 7 and torch.equal(kept["xest_trace"][7], kept["xest_trace"][7])
    assert rel_l2(kept["xest_trace"][kept["nstep"]], kept["result"]) == 0.0
    monkeypatch.setenv("PSIGNN_SOLVER_CACHE_GB", "0")
    solver.broyden(fm, fm.h0, threshold=7, eps=0.0, keep_trace=False)
    assert fm._idle_broyden is None

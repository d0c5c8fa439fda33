/*
 * psignn_hip.h — C ABI of libpsignn_hip.so, the MI355X (gfx950) implementation of the PSI-GNN
 * fixed-point inference hot path.
 *
 * The reference (mnastorg/PSI-GNN) has no FFI layer: its hot path is Python calling
 * torch_geometric / torch_sparse / ATen kernels.  Each entry point below replaces one such
 * call site; the citation after "replaces:" is the reference file:line (relative to the
 * reference repository root).  All pointers named d_* are DEVICE pointers (hipMalloc'ed or
 * owned by any framework's allocator); h_* are host pointers.  `stream` is a hipStream_t
 * passed as void* (NULL = default stream).  Every function returns 0 on success or a negative
 * PSIGNN_E* code; psignn_last_error() returns a message for the calling thread.
 *
 * Unless stated otherwise calls are asynchronous on `stream` and perform no host sync.
 * Tensors are dense row-major float32; node states are (N, 10) — the latent width d = 10 is
 * the only value the reference ever trains (SURVEY.md "d") and is fixed at compile time.
 */
#ifndef PSIGNN_HIP_H
#define PSIGNN_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSIGNN_D 10            /* latent_dim */
#define PSIGNN_EDGE_F 3        /* edge_features_dim */

#define PSIGNN_OK 0
#define PSIGNN_EINVAL (-1)     /* bad argument / shape */
#define PSIGNN_EHIP (-2)       /* HIP runtime error */
#define PSIGNN_EINDEX (-3)     /* edge index out of range */
#define PSIGNN_ENOMEM (-4)

const char* psignn_last_error(void);
int psignn_version(void);
/* Re-read the PSIGNN_* environment knobs at their next use (they are cached after the first read).  The knobs select
 * alternative forms of a kernel for A/B measurements and tests (MFMA stage 1, Hilbert tiling, gather kernels for the mixed
 * family, ...); a normal run sets none of them.  No reference counterpart. */
void psignn_reload_knobs(void);

/* ------------------------------------------------------------------------------------------
 * Mesh plan: everything iteration-invariant, computed once per mesh on the device.
 *
 * replaces: torch_geometric.utils.remove_self_loops executed twice per f call
 *           (dirichlet/psignn/model.py:342,360), torch.where(tags == 1) re-run every f call
 *           (model.py:281; mixed/psignn/model.py:218-219), and the COO->CSR conversion inside
 *           SparseTensor(...) (model.py:159-163).
 *
 * Layout built (all int32 / float32, device memory owned by the plan):
 *   csr: non-self edges grouped by row  r = edge_index[0]  (aggregation side of Phi_from)
 *   csc: non-self edges grouped by col  c = edge_index[1]  (aggregation side of Phi_to)
 *   inside a group edges are ordered by (other endpoint, original edge id) — a canonical order
 *   independent of the input edge order; edge_attr is stored once per ordering;
 *   full CSR of A (self loops included, a_ij values) for the residual SpMV;
 *   node flags: bit0 = Dirichlet, bit1 = Neumann.
 * ------------------------------------------------------------------------------------------ */
typedef struct psignn_plan psignn_plan_t;

/* tags: (N, tags_cols) float32 with tags_cols = 1 (dirichlet: tags[:,0]==1 -> Dirichlet) or
 * 3 (mixed one-hot [interior, dirichlet, neumann]).  d_a_ij may be NULL (no residual SpMV).
 * d_pos: (N,2) node coordinates (batch.pos) or NULL; used only to renumber nodes into spatially
 * compact tiles (speed, never results).  tile_target: nodes per tile (0 = default, < 0 = no tiles).
 * Synchronous: returns after the plan is complete (a few small device->host reads). */
int psignn_plan_create(psignn_plan_t** out, int64_t n_nodes, int64_t n_edges,
                       const int64_t* d_edge_index /* (2, E) */, const float* d_edge_attr /* (E,3) */,
                       const float* d_a_ij /* (E) or NULL */, const float* d_tags, int tags_cols,
                       const float* d_pos, int tile_target, void* stream);
void psignn_plan_destroy(psignn_plan_t* plan);

/* Tile structures: when present, solver state is kept in "plan order"
 * (nodes renumbered tile by tile).  psignn_plan_permute moves (N, cols) float rows between the
 * caller's numbering and plan order: to_plan = 1: dst[new] = src[perm[new]]; 0: dst[old] = src[inv[old]].
 * On an untiled plan it is a copy. */
int psignn_plan_is_tiled(const psignn_plan_t* plan);
int64_t psignn_plan_num_tiles(const psignn_plan_t* plan);
int64_t psignn_plan_ell_rows(const psignn_plan_t* plan);      /* 64-lane ELL slot-rows (padding included) */
int psignn_plan_max_tile_rows(const psignn_plan_t* plan);     /* max tile + halo rows staged in LDS */
int psignn_plan_permute(const psignn_plan_t* plan, const float* d_src, int cols, float* d_dst, int to_plan,
                        void* stream);

int64_t psignn_plan_num_nodes(const psignn_plan_t* plan);
int64_t psignn_plan_num_edges(const psignn_plan_t* plan);          /* E, as given */
int64_t psignn_plan_num_nonself_edges(const psignn_plan_t* plan);  /* E' */

/* Copy one plan array to host memory (tests: bit-exact comparison with the numpy oracle).
 * which: 0 csr_ptr(N+1 i32) 1 csr_nbr(E' i32) 2 csr_eid(E' i32) 3 csc_ptr 4 csc_nbr 5 csc_eid
 *        6 node_flags (N u8) 7 csr_attr (E'*3 f32) 8 csc_attr 9 a_ptr (N+1 i32) 10 a_col (E i32)
 *        11 a_val (E f32); tiled plans only: 12 perm (N i32, perm[new] = old) 13 tile_ptr (T+1 i32)
 *        14 halo_cnt (T i32) 15 halo (T*512 i32) 16 slice_off (S+1 i32) 17 slice_deg (S u8)
 *        18 ell (rows*64*4 u32: pair-merged neighbour slots) 20 tile_slice (T+1 i32).  Synchronous. */
int psignn_plan_export(const psignn_plan_t* plan, int which, void* h_dst, size_t dst_bytes);

/* ------------------------------------------------------------------------------------------
 * Weights: one flat float32 device buffer in the order documented in DESIGN.md §weights
 * (nn.Linear (out,in) row-major blocks, concatenated).  psignn_weights_size() gives its length.
 * replaces: the nn.Module parameter tensors of Function / Phi_to / Phi_from / MLP
 *           (dirichlet/psignn/model.py:265-277,316-368; mixed/psignn/model.py:198-214).
 * ------------------------------------------------------------------------------------------ */
int64_t psignn_weights_size(int mixed, int n_layers);

/* ------------------------------------------------------------------------------------------
 * f_theta: one application of the GNN block.
 * replaces: Function.forward (dirichlet/psignn/model.py:279-300 == tests/model_psignn.py:269-290;
 *           mixed/psignn/model.py:216-245) including both/all three MessagePassing.propagate
 *           passes, the gated update MLP, LayerNorm and the Dirichlet/Neumann row handling.
 * d_prb: (N,2) dirichlet / (N,3) mixed.  d_normals: (N,2) mixed only, else NULL.
 * d_work: scratch of psignn_f_workspace_floats(plan) floats.  d_out must not alias d_h.
 * ------------------------------------------------------------------------------------------ */
int64_t psignn_f_workspace_floats(const psignn_plan_t* plan);
int psignn_f_forward(const psignn_plan_t* plan, const float* d_weights, int n_layers,
                     const float* d_h, const float* d_h_initial, const float* d_prb,
                     const float* d_normals, float* d_out, float* d_work, void* stream);

/* Same, with h, h_initial, prb, normals and out already in plan order (no permutation passes):
 * the form iterative callers use after one psignn_plan_permute per tensor. */
int psignn_f_forward_p(const psignn_plan_t* plan, const float* d_weights, int n_layers,
                       const float* d_h, const float* d_h_initial, const float* d_prb,
                       const float* d_normals, float* d_out, float* d_work, void* stream);

/* n successive applications x <- f(x) in plan order without host involvement between them.
 * replaces: the loop body of forward_iteration (utilities/solver.py:301-341) minus its per-step norms.
 * d_x: x_0 on entry, x_n on return; d_tmp: a second (N, d) buffer. */
int psignn_picard_p(const psignn_plan_t* plan, const float* d_weights, int n_layers, float* d_x, float* d_tmp,
                    const float* d_h_initial, const float* d_prb, const float* d_normals, float* d_work, int n,
                    void* stream);

/* Single message-passing aggregation (tests / diagnostics): which = 0 Phi_to, 1 Phi_from,
 * 2 Phi_neumann (mixed).  replaces: Phi_to.forward / Phi_from.forward (model.py:334-368). */
int psignn_phi(const psignn_plan_t* plan, const float* d_weights, int n_layers, int layer, int which,
               const float* d_h, float* d_out, float* d_work, void* stream);

/* Jacobian-vector product of f at h along v (analytic, no finite differences).
 * replaces: nothing executable in the reference (scipy.optimize.newton_krylov is imported at
 *           utilities/solver.py:6 but never called); it is the transpose of the VJP that
 *           autograd.grad(new_H, H, v) computes at model.py:214,432,449. */
int psignn_f_jvp(const psignn_plan_t* plan, const float* d_weights, int n_layers,
                 const float* d_h, const float* d_prb, const float* d_normals,
                 const float* d_v, float* d_out, float* d_work, void* stream);

/* Linearisation of f at a fixed state h, for Krylov solvers that apply J_f(h) to many vectors (Newton-Krylov, BASELINE config 5):
 * psignn_lin_build evaluates the value path of f once and stores what the Jacobian needs (relu masks of every edge direction as
 * wave-level bit masks, per-node gate / update / LayerNorm quantities); psignn_lin_jvp then applies J_f(h) as a linear operator,
 * at about half the cost of psignn_f_jvp.  Tiled plans; dirichlet: single-layer blocks; mixed (d_normals_plan required; the tiles
 * holding Neumann nodes run the direct kernel at a copy of the state kept by the build).  h, prb, normals, v, out in PLAN order
 * (psignn_plan_permute).  The handle keeps a pointer to the plan: destroy it before the plan.
 * replaces: nothing executable in the reference (see psignn_f_jvp); same product as psignn_f_jvp up to fp32 summation order. */
typedef struct psignn_lin psignn_lin_t;
int psignn_lin_create(psignn_lin_t** out, const psignn_plan_t* plan);
void psignn_lin_destroy(psignn_lin_t* lin);
size_t psignn_lin_bytes(const psignn_lin_t* lin);
int psignn_lin_build(psignn_lin_t* lin, const float* d_weights, int n_layers, const float* d_h_plan, const float* d_prb_plan,
                     const float* d_normals_plan, void* stream);
int psignn_lin_jvp(const psignn_lin_t* lin, const float* d_weights, int n_layers, const float* d_v_plan, float* d_out_plan,
                   void* stream);

/* Vector-Jacobian product out = w^T (df/dh) at h (both families), as two gather passes over the plan's
 * CSR/CSC lists (no atomics).  d_normals: (N,2) for mixed plans, else NULL.
 * replaces: torch.autograd.grad(new_H_star, H_star, y) inside the implicit backward hook
 *           (dirichlet/psignn/model.py:210-223), jac_loss_estimate (:416-435) and power_method (:437-452). */
int psignn_f_vjp(const psignn_plan_t* plan, const float* d_weights, int n_layers, const float* d_h,
                 const float* d_prb, const float* d_normals, const float* d_w, float* d_out, float* d_work,
                 void* stream);

/* JVP with h, prb, normals, v and out in plan order (the tiled LDS-staged kernel: single-layer dirichlet plans, and
 * mixed plans of any depth, whose iterated layer is the last one).  d_normals: (N,2) in plan order for mixed plans, else NULL. */
int psignn_f_jvp_p(const psignn_plan_t* plan, const float* d_weights, int n_layers, const float* d_h, const float* d_prb,
                   const float* d_normals, const float* d_v, float* d_out, void* stream);

/* Same with h, prb, w and out in plan order (tiled kernels where the plan has tiles; the form the adjoint solve uses). */
int psignn_f_vjp_p(const psignn_plan_t* plan, const float* d_weights, int n_layers, const float* d_h,
                   const float* d_prb, const float* d_normals, const float* d_w, float* d_out, float* d_work,
                   void* stream);

/* Parameter gradient  w^T (d f / d theta)  at h (plan order; tiled single-layer dirichlet plans), together with
 * w^T (d f / d h) -> d_out_h.  Replaces loss.backward() through new_H = f(H*, H_init, batch) with the hooked
 * cotangent (dirichlet/psignn/model.py:203-225, training_class.py:150-163).  d_grad receives
 * psignn_param_grad_size(mixed, n_layers) floats laid out like the leading (un-derived) section of the packed
 * weights: shared{ln_gamma, ln_beta, alpha_w, alpha_b} | phi_to{W1,b1,W2,b2} | phi_from | update{U1,c1,U2,c2};
 * derived (fold) slots stay zero.  d_work: psignn_f_param_vjp_workspace_floats(plan) floats. */
int64_t psignn_param_grad_size(int mixed, int n_layers);
int64_t psignn_f_param_vjp_workspace_floats(const psignn_plan_t* plan);
/* the same in the caller's numbering and for every single-layer plan: tiled plans of both families run the tile kernels in
 * record mode (mixed family: d_normals required; d_grad then also covers phi_neumann{W1,b1,W2,b2} |
 * update_neumann{N1,nb1,N2,nb2}; mixed/psignn/model.py:216-245 under loss.backward()), untiled plans the global-gather kernels. */
int psignn_f_param_vjp(const psignn_plan_t* plan, const float* d_weights, int n_layers, const float* d_h,
                       const float* d_prb, const float* d_normals, const float* d_w, float* d_grad, float* d_out_h,
                       float* d_work, void* stream);
int psignn_f_param_vjp_p(const psignn_plan_t* plan, const float* d_weights, int n_layers, const float* d_h,
                         const float* d_prb, const float* d_w, float* d_grad, float* d_out_h, float* d_work,
                         void* stream);

/* Backward of the VJP: gradient of  phi = gbar . (J_f(h)^T v) = v^T J_f(h) gbar  (gbar held constant) w.r.t. the
 * parameters (d_grad, layout above) and w.r.t. h (d_grad_h, (N,10)).  Replaces autograd's double backward through
 * jac_loss_estimate -- autograd.grad(f0, z0, v, create_graph=True), dirichlet/psignn/model.py:416-435 -- when the
 * Jacobian regulariser is part of the loss (jac_weight, training_class.py:156-159): with g = J^T v and
 * jac_loss = |g|^2 / (N d), pass gbar = (d loss / d jac_loss) * 2 g / (N d).  Caller's numbering; single-layer
 * blocks of both families (mixed: d_normals required, d_grad also covers phi_neumann | update_neumann).  d_work: psignn_f_vjp_backward_workspace_floats(plan) floats. */
int64_t psignn_f_vjp_backward_workspace_floats(const psignn_plan_t* plan);
int psignn_f_vjp_backward(const psignn_plan_t* plan, const float* d_weights, int n_layers, const float* d_h,
                          const float* d_prb, const float* d_normals, const float* d_v, const float* d_gbar, float* d_grad,
                          float* d_grad_h, float* d_work, void* stream);

/* Backward of psignn_mlp2 and of psignn_residual: what autograd runs for the autoencoder and residual terms of
 * the training loss (dirichlet/psignn/model.py:58-99).  d_gflat = gradients of [W1 | b1 | W2 | b2];
 * d_gx (n, din) may be NULL.  psignn_residual_t: out = A^T r with the caller's a_ij (E floats, edge_index order). */
int64_t psignn_mlp2_backward_workspace_floats(int64_t n);
int psignn_mlp2_backward(const float* d_x, const float* d_gy, int64_t n, int din, int hid, int dout, const float* d_w1,
                         const float* d_b1, const float* d_w2, float* d_gx, float* d_gflat, float* d_work, void* stream);
int psignn_residual_t(const psignn_plan_t* plan, const float* d_a_ij, const float* d_r, float* d_out, void* stream);

/* DS-GPS, the unrolled recurrent baseline (dirichlet/dsgps/model.py:130-163, mixed/dsgps/model.py:50-95): k updates
 *   h <- h + sigmoid(Wz c + bz) * tanh(Wc [sigmoid(Wr c + br) * h | mess_to | mess_from | prb] + bc),
 *   c = [h | Phi_to(h) | Phi_from(h) | prb], Dirichlet rows <- d_h0 rows; mixed plans: Neumann rows <-
 *   update_neumann([h | Phi_neumann(h) | prb | normal]) (d_normals required),
 * starting from d_h0 (the encoder state).  Tiled plans.  d_weights: psignn_dsgps_weights_size(mixed) floats
 * (layout in csrc/dsgps_tile.hip, built by engine.pack_dsgps); d_work: 4 * N * 10 floats.
 * psignn_dsgps_step_p: a single update with all node tensors in plan order. */
int64_t psignn_dsgps_weights_size(int mixed);
int psignn_dsgps_forward(const psignn_plan_t* plan, const float* d_weights, int k, const float* d_h0, const float* d_prb,
                         const float* d_normals, float* d_out, float* d_work, void* stream);
int psignn_dsgps_step_p(const psignn_plan_t* plan, const float* d_weights, const float* d_h, const float* d_h0,
                        const float* d_prb, const float* d_normals, float* d_out, void* stream);
/* Backward of one DS-GPS update h' = step(h) (both families, caller's numbering): what loss.backward() runs per unrolled
 * update of ModelDSGPS.forward (dirichlet/dsgps/model.py:72-89, mixed/dsgps/model.py:72-93; training_class.py of those
 * directories).  d_w: cotangent on h'.  d_out_h = w^T dh'/dh (the Dirichlet rows' share goes to H_0: it is d_w on those
 * rows); d_grad: psignn_dsgps_grad_size(mixed) floats = the f_theta gradient layout (phi_to / phi_from slots, mixed: also
 * phi_neumann | update_neumann; psignn_param_grad_size(mixed, 1)) followed by [Wz (10 x (30+P)) | bz | Wr | br | Wc | bc].
 * d_phi_weights: those modules in the f_theta weight layout (psignn_weights_size(mixed, 1) floats; only their blocks are
 * read); d_gate_weights: [Wz | bz | Wr | br | Wc | bc] as nn.Linear stores them.  d_normals: mixed plans.
 * d_work: psignn_dsgps_step_backward_workspace_floats(plan). */
int64_t psignn_dsgps_grad_size(int mixed);
int64_t psignn_dsgps_step_backward_workspace_floats(const psignn_plan_t* plan);
int psignn_dsgps_step_backward(const psignn_plan_t* plan, const float* d_phi_weights, const float* d_gate_weights,
                               const float* d_h, const float* d_prb, const float* d_normals, const float* d_w, float* d_grad,
                               float* d_out_h, float* d_work, void* stream);

/* DSS, the Deep Statistical Solver baseline (dirichlet/dss/model.py:97-120): k updates with per-step weights
 *   h <- h + alpha * Psi_t([h | Phi_to_t(h) | Phi_from_t(h) | b'_norm]),  H_0 = 0,
 * on a tiled plan created with edge_attr = (0, 0, a_ij_norm) (the scalar edge feature of DSS).  d_weights:
 * psignn_dss_weights_size(k) floats (layout in csrc/dss_tile.hip, built by engine.pack_dss); d_work: N * 23 floats. */
int64_t psignn_dss_weights_size(int k);
int psignn_dss_forward(const psignn_plan_t* plan, const float* d_weights, int k, float alpha, const float* d_bprime_norm,
                       float* d_out, float* d_work, void* stream);
/* update t alone, state and b'_norm in plan order (the per-step loop of DeepStatisticalSolver.forward, model.py:77-93) */
int psignn_dss_step_p(const psignn_plan_t* plan, const float* d_weights, int t, float alpha, const float* d_h,
                      const float* d_bprime_norm_p, float* d_out, void* stream);
/* Backward of one DSS update h' = h + alpha Psi_t([h | Phi_to_t(h) | Phi_from_t(h) | b'_norm]) (caller's numbering): what
 * loss.backward() runs per update of DeepStatisticalSolver.forward (dirichlet/dss/model.py:75-83).  d_weights_t: update t's
 * modules in the f_theta weight layout with three node inputs (psignn_weights_size(1, 1)-style layout without Neumann blocks:
 * phi W1 padded to (10, 23) with the edge-feature weight in column 22, Psi in the update slots; engine.pack_dss_train builds
 * it).  d_grad: psignn_dss_grad_size() floats in that layout; d_out_h = w^T dh'/dh.
 * d_work: psignn_dss_step_backward_workspace_floats(plan). */
int64_t psignn_dss_grad_size(void);
int64_t psignn_dss_step_backward_workspace_floats(const psignn_plan_t* plan);
int psignn_dss_step_backward(const psignn_plan_t* plan, const float* d_weights_t, float alpha, const float* d_h,
                             const float* d_bprime_norm, const float* d_w, float* d_grad, float* d_out_h, float* d_work,
                             void* stream);

/* ------------------------------------------------------------------------------------------
 * Small dense pieces around the solve.
 * replaces: Encoder / Decoder MLPs (model.py:370-392), residual_loss SpMV (model.py:157-167).
 * ------------------------------------------------------------------------------------------ */
/* out (N, dout) = W2 relu(W1 x + b1) + b2 with W1 (hid,din), W2 (dout,hid); din,hid,dout <= 16 */
int psignn_mlp2(const float* d_x, int64_t n, int din, int hid, int dout,
                const float* d_w1, const float* d_b1, const float* d_w2, const float* d_b2,
                float* d_out, void* stream);
/* d_out (N) = A u - y  (A incl. diagonal). */
int psignn_residual(const psignn_plan_t* plan, const float* d_u, const float* d_y, float* d_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * On-device Broyden root-find of g(x) = f(x) - x.
 * replaces: broyden(f, x0, threshold, eps, stop_mode="rel", ls=False)
 *           (dirichlet/psignn/utilities/solver.py:116-207) with matvec/rmatvec (:96-114) and the
 *           two .item() syncs per iteration (:162-163).
 * The inverse-Jacobian factors are stored U:(thr, N*d), V:(thr, N*d) row-contiguous.
 * ------------------------------------------------------------------------------------------ */
typedef struct psignn_broyden psignn_broyden_t;

typedef struct {
  int32_t nstep;        /* index of the lowest-residual iterate ("nstep" of the reference dict) */
  int32_t n_iter;       /* iterations executed (= len(xest_trace) - 1) */
  int32_t prot_break;
  int32_t stop_reason;  /* 0 threshold, 1 rel<eps, 2 plateau, 3 protective break */
  double lowest;        /* lowest rel residual */
  double lowest_abs;
} psignn_solve_info_t;

/* keep_trace != 0 stores every iterate ((thr+1) * N * d floats) for xest_trace. */
int psignn_broyden_create(psignn_broyden_t** out, const psignn_plan_t* plan, int threshold, int keep_trace);
void psignn_broyden_destroy(psignn_broyden_t* s);
/* stop_mode of the following solves: 0 = "rel" (default; all reference call sites), 1 = "abs" (utilities/solver.py:116,140,174):
 * objective, lowest-iterate tracking and the protective-break factor (1e6 instead of 1e3) follow it. */
int psignn_broyden_set_stop_mode(psignn_broyden_t* solver, int abs_mode);
size_t psignn_broyden_bytes(const psignn_broyden_t* s);

/* Solve; synchronous at the end (the result must be complete when it returns).  The host polls the
 * device-side status every `poll_every` iterations (<=0: default 8).
 * d_result: (N, d) lowest-residual iterate.  h_rel_trace / h_abs_trace: host arrays of
 * `threshold` doubles (may be NULL). */
int psignn_broyden_solve(psignn_broyden_t* s, const float* d_weights, int n_layers,
                         const float* d_h_initial, const float* d_prb, const float* d_normals,
                         double eps, int poll_every, float* d_result, psignn_solve_info_t* h_info,
                         double* h_rel_trace, double* h_abs_trace, void* stream);
/* Batched solve of n independent meshes (one GPU's share of a batch: BASELINE configs[3], 8 x 50k-node meshes): the meshes
 * iterate in lockstep and every per-iteration pass (fused f step, dots, reduce + stop tests, axpy, final) is ONE launch over
 * all of them; each mesh keeps its own status block, traces and stop test, and its result is bit-identical to
 * psignn_broyden_solve with the same solver on that mesh alone.  solvers[i] was created from mesh i's plan (tiled plans of ONE
 * family -- all dirichlet or all mixed, d_normals then holds the unit normals, else NULL --, one threshold and vector-size class
 * for the shard: psignn_broyden_batchable; otherwise PSIGNN_EINVAL).
 * replaces: the reference's one-union-Batch-per-device DataParallel call (dirichlet/psignn/main.py:106, mixed/psignn/main.py:106,
 *           dirichlet/psignn/test/test_func.py:68-120) for independent per-mesh solves.
 * Arrays of n device / host pointers; h_rel_trace[i] / h_abs_trace[i]: `threshold` doubles each (arrays may be NULL). */
/* Solver for mesh `plan` that will be used inside psignn_broyden_solve_batch together with others: shard_elems = sum of
 * N * d over the shard.  Its reduction shapes (vector width, split of the sweeps over the stored pairs) are sized for the
 * shard, not for the single mesh; psignn_broyden_solve on such a solver gives the same bits as the batched solve. */
int psignn_broyden_create_for_batch(psignn_broyden_t** out, const psignn_plan_t* plan, int threshold, int keep_trace,
                                    int64_t shard_elems);
int psignn_broyden_solve_batch(int n, psignn_broyden_t** solvers, const float* d_weights, int n_layers,
                               const float* const* d_h_initial, const float* const* d_prb, const float* const* d_normals,
                               double eps, int poll_every, float* const* d_results, psignn_solve_info_t* h_infos,
                               double* const* h_rel_trace, double* const* h_abs_trace, void* stream);
/* 1 when psignn_broyden_solve_batch takes these solvers together (tiled plans of one boundary-condition family, one size class),
 * else 0 -- asked on the host before a shard is handed over, so that "not batchable" is a decision and not an error code.
 * replaces: nothing in the reference (its DataParallel call takes any list of graphs as one union batch,
 *           dirichlet/psignn/main.py:106; mixed/psignn/main.py:106). */
int psignn_broyden_batchable(int n, psignn_broyden_t* const* solvers);
/* Adjoint fixed point y = J_f(h*)^T y + grad with the same Broyden machinery, the VJP kernel as the map, y_0 = 0.
 * replaces: the backward hook of DeepEquilibrium.forward (dirichlet/psignn/model.py:210-223), i.e.
 *           solver(lambda y: autograd.grad(new_H, H, y) + grad, zeros, bw_thres, bw_tol).
 * All tensors in the caller's numbering. */
int psignn_broyden_solve_adjoint(psignn_broyden_t* s, const float* d_weights, int n_layers, const float* d_h_star,
                                 const float* d_prb, const float* d_normals, const float* d_grad, double eps,
                                 int poll_every, float* d_result, psignn_solve_info_t* h_info,
                                 double* h_rel_trace, double* h_abs_trace, void* stream);
/* Copy iterate i (0..n_iter) of the last solve to d_dst (needs keep_trace). */
int psignn_broyden_get_iterate(const psignn_broyden_t* s, int i, float* d_dst, void* stream);
/* Copy stored rank-one pair j (0 .. pairs stored - 1) of the last solve to d_dst: which = 0 -> U_j, 1 -> V_j; which = 2 -> the current
 * `update` vector (solver.py:136,192; j ignored).  Caller's numbering.
 * replaces: reading Us[..., j] / VTs[:, j] of broyden() (utilities/solver.py:134-135, written at :190-191); the reference keeps them
 *           as locals of the solver call -- here they stay on the device between solves.  Used by the parity tests to check
 *           the Broyden recurrences of every update form on the device's own state. */
int psignn_broyden_get_pair(const psignn_broyden_t* s, int j, int which, float* d_dst, void* stream);

/* Generic low-rank machinery for user-supplied f (Python callables): the solver-API drop-in
 * `broyden(f, x0, threshold, eps)` drives these from the host with one f call per iteration.
 *   step_begin: x_new = x + update                      -> returns pointer-free: writes d_x_new
 *   step_end:   given fx = f(x_new): residual norms, stop test, rank-1 update, next update. */
int psignn_broyden_ext_begin(psignn_broyden_t* s, const float* d_x0, const float* d_fx0, void* stream);
int psignn_broyden_ext_next_x(psignn_broyden_t* s, float* d_x_new, void* stream);
/* Line search of `broyden(..., ls=True)` (line_search / scalar_search_armijo, utilities/solver.py:20-94): the host
 * evaluates phi(s) = |g(x + s * update)|^2 at trial points (ext_trial_x writes x + s * update, no state change) and
 * commits the accepted step length with ext_scale_step (update <- s * update) before ext_next_x / ext_update. */
int psignn_broyden_ext_trial_x(psignn_broyden_t* s, double step, float* d_x_trial, void* stream);
int psignn_broyden_ext_scale_step(psignn_broyden_t* s, double step, void* stream);
/* returns 1 in *h_done when the stop test fired (synchronous read of the status). */
int psignn_broyden_ext_update(psignn_broyden_t* s, const float* d_fx_new, double eps, int* h_done, void* stream);
int psignn_broyden_ext_finish(psignn_broyden_t* s, float* d_result, psignn_solve_info_t* h_info,
                              double* h_rel_trace, double* h_abs_trace, void* stream);
/* Same, for a problem that is not tied to a mesh plan (any vector length). */
int psignn_broyden_create_n(psignn_broyden_t** out, int64_t n_elems, int seq_len, int threshold, int keep_trace);

/* ------------------------------------------------------------------------------------------
 * Picard iteration and Anderson acceleration: the vector work, norms, stop tests and the small bordered solve on the
 * device; the caller evaluates f between the calls (the HIP GNN block, or any function of device tensors).
 * replaces: forward_iteration (dirichlet/psignn/utilities/solver.py:301-341) and anderson (:215-293, m = 2, lam = 1e-4,
 *           beta = 1 at every call site) -- their torch.linalg.norm / .item() per iteration, bmm, linalg.solve and the
 *           alpha @ F mixing.
 * n_elems = N * d.  m: history length of Anderson (1..8; Picard ignores it).  threshold: the reference's `threshold`.
 * keep_trace != 0 stores every iterate (threshold + 2 vectors) for xest_trace.  Device status block: after the stop test
 * has fired every later call is a no-op, so a caller may run a few iterations ahead of psignn_fpiter_poll.
 * ------------------------------------------------------------------------------------------ */
typedef struct psignn_fpiter psignn_fpiter_t;
int psignn_fpiter_create(psignn_fpiter_t** out, int64_t n_elems, int m, int threshold, int keep_trace);
void psignn_fpiter_destroy(psignn_fpiter_t* s);
size_t psignn_fpiter_bytes(const psignn_fpiter_t* s);
int psignn_fpiter_poll(psignn_fpiter_t* s, int* h_done, void* stream);     /* synchronous read of the done flag */
/* Picard: begin(z0); then repeat { current_x -> x ; fx = f(x) ; update(fx) }: abs = |x - fx|, rel = abs / |fx| appended
 * to the traces, stop when rel <= eps or after threshold + 1 evaluations, the current iterate becomes fx.
 * h_done may be NULL (no host read). */
int psignn_picard_begin(psignn_fpiter_t* s, const float* d_x0, void* stream);
int psignn_picard_current_x(psignn_fpiter_t* s, float* d_x, void* stream);
int psignn_picard_update(psignn_fpiter_t* s, const float* d_fx, double eps, int* h_done, void* stream);
/* Anderson: begin(x0, f(x0), f(f(x0))); then for k = 2 .. threshold - 1 { next_x -> x_k ; fx = f(x_k) ; update(fx) }.
 * next_x: Gram matrix of the residual history, alpha from the bordered system [[0, 1^T], [1, G G^T + lam I]], x_k =
 * beta sum alpha_i F_i + (1 - beta) sum alpha_i X_i.  update: rel = |fx - x_k| / (1e-5 + |fx|), traces, lowest iterate
 * by stop_mode (stop_abs: 0 "rel", 1 "abs"), stop when the objective < eps. */
int psignn_anderson_begin(psignn_fpiter_t* s, const float* d_x0, const float* d_f0, const float* d_f1, double lam, double beta,
                          int stop_abs, void* stream);
int psignn_anderson_next_x(psignn_fpiter_t* s, float* d_x_new, void* stream);
int psignn_anderson_update(psignn_fpiter_t* s, const float* d_fx_new, double eps, int* h_done, void* stream);
/* Result (Picard: the last iterate, nstep = ite, lowest = last rel; Anderson: the lowest iterate, nstep = its loop index),
 * traces (n_iter entries each; arrays of threshold + 2 doubles) and, for Anderson, per loop iteration the loop index of the
 * lowest iterate so far (h_low_idx, threshold + 2 int32, may be NULL).  Synchronous. */
int psignn_fpiter_finish(psignn_fpiter_t* s, float* d_result, psignn_solve_info_t* h_info, double* h_rel_trace,
                         double* h_abs_trace, int32_t* h_low_idx, void* stream);
/* Iterate i of the last run (keep_trace): Picard z_i; Anderson the trial point of loop index i (i >= 2). */
int psignn_fpiter_get_iterate(const psignn_fpiter_t* s, int i, float* d_dst, void* stream);

/* ------------------------------------------------------------------------------------------
 * GMRES on the device for the Newton-Krylov solver (BASELINE configs[4]: "Newton-Krylov JVP path").
 * replaces: nothing executable -- scipy.optimize.newton_krylov is imported at utilities/solver.py:6 and never called; with
 *           fp32 finite-difference products scipy's solver does not converge on this problem, the
 *           analytic JVP kernel (psignn_f_jvp_p) is the operator here.
 * The caller owns the basis d_basis: (m_max + 1) rows of `ld` floats (ld >= n_elems, a multiple of 4).  Per Arnoldi step j
 * it writes the raw operator product of basis row j into row j + 1 and calls psignn_gmres_step, which orthogonalises it
 * (classical Gram-Schmidt twice), updates the Hessenberg least-squares problem by Givens rotations and raises the device
 * stop flag once |residual| <= eta |b|.  Krylov operator: A v = product - shift * v (shift = 1 for A = J_f - I).
 * ------------------------------------------------------------------------------------------ */
typedef struct psignn_gmres psignn_gmres_t;
int psignn_gmres_create(psignn_gmres_t** out, int64_t n_elems, int64_t ld, int m_max, float* d_basis);
void psignn_gmres_destroy(psignn_gmres_t* s);
/* g = fx - x -> d_g (may be NULL), -g -> d_neg_g (may be NULL); h_norms[0] = |g|, h_norms[1] = |fx|.  Synchronous. */
int psignn_residual_norms(psignn_gmres_t* s, const float* d_x, const float* d_fx, float* d_g, float* d_neg_g, double* h_norms,
                          void* stream);
int psignn_gmres_begin(psignn_gmres_t* s, const float* d_b, void* stream);               /* basis row 0 = b / |b| */
int psignn_gmres_step(psignn_gmres_t* s, int j, double shift, double eta, int* h_done /* may be NULL */, void* stream);
/* d_dst = d_base + scale * z, z = the least-squares solution over the first k steps (k <= 0: all completed; d_base may be
 * NULL).  h_info (may be NULL): [steps completed, |b|, |residual|].  d_dst may alias d_base. */
int psignn_gmres_solution(psignn_gmres_t* s, int k, const float* d_base, double scale, float* d_dst, double* h_info, void* stream);
int psignn_gmres_history(psignn_gmres_t* s, double* h_res /* m_max + 1 */, void* stream);
/* Steps of the current solve whose second Gram-Schmidt pass ran (the pass is conditional: Daniel-Gragg-Kaufman-Stewart test on
 * the device, |w'|^2 < 1/2 |w|^2; PSIGNN_GMRES_REORTH=always makes it unconditional). */
int psignn_gmres_reorth_count(psignn_gmres_t* s, int* h_count, void* stream);

/* ------------------------------------------------------------------------------------------
 * Per-kernel timing with HIP events on the launch stream (used by bench.py for the roofline line;
 * replaces: the reference's only instrumentation, time.time() around the model call,
 * tests/special_geo/spec_geo_2.py:313-317).  Off by default.
 * ------------------------------------------------------------------------------------------ */
void psignn_prof_enable(int on);
int psignn_prof_collect(void);   /* sync + aggregate per kernel name; returns the number of names */
int psignn_prof_get(int i, char* name, int cap, int64_t* calls, double* total_ms);
/* The same plus the sum of the ALGORITHMIC bytes of those launches as the launch sites state them (stored pairs swept, kept
 * window, meshes of the shard: what actually ran) -- 0 for kernels that state none.  bench.py's roofline figures come from here. */
int psignn_prof_get2(int i, char* name, int cap, int64_t* calls, double* total_ms, int64_t* alg_bytes);
/* Launch i of the last collect, in launch order: kernel name, duration, stated algorithmic bytes.  name == NULL: returns the
 * number of launches.  (scripts/summarise_pmc.py lines rocprofv3's per-dispatch counters up with this log.) */
int psignn_prof_launch(int i, char* name, int cap, double* ms, int64_t* alg_bytes);
/* Diagnostics: d_buf = device array of n_tiles * 4 * 8 int64 (or NULL to switch off).  While set, every wave of the f tile
 * kernel stores shader-clock stamps at its phase boundaries (0 start, 1 stage 1 done, 2 past the barrier, 3 neighbour sums
 * done, 4 node update done, 5 stored) -- scripts/tile_phases.py turns them into a per-phase time budget. */
void psignn_prof_tile_stamps(void* d_buf);

#ifdef __cplusplus
}
#endif
#endif /* PSIGNN_HIP_H */

// Shared declarations for libpsignn_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include "../../include/psignn_hip.h"

#define D PSIGNN_D

void psignn_set_error(const char* fmt, ...);

// Run-time knobs (PSIGNN_* environment variables; A/B scaffolding and test selectors, never needed for a normal run) are read
// once and cached; psignn_reload_knobs() bumps the epoch so that the next use re-reads them (tests switch forms in-process).
extern int g_knob_epoch;
#define KNOB_INT(var, expr)                   \
  static int var##_epoch = -1;                \
  static int var = 0;                         \
  if (var##_epoch != g_knob_epoch) {          \
    var = (expr);                             \
    var##_epoch = g_knob_epoch;               \
  }

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      psignn_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return PSIGNN_EHIP;                                                               \
    }                                                                                   \
  } while (0)

#define ARG_CHECK(cond, msg)                                   \
  do {                                                         \
    if (!(cond)) {                                             \
      psignn_set_error("%s: %s", __func__, msg);               \
      return PSIGNN_EINVAL;                                    \
    }                                                          \
  } while (0)

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------------
// Weight pack layout (floats).  All blocks are nn.Linear (out,in) row-major.
//   shared : ln_gamma[10] ln_beta[10] alpha_w[30+P] alpha_b[1]            (padded to SHARED_SZ)
//   layer l: phi_to{W1[10x23] b1[10] W2[10x10] b2[10]}  phi_from{...}  update{U1[10x(30+P)] c1[10] U2[10x10] c2[10]}
//            fold{G_to[10x10] g_to[10] G_fr[10x10] g_fr[10] a_to[10] a_fr[10] ab_to ab_fr}          (FOLD_SZ 244)
//   mixed  : phi_neu{...350}  upd_neu{N1[10x25] n1[10] N2[10x10] n2[10]}  nfold{G_n[10x10] g_n[10]} (NFOLD_SZ 112)
// P = second_member_dim = 2 (dirichlet) / 3 (mixed).
// fold blocks are derived on the host (engine.pack_weights): the second Phi layer is linear, so
//   U1[:, mp_to cols] (W2_to S + deg b2_to) = G_to S + deg g_to   with G_to = U1_to W2_to, g_to = U1_to b2_to
// and likewise for the alpha gate (a_to = w_alpha,to W2_to, ab_to = w_alpha,to . b2_to) and the Neumann MLP.
// ---------------------------------------------------------------------------------------------
template <int P>
struct WLayout {
  static constexpr int CAT = 3 * D + P;       // 32 / 33
  static constexpr int EIN = 2 * D + 3;       // 23
  static constexpr int LN_G = 0, LN_B = 10, AL_W = 20, AL_B = 20 + CAT;
  static constexpr int SHARED_SZ = 64;
  static constexpr int PHI_SZ = D * EIN + D + D * D + D;  // 350
  static constexpr int PHI_W1 = 0, PHI_B1 = D * EIN, PHI_W2 = D * EIN + D, PHI_B2 = D * EIN + D + D * D;
  static constexpr int UPD_SZ = D * CAT + D + D * D + D;
  static constexpr int UPD_W1 = 0, UPD_B1 = D * CAT, UPD_W2 = D * CAT + D, UPD_B2 = D * CAT + D + D * D;
  static constexpr int FOLD_SZ = 244;
  static constexpr int F_GTO = 0, F_gTO = 100, F_GFR = 110, F_gFR = 210, F_ATO = 220, F_AFR = 230, F_ABTO = 240, F_ABFR = 241;
  static constexpr int LAYER_SZ = 2 * PHI_SZ + UPD_SZ + FOLD_SZ;
  static constexpr int L_TO = 0, L_FROM = PHI_SZ, L_UPD = 2 * PHI_SZ, L_FOLD = 2 * PHI_SZ + UPD_SZ;
  static constexpr int NFOLD_SZ = 112, NF_G = 0, NF_g = 100;
  static constexpr int NEU_CAT = 2 * D + P + 2;  // 25 (mixed only)
  static constexpr int NEU_SZ = D * NEU_CAT + D + D * D + D;
  static constexpr int NEU_W1 = 0, NEU_B1 = D * NEU_CAT, NEU_W2 = D * NEU_CAT + D, NEU_B2 = D * NEU_CAT + D + D * D;
  __host__ __device__ static constexpr int layer(int l) { return SHARED_SZ + l * LAYER_SZ; }
  __host__ __device__ static constexpr int phi_neu(int nl) { return SHARED_SZ + nl * LAYER_SZ; }
  __host__ __device__ static constexpr int upd_neu(int nl) { return SHARED_SZ + nl * LAYER_SZ + PHI_SZ; }
  __host__ __device__ static constexpr int nfold(int nl) { return SHARED_SZ + nl * LAYER_SZ + PHI_SZ + NEU_SZ; }
  __host__ __device__ static constexpr int base_total(int nl, bool mixed) {
    return SHARED_SZ + nl * LAYER_SZ + (mixed ? PHI_SZ + NEU_SZ + NFOLD_SZ : 0);
  }
  // ---- transposed section (tile kernel): every matrix again as [in k][out o], o fastest, so that the outputs
  // (o, o+1) of one input k are adjacent -> one SGPR pair feeds a v_pk_fma_f32.  FP32 peak on CDNA needs the
  // packed form; the scalar form issues at half the rate.  Derived on the host (engine.pack_weights).
  static constexpr int T_W1J_TO = 0, T_W1J_FR = 100, T_W1I_TO = 200, T_W1I_FR = 300, T_A_TO = 400, T_A_FR = 430,
                       T_B1_TO = 460, T_B1_FR = 470, T_U1H = 480, T_GTO = 580, T_GFR = 680, T_U1P = 780, T_HB = 810,
                       T_gTO = 820, T_gFR = 830, T_U2 = 840, T_C2 = 940, TPL_SZ = 950;
  static constexpr int N_W1J = 0, N_W1I = 100, N_A = 200, N_B1 = 230, N_N1H = 240, N_GN = 340, N_N1P = 440,
                       N_NB1 = 490, N_gN = 500, N_N2 = 510, N_NB2 = 610, TPN_SZ = 620;
  __host__ __device__ static constexpr int tp_layer(int nl, bool mixed, int l) { return base_total(nl, mixed) + l * TPL_SZ; }
  __host__ __device__ static constexpr int tp_neu(int nl) { return base_total(nl, true) + nl * TPL_SZ; }
  __host__ __device__ static constexpr int total(int nl, bool mixed) {
    return base_total(nl, mixed) + nl * TPL_SZ + (mixed ? TPN_SZ : 0);
  }
};

// Iteration-invariant pointers of a tiled plan, kept in DEVICE memory and handed to the tile kernels as one pointer.
// As separate kernel arguments they are all live from the kernel's first instruction; the f kernel needs ~60 SGPRs for
// weights in its hot phases, so the compiler parked those arguments in VGPR lanes (v_writelane / v_readlane: VALU issue
// slots).  Behind a pointer each one is a scalar load next to its use.
struct TileCtx {
  const int32_t *tile_ptr, *tile_slice, *halo, *halo_cnt, *slice_off;
  const uint8_t* slice_deg;
  const uint4* ell;
  const uint8_t* flags_p;
  // tiles are consecutive chunks of `tile_nodes` nodes (a multiple of 64; the last tile may be short): a tile's node range and
  // its first slice follow from its index -- no dependent scalar load in front of the kernel's first vector loads
  int32_t tile_nodes;
  int32_t n_nodes;
};

// ---------------------------------------------------------------------------------------------
// Mesh plan (device memory owned here).
// ---------------------------------------------------------------------------------------------
struct psignn_plan {
  int64_t N = 0, E = 0, Ep = 0;
  int mixed = 0;
  int32_t *csr_ptr = nullptr, *csr_nbr = nullptr, *csr_eid = nullptr;
  int32_t *csc_ptr = nullptr, *csc_nbr = nullptr, *csc_eid = nullptr;
  float *csr_attr = nullptr, *csc_attr = nullptr;  // (E',3)
  uint8_t* flags = nullptr;                        // (N)
  int32_t *a_ptr = nullptr, *a_col = nullptr;      // full CSR of A (self loops included)
  float* a_val = nullptr;
  int max_deg = 0;

  // ---- tile structures (tiles.hip); valid when tiled != 0 -------------------------------------
  // Nodes are renumbered so that a tile (<= TILE_MAX consecutive new ids) is spatially compact; a tile's
  // out-of-tile neighbours form its halo.  Solver state lives in the new ("plan") order.
  int tiled = 0;
  int64_t n_tiles = 0, n_slices = 0, ell_rows = 0;
  int max_rows = 0;                            // max over tiles of n_t + n_halo (LDS rows)
  int32_t *perm = nullptr, *inv = nullptr;     // perm[new] = old ; inv[old] = new
  int32_t* tile_ptr = nullptr;                 // (n_tiles+1) new-id ranges
  int32_t* tile_slice = nullptr;               // (n_tiles+1) first 64-lane slice of each tile
  int32_t *halo = nullptr, *halo_cnt = nullptr;  // (n_tiles, HALO_CAP) sorted new ids ; (n_tiles)
  int32_t* slice_off = nullptr;                // (n_slices+1) first ELL slot-row of each slice
  uint8_t* slice_deg = nullptr;                // (n_slices) slot-rows of the slice (max neighbour slots of its nodes)
  uint4* ell = nullptr;                        // (ell_rows, 64) pair-merged slots {row|IN|OUT, a0, a1, a2}, tiles.hip
  uint8_t* flags_p = nullptr;                  // node flags in plan order
  // mixed plans: tiles without Neumann nodes first, then the (few, boundary) tiles with Neumann nodes -- the f kernel
  // runs the first group without the Phi_neumann columns in LDS (80-byte rows, one more workgroup per CU)
  int32_t* tile_order = nullptr;               // (n_tiles) tile ids
  mutable int32_t* tile_order_cost = nullptr;  // experiment (PSIGNN_TILE_ORDER=cost, fgnn_tile.hip): per XCD run, costliest tiles first
  int64_t n_tiles_plain = 0;                   // tiles in the first group
  float cell_size = 0.f, xmin = 0.f, ymin = 0.f;
  int nx = 0, ny = 0;
  TileCtx* d_ctx = nullptr;                    // device copy of the tile pointers (one kernel argument instead of eight)
  TileCtx h_ctx{};                             // the same on the host (kernels that take the struct by value)
};


#define TILE_MAX 256      // nodes per tile = threads per block of the tile kernel
#define HALO_CAP 512      // halo entries stored per tile
#define ELL_EMPTY 0xFFFFu

int psignn_tiles_build(psignn_plan* p, const float* d_pos, int tile_target, hipStream_t st);
void psignn_tiles_free(psignn_plan* p);

// ---------------------------------------------------------------------------------------------
// Optional per-kernel timing with HIP events on the launch stream (bench.py's roofline numbers).
// Off by default: LAUNCH() is then a plain launch.
// ---------------------------------------------------------------------------------------------
extern int g_prof_on;
// the NEXT profiled launch's ALGORITHMIC bytes (every operand read once, every result written once: DESIGN section 4), stated at the
// launch site from what is actually launched (stored pairs swept, kept window, meshes of the shard) -- bench.py builds its roofline
// numbers from these records, not from a re-derivation of the solver's schedule.  Consumed (and reset to 0) by prof_begin.
extern int64_t g_prof_next_bytes;
#define PROF_BYTES(b)                           \
  do {                                          \
    if (g_prof_on) g_prof_next_bytes = (int64_t)(b); \
  } while (0)
void prof_begin(const char* name, hipStream_t st);
void prof_end(hipStream_t st);
#define LAUNCH(name, st, ...)            \
  do {                                   \
    if (g_prof_on) prof_begin(name, st); \
    __VA_ARGS__;                         \
    if (g_prof_on) prof_end(st);         \
  } while (0)

#define FLAG_DIRICHLET 1
#define FLAG_NEUMANN 2

// ---------------------------------------------------------------------------------------------
// Batched Broyden (solver.hip psignn_broyden_solve_batch): one descriptor per mesh of a shard, in device memory.  Every
// per-iteration kernel is launched ONCE for the whole shard with blockIdx.z = mesh; a block loads its mesh's descriptor
// and then runs exactly the code (same block -> element mapping, same partial-sum shapes) of the single-mesh kernels, so
// each mesh's result is bit-identical to its own solve.
// ---------------------------------------------------------------------------------------------
struct BatchDesc {
  int64_t M, ld;
  int32_t nblk, npart, nblk_ax, jgroups, thr, seq_len, keep_trace, n_tiles, tile_base;
  int32_t* st;                 // the mesh's Status block, int32 view
  float *U, *V, *xbuf, *g0, *g1, *upd, *part, *coef, *nrm_part, *jpart;
  double *rel_trace, *abs_trace;
  const struct TileCtx* ctx;
  const float *h0p, *prbp;
  float* part2;                // three-sweep update: block partials of vT.dg, vT.g
  int32_t nblk_u, npart_u;     // its blocks / per-wave partials per stored pair
  float* parta;                // folded sweep 3: per-wave partials of the next iteration's a, contiguous per stored pair
  int32_t nblk4, pad_;         // its blocks (4 floats per lane)
  const float* nrmp;           // mixed family: unit normals in plan order (NULL for dirichlet plans)
  int64_t pstride;             // plane stride of the dot partials (solver.hip: part = 3 planes of (blocks, ldp))
};

// PSIGNN_TILE_LDS_MIN = lower limit in bytes of the dynamic LDS request of the dirichlet f / JVP launches (<= 65536): caps the
// workgroups per CU (160 KB / request) without touching the code -- the occupancy sweeps of profiles/r3_f_model.txt.
static inline size_t tile_lds_min() {
  KNOB_INT(mn, [] { const char* e = getenv("PSIGNN_TILE_LDS_MIN"); return e ? atoi(e) : 0; }());
  return (size_t)(mn < 0 ? 0 : (mn > 65536 ? 65536 : mn));
}

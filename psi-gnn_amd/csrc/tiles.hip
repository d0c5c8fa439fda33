// Tile structures of the mesh plan (gfx950): spatial node renumbering, tiles with halos, sliced-ELL
// neighbour lists with LDS-local indices.  One-time, on the device; integer / index work only.
//
// Why: the GNN block gathers a projected 40-byte row per edge endpoint.  From global memory those gathers
// are uncoalesced (one cache line per lane) and the texture-address path, not HBM, bounds the kernel.
// With nodes renumbered so that <= 256 consecutive ids form a compact patch of the mesh, a workgroup
// stages the rows of its patch + halo in LDS once and every gather becomes a ds_read.
//
//   1. order the nodes along a Hilbert curve: fine square cells (~4 nodes each, a 2^k x 2^k grid over the bounding box),
//      cell key = Hilbert index, counting sort by key, node id ascending inside a cell
//   2. tiles = consecutive chunks of exactly tile_target (256) nodes of that order -- every workgroup of the tile kernels
//      has all its lanes busy (cells sized for "at most 256" left 12 % of the lanes idle: 226 nodes per tile on the
//      1M-node mesh) and a chunk of a Hilbert curve is a compact blob at any local mesh density; node ids are then
//      sorted ascending inside each tile; tiles -> 64-lane slices
//   3. per tile: halo = sorted distinct out-of-tile neighbours (both directions)
//   4. per slice: pair-merged ELL slot-rows x 64 lanes: one 16-byte slot per neighbour {LDS row, IN/OUT, attr}
//      (see 'pair-merged ELL' below), in the canonical neighbour order of the CSR/CSC plan
// If a structure limit is exceeded (cell > SORT_CAP nodes, halo > HALO_CAP, degree > 255) the plan
// stays untiled and the global-gather kernels (fgnn.hip) are used.
#include "common.h"
#include <math.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define SORT_CAP 4096
#define CAND_CAP 4096

// ---------------------------------------------------------------- bounding box
__device__ __forceinline__ uint32_t f2o(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static float o2f(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

__global__ void k_bbox(int64_t N, const float* __restrict__ pos, uint32_t* __restrict__ box /* xmin ymin xmax ymax */) {
  float xmn = INFINITY, ymn = INFINITY, xmx = -INFINITY, ymx = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    float x = pos[2 * i], y = pos[2 * i + 1];
    xmn = fminf(xmn, x); xmx = fmaxf(xmx, x);
    ymn = fminf(ymn, y); ymx = fmaxf(ymx, y);
  }
  for (int o = 32; o > 0; o >>= 1) {
    xmn = fminf(xmn, __shfl_xor(xmn, o)); xmx = fmaxf(xmx, __shfl_xor(xmx, o));
    ymn = fminf(ymn, __shfl_xor(ymn, o)); ymx = fmaxf(ymx, __shfl_xor(ymx, o));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&box[0], f2o(xmn)); atomicMin(&box[1], f2o(ymn));
    atomicMax(&box[2], f2o(xmx)); atomicMax(&box[3], f2o(ymx));
  }
}

// Ordering key of a node.  inv_cs > 0: Hilbert index of its cell (cx, cy) on a 2^k x 2^k grid (the classic xy -> d walk,
// k <= 12; nx = ny = 2^k).  inv_cs < 0: "snake strips" -- the domain is cut into ny horizontal strips of height 1 / g_inv_h and
// each strip into nx cells of width -1 / inv_cs (inv_h = 1 / strip height); strips are walked alternately left-to-right and right-to-left.
__device__ __forceinline__ int cell_of(float x, float y, float xmin, float ymin, float inv_cs, float inv_h, int nx, int ny) {
  if (inv_cs < 0.f) {
    int cx = (int)floorf((x - xmin) * -inv_cs);
    int cy = (int)floorf((y - ymin) * inv_h);
    cx = min(max(cx, 0), nx - 1);
    cy = min(max(cy, 0), ny - 1);
    return cy * nx + ((cy & 1) ? nx - 1 - cx : cx);
  }
  int cx = (int)floorf((x - xmin) * inv_cs);
  int cy = (int)floorf((y - ymin) * inv_cs);
  cx = min(max(cx, 0), nx - 1);
  cy = min(max(cy, 0), ny - 1);
  int d = 0;
  for (int s = nx >> 1; s > 0; s >>= 1) {
    const int rx = (cx & s) ? 1 : 0, ry = (cy & s) ? 1 : 0;
    d += s * s * ((3 * rx) ^ ry);
    if (!ry) {  // rotate the quadrant
      if (rx) {
        cx = nx - 1 - cx;
        cy = nx - 1 - cy;
      }
      const int t = cx;
      cx = cy;
      cy = t;
    }
  }
  return d;
}

__global__ void k_cell_count(int64_t N, const float* __restrict__ pos, float xmin, float ymin, float inv_cs, float inv_h, int nx,
                             int ny, int32_t* __restrict__ cnt) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  atomicAdd(&cnt[cell_of(pos[2 * i], pos[2 * i + 1], xmin, ymin, inv_cs, inv_h, nx, ny)], 1);
}
__global__ void k_count_occupied(int64_t n, const int32_t* __restrict__ cnt, int32_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int v = (i < n && cnt[i] > 0) ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v);
}
__global__ void k_count_max(int64_t n, const int32_t* __restrict__ cnt, int32_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int v = i < n ? cnt[i] : 0;
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  if ((threadIdx.x & 63) == 0 && v) atomicMax(out, v);
}
__global__ void k_cell_fill(int64_t N, const float* __restrict__ pos, float xmin, float ymin, float inv_cs, float inv_h, int nx,
                            int ny, const int32_t* __restrict__ cptr, int32_t* __restrict__ cur,
                            int32_t* __restrict__ list) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int c = cell_of(pos[2 * i], pos[2 * i + 1], xmin, ymin, inv_cs, inv_h, nx, ny);
  list[cptr[c] + atomicAdd(&cur[c], 1)] = (int32_t)i;
}
// One block per cell: rank sort of the (unique) node ids held by the cell.
__global__ __launch_bounds__(256) void k_cell_sort(const int32_t* __restrict__ cptr, int32_t* __restrict__ list,
                                                   int32_t* __restrict__ err) {
  __shared__ int32_t ids[SORT_CAP];
  int32_t s = cptr[blockIdx.x], n = cptr[blockIdx.x + 1] - s;
  if (n <= 1) return;
  if (n > SORT_CAP) {
    if (threadIdx.x == 0) atomicOr(err, 1);
    return;
  }
  for (int i = threadIdx.x; i < n; i += 256) ids[i] = list[s + i];
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    int32_t x = ids[i];
    int r = 0;
    for (int j = 0; j < n; ++j) r += ids[j] < x;
    list[s + r] = x;
  }
}

__global__ void k_inverse_perm(int64_t N, const int32_t* __restrict__ perm, int32_t* __restrict__ inv,
                               const uint8_t* __restrict__ flags, uint8_t* __restrict__ flags_p) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int32_t o = perm[i];
  inv[o] = (int32_t)i;
  flags_p[i] = flags[o];
}
__global__ void k_iota(int64_t N, int32_t* __restrict__ a) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) a[i] = (int32_t)i;
}

// ---------------------------------------------------------------- halo
// One block per tile: distinct out-of-tile neighbours (new ids), ascending.
__global__ __launch_bounds__(256) void k_halo(const int32_t* __restrict__ tile_ptr, const int32_t* __restrict__ perm,
                                              const int32_t* __restrict__ inv, const int32_t* __restrict__ csr_ptr,
                                              const int32_t* __restrict__ csr_nbr, const int32_t* __restrict__ csc_ptr,
                                              const int32_t* __restrict__ csc_nbr, int32_t* __restrict__ halo,
                                              int32_t* __restrict__ halo_cnt, int32_t* __restrict__ misc /* [0] err [1] max rows */) {
  __shared__ int32_t cand[CAND_CAP];
  __shared__ uint8_t keep[CAND_CAP];
  __shared__ int32_t n_cand, n_keep;
  const int tile = blockIdx.x;
  const int32_t t0 = tile_ptr[tile], t1 = tile_ptr[tile + 1];
  if (threadIdx.x == 0) { n_cand = 0; n_keep = 0; }
  __syncthreads();
  if ((int)threadIdx.x < t1 - t0) {
    int32_t old = perm[t0 + threadIdx.x];
    for (int pass = 0; pass < 2; ++pass) {
      const int32_t* ptr = pass ? csr_ptr : csc_ptr;
      const int32_t* nbr = pass ? csr_nbr : csc_nbr;
      for (int32_t e = ptr[old]; e < ptr[old + 1]; ++e) {
        int32_t nb = inv[nbr[e]];
        if (nb < t0 || nb >= t1) {
          int p = atomicAdd(&n_cand, 1);
          if (p < CAND_CAP) cand[p] = nb;
        }
      }
    }
  }
  __syncthreads();
  int n = n_cand;
  if (n > CAND_CAP) {
    if (threadIdx.x == 0) atomicOr(&misc[0], 2);
    n = CAND_CAP;
  }
  for (int i = threadIdx.x; i < n; i += 256) {
    int32_t x = cand[i];
    bool first = true;
    for (int j = 0; j < i; ++j) first = first && (cand[j] != x);
    keep[i] = first;
    if (first) atomicAdd(&n_keep, 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    if (!keep[i]) continue;
    int32_t x = cand[i];
    int r = 0;
    for (int j = 0; j < n; ++j) r += (keep[j] && cand[j] < x);
    if (r < HALO_CAP) halo[(int64_t)tile * HALO_CAP + r] = x;
  }
  if (threadIdx.x == 0) {
    halo_cnt[tile] = n_keep;
    if (n_keep > HALO_CAP) atomicOr(&misc[0], 4);
    atomicMax(&misc[1], (t1 - t0) + n_keep);
  }
}

// ---------------------------------------------------------------- slices / pair-merged ELL
// A node's neighbour u usually appears twice: as in-edge (u -> v) and as out-edge (v -> u), and for mesh
// data the two edge_attr triples are exact mirror images, a_uv = (-a_vu[0], -a_vu[1], a_vu[2]) (differences
// of coordinates and a length).  Such a pair is stored as ONE slot {LDS row of u, IN|OUT, a_vu}: the kernel
// reads 16 bytes and one LDS row per neighbour instead of per edge direction.  The mirror property is
// checked bit-for-bit per pair when the plan is built; anything else (one-directional edges such as those
// into Dirichlet rows, duplicate edges, attrs that are not mirrors) becomes an IN-only or OUT-only slot, so
// the result is exact for arbitrary input.  Slot order = merge of the canonical in- and out-lists, which keeps
// each direction's summation order.
#define SLOT_IN 0x10000u
#define SLOT_OUT 0x20000u

// emit(r, kind, i, j) is called once per slot in order; kind: 1 in-only, 2 out-only, 3 merged pair;
// i / j = positions in the CSC (in) / CSR (out) lists.
template <typename F>
__device__ __forceinline__ int merge_slots(int32_t ib, int32_t ie, int32_t jb, int32_t je,
                                           const int32_t* __restrict__ csc_nbr, const float* __restrict__ csc_attr,
                                           const int32_t* __restrict__ csr_nbr, const float* __restrict__ csr_attr, F emit) {
  int n = 0;
  int32_t i = ib, j = jb;
  while (i < ie || j < je) {
    const int32_t ni = i < ie ? csc_nbr[i] : INT32_MAX;
    const int32_t nj = j < je ? csr_nbr[j] : INT32_MAX;
    if (ni == nj) {
      const bool single = (i + 1 >= ie || csc_nbr[i + 1] != ni) && (j + 1 >= je || csr_nbr[j + 1] != nj);
      const uint32_t* ai = reinterpret_cast<const uint32_t*>(csc_attr) + 3 * (int64_t)i;
      const uint32_t* aj = reinterpret_cast<const uint32_t*>(csr_attr) + 3 * (int64_t)j;
      const bool mirror = ai[0] == (aj[0] ^ 0x80000000u) && ai[1] == (aj[1] ^ 0x80000000u) && ai[2] == aj[2];
      if (single && mirror) {
        emit(n, 3, i, j);
        ++i; ++j;
      } else {
        emit(n, 1, i, j);
        ++i;
      }
    } else if (ni < nj) {
      emit(n, 1, i, j);
      ++i;
    } else {
      emit(n, 2, i, j);
      ++j;
    }
    ++n;
  }
  return n;
}

// One wave per slice: max slot count of its (up to) 64 nodes.
__global__ __launch_bounds__(256) void k_slice_deg(int64_t n_slices, const int32_t* __restrict__ slice_tile,
                                                   const int32_t* __restrict__ tile_slice,
                                                   const int32_t* __restrict__ tile_ptr, const int32_t* __restrict__ perm,
                                                   const int32_t* __restrict__ csr_ptr, const int32_t* __restrict__ csr_nbr,
                                                   const float* __restrict__ csr_attr, const int32_t* __restrict__ csc_ptr,
                                                   const int32_t* __restrict__ csc_nbr, const float* __restrict__ csc_attr,
                                                   uint8_t* __restrict__ slice_deg, int32_t* __restrict__ misc) {
  int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= n_slices) return;
  int lane = threadIdx.x & 63;
  int tile = slice_tile[s];
  int32_t node = tile_ptr[tile] + 64 * (int32_t)(s - tile_slice[tile]) + lane;
  int ns = 0;
  if (node < tile_ptr[tile + 1]) {
    int32_t old = perm[node];
    ns = merge_slots(csc_ptr[old], csc_ptr[old + 1], csr_ptr[old], csr_ptr[old + 1], csc_nbr, csc_attr, csr_nbr, csr_attr,
                     [](int, int, int32_t, int32_t) {});
  }
  for (int o = 32; o > 0; o >>= 1) ns = max(ns, __shfl_xor(ns, o));
  if (lane == 0) {
    if (ns > 255) atomicOr(&misc[0], 8);
    slice_deg[s] = (uint8_t)min(ns, 255);
  }
}

__device__ __forceinline__ int lower_bound_i32(const int32_t* __restrict__ a, int n, int32_t x) {
  int lo = 0, hi = n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (a[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// ell: (ell_rows, 64) slots of 4 words {LDS row | SLOT_IN | SLOT_OUT, a0, a1, a2}; an empty slot is {0xFFFF,0,0,0}.
// a = edge_attr of the OUT edge (v -> u); for an IN-only slot a = mirror of the in-edge's attr, so that the
// kernel's mirror of a gives the in-edge's attr back bit-exactly.
__global__ __launch_bounds__(256) void k_ell_fill(int64_t n_slices, const int32_t* __restrict__ slice_tile,
                                                  const int32_t* __restrict__ tile_slice,
                                                  const int32_t* __restrict__ tile_ptr, const int32_t* __restrict__ perm,
                                                  const int32_t* __restrict__ inv, const int32_t* __restrict__ halo,
                                                  const int32_t* __restrict__ halo_cnt,
                                                  const int32_t* __restrict__ csr_ptr, const int32_t* __restrict__ csr_nbr,
                                                  const float* __restrict__ csr_attr, const int32_t* __restrict__ csc_ptr,
                                                  const int32_t* __restrict__ csc_nbr, const float* __restrict__ csc_attr,
                                                  const int32_t* __restrict__ slice_off, const uint8_t* __restrict__ slice_deg,
                                                  uint4* __restrict__ ell) {
  int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= n_slices) return;
  int lane = threadIdx.x & 63;
  int tile = slice_tile[s];
  const int32_t t0 = tile_ptr[tile], t1 = tile_ptr[tile + 1];
  const int32_t n_t = t1 - t0;
  int32_t node = t0 + 64 * (int32_t)(s - tile_slice[tile]) + lane;
  const int32_t* hl = halo + (int64_t)tile * HALO_CAP;
  const int hc = min(halo_cnt[tile], HALO_CAP);
  const int64_t row0 = slice_off[s];
  const int dmax = slice_deg[s];
  int ns = 0;
  if (node < t1) {
    int32_t old = perm[node];
    ns = merge_slots(csc_ptr[old], csc_ptr[old + 1], csr_ptr[old], csr_ptr[old + 1], csc_nbr, csc_attr, csr_nbr, csr_attr,
                     [&](int r, int kind, int32_t i, int32_t j) {
                       if (r >= dmax) return;
                       const int32_t nb = inv[(kind & 2) ? csr_nbr[j] : csc_nbr[i]];
                       uint32_t li = (nb >= t0 && nb < t1) ? (uint32_t)(nb - t0) : (uint32_t)(n_t + lower_bound_i32(hl, hc, nb));
                       uint4 v;
                       v.x = li | ((kind & 1) ? SLOT_IN : 0u) | ((kind & 2) ? SLOT_OUT : 0u);
                       if (kind & 2) {
                         const uint32_t* a = reinterpret_cast<const uint32_t*>(csr_attr) + 3 * (int64_t)j;
                         v.y = a[0]; v.z = a[1]; v.w = a[2];
                       } else {
                         const uint32_t* a = reinterpret_cast<const uint32_t*>(csc_attr) + 3 * (int64_t)i;
                         v.y = a[0] ^ 0x80000000u; v.z = a[1] ^ 0x80000000u; v.w = a[2];
                       }
                       ell[(row0 + r) * 64 + lane] = v;
                     });
  }
  for (int r = ns; r < dmax; ++r) ell[(row0 + r) * 64 + lane] = make_uint4(ELL_EMPTY, 0u, 0u, 0u);
}

// ---------------------------------------------------------------- host
int psignn_exclusive_scan(const int32_t* in, int64_t n, int32_t* out, int32_t* bsum, hipStream_t st);

void psignn_tiles_free(psignn_plan* p) {
  void* ptrs[] = {p->perm, p->inv, p->tile_ptr, p->tile_slice, p->halo, p->halo_cnt,
                  p->slice_off, p->slice_deg, p->ell, p->flags_p, p->tile_order, p->d_ctx, p->tile_order_cost};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  p->perm = p->inv = p->tile_ptr = p->tile_slice = p->halo = p->halo_cnt = p->slice_off = nullptr;
  p->slice_deg = nullptr; p->ell = nullptr; p->flags_p = nullptr; p->tile_order = nullptr; p->d_ctx = nullptr; p->tile_order_cost = nullptr;
  p->tiled = 0;
}

#define HT(expr)                                                                        \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      psignn_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      rc = PSIGNN_EHIP;                                                                 \
      goto done;                                                                        \
    }                                                                                   \
  } while (0)

// Builds the tile structures.  Returns 0 and sets p->tiled = 1 on success; returns 0 with p->tiled = 0
// when a structural limit was hit (the caller keeps the untiled plan); negative on HIP errors.
__global__ void k_tile_has_neumann(int64_t n_tiles, const int32_t* __restrict__ tile_ptr, const uint8_t* __restrict__ flags_p,
                                   uint8_t* __restrict__ has) {
  const int64_t t = blockIdx.x;
  if (t >= n_tiles) return;
  int any = 0;
  for (int32_t i = tile_ptr[t] + threadIdx.x; i < tile_ptr[t + 1]; i += blockDim.x) any |= flags_p[i] & FLAG_NEUMANN;
  any = __syncthreads_or(any);
  if (threadIdx.x == 0) has[t] = any ? 1 : 0;
}

// mode 0: snake strips (full tiles with the halo of a square patch on quasi-uniform meshes); 1: Hilbert curve (compact blobs
// at any local density, ~25 % more halo rows)
static int tiles_build_mode(psignn_plan* p, const float* d_pos, int tile_target, hipStream_t st, int mode) {
  const int64_t N = p->N;
  const unsigned TB = 256;
  int rc = 0;
  if (tile_target <= 0 || tile_target > TILE_MAX) tile_target = TILE_MAX;
  int32_t *cnt = nullptr, *cptr = nullptr, *cur = nullptr, *bsum = nullptr, *misc = nullptr, *slice_tile = nullptr;
  uint32_t* box = nullptr;
  std::vector<int32_t> h_cptr, h_tile_ptr, h_tile_slice, h_slice_tile, h_slice_off;
  std::vector<uint8_t> h_deg;
  int64_t ncell = 0;
  int32_t h_misc[2] = {0, 0};
  const unsigned gn = (unsigned)cdiv(N, TB);

  HT(hipMalloc((void**)&p->perm, N * 4));
  HT(hipMalloc((void**)&p->inv, N * 4));
  HT(hipMalloc((void**)&p->flags_p, N));
  HT(hipMalloc((void**)&misc, 8));
  HT(hipMemsetAsync(misc, 0, 8, st));

  if (d_pos) {
    // ---- 1. cells
    uint32_t h_box[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u};
    HT(hipMalloc((void**)&box, 16));
    HT(hipMemcpyAsync(box, h_box, 16, hipMemcpyHostToDevice, st));
    k_bbox<<<1024, TB, 0, st>>>(N, d_pos, box);
    HT(hipMemcpyAsync(h_box, box, 16, hipMemcpyDeviceToHost, st));
    HT(hipStreamSynchronize(st));
    float xmin = o2f(h_box[0]), ymin = o2f(h_box[1]), xmax = o2f(h_box[2]), ymax = o2f(h_box[3]);
    if (!(isfinite(xmin) && isfinite(ymin) && isfinite(xmax) && isfinite(ymax))) goto done;  // NaN/inf positions: stay untiled
    double w = fmax((double)xmax - xmin, 1e-30), hgt = fmax((double)ymax - ymin, 1e-30);
    int nx, ny;
    double cs, inv_h = 0.0;
    if (mode == 1) {
      // Fine cells of ~4 nodes on a 2^k x 2^k grid over the bounding square (side L = max(w, h)); the part of the grid
      // inside the bounding box has G^2 w h / L^2 cells.  Ordering only: any k gives a correct plan.
      const double L = fmax(w, hgt);
      const double cells_wanted = fmax((double)N / 4.0, 1.0);
      int k = (int)ceil(0.5 * log2(fmax(cells_wanted * L * L / (w * hgt), 1.0)));
      const int kcap = (int)ceil(0.5 * log2(16.0 * (double)N));   // at most ~16 cells per node, however thin the domain
      k = std::min(k, std::min(kcap, 12));
      k = k < 0 ? 0 : k;
      nx = ny = 1 << k;
      cs = L / nx * (1.0 + 1e-6);
    } else {
      // Strips: node density from the occupied cells of a coarse grid (domains do not fill their bounding box), strip height
      // = side of a square patch of tile_target nodes, cells of ~8 nodes along the strip.
      double c0 = sqrt(64.0 * w * hgt / (double)N);
      int gx = (int)fmin(fmax(ceil(w / c0), 1.0), 4096.0), gy = (int)fmin(fmax(ceil(hgt / c0), 1.0), 4096.0);
      c0 = fmax(w / gx, hgt / gy) * (1.0 + 1e-6);
      const int64_t nc0 = (int64_t)gx * gy;
      HT(hipMalloc((void**)&cnt, (nc0 + 1) * 4));
      HT(hipMemsetAsync(cnt, 0, (nc0 + 1) * 4, st));
      k_cell_count<<<gn, TB, 0, st>>>(N, d_pos, xmin, ymin, -(float)(1.0 / c0), (float)(1.0 / c0), gx, gy, cnt);
      k_count_occupied<<<(unsigned)cdiv(nc0, TB), TB, 0, st>>>(nc0, cnt, cnt + nc0);
      int32_t occ = 0;
      HT(hipMemcpyAsync(&occ, cnt + nc0, 4, hipMemcpyDeviceToHost, st));
      HT(hipStreamSynchronize(st));
      (void)hipFree(cnt);
      cnt = nullptr;
      const double rho = (double)N / (fmax((double)occ, 1.0) * c0 * c0);
      const double H = sqrt((double)tile_target / rho);
      ny = (int)fmin(fmax(ceil(hgt / H), 1.0), 65536.0);
      const double cw = 8.0 / (rho * H);
      nx = (int)fmin(fmax(ceil(w / cw), 1.0), 65536.0);
      while ((int64_t)nx * ny > 16 * N + 1024 && nx > 1) nx = (nx + 1) / 2;   // never more than ~16 cells per node
      inv_h = (double)ny / (hgt * (1.0 + 1e-6));
      cs = -(w * (1.0 + 1e-6)) / nx;          // negative: marks the strip mode for cell_of (|cs| = cell width)
    }
    ncell = (int64_t)nx * ny;
    HT(hipMalloc((void**)&cnt, (ncell + 1) * 4));
    HT(hipMemsetAsync(cnt, 0, (ncell + 1) * 4, st));
    k_cell_count<<<gn, TB, 0, st>>>(N, d_pos, xmin, ymin, (float)(1.0 / cs), (float)inv_h, nx, ny, cnt);
    p->cell_size = (float)fabs(cs); p->xmin = xmin; p->ymin = ymin; p->nx = nx; p->ny = ny;
    HT(hipMalloc((void**)&cptr, (ncell + 1) * 4));
    HT(hipMalloc((void**)&cur, ncell * 4));
    HT(hipMalloc((void**)&bsum, (cdiv(ncell, 1024) + 2) * 4));
    HT(hipMemsetAsync(cur, 0, ncell * 4, st));
    if ((rc = psignn_exclusive_scan(cnt, ncell, cptr, bsum, st)) != 0) goto done;
    k_cell_fill<<<gn, TB, 0, st>>>(N, d_pos, xmin, ymin, (float)(1.0 / cs), (float)inv_h, nx, ny, cptr, cur, p->perm);
    // node ids ascending inside a cell: the order (hence which nodes a chunk boundary cuts off) is deterministic
    for (int64_t c0 = 0; c0 < ncell; c0 += 1 << 20)
      k_cell_sort<<<(unsigned)std::min<int64_t>(ncell - c0, 1 << 20), TB, 0, st>>>(cptr + c0, p->perm, misc);
    HT(hipMemcpyAsync(h_misc, misc, 8, hipMemcpyDeviceToHost, st));
    HT(hipStreamSynchronize(st));
    if (h_misc[0]) goto done;  // a cell above SORT_CAP nodes (coincident points): stay untiled
  } else {
    // no coordinates: keep the given numbering, tiles = consecutive chunks
    k_iota<<<gn, TB, 0, st>>>(N, p->perm);
  }
  // ---- 2. tiles = consecutive chunks of tile_target nodes of that order; slices (host: a few thousand entries)
  for (int64_t b = 0; b < N; b += tile_target) h_tile_ptr.push_back((int32_t)b);
  h_tile_ptr.push_back((int32_t)N);
  p->n_tiles = (int64_t)h_tile_ptr.size() - 1;
  h_tile_slice.resize(p->n_tiles + 1);
  h_tile_slice[0] = 0;
  for (int64_t t = 0; t < p->n_tiles; ++t) {
    int ns = (h_tile_ptr[t + 1] - h_tile_ptr[t] + 63) / 64;
    h_tile_slice[t + 1] = h_tile_slice[t] + ns;
    for (int j = 0; j < ns; ++j) h_slice_tile.push_back((int32_t)t);
  }
  p->n_slices = h_tile_slice[p->n_tiles];
  HT(hipMalloc((void**)&p->tile_ptr, (p->n_tiles + 1) * 4));
  HT(hipMalloc((void**)&p->tile_slice, (p->n_tiles + 1) * 4));
  HT(hipMalloc((void**)&slice_tile, p->n_slices * 4 + 4));
  HT(hipMalloc((void**)&p->halo, p->n_tiles * HALO_CAP * 4));
  HT(hipMalloc((void**)&p->halo_cnt, p->n_tiles * 4));
  HT(hipMalloc((void**)&p->slice_deg, p->n_slices + 2));
  HT(hipMalloc((void**)&p->slice_off, (p->n_slices + 1) * 4));
  HT(hipMemcpyAsync(p->tile_ptr, h_tile_ptr.data(), (p->n_tiles + 1) * 4, hipMemcpyHostToDevice, st));
  HT(hipMemcpyAsync(p->tile_slice, h_tile_slice.data(), (p->n_tiles + 1) * 4, hipMemcpyHostToDevice, st));
  HT(hipMemcpyAsync(slice_tile, h_slice_tile.data(), p->n_slices * 4, hipMemcpyHostToDevice, st));
  if (d_pos)  // node ids ascending inside a tile (a tile's lanes then follow the caller's numbering, like its LDS rows)
    k_cell_sort<<<(unsigned)p->n_tiles, TB, 0, st>>>(p->tile_ptr, p->perm, misc);
  k_inverse_perm<<<gn, TB, 0, st>>>(N, p->perm, p->inv, p->flags, p->flags_p);
  // ---- 3. halos
  k_halo<<<(unsigned)p->n_tiles, TB, 0, st>>>(p->tile_ptr, p->perm, p->inv, p->csr_ptr, p->csr_nbr, p->csc_ptr,
                                               p->csc_nbr, p->halo, p->halo_cnt, misc);
  // ---- 4. slices / ELL
  k_slice_deg<<<(unsigned)cdiv(p->n_slices, 4), TB, 0, st>>>(p->n_slices, slice_tile, p->tile_slice, p->tile_ptr,
                                                              p->perm, p->csr_ptr, p->csr_nbr, p->csr_attr, p->csc_ptr,
                                                              p->csc_nbr, p->csc_attr, p->slice_deg, misc);
  h_deg.resize(p->n_slices);
  HT(hipMemcpyAsync(h_deg.data(), p->slice_deg, p->n_slices, hipMemcpyDeviceToHost, st));
  HT(hipMemcpyAsync(h_misc, misc, 8, hipMemcpyDeviceToHost, st));
  HT(hipStreamSynchronize(st));
  if (h_misc[0]) goto done;  // halo or degree limit exceeded: stay untiled
  p->max_rows = h_misc[1];
  h_slice_off.resize(p->n_slices + 1);
  h_slice_off[0] = 0;
  for (int64_t s = 0; s < p->n_slices; ++s) {
    int64_t nxt = (int64_t)h_slice_off[s] + h_deg[s];
    if (nxt > (int64_t)INT32_MAX / 256) goto done;
    h_slice_off[s + 1] = (int32_t)nxt;
  }
  p->ell_rows = h_slice_off[p->n_slices];
  HT(hipMemcpyAsync(p->slice_off, h_slice_off.data(), (p->n_slices + 1) * 4, hipMemcpyHostToDevice, st));
  HT(hipMalloc((void**)&p->ell, (size_t)(p->ell_rows + 1) * 64 * 16));
  k_ell_fill<<<(unsigned)cdiv(p->n_slices, 4), TB, 0, st>>>(p->n_slices, slice_tile, p->tile_slice, p->tile_ptr, p->perm,
                                                             p->inv, p->halo, p->halo_cnt, p->csr_ptr, p->csr_nbr,
                                                             p->csr_attr, p->csc_ptr, p->csc_nbr, p->csc_attr,
                                                             p->slice_off, p->slice_deg, p->ell);
  HT(hipStreamSynchronize(st));
  HT(hipGetLastError());
  if (p->mixed) {  // tile order: plain tiles first, tiles with Neumann nodes last
    uint8_t* d_has = nullptr;
    HT(hipMalloc((void**)&d_has, p->n_tiles + 1));
    k_tile_has_neumann<<<(unsigned)p->n_tiles, 64, 0, st>>>(p->n_tiles, p->tile_ptr, p->flags_p, d_has);
    std::vector<uint8_t> h_has(p->n_tiles);
    hipError_t e1 = hipMemcpyAsync(h_has.data(), d_has, p->n_tiles, hipMemcpyDeviceToHost, st);
    hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(d_has);
    HT(e1);
    HT(e2);
    std::vector<int32_t> order;
    order.reserve(p->n_tiles);
    for (int64_t t = 0; t < p->n_tiles; ++t)
      if (!h_has[t]) order.push_back((int32_t)t);
    p->n_tiles_plain = (int64_t)order.size();
    for (int64_t t = 0; t < p->n_tiles; ++t)
      if (h_has[t]) order.push_back((int32_t)t);
    HT(hipMalloc((void**)&p->tile_order, p->n_tiles * 4));
    HT(hipMemcpy(p->tile_order, order.data(), p->n_tiles * 4, hipMemcpyHostToDevice));
  }
  {
    TileCtx h{p->tile_ptr, p->tile_slice, p->halo, p->halo_cnt, p->slice_off, p->slice_deg, p->ell, p->flags_p,
              (tile_target % 64 == 0) ? tile_target : 0, (int32_t)N};
    HT(hipMalloc((void**)&p->d_ctx, sizeof(TileCtx)));
    HT(hipMemcpy(p->d_ctx, &h, sizeof(TileCtx), hipMemcpyHostToDevice));
    p->h_ctx = h;
  }
  p->tiled = 1;
done:
  for (void* q : {(void*)cnt, (void*)cptr, (void*)cur, (void*)bsum, (void*)misc, (void*)slice_tile, (void*)box})
    if (q) (void)hipFree(q);
  if (!p->tiled) psignn_tiles_free(p);
  return rc;
}

// Tile structures: strips first, the Hilbert order where strips exceed a structure limit (halo > 512: graded meshes whose
// strips degenerate into long thin tiles), untiled where that fails too.  PSIGNN_TILING = strips | hilbert forces one.
int psignn_tiles_build(psignn_plan* p, const float* d_pos, int tile_target, hipStream_t st) {
  KNOB_INT(forced, [] {
    const char* e = getenv("PSIGNN_TILING");
    return !e ? -1 : (strcmp(e, "hilbert") == 0 ? 1 : 0);
  }());
  if (!d_pos || forced >= 0) return tiles_build_mode(p, d_pos, tile_target, st, forced > 0 ? 1 : 0);
  int rc = tiles_build_mode(p, d_pos, tile_target, st, 0);
  if (rc == 0 && !p->tiled) rc = tiles_build_mode(p, d_pos, tile_target, st, 1);
  return rc;
}

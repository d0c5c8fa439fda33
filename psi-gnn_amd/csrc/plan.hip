// Mesh plan: one-time, on-device preprocessing of a mesh graph (gfx950).
//
// Hoists out of the fixed-point loop everything the reference recomputes in every f call:
// self-loop removal (dirichlet/psignn/model.py:342,360), the Dirichlet/Neumann index sets
// (model.py:281; mixed/psignn/model.py:218-219) and the sparse-matrix build of residual_loss
// (model.py:159-163).  Output: int32 CSR (by row) and CSC (by col) of the non-self edges with a
// canonical in-group order (other endpoint, original edge id), edge_attr permuted into both
// orders, full CSR of A, node flags.  Integer work only: histogram (int atomics, order-free
// result), exclusive scan, bucket fill, per-bucket canonical sort.
#include "common.h"
#include <stdarg.h>
#include <vector>

static thread_local std::string g_err;
void psignn_set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}
extern "C" const char* psignn_last_error(void) { return g_err.c_str(); }
extern "C" int psignn_version(void) { return 100; }
int g_knob_epoch = 0;
extern "C" void psignn_reload_knobs(void) { ++g_knob_epoch; }

// ------------------------------------------------------------------ event profiler
#include <map>
int g_prof_on = 0;
namespace {
struct ProfRec { int name; hipEvent_t a, b; int64_t bytes; };
std::vector<ProfRec> g_recs;
std::vector<std::string> g_names;
std::vector<hipEvent_t> g_pool;
struct ProfAgg { std::string name; int64_t calls; double ms; int64_t bytes; };
std::vector<ProfAgg> g_agg;
struct ProfLaunch { int name; float ms; int64_t bytes; };
std::vector<ProfLaunch> g_log;   // the launches of the last collect, in launch order
hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e; (void)hipEventCreate(&e); return e;
}
int name_id(const char* n) {
  for (size_t i = 0; i < g_names.size(); ++i) if (g_names[i] == n) return (int)i;
  g_names.push_back(n); return (int)g_names.size() - 1;
}
}  // namespace
int64_t g_prof_next_bytes = 0;
void prof_begin(const char* name, hipStream_t st) {
  ProfRec r{name_id(name), get_event(), get_event(), g_prof_next_bytes};
  g_prof_next_bytes = 0;
  (void)hipEventRecord(r.a, st);
  g_recs.push_back(r);
}
void prof_end(hipStream_t st) { (void)hipEventRecord(g_recs.back().b, st); }

extern "C" void psignn_prof_enable(int on) { g_prof_on = on; }
// Waits for the recorded work, aggregates per kernel name, clears the records.  Returns #names.
extern "C" int psignn_prof_collect(void) {
  std::map<int, ProfAgg> m;
  g_log.clear();
  for (auto& r : g_recs) {
    (void)hipEventSynchronize(r.b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.a, r.b);
    g_log.push_back({r.name, ms, r.bytes});
    auto& a = m[r.name];
    a.name = g_names[r.name]; a.calls += 1; a.ms += ms; a.bytes += r.bytes;
    g_pool.push_back(r.a); g_pool.push_back(r.b);
  }
  g_recs.clear();
  g_agg.clear();
  for (auto& kv : m) g_agg.push_back(kv.second);
  return (int)g_agg.size();
}
extern "C" int psignn_prof_get(int i, char* name, int cap, int64_t* calls, double* total_ms) {
  if (i < 0 || i >= (int)g_agg.size() || !name || cap < 2) return PSIGNN_EINVAL;
  snprintf(name, cap, "%s", g_agg[i].name.c_str());
  if (calls) *calls = g_agg[i].calls;
  if (total_ms) *total_ms = g_agg[i].ms;
  return PSIGNN_OK;
}

// the same plus the sum of the launches' algorithmic bytes (0 for kernels that state none)
extern "C" int psignn_prof_get2(int i, char* name, int cap, int64_t* calls, double* total_ms, int64_t* alg_bytes) {
  int rc = psignn_prof_get(i, name, cap, calls, total_ms);
  if (rc == PSIGNN_OK && alg_bytes) *alg_bytes = g_agg[i].bytes;
  return rc;
}

// launch i of the last collect, in launch order (returns the number of launches when name == NULL)
extern "C" int psignn_prof_launch(int i, char* name, int cap, double* ms, int64_t* alg_bytes) {
  if (!name) return (int)g_log.size();
  if (i < 0 || i >= (int)g_log.size() || cap < 2) return PSIGNN_EINVAL;
  snprintf(name, cap, "%s", g_names[g_log[i].name].c_str());
  if (ms) *ms = g_log[i].ms;
  if (alg_bytes) *alg_bytes = g_log[i].bytes;
  return PSIGNN_OK;
}

// ------------------------------------------------------------------ kernels
__global__ void k_count_edges(int64_t E, int64_t N, const int64_t* __restrict__ ei,
                              int32_t* __restrict__ outdeg, int32_t* __restrict__ indeg,
                              int32_t* __restrict__ fulldeg, int32_t* __restrict__ err) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t r = ei[e], c = ei[E + e];
  if (r < 0 || r >= N || c < 0 || c >= N) {
    atomicOr(err, 1);
    return;
  }
  atomicAdd(&fulldeg[r], 1);
  if (r != c) {
    atomicAdd(&outdeg[r], 1);
    atomicAdd(&indeg[c], 1);
  }
}

// Exclusive scan, 3 phases.  Block = 256 threads x 4 items.
#define SCAN_ITEMS 1024
__global__ void k_scan_block_sums(const int32_t* __restrict__ in, int64_t n, int32_t* __restrict__ bsum) {
  __shared__ int32_t red[256];
  int64_t base = (int64_t)blockIdx.x * SCAN_ITEMS;
  int32_t s = 0;
  for (int i = threadIdx.x; i < SCAN_ITEMS; i += 256) {
    int64_t idx = base + i;
    if (idx < n) s += in[idx];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) bsum[blockIdx.x] = red[0];
}
// single block: in-place exclusive scan of bsum[0..nb), total -> bsum[nb]
__global__ void k_scan_sums(int32_t* __restrict__ bsum, int64_t nb) {
  __shared__ int32_t sh[256];
  __shared__ int32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < nb; base += 256) {
    int64_t idx = base + threadIdx.x;
    int32_t v = idx < nb ? bsum[idx] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      int32_t t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    int32_t incl = sh[threadIdx.x];
    if (idx < nb) bsum[idx] = carry + incl - v;
    __syncthreads();
    if (threadIdx.x == 255) carry += incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) bsum[nb] = carry;
}
__global__ void k_scan_final(const int32_t* __restrict__ in, int64_t n, const int32_t* __restrict__ bsum,
                             int32_t* __restrict__ out /* n+1 */) {
  __shared__ int32_t sh[256];
  int64_t base = (int64_t)blockIdx.x * SCAN_ITEMS + (int64_t)threadIdx.x * 4;
  int32_t v[4];
  int32_t s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i] = (base + i < n) ? in[base + i] : 0;
    s += v[i];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    int32_t t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  int32_t run = bsum[blockIdx.x] + sh[threadIdx.x] - s;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (base + i < n) out[base + i] = run;
    run += v[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = bsum[gridDim.x];
}

__global__ void k_fill(int64_t E, const int64_t* __restrict__ ei, const int32_t* __restrict__ csr_ptr,
                       const int32_t* __restrict__ csc_ptr, const int32_t* __restrict__ a_ptr,
                       int32_t* __restrict__ cur_r, int32_t* __restrict__ cur_c, int32_t* __restrict__ cur_a,
                       int32_t* __restrict__ csr_nbr, int32_t* __restrict__ csr_eid,
                       int32_t* __restrict__ csc_nbr, int32_t* __restrict__ csc_eid,
                       int32_t* __restrict__ a_col, int32_t* __restrict__ a_eid) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int32_t r = (int32_t)ei[e], c = (int32_t)ei[E + e];
  int32_t sa = a_ptr[r] + atomicAdd(&cur_a[r], 1);
  a_col[sa] = c;
  a_eid[sa] = (int32_t)e;
  if (r != c) {
    int32_t s1 = csr_ptr[r] + atomicAdd(&cur_r[r], 1);
    csr_nbr[s1] = c;
    csr_eid[s1] = (int32_t)e;
    int32_t s2 = csc_ptr[c] + atomicAdd(&cur_c[c], 1);
    csc_nbr[s2] = r;
    csc_eid[s2] = (int32_t)e;
  }
}

// One thread per bucket: insertion sort by (nbr, eid).  Buckets are mesh-vertex degrees (~6).
__global__ void k_canon(int64_t N, const int32_t* __restrict__ ptr, int32_t* __restrict__ nbr,
                        int32_t* __restrict__ eid, int32_t* __restrict__ maxdeg) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= N) return;
  int32_t s = ptr[v], e = ptr[v + 1];
  for (int32_t i = s + 1; i < e; ++i) {
    int32_t kn = nbr[i], ke = eid[i];
    int32_t j = i - 1;
    while (j >= s && (nbr[j] > kn || (nbr[j] == kn && eid[j] > ke))) {
      nbr[j + 1] = nbr[j];
      eid[j + 1] = eid[j];
      --j;
    }
    nbr[j + 1] = kn;
    eid[j + 1] = ke;
  }
  if (maxdeg) atomicMax(maxdeg, e - s);
}

__global__ void k_gather_attr(int64_t Ep, const int32_t* __restrict__ eid, const float* __restrict__ attr,
                              float* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Ep) return;
  int64_t e = eid[i];
  out[3 * i + 0] = attr[3 * e + 0];
  out[3 * i + 1] = attr[3 * e + 1];
  out[3 * i + 2] = attr[3 * e + 2];
}
__global__ void k_gather_val(int64_t E, const int32_t* __restrict__ eid, const float* __restrict__ val,
                             float* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < E) out[i] = val[eid[i]];
}
__global__ void k_flags(int64_t N, const float* __restrict__ tags, int cols, uint8_t* __restrict__ flags) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= N) return;
  uint8_t f = 0;
  if (cols == 1) {
    if (tags[v] == 1.0f) f = FLAG_DIRICHLET;
  } else {
    if (tags[v * 3 + 1] == 1.0f) f |= FLAG_DIRICHLET;
    if (tags[v * 3 + 2] == 1.0f) f |= FLAG_NEUMANN;
  }
  flags[v] = f;
}

// ------------------------------------------------------------------ host
int psignn_exclusive_scan(const int32_t* in, int64_t n, int32_t* out, int32_t* bsum, hipStream_t st) {
  int64_t nb = cdiv(n, SCAN_ITEMS);
  k_scan_block_sums<<<dim3((unsigned)nb), 256, 0, st>>>(in, n, bsum);
  k_scan_sums<<<1, 256, 0, st>>>(bsum, nb);
  k_scan_final<<<dim3((unsigned)nb), 256, 0, st>>>(in, n, bsum, out);
  HIP_TRY(hipGetLastError());
  return 0;
}

template <typename T>
static int dmalloc(T** p, size_t n) {
  if (n == 0) n = 1;
  HIP_TRY(hipMalloc((void**)p, n * sizeof(T)));
  return 0;
}

extern "C" void psignn_plan_destroy(psignn_plan_t* p) {
  if (!p) return;
  void* ptrs[] = {p->csr_ptr, p->csr_nbr, p->csr_eid, p->csc_ptr, p->csc_nbr, p->csc_eid, p->csr_attr,
                  p->csc_attr, p->flags,   p->a_ptr,   p->a_col,   p->a_val};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  psignn_tiles_free(p);
  delete p;
}

extern "C" int psignn_plan_create(psignn_plan_t** out, int64_t N, int64_t E, const int64_t* d_ei,
                                  const float* d_attr, const float* d_aij, const float* d_tags, int tags_cols,
                                  const float* d_pos, int tile_target, void* stream) {
  ARG_CHECK(out != nullptr, "out is NULL");
  *out = nullptr;
  ARG_CHECK(N > 0 && N < (int64_t)INT32_MAX, "n_nodes out of range");
  ARG_CHECK(E >= 0 && E < (int64_t)INT32_MAX, "n_edges out of range");
  ARG_CHECK(E == 0 || (d_ei && d_attr), "edge_index / edge_attr is NULL");
  ARG_CHECK(d_tags != nullptr, "tags is NULL");
  ARG_CHECK(tags_cols == 1 || tags_cols == 3, "tags must have 1 (dirichlet) or 3 (mixed) columns");
  hipStream_t st = (hipStream_t)stream;
  psignn_plan* p = new psignn_plan();
  p->N = N;
  p->E = E;
  p->mixed = tags_cols == 3;
  int rc = 0;
  int32_t *deg = nullptr, *bsum = nullptr, *a_eid = nullptr, *misc = nullptr;
  const unsigned TB = 256;
  auto fail = [&](int code) {
    if (deg) (void)hipFree(deg);
    if (bsum) (void)hipFree(bsum);
    if (a_eid) (void)hipFree(a_eid);
    if (misc) (void)hipFree(misc);
    psignn_plan_destroy(p);
    return code;
  };
#define TRY(x) \
  if ((rc = (x)) != 0) return fail(rc)
  // deg: outdeg | indeg | fulldeg  (later reused as fill cursors)
  TRY(dmalloc(&deg, 3 * (size_t)N));
  TRY(dmalloc(&bsum, (size_t)cdiv(N, SCAN_ITEMS) + 2));
  TRY(dmalloc(&misc, 2));  // [0] error flag, [1] max degree
  TRY(dmalloc(&p->csr_ptr, (size_t)N + 1));
  TRY(dmalloc(&p->csc_ptr, (size_t)N + 1));
  TRY(dmalloc(&p->a_ptr, (size_t)N + 1));
  TRY(dmalloc(&p->flags, (size_t)N));
  if (hipMemsetAsync(deg, 0, 3 * N * sizeof(int32_t), st) != hipSuccess ||
      hipMemsetAsync(misc, 0, 2 * sizeof(int32_t), st) != hipSuccess) {
    psignn_set_error("hipMemsetAsync failed");
    return fail(PSIGNN_EHIP);
  }
  if (E > 0) k_count_edges<<<dim3((unsigned)cdiv(E, TB)), TB, 0, st>>>(E, N, d_ei, deg, deg + N, deg + 2 * N, misc);
  TRY(psignn_exclusive_scan(deg, N, p->csr_ptr, bsum, st));
  TRY(psignn_exclusive_scan(deg + N, N, p->csc_ptr, bsum, st));
  TRY(psignn_exclusive_scan(deg + 2 * N, N, p->a_ptr, bsum, st));
  k_flags<<<dim3((unsigned)cdiv(N, TB)), TB, 0, st>>>(N, d_tags, tags_cols, p->flags);
  int32_t h_misc[2] = {0, 0}, h_ep = 0, h_ea = 0;
  if (hipMemcpyAsync(h_misc, misc, sizeof(h_misc), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipMemcpyAsync(&h_ep, p->csr_ptr + N, sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipMemcpyAsync(&h_ea, p->a_ptr + N, sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) {
    psignn_set_error("plan: device->host read failed: %s", hipGetErrorString(hipGetLastError()));
    return fail(PSIGNN_EHIP);
  }
  if (h_misc[0]) {
    psignn_set_error("edge_index holds a node id outside [0, %lld)", (long long)N);
    return fail(PSIGNN_EINDEX);
  }
  if (h_ea != (int32_t)E) {
    psignn_set_error("internal: full-CSR count %d != E %lld", h_ea, (long long)E);
    return fail(PSIGNN_EHIP);
  }
  p->Ep = h_ep;
  size_t Ep = (size_t)h_ep;
  TRY(dmalloc(&p->csr_nbr, Ep));
  TRY(dmalloc(&p->csr_eid, Ep));
  TRY(dmalloc(&p->csc_nbr, Ep));
  TRY(dmalloc(&p->csc_eid, Ep));
  TRY(dmalloc(&p->csr_attr, 3 * Ep));
  TRY(dmalloc(&p->csc_attr, 3 * Ep));
  TRY(dmalloc(&p->a_col, (size_t)E));
  TRY(dmalloc(&p->a_val, (size_t)E));
  TRY(dmalloc(&a_eid, (size_t)E));
  if (hipMemsetAsync(deg, 0, 3 * N * sizeof(int32_t), st) != hipSuccess) return fail(PSIGNN_EHIP);
  if (E > 0) {
    k_fill<<<dim3((unsigned)cdiv(E, TB)), TB, 0, st>>>(E, d_ei, p->csr_ptr, p->csc_ptr, p->a_ptr, deg, deg + N,
                                                        deg + 2 * N, p->csr_nbr, p->csr_eid, p->csc_nbr,
                                                        p->csc_eid, p->a_col, a_eid);
    unsigned gn = (unsigned)cdiv(N, TB);
    k_canon<<<gn, TB, 0, st>>>(N, p->csr_ptr, p->csr_nbr, p->csr_eid, misc + 1);
    k_canon<<<gn, TB, 0, st>>>(N, p->csc_ptr, p->csc_nbr, p->csc_eid, misc + 1);
    k_canon<<<gn, TB, 0, st>>>(N, p->a_ptr, p->a_col, a_eid, nullptr);
    if (Ep > 0) {
      unsigned ge = (unsigned)cdiv((int64_t)Ep, TB);
      k_gather_attr<<<ge, TB, 0, st>>>((int64_t)Ep, p->csr_eid, d_attr, p->csr_attr);
      k_gather_attr<<<ge, TB, 0, st>>>((int64_t)Ep, p->csc_eid, d_attr, p->csc_attr);
    }
    if (d_aij)
      k_gather_val<<<dim3((unsigned)cdiv(E, TB)), TB, 0, st>>>(E, a_eid, d_aij, p->a_val);
    else if (hipMemsetAsync(p->a_val, 0, E * sizeof(float), st) != hipSuccess)
      return fail(PSIGNN_EHIP);
  }
  if (hipMemcpyAsync(h_misc, misc, sizeof(h_misc), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) {
    psignn_set_error("plan build kernels failed: %s", hipGetErrorString(hipGetLastError()));
    return fail(PSIGNN_EHIP);
  }
  p->max_deg = h_misc[1];
  (void)hipFree(deg);
  (void)hipFree(bsum);
  (void)hipFree(a_eid);
  (void)hipFree(misc);
#undef TRY
  if (tile_target >= 0 && Ep > 0) {
    rc = psignn_tiles_build(p, d_pos, tile_target, st);
    if (rc) {
      psignn_plan_destroy(p);
      return rc;
    }
  }
  *out = p;
  return PSIGNN_OK;
}

extern "C" int psignn_plan_is_tiled(const psignn_plan_t* p) { return p ? p->tiled : 0; }
extern "C" int64_t psignn_plan_num_tiles(const psignn_plan_t* p) { return p && p->tiled ? p->n_tiles : 0; }
extern "C" int64_t psignn_plan_ell_rows(const psignn_plan_t* p) { return p && p->tiled ? p->ell_rows : 0; }
extern "C" int psignn_plan_max_tile_rows(const psignn_plan_t* p) { return p && p->tiled ? p->max_rows : 0; }

extern "C" int64_t psignn_plan_num_nodes(const psignn_plan_t* p) { return p ? p->N : -1; }
extern "C" int64_t psignn_plan_num_edges(const psignn_plan_t* p) { return p ? p->E : -1; }
extern "C" int64_t psignn_plan_num_nonself_edges(const psignn_plan_t* p) { return p ? p->Ep : -1; }

extern "C" int psignn_plan_export(const psignn_plan_t* p, int which, void* h_dst, size_t dst_bytes) {
  ARG_CHECK(p && h_dst, "NULL argument");
  const void* src = nullptr;
  size_t bytes = 0;
  size_t N = (size_t)p->N, E = (size_t)p->E, Ep = (size_t)p->Ep;
  switch (which) {
    case 0: src = p->csr_ptr; bytes = (N + 1) * 4; break;
    case 1: src = p->csr_nbr; bytes = Ep * 4; break;
    case 2: src = p->csr_eid; bytes = Ep * 4; break;
    case 3: src = p->csc_ptr; bytes = (N + 1) * 4; break;
    case 4: src = p->csc_nbr; bytes = Ep * 4; break;
    case 5: src = p->csc_eid; bytes = Ep * 4; break;
    case 6: src = p->flags; bytes = N; break;
    case 7: src = p->csr_attr; bytes = Ep * 12; break;
    case 8: src = p->csc_attr; bytes = Ep * 12; break;
    case 9: src = p->a_ptr; bytes = (N + 1) * 4; break;
    case 10: src = p->a_col; bytes = E * 4; break;
    case 11: src = p->a_val; bytes = E * 4; break;
    case 12: src = p->perm; bytes = p->tiled ? N * 4 : 0; break;
    case 13: src = p->tile_ptr; bytes = p->tiled ? (size_t)(p->n_tiles + 1) * 4 : 0; break;
    case 14: src = p->halo_cnt; bytes = p->tiled ? (size_t)p->n_tiles * 4 : 0; break;
    case 15: src = p->halo; bytes = p->tiled ? (size_t)p->n_tiles * HALO_CAP * 4 : 0; break;
    case 16: src = p->slice_off; bytes = p->tiled ? (size_t)(p->n_slices + 1) * 4 : 0; break;
    case 17: src = p->slice_deg; bytes = p->tiled ? (size_t)p->n_slices : 0; break;
    case 18: src = p->ell; bytes = p->tiled ? (size_t)p->ell_rows * 64 * 16 : 0; break;
    case 20: src = p->tile_slice; bytes = p->tiled ? (size_t)(p->n_tiles + 1) * 4 : 0; break;
    default: ARG_CHECK(false, "unknown array id");
  }
  ARG_CHECK(dst_bytes >= bytes, "destination too small");
  if (bytes) HIP_TRY(hipMemcpy(h_dst, src, bytes, hipMemcpyDeviceToHost));
  return PSIGNN_OK;
}

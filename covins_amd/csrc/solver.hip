// solver.hip — host side of libcovgpu: context, HBM residency, trust-region driver, extern "C" entry points.
//
// This is the thin extern "C" shim of BASELINE.json's north_star. It replaces what ceres::Solve does between
// optimization_be.cpp:560-567 (GBA), :257-265 (GBA outlier round) and :1024-1031 (PGO): only scalars cross
// PCIe inside the loop (three small read-backs per trust-region iteration); residuals, Jacobians, the Schur
// complement, the reduced system and every vector of the step live in HBM for the whole solve.
// Loop structure and constants follow SURVEY.md A.6 (Ceres 1.x TrustRegionMinimizer + DoglegStrategy /
// LevenbergMarquardtStrategy defaults).
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "common.hpp"
#include "nd_plan.hpp"

using namespace covgpu;

static thread_local std::string g_err;
extern "C" const char* covgpu_last_error(void) { return g_err.c_str(); }

#define HIPCHK(expr)                                                                            \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      g_err = std::string(#expr) + ": " + hipGetErrorString(e_);                                \
      return e_ == hipErrorOutOfMemory ? COVGPU_ERR_OUT_OF_MEMORY : COVGPU_ERR_NO_DEVICE;       \
    }                                                                                           \
  } while (0)

struct covgpu_profile_t {
  double t_build_ms = 0, t_factor_ms = 0, t_syrk_ms = 0, syrk_flops = 0;
  long n_build = 0, n_factor = 0, n_syrk = 0;
};

// elimination tree of the last single-GPU GBA upload and what it was built for (upload_impl)
struct PlanCache {
  bool valid = false; int K = 0; bool vi = false; int leaf = 0;
  std::vector<int> chain_ptr, pos_kf;
  std::vector<uint64_t> keys;   // sorted (position i << 32 | position j) of every covisible / loop-edge pair
  NdHostPlan hp;
};

constexpr int COVGPU_ERR_GATE_TIMEOUT = -1000;   // internal (solve_any): never returned through the C ABI
struct covgpu_group;
struct covgpu_context {
  int device = 0;
  hipStream_t st = nullptr;
  DevProblem P;
  bool have = false, pgo = false;
  std::vector<void*> allocs;
  size_t alloc_bytes = 0;  // device bytes behind `allocs` (the footprint covgpu_get_layout reports)
  double* h_scal = nullptr;  // pinned mirror of P.scal + flag
  double* h_tr = nullptr;    // pinned mirror of P.tr (device-side trust region)
  double* h_box = nullptr;   // pinned + mapped [TR_COUNT + 1]: the last kernel of an iteration posts P.tr and a sequence number here (k_tr_accept), the host polls it
  double* d_box = nullptr;   // its device address (nullptr: not available — D2H copy + stream synchronisation)
  double box_seq = 0.0;
  int profiling = 0;
  covgpu_profile_t prof;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  CholAux chol;
  PgoPlan pgo_plan;  // block-arrow pose-graph solve (k_pgo.hip)
  PlanCache plan_cache;  // elimination tree of the previous single-GPU GBA upload (reused when the new problem's couplings are a subset)
  NdDev nd;          // multifrontal GBA solve (k_front.hip)
  // agent-sharded solve: the global plan (variables as 2 * IR keyframe + kind, node -> rank), this rank's
  // identity and its collective
  bool sharded = false;
  int rank = 0, world = 1;
  std::shared_ptr<NdHostPlan> shard_plan;
  struct Reducer* reducer = nullptr;
  double* d_red = nullptr;     // [SC_COUNT + 2 world] scratch of the scalar all-reduce
  double cur_damp = 0.0;       // damping of the system being built (the top unknowns get theirs after the all-reduce)
  // a collective (or a scratch allocation of the linear solve) that failed while the iteration was being enqueued (broken / timed-out
  // group barrier, scratch hipMalloc, non-zero ncclAllReduce): latched here, checked after the iteration's host synchronisation —
  // the solve then returns an error instead of an estimate computed from un-reduced top fronts
  bool coll_failed = false;
  std::string coll_err;
  covgpu_group* group = nullptr;   // the in-process group this context's reducer belongs to (aborted when this rank gives up)
  int* d_pairkey = nullptr;    // [K] key of every keyframe in the covisible-pair numbering (chain position, -1: constant pose), kept for the second round of a call
  std::vector<int> h_perm;     // [K] keyframe -> chain position of the resident problem
  std::atomic<int>* peer_fail = nullptr;   // covgpu_gba_solve_multi: raised by any rank of the call that gave up; polled while waiting
};

// ---------------------------------------------------------------------------------------------------- collectives
// sum (op 0) / max (op 1) of n device doubles over all ranks, in place, ENQUEUED on the stream (no host synchronisation in
// the RCCL form). Two native forms: RCCL (one process per GPU, or several devices in one process) and an in-process group
// of host threads whose contexts share one device (virtual ranks: how the sharded path runs on a one-GPU test box).
struct Reducer {
  int rank = 0, world = 1;
  size_t calls = 0, bytes = 0;
  virtual ~Reducer() {}
  virtual int allreduce(double* dev, size_t n, int op, hipStream_t st) = 0;
  virtual void abort() {}                       // make every peer's pending and future collectives return
  virtual std::string last_error() { return "all-reduce failed"; }
};

struct GroupPtrs { double* p[16]; };
__global__ __launch_bounds__(256) void k_group_reduce(GroupPtrs g, int world, size_t n, int op, double* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    double v = g.p[0][i];
    for (int r = 1; r < world; ++r) v = op == 0 ? v + g.p[r][i] : fmax(v, g.p[r][i]);  // rank order: identical on every rank
    out[i] = v;
  }
}
struct covgpu_group {
  int world = 1;
  std::mutex m; std::condition_variable cv; int count = 0; long gen = 0; bool broken = false;
  GroupPtrs ptrs;
  bool wait() {  // barrier of the group's host threads; false if a member gave up
    std::unique_lock<std::mutex> lk(m);
    const long g0 = gen;
    if (++count == world) { count = 0; ++gen; cv.notify_all(); return !broken; }
    cv.wait_for(lk, std::chrono::seconds(600), [&] { return gen != g0 || broken; });
    if (gen == g0) { broken = true; cv.notify_all(); return false; }
    return !broken;
  }
};
struct GroupReducer : Reducer {
  covgpu_group* g = nullptr;
  double* scratch = nullptr; size_t scratch_n = 0;
  ~GroupReducer() override { if (scratch) (void)hipFree(scratch); }
  void abort() override { std::lock_guard<std::mutex> lk(g->m); g->broken = true; g->cv.notify_all(); }
  std::string last_error() override { return "in-process group all-reduce failed (a member gave up, timed out, or the scratch allocation failed)"; }
  int allreduce(double* dev, size_t n, int op, hipStream_t st) override {
    if (n == 0) return 0;
    if (scratch_n < n) { if (scratch) (void)hipFree(scratch); if (hipMalloc((void**)&scratch, n * sizeof(double)) != hipSuccess) return 1; scratch_n = n; }
    (void)hipStreamSynchronize(st);  // this rank's contribution is final
    { std::lock_guard<std::mutex> lk(g->m); g->ptrs.p[rank] = dev; }
    if (!g->wait()) return 1;
    hipLaunchKernelGGL(k_group_reduce, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, st, g->ptrs, world, n, op, scratch);
    (void)hipStreamSynchronize(st);
    if (!g->wait()) return 1;        // everybody has read everybody's buffer
    (void)hipMemcpyAsync(dev, scratch, n * sizeof(double), hipMemcpyDeviceToDevice, st);
    ++calls; bytes += n * sizeof(double);
    return 0;
  }
};

// RCCL, loaded at run time: the single-GPU path of libcovgpu has no dependency on it
struct RcclApi {
  void* h = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, struct RcclId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*CommAbort)(void*) = nullptr;
  int (*CommGetAsyncError)(void*, int*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
struct RcclId { char internal[128]; };   // ncclUniqueId (rccl.h:40-43)
static RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { api.h = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (api.h) break; }
    if (!api.h) return;
    api.GetUniqueId = (int (*)(void*))dlsym(api.h, "ncclGetUniqueId");
    api.CommInitRank = (int (*)(void**, int, RcclId, int))dlsym(api.h, "ncclCommInitRank");
    api.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(api.h, "ncclAllReduce");
    api.CommDestroy = (int (*)(void*))dlsym(api.h, "ncclCommDestroy");
    api.GetErrorString = (const char* (*)(int))dlsym(api.h, "ncclGetErrorString");
    api.CommAbort = (int (*)(void*))dlsym(api.h, "ncclCommAbort");
    api.CommGetAsyncError = (int (*)(void*, int*))dlsym(api.h, "ncclCommGetAsyncError");
    if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy) { dlclose(api.h); api.h = nullptr; }
  });
  return api.h ? &api : nullptr;
}
struct RcclReducer : Reducer {
  void* comm = nullptr;
  bool aborted = false;
  int last_rc = 0;
  ~RcclReducer() override { if (comm && rccl_api()) rccl_api()->CommDestroy(comm); }
  int allreduce(double* dev, size_t n, int op, hipStream_t st) override {
    if (n == 0) return 0;
    if (aborted) return 1;
    ++calls; bytes += n * sizeof(double);
    last_rc = rccl_api()->AllReduce(dev, dev, n, /*ncclFloat64*/ 8, op == 0 ? /*ncclSum*/ 0 : /*ncclMax*/ 2, comm, st);
    return last_rc;
  }
  // asynchronous error of the communicator (a peer died, a link failed): polled by the host while it waits for the iteration
  int async_error() {
    int e = 0;
    if (comm && rccl_api()->CommGetAsyncError && rccl_api()->CommGetAsyncError(comm, &e) == 0) return e;
    return 0;
  }
  void abort() override {   // tears the communicator down at once: collective kernels in flight on any stream of this rank return
    if (comm && rccl_api()->CommAbort) { rccl_api()->CommAbort(comm); comm = nullptr; }
    aborted = true;
  }
  std::string last_error() override {
    return std::string("RCCL all-reduce failed: ") + (rccl_api()->GetErrorString && last_rc ? rccl_api()->GetErrorString(last_rc) : "communicator aborted or asynchronous error");
  }
};

static void coll_latch(covgpu_context* c) {
  if (!c->coll_failed) { c->coll_failed = true; c->coll_err = c->reducer ? c->reducer->last_error() : "all-reduce failed"; }
}
static void ctx_reduce(void* vc, double* dev, size_t n, int op, hipStream_t st) {
  covgpu_context* c = (covgpu_context*)vc;
  if (c->reducer && n && c->reducer->allreduce(dev, n, op, st) != 0) coll_latch(c);
}
static void clear_shard(covgpu_context* c) {
  delete c->reducer; c->reducer = nullptr;
  c->coll_failed = false; c->coll_err.clear(); c->group = nullptr;
  c->sharded = false; c->rank = 0; c->world = 1; c->shard_plan.reset();
  c->chol.reduce = nullptr; c->chol.reduce_ctx = nullptr;
}
extern "C" int covgpu_set_shard_none(covgpu_context* c) { clear_shard(c); return COVGPU_OK; }

extern "C" int covgpu_group_create(int32_t world, covgpu_group** out) {
  if (world < 1 || world > 16) { g_err = "covgpu_group_create: world must be 1..16"; return COVGPU_ERR_INVALID_ARG; }
  covgpu_group* g = new covgpu_group(); g->world = world;
  for (auto& q : g->ptrs.p) q = nullptr;
  *out = g;
  return COVGPU_OK;
}
extern "C" void covgpu_group_destroy(covgpu_group* g) { delete g; }
extern "C" void covgpu_group_abort(covgpu_group* g) { std::lock_guard<std::mutex> lk(g->m); g->broken = true; g->cv.notify_all(); }

struct covgpu_nd_plan { NdHostPlan hp; std::vector<int> pos_kf; };
static int set_shard_common(covgpu_context* c, const covgpu_nd_plan* plan, int rank, int world, Reducer* red) {
  if (!plan || plan->hp.node_rank.empty()) { delete red; g_err = "covgpu_set_shard: the plan carries no rank assignment (use covgpu_shard_plan)"; return COVGPU_ERR_INVALID_ARG; }
  if (rank < 0 || rank >= world) { delete red; g_err = "covgpu_set_shard: rank out of range"; return COVGPU_ERR_INVALID_ARG; }
  for (int r : plan->hp.node_rank) if (r >= world) { delete red; g_err = "covgpu_set_shard: the plan was made for more ranks"; return COVGPU_ERR_INVALID_ARG; }
  clear_shard(c);
  // keep the plan in IR terms (variable = 2 * keyframe + kind): a rank's sub-problem orders its keyframes differently
  auto hp = std::make_shared<NdHostPlan>(plan->hp);
  std::vector<int> to_ir(2 * (size_t)hp->K);
  for (int q = 0; q < hp->K; ++q) { to_ir[2 * q] = 2 * plan->pos_kf[q]; to_ir[2 * q + 1] = 2 * plan->pos_kf[q] + 1; }
  nd_plan_remap(*hp, to_ir);
  c->shard_plan = hp;
  red->rank = rank; red->world = world;
  c->reducer = red; c->sharded = true; c->rank = rank; c->world = world;
  c->chol.reduce = ctx_reduce; c->chol.reduce_ctx = c;
  return COVGPU_OK;
}
extern "C" int covgpu_set_shard_group(covgpu_context* c, const covgpu_nd_plan* plan, int32_t rank, covgpu_group* g) {
  if (!g) { g_err = "covgpu_set_shard_group: NULL group"; return COVGPU_ERR_INVALID_ARG; }
  GroupReducer* r = new GroupReducer(); r->g = g;
  const int rc = set_shard_common(c, plan, rank, g->world, r);
  if (rc == COVGPU_OK) c->group = g;
  return rc;
}
extern "C" int covgpu_rccl_unique_id(uint8_t* out128) {
  RcclApi* api = rccl_api();
  if (!api) { g_err = "librccl could not be loaded"; return COVGPU_ERR_NO_DEVICE; }
  RcclId id;
  const int rc = api->GetUniqueId(&id);
  if (rc != 0) { g_err = std::string("ncclGetUniqueId: ") + (api->GetErrorString ? api->GetErrorString(rc) : "error"); return COVGPU_ERR_NO_DEVICE; }
  std::memcpy(out128, id.internal, 128);
  return COVGPU_OK;
}
extern "C" int covgpu_set_shard_rccl(covgpu_context* c, const covgpu_nd_plan* plan, int32_t rank, int32_t world, const uint8_t* id128) {
  RcclApi* api = rccl_api();
  if (!api) { g_err = "librccl could not be loaded"; return COVGPU_ERR_NO_DEVICE; }
  HIPCHK(hipSetDevice(c->device));
  RcclId id; std::memcpy(id.internal, id128, 128);
  RcclReducer* r = new RcclReducer();
  const int rc = api->CommInitRank(&r->comm, world, id, rank);
  if (rc != 0) { delete r; g_err = std::string("ncclCommInitRank: ") + (api->GetErrorString ? api->GetErrorString(rc) : "error"); return COVGPU_ERR_NO_DEVICE; }
  return set_shard_common(c, plan, rank, world, r);
}
// host convenience through the context's collective (barriers / timing aggregates of a multi-process run): in place
extern "C" int covgpu_allreduce_host(covgpu_context* c, double* host, int64_t n, int32_t op) {
  if (!c->reducer || n <= 0) return COVGPU_OK;
  HIPCHK(hipSetDevice(c->device));
  double* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, n * sizeof(double)));
  HIPCHK(hipMemcpyAsync(d, host, n * sizeof(double), hipMemcpyHostToDevice, c->st));
  const int rc = c->reducer->allreduce(d, (size_t)n, op, c->st);
  HIPCHK(hipMemcpyAsync(host, d, n * sizeof(double), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  (void)hipFree(d);
  if (rc != 0) { g_err = "all-reduce failed"; return COVGPU_ERR_NO_DEVICE; }
  return COVGPU_OK;
}
// out[4] = { collectives issued, bytes all-reduced (per rank), rank, world }
extern "C" void covgpu_shard_stats(covgpu_context* c, int64_t* out) {
  out[0] = c->reducer ? (int64_t)c->reducer->calls : 0; out[1] = c->reducer ? (int64_t)c->reducer->bytes : 0; out[2] = c->rank; out[3] = c->world;
}

extern "C" void covgpu_default_options(covgpu_options* o) {
  std::memset(o, 0, sizeof(*o));
  o->strategy = COVGPU_DOGLEG;   // optimization_be.cpp:564
  o->max_iterations = 10;        // config_backend.yaml:115  opt.gba_iteration_limit
  o->reproj_loss_a = 1.0;        // optimization_be.cpp:302
  o->initial_radius = 1e4; o->max_radius = 1e16; o->min_relative_decrease = 1e-3;
  o->function_tolerance = 1e-6; o->parameter_tolerance = 1e-8; o->gradient_tolerance = 1e-10;
  const double sf = std::sqrt(200.0);  // EuRoC.yaml:40-44 at 200 Hz (orb_slam3/src/Tracking.cc:1203-1211)
  o->sigma_g = 1.7e-4 * sf; o->sigma_a = 2.0e-3 * sf; o->sigma_gw = 1.9393e-5 / sf; o->sigma_aw = 3.0e-3 / sf;
  o->gravity = 9.81;             // orb_slam3/include/ImuTypes.h:43
}

extern "C" int32_t covgpu_reduced_dim(const covgpu_options* opt, const covgpu_problem* p) {
  return (opt->visual_only ? 6 : 15) * p->num_kf;
}

extern "C" int32_t covgpu_pgo_partition(int32_t num_kf, int32_t num_edge, const int32_t* edge_i, const int32_t* edge_j, int32_t* block_of_kf) {
  for (int k = 0; k < num_kf; ++k) block_of_kf[k] = -1;
  for (int e = 0; e < num_edge; ++e)
    if (edge_i[e] < 0 || edge_i[e] >= num_kf || edge_j[e] < 0 || edge_j[e] >= num_kf) return 0;
  PgoHostPlan hp;
  if (!pgo_plan_analyse(num_kf, num_edge, edge_i, edge_j, hp)) return 0;
  for (size_t a = 0; a < hp.block_kf.size(); ++a)
    for (int kf : hp.block_kf[a]) block_of_kf[kf] = (int32_t)a;
  return (int32_t)hp.block_kf.size();
}

extern "C" int covgpu_create(const covgpu_options* opt, covgpu_context** out) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    g_err = "no HIP device visible: libcovgpu has no CPU fallback";
    return COVGPU_ERR_NO_DEVICE;
  }
  covgpu_context* c = new covgpu_context();
  c->device = opt ? opt->device : 0;
  if (c->device < 0 || c->device >= ndev) { g_err = "device ordinal out of range"; delete c; return COVGPU_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  {
    // the main stream carries the serial panel chain of the factorisation: give it dispatch priority over the
    // auxiliary stream's bulk trailing updates (k_chol.hip look-ahead)
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    HIPCHK(hipStreamCreateWithPriority(&c->st, hipStreamDefault, hi));
  }
  HIPCHK(hipHostMalloc((void**)&c->h_scal, (SC_COUNT + 4) * sizeof(double), hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&c->h_tr, TR_COUNT * sizeof(double), hipHostMallocDefault));
  {
    static const bool box_on = getenv("COVGPU_MAILBOX") == nullptr || atoi(getenv("COVGPU_MAILBOX")) != 0;
    if (box_on && hipHostMalloc((void**)&c->h_box, (TR_COUNT + 1) * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
      std::memset(c->h_box, 0, (TR_COUNT + 1) * sizeof(double));
      if (hipHostGetDevicePointer((void**)&c->d_box, c->h_box, 0) != hipSuccess) c->d_box = nullptr;
    } else { (void)hipGetLastError(); c->h_box = nullptr; }
  }
  for (auto& e : c->ev) HIPCHK(hipEventCreate(&e));
  std::memset(&c->P, 0, sizeof(c->P));
  *out = c;
  return COVGPU_OK;
}

static void free_problem(covgpu_context* c) {
  for (void* p : c->allocs) (void)hipFree(p);
  c->allocs.clear();
  c->alloc_bytes = 0;
  c->chol.tri_clear();
  c->have = false;
  c->pgo_plan.active = false;  // its device buffers were in `allocs`
  c->d_pairkey = nullptr;
  c->nd = NdDev();
}

extern "C" void covgpu_destroy(covgpu_context* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  free_problem(c);
  for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
  c->chol.destroy();
  if (c->h_scal) (void)hipHostFree(c->h_scal);
  if (c->h_tr) (void)hipHostFree(c->h_tr);
  if (c->h_box) (void)hipHostFree(c->h_box);
  clear_shard(c);
  if (c->st) (void)hipStreamDestroy(c->st);
  delete c;
}

extern "C" void covgpu_set_profiling(covgpu_context* c, int on) {
  c->profiling = on; c->prof = covgpu_profile_t();
  c->chol.profile = on != 0; c->chol.syrk_ms = 0; c->chol.syrk_flops = 0; c->chol.n_syrk = 0;
  c->chol.potrf_ms = 0; c->chol.potrf_flops = 0; c->chol.n_potrf = 0;
}
// out[0..7] = build ms, n_build, factor ms, n_factor, syrk ms, n_syrk launches, syrk flops, tri-solve ms
extern "C" void covgpu_get_profile(covgpu_context* c, double* out) {
  out[0] = c->prof.t_build_ms; out[1] = (double)c->prof.n_build; out[2] = c->prof.t_factor_ms; out[3] = (double)c->prof.n_factor;
  out[4] = c->chol.syrk_ms; out[5] = (double)c->chol.n_syrk; out[6] = c->chol.syrk_flops;
  out[7] = (double)(c->have ? c->P.npairs + c->P.nepairs : 0);  // off-diagonal 6x6 pose-pose blocks of the reduced system (nnzS - K)
}
// out[0..7] as covgpu_get_profile; out[8] = k_potrf_panel ms (sum over its launches), [9] = its launches, [10] = its algorithmic flops,
// [11] = flops of ONE multifrontal factorisation of the resident problem's plan (Cholesky of every front incl. its border updates),
// [12..15] = 0
extern "C" void covgpu_get_profile2(covgpu_context* c, double* out) {
  covgpu_get_profile(c, out);
  out[8] = c->chol.potrf_ms; out[9] = (double)c->chol.n_potrf; out[10] = c->chol.potrf_flops; out[11] = c->have ? c->nd.plan_flops : 0.0;
  out[12] = out[13] = out[14] = out[15] = 0.0;
}

// out[16] = { shard world, shard rank, scalar unknowns of the replicated top, top levels, KiB all-reduced per linear solve, stream ordering (1 flags | 0 events | -1 fell back: events and no in-launch hand-overs),
//             dense order npad, covisible pairs, edge pairs, IMU chains, device MiB allocated for the problem,
//             fronts, levels, serial 256-column panels, order of the root level, MiB of fronts } (include/covgpu.h)
extern "C" void covgpu_get_layout(covgpu_context* c, int64_t* out) {
  for (int i = 0; i < 16; ++i) out[i] = 0;
  if (!c->have) return;
  const DevProblem& P = c->P;
  out[0] = c->sharded ? c->world : 0; out[1] = c->sharded ? c->rank : 0; out[2] = c->nd.ntop;   // ranks | rank | scalar unknowns of the replicated top nodes
  out[3] = (int64_t)c->nd.lev.size() - c->nd.top_lev0;                                          // top levels
  out[4] = (int64_t)(((c->nd.top_pack != nullptr ? (size_t)c->nd.n_top_tiles * kTile * kTile : c->nd.M_elems - c->nd.M_sub) + c->nd.rhs_top + 2 * (size_t)c->nd.ntop) * sizeof(double)) >> 10;  // KiB all-reduced per linear solve
  out[5] = c->chol.pipe_broken ? -1 : c->chol.gates_on ? 1 : 0;   // stream ordering: 1 device flags | 0 HIP events (COVGPU_GATES=0) | -1 a gate or the backward pipeline timed out: events and the launch-per-tile substitution from then on
  out[6] = P.npad; out[7] = P.npairs; out[8] = P.nepairs; out[9] = P.nchains; out[10] = (int64_t)(c->alloc_bytes >> 20);
  if (P.nd) {  // multifrontal form: nodes, levels, serial 256-column panels (sum of the levels' interior orders / 256), root order, front bytes (MiB)
    out[11] = P.nd_nnodes; out[12] = P.nd_nlev;
    for (const NdLevel& L : c->nd.lev) out[13] += L.nI / 256;
    out[14] = c->nd.lev.empty() ? 0 : c->nd.lev.back().nI; out[15] = (int64_t)((c->nd.M_elems * sizeof(double)) >> 20);
  }
}

template <typename T>
static int dev_alloc(covgpu_context* c, T** ptr, size_t count) {
  *ptr = nullptr;
  if (count == 0) count = 1;
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, count * sizeof(T));
  if (e != hipSuccess) { g_err = std::string("hipMalloc: ") + hipGetErrorString(e); return COVGPU_ERR_OUT_OF_MEMORY; }
  c->allocs.push_back(p);
  c->alloc_bytes += count * sizeof(T);
  *ptr = (T*)p;
  return COVGPU_OK;
}
template <typename T>
static int dev_upload(covgpu_context* c, T** ptr, const T* host, size_t count) {
  int rc = dev_alloc(c, ptr, count);
  if (rc) return rc;
  if (count && host) HIPCHK(hipMemcpyAsync(*ptr, host, count * sizeof(T), hipMemcpyHostToDevice, c->st));
  else if (count) HIPCHK(hipMemsetAsync(*ptr, 0, count * sizeof(T), c->st));
  return COVGPU_OK;
}
#define RC(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)
// no C++ exception may cross the extern "C" boundary (std::bad_alloc from the host staging vectors, std::system_error from
// std::thread): map them to status codes
template <typename F>
static int guarded(F&& body) {
  try { return body(); }
  catch (const std::bad_alloc&) { g_err = "host allocation failed"; return COVGPU_ERR_OUT_OF_MEMORY; }
  catch (const std::exception& e) { g_err = std::string("host exception: ") + e.what(); return COVGPU_ERR_INVALID_ARG; }
}

static int validate(const covgpu_problem* p, bool pgo, bool vi) {
  auto bad = [](const char* m) { g_err = std::string("invalid problem: ") + m; return COVGPU_ERR_INVALID_ARG; };
  if (!p || p->num_kf <= 0) return bad("no keyframes");
  if (p->num_cam < 0 || p->num_lm < 0 || p->num_obs < 0 || p->num_imu < 0 || p->num_edge < 0 || p->num_imu_samples < 0) return bad("negative count");
  if (!p->kf_pose || !p->kf_fixed || !p->kf_cam) return bad("NULL keyframe array");
  if (vi && !p->kf_speed_bias) return bad("NULL speed-bias array in visual-inertial mode");
  const int K = p->num_kf;
  if (!pgo) {
    if (p->num_lm > 0 && (!p->lm_pos || !p->lm_obs_ptr)) return bad("NULL landmark array");
    if (p->num_obs > 0 && (!p->obs_kf || !p->obs_uv || !p->obs_sigma)) return bad("NULL observation array");
    if (p->num_lm > 0 && (p->lm_obs_ptr[0] != 0 || p->lm_obs_ptr[p->num_lm] != p->num_obs)) return bad("lm_obs_ptr does not span the observations");
    for (int l = 0; l < p->num_lm; ++l) if (p->lm_obs_ptr[l + 1] < p->lm_obs_ptr[l]) return bad("lm_obs_ptr not monotone");
    for (int o = 0; o < p->num_obs; ++o) if (p->obs_kf[o] < 0 || p->obs_kf[o] >= K) return bad("obs_kf out of range");
    if (p->num_cam <= 0 || !p->cam_extr || !p->cam_intr || !p->cam_dist || !p->cam_dist_type) return bad("NULL camera array");
    for (int k = 0; k < K; ++k) if (p->kf_cam[k] < 0 || p->kf_cam[k] >= p->num_cam) return bad("kf_cam out of range");
    for (int a = 0; a < p->num_cam; ++a) if (p->cam_dist_type[a] != COVGPU_DIST_RADTAN && p->cam_dist_type[a] != COVGPU_DIST_EQUIDISTANT) return bad("unknown distortion type");
    if (vi && p->num_imu > 0 && (!p->imu_kf_i || !p->imu_kf_j || !p->imu_sample_ptr || !p->imu_first)) return bad("NULL IMU array");
    if (vi && p->num_imu > 0 && p->imu_sample_ptr[p->num_imu] > 0 && !p->imu_samples) return bad("NULL IMU sample array");
    if (vi && p->num_imu > 0 && (p->imu_sample_ptr[0] != 0 || p->imu_sample_ptr[p->num_imu] != p->num_imu_samples)) return bad("imu_sample_ptr does not span the IMU samples");
    if (vi) for (int f = 0; f < p->num_imu; ++f) {
      if (p->imu_kf_i[f] < 0 || p->imu_kf_i[f] >= K || p->imu_kf_j[f] < 0 || p->imu_kf_j[f] >= K) return bad("imu keyframe out of range");
      if (p->imu_sample_ptr[f + 1] < p->imu_sample_ptr[f]) return bad("imu_sample_ptr not monotone");
    }
  }
  if (p->num_edge > 0 && (!p->edge_i || !p->edge_j || !p->edge_meas || !p->edge_sqrt_info || !p->edge_loss_a)) return bad("NULL edge array");
  for (int e = 0; e < p->num_edge; ++e)
    if (p->edge_i[e] < 0 || p->edge_i[e] >= K || p->edge_j[e] < 0 || p->edge_j[e] >= K) return bad("edge keyframe out of range");
  return COVGPU_OK;
}

// IMU chains (one per agent: predecessor -> successor factors, optimization_be.cpp:369-416) laid out back to back:
// perm[kf] = chain-major position, pos_kf = inverse, chain_ptr = positions of each chain. Without IMU factors every
// keyframe is its own position (identity order, one "chain").
static int build_chains(const covgpu_problem* p, bool vi, std::vector<int>& perm, std::vector<int>& pos_kf, std::vector<int>& chain_ptr) {
  const int K = p->num_kf, I = vi ? p->num_imu : 0;
  perm.assign(K, 0); pos_kf.assign(K, 0); chain_ptr.assign(1, 0);
  if (vi) {
    std::vector<int> succ(K, -1), has_pred(K, 0);
    for (int f = 0; f < I; ++f) {
      const int i = p->imu_kf_i[f], j = p->imu_kf_j[f];
      if (i == j || succ[i] != -1 || has_pred[j]) { g_err = "invalid problem: IMU factors must form simple predecessor chains"; return COVGPU_ERR_INVALID_ARG; }
      succ[i] = j; has_pred[j] = 1;
    }
    int pos = 0;
    for (int k = 0; k < K; ++k) {
      if (has_pred[k]) continue;
      for (int c = k; c != -1; c = succ[c]) { perm[c] = pos; pos_kf[pos] = c; ++pos; }
      chain_ptr.push_back(pos);
    }
    if (pos != K) { g_err = "invalid problem: IMU factors contain a cycle"; return COVGPU_ERR_INVALID_ARG; }
  } else {
    for (int k = 0; k < K; ++k) { perm[k] = k; pos_kf[k] = k; }
    chain_ptr.push_back(K);
  }
  return COVGPU_OK;
}

// Pose graph (round 6: the multifrontal solve of k_front.hip serves PoseGraphOptimization too): there are no IMU factors to read an agent's
// time axis from, and the map hands keyframes over agent-interleaved (typedefs_base.hpp:178) — index distance says nothing. The edge list does:
// an edge whose endpoints share no neighbour is a bridge between otherwise unrelated parts of the graph, a loop closure
// (optimization_be.cpp:912-944); the odometry edges of one agent (every keyframe tied to its five predecessors, :947-1021) always share
// neighbours. "Chains" = connected components of the graph without its bridges, each laid out in breadth-first order from a pseudo-peripheral
// keyframe — along an odometry chain that IS the time axis up to a few places, and nd_plan's halving of a chain then cuts ~5 keyframes.
static void build_chains_pgo(const covgpu_problem* p, std::vector<int>& perm, std::vector<int>& pos_kf, std::vector<int>& chain_ptr) {
  const int K = p->num_kf, E = p->num_edge;
  perm.assign(K, 0); pos_kf.assign(K, 0); chain_ptr.assign(1, 0);
  std::vector<std::vector<int>> adj(K);
  for (int e = 0; e < E; ++e) if (p->edge_i[e] != p->edge_j[e]) { adj[p->edge_i[e]].push_back(p->edge_j[e]); adj[p->edge_j[e]].push_back(p->edge_i[e]); }
  for (auto& a : adj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); }
  auto bridge = [&](int u, int v) {
    const std::vector<int>&A = adj[u], &B = adj[v];
    for (size_t x = 0, y = 0; x < A.size() && y < B.size();) { if (A[x] == B[y]) return false; if (A[x] < B[y]) ++x; else ++y; }
    return A.size() > 1 || B.size() > 1;
  };
  std::vector<std::vector<int>> nb(K);   // neighbours over non-bridge edges
  for (int u = 0; u < K; ++u) for (int v : adj[u]) if (!bridge(u, v)) nb[u].push_back(v);
  std::vector<int> level(K, -1), queue;
  std::vector<char> placed(K, 0);
  auto bfs = [&](int start) {
    queue.assign(1, start); level[start] = 0;
    for (size_t h = 0; h < queue.size(); ++h) { const int u = queue[h]; for (int v : nb[u]) if (level[v] < 0) { level[v] = level[u] + 1; queue.push_back(v); } }
    return queue.back();
  };
  int pos = 0;
  for (int k = 0; k < K; ++k) {
    if (placed[k]) continue;
    const int far = bfs(k);
    for (int v : queue) level[v] = -1;
    bfs(far);   // second sweep from a pseudo-peripheral keyframe: its breadth-first order is the layout
    for (int v : queue) { level[v] = -1; placed[v] = 1; perm[v] = pos; pos_kf[pos] = v; ++pos; }
    chain_ptr.push_back(pos);
  }
}

// unique covisible pairs (free keyframes only) and edge pairs, as chain-major positions i > j (host-only plan functions;
// upload_impl builds the same pair list with the per-pair observation lists the device needs)
static bool host_pairs(const covgpu_problem* p, const std::vector<int>& perm, std::vector<int>& pi, std::vector<int>& pj, std::vector<int>& ei,
                       std::vector<int>& ej) {
  std::vector<long long> keys;
  for (int l = 0; l < p->num_lm; ++l)
    for (int a = p->lm_obs_ptr[l]; a < p->lm_obs_ptr[l + 1]; ++a) {
      if (p->kf_fixed[p->obs_kf[a]]) continue;
      for (int b = p->lm_obs_ptr[l]; b < p->lm_obs_ptr[l + 1]; ++b) {
        if (p->kf_fixed[p->obs_kf[b]]) continue;
        const int pa = perm[p->obs_kf[a]], pb = perm[p->obs_kf[b]];
        if (pb < pa) keys.push_back(((long long)pa << 32) | (unsigned)pb);
      }
    }
  std::sort(keys.begin(), keys.end()); keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
  pi.resize(keys.size()); pj.resize(keys.size()); ei.clear(); ej.clear();
  for (size_t q = 0; q < keys.size(); ++q) { pi[q] = (int)(keys[q] >> 32); pj[q] = (int)(keys[q] & 0xffffffffll); }
  for (int e = 0; e < p->num_edge; ++e) {
    const int a = perm[p->edge_i[e]], b = perm[p->edge_j[e]];
    if (a == b) return false;
    ei.push_back(std::max(a, b)); ej.push_back(std::min(a, b));
  }
  return true;
}

// Host-only: the nested-dissection plan of the reduced camera system (nd_plan.hpp, k_front.hip) for inspection and the CPU
// tests (tests/test_nd_plan.py replays the elimination in numpy on the oracle's system).
static int nd_leaf_dims(int requested) {   // 0: nd_plan_build chooses among its candidates (nd_plan.hip)
  if (requested > 0) return requested;
  const char* e = getenv("COVGPU_ND_LEAF");
  const int v = e ? atoi(e) : 0;
  return v > 0 ? v : 0;
}
static int nd_plan_create_mode(const covgpu_options* opt, const covgpu_problem* p, int32_t leaf_dims, covgpu_nd_plan** out, int top_mode);
extern "C" int covgpu_nd_plan_create(const covgpu_options* opt, const covgpu_problem* p, int32_t leaf_dims, covgpu_nd_plan** out) { return nd_plan_create_mode(opt, p, leaf_dims, out, -1); }
static int nd_plan_create_mode(const covgpu_options* opt, const covgpu_problem* p, int32_t leaf_dims, covgpu_nd_plan** out, int top_mode) {
  return guarded([&] {
    const bool vi = !opt->visual_only;
    *out = nullptr;
    RC(validate(p, false, vi));
    std::vector<int> perm, pos_kf, chain_ptr, pi, pj, ei, ej;
    RC(build_chains(p, vi, perm, pos_kf, chain_ptr));
    if (!host_pairs(p, perm, pi, pj, ei, ej)) { g_err = "invalid problem: self edge"; return (int)COVGPU_ERR_INVALID_ARG; }
    covgpu_nd_plan* pl = new covgpu_nd_plan();
    pl->pos_kf = pos_kf;
    const auto t_pl = std::chrono::steady_clock::now();
    struct PT { std::chrono::steady_clock::time_point t0; ~PT() { if (getenv("COVGPU_PLAN_TIMING")) std::fprintf(stderr, "[covgpu] nd_plan_build %.1f ms\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3); } } pt{t_pl};
    if (!nd_plan_build(p->num_kf, vi, (int)chain_ptr.size() - 1, chain_ptr.data(), (int)pi.size(), pi.data(), pj.data(), (int)ei.size(), ei.data(), ej.data(),
                       nd_leaf_dims(leaf_dims), pl->hp, top_mode)) {
      delete pl; g_err = "nested-dissection plan: a coupling joins two branches"; return (int)COVGPU_ERR_INVALID_ARG;
    }
    *out = pl;
    return (int)COVGPU_OK;
  });
}
// Host-only: the plan PoseGraphOptimization's solve runs on (round 6): 6-dof blocks, chains read from the edge graph (build_chains_pgo), couplings = the edge pairs
extern "C" int covgpu_nd_plan_create_pgo(const covgpu_options* opt, const covgpu_problem* p, int32_t leaf_dims, covgpu_nd_plan** out) {
  (void)opt;
  return guarded([&] {
    *out = nullptr;
    RC(validate(p, true, false));
    std::vector<int> perm, pos_kf, chain_ptr;
    build_chains_pgo(p, perm, pos_kf, chain_ptr);
    std::vector<uint64_t> keys;
    for (int e = 0; e < p->num_edge; ++e) {
      const int a = perm[p->edge_i[e]], b = perm[p->edge_j[e]];
      if (a == b) { g_err = "invalid problem: self edge"; return (int)COVGPU_ERR_INVALID_ARG; }
      keys.push_back(((uint64_t)(uint32_t)std::max(a, b) << 32) | (uint32_t)std::min(a, b));
    }
    std::sort(keys.begin(), keys.end()); keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    std::vector<int> ei(keys.size()), ej(keys.size());
    for (size_t q = 0; q < keys.size(); ++q) { ei[q] = (int)(keys[q] >> 32); ej[q] = (int)(keys[q] & 0xffffffffull); }
    covgpu_nd_plan* pl = new covgpu_nd_plan();
    pl->pos_kf = pos_kf;
    if (!nd_plan_build(p->num_kf, false, (int)chain_ptr.size() - 1, chain_ptr.data(), 0, nullptr, nullptr, (int)ei.size(), ei.data(), ej.data(), nd_leaf_dims(leaf_dims), pl->hp, -1)) {
      delete pl; g_err = "nested-dissection plan: a coupling joins two branches"; return (int)COVGPU_ERR_INVALID_ARG;
    }
    *out = pl;
    return (int)COVGPU_OK;
  });
}
extern "C" void covgpu_nd_plan_destroy(covgpu_nd_plan* pl) { delete pl; }
// out[16] = { nodes, levels, depth, own entries, front-structure entries, front elements (all batches), flops, largest own dims,
//             largest border dims, root own dims, 0 ... }
extern "C" void covgpu_nd_plan_info(const covgpu_nd_plan* pl, int64_t* out) {
  for (int i = 0; i < 16; ++i) out[i] = 0;
  const NdHostPlan& h = pl->hp;
  out[0] = h.nnodes; out[1] = h.nlev; out[2] = h.maxdepth;
  for (int n = 0; n < h.nnodes; ++n) {
    out[3] += (int64_t)h.own[n].size(); out[4] += (int64_t)h.strct[n].size();
    out[7] = std::max<int64_t>(out[7], h.own_dims[n]); out[8] = std::max<int64_t>(out[8], h.st_dims[n]);
    if (h.parent[n] < 0) out[9] = std::max<int64_t>(out[9], h.own_dims[n]);
  }
  out[5] = (int64_t)h.front_elems; out[6] = (int64_t)h.flops;
  out[10] = h.top_mode; out[11] = h.leaf; out[12] = h.group_frac100;   // which candidate tree (nd_plan_build): COVGPU_ND_TOP / COVGPU_ND_LEAF / COVGPU_ND_GROUP_FRAC (= out[12] / 100) reproduce it
}
// per node: parent, level, own_ptr / st_ptr [nodes + 1]; variables as 2 * IR keyframe + (0 pose | 1 speed-bias)
extern "C" void covgpu_nd_plan_arrays(const covgpu_nd_plan* pl, int32_t* parent, int32_t* level, int32_t* own_ptr, int32_t* own_var, int32_t* st_ptr,
                                      int32_t* st_var) {
  const NdHostPlan& h = pl->hp;
  int o = 0, s = 0;
  auto ir = [&](int v) { return 2 * pl->pos_kf[v >> 1] + (v & 1); };
  for (int n = 0; n < h.nnodes; ++n) {
    parent[n] = h.parent[n]; level[n] = h.level[n]; own_ptr[n] = o; st_ptr[n] = s;
    for (int v : h.own[n]) own_var[o++] = ir(v);
    for (int v : h.strct[n]) st_var[s++] = ir(v);
  }
  own_ptr[h.nnodes] = o; st_ptr[h.nnodes] = s;
}

// Host-only: the multi-GPU split of ONE map = the nested-dissection plan of the FULL problem
// plus, per tree node, the rank that owns it (nd_shard_assign: top of the tree replicated, subtrees dealt to the ranks), and
// the owner of every landmark / IMU factor / between factor: the rank of the DEEPEST tree node among the unknowns the
// residual touches (those unknowns form a clique of the coupling graph, so they lie on one root path: everything a residual
// contributes lands in fronts of its owner's subtrees or in top fronts — the only part that is all-reduced); residuals that
// touch top unknowns only are dealt round-robin. Returns the number of subtrees (0: nothing to split).
extern "C" int32_t covgpu_shard_plan(const covgpu_options* opt, const covgpu_problem* p, int32_t world, covgpu_nd_plan** plan_out, int32_t* lm_rank,
                                     int32_t* imu_rank, int32_t* edge_rank) {
  *plan_out = nullptr;
  if (world < 1) return 0;
  // Candidates (round 5): the tree with ONE separator of all agents at the top (its children, one region per agent, are the subtrees) and the tree with two
  // groups of agents at the top, each with the replicated top capped at 48 MiB (round 4's cap) and at 512 MiB of fronts; the cheapest by nd_shard_cost —
  // (replicated top + busiest rank's subtrees) at 30 TFLOP/s + the panel chains + the ring all-reduce of the top — is kept. On the corrected 5-agent map the
  // one-separator top is 3 726 unknowns = 61 % of the flops on every rank; the two-groups tree gives two ranks a 1 968-order top (10 %) and, with its two
  // second-level separators opened, four ranks a top of 45 %. COVGPU_SHARD_TREE = 0 / 1 forces the tree, COVGPU_SHARD_CAP_MIB the cap. Deterministic.
  covgpu_nd_plan* pl = nullptr;
  {
    const char* e_tree = getenv("COVGPU_SHARD_TREE");
    const char* e_cap = getenv("COVGPU_SHARD_CAP_MIB");
    std::vector<int> modes = e_tree ? std::vector<int>{atoi(e_tree) != 0 ? 1 : 0} : std::vector<int>{0, 1};
    std::vector<double> caps = e_cap ? std::vector<double>{atof(e_cap)} : std::vector<double>{48.0, 512.0};
    double best_cost = 0.0;
    for (int mode : modes) {
      covgpu_nd_plan* cand = nullptr;
      if (nd_plan_create_mode(opt, p, 0, &cand, mode) != COVGPU_OK) continue;
      for (double cap : caps) {
        NdHostPlan trial = cand->hp;
        nd_shard_assign(trial, world, cap * 1048576.0);
        if (trial.nsub == 0) continue;
        const double cost = nd_shard_cost(trial, world);
        if (pl == nullptr || cost < best_cost) {
          if (pl == nullptr) { pl = new covgpu_nd_plan(); pl->pos_kf = cand->pos_kf; }
          pl->hp = std::move(trial); best_cost = cost;
        }
      }
      delete cand;
    }
    if (pl == nullptr) return 0;
  }
  NdHostPlan& hp = pl->hp;
  const bool vi = !opt->visual_only;
  std::vector<int> perm(p->num_kf);
  for (int q = 0; q < p->num_kf; ++q) perm[pl->pos_kf[q]] = q;
  // owner of a set of variables (position-based ids): rank of the deepest node, -1 if they are all top
  auto owner = [&](std::initializer_list<int> vars) {
    int best = -1, depth = -1;
    for (int v : vars) { const int n = hp.vnode[v]; if (n >= 0 && hp.depth[n] > depth) { depth = hp.depth[n]; best = n; } }
    return best < 0 ? -1 : hp.node_rank[best];
  };
  for (int l = 0; l < p->num_lm; ++l) {
    int best = -1, depth = -1;
    for (int o = p->lm_obs_ptr[l]; o < p->lm_obs_ptr[l + 1]; ++o) {
      const int kf = p->obs_kf[o];
      if (p->kf_fixed[kf]) continue;  // (a constant keyframe has no pose unknown: it binds nobody)
      const int n = hp.vnode[2 * perm[kf]];
      if (hp.depth[n] > depth) { depth = hp.depth[n]; best = n; }
    }
    const int r = best < 0 ? -1 : hp.node_rank[best];
    lm_rank[l] = r >= 0 ? r : l % world;
  }
  for (int f = 0; f < p->num_imu; ++f) {
    const int a = perm[p->imu_kf_i[f]], b = perm[p->imu_kf_j[f]];
    const int r = vi ? owner({2 * a, 2 * a + 1, 2 * b, 2 * b + 1}) : -1;
    imu_rank[f] = r >= 0 ? r : f % world;
  }
  for (int e = 0; e < p->num_edge; ++e) {
    const int r = owner({2 * perm[p->edge_i[e]], 2 * perm[p->edge_j[e]]});
    edge_rank[e] = r >= 0 ? r : e % world;
  }
  *plan_out = pl;
  return hp.nsub;
}
extern "C" void covgpu_nd_plan_ranks(const covgpu_nd_plan* pl, int32_t* node_rank) {  // per tree node: owning rank, -1 = top (replicated); all 0 without an assignment
  for (int n = 0; n < pl->hp.nnodes; ++n) node_rank[n] = pl->hp.node_rank.empty() ? 0 : pl->hp.node_rank[n];
}
// after a sharded solve: which rank's download holds keyframe k's pose / speed-bias (-1: a top unknown — identical on every rank)
extern "C" void covgpu_nd_plan_owner(const covgpu_nd_plan* pl, int32_t* pose_rank, int32_t* sb_rank) {
  const NdHostPlan& hp = pl->hp;
  for (int q = 0; q < hp.K; ++q) {
    const int kf = pl->pos_kf[q];
    const int np = hp.vnode[2 * q], ns = hp.vnode[2 * q + 1];
    pose_rank[kf] = (np >= 0 && !hp.node_rank.empty()) ? hp.node_rank[np] : -1;
    sb_rank[kf] = (ns >= 0 && !hp.node_rank.empty()) ? hp.node_rank[ns] : -1;
  }
}

static int upload_impl(covgpu_context* c, const covgpu_options* opt, const covgpu_problem* p, bool pgo, bool allow_arrow = true) {
  HIPCHK(hipSetDevice(c->device));
  // dev aid (COVGPU_PLAN_TIMING=1): where the host side of an upload goes
  static const bool tm_on = getenv("COVGPU_PLAN_TIMING") != nullptr;
  auto tm_last = std::chrono::steady_clock::now();
  auto tm = [&](const char* what) {
    if (!tm_on) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[covgpu] upload: %-28s %.2f ms\n", what, std::chrono::duration<double>(now - tm_last).count() * 1e3);
    tm_last = now;
  };
  const bool vi = !pgo && !opt->visual_only;
  RC(validate(p, pgo, vi));
  free_problem(c);
  DevProblem& P = c->P;
  std::memset(&P, 0, sizeof(P));
  P.K = p->num_kf; P.A = p->num_cam;
  P.L = pgo ? 0 : p->num_lm; P.O = pgo ? 0 : p->num_obs;
  P.I = vi ? p->num_imu : 0; P.E = p->num_edge;
  P.S = P.I ? p->imu_sample_ptr[P.I] : 0;
  P.vi = vi; P.D = vi ? 15 : 6; P.n = P.D * P.K;
  {  // lanes per landmark of the landmark-major kernels, from the mean track length (COVGPU_LM_GROUP forces 4 / 8 / 16)
    const double mean_track = P.L > 0 ? (double)P.O / P.L : 0.0;
    P.lm_group = mean_track <= 5.0 ? 4 : (mean_track <= 8.0 ? 8 : 16);   // configs[4] (4.1): linearise+Schur 4.65 ms at 16 lanes, 4.24 at 8, 4.11 at 4; 5-agent map (10.0): 16 and 8 alike
    if (const char* e = getenv("COVGPU_LM_GROUP")) { const int g = atoi(e); if (g == 4 || g == 8 || g == 16) P.lm_group = g; }
  }
  P.npad = ((6 * P.K + kTile - 1) / kTile) * kTile;  // dense stage = pose-pose system only (k_struct.hip)
  P.N = P.n + 3 * P.L;
  // IMU chains -> chain-major keyframe order
  std::vector<int> perm, pos_kf, chain_ptr;
  // pose graph on the elimination tree (round 6; COVGPU_PGO_ND=0: round 2's block-arrow scheme of k_pgo.hip on a dense matrix)
  const bool pgo_nd_on = getenv("COVGPU_PGO_ND") == nullptr || atoi(getenv("COVGPU_PGO_ND")) != 0;   // (read per upload: the parity test switches it)
  const char* e_pgo_dense = getenv("COVGPU_PGO_DENSE");
  const bool pgo_nd = pgo && pgo_nd_on && allow_arrow && p->num_edge > 0 && !(e_pgo_dense && e_pgo_dense[0] == '1') && !c->sharded;
  if (pgo_nd) build_chains_pgo(p, perm, pos_kf, chain_ptr);
  else RC(build_chains(p, vi, perm, pos_kf, chain_ptr));
  std::vector<int> chain_end(P.K);
  P.nchains = (int)chain_ptr.size() - 1;
  P.max_chain_len = 1;
  for (int ci = 0; ci < P.nchains; ++ci) P.max_chain_len = std::max(P.max_chain_len, chain_ptr[ci + 1] - chain_ptr[ci]);
  for (int c = 0; c < P.nchains; ++c)
    for (int q = chain_ptr[c]; q < chain_ptr[c + 1]; ++q) chain_end[q] = chain_ptr[c + 1];
  P.reproj_loss_a = opt->reproj_loss_a; P.gravity = opt->gravity;
  // ---- form of the reduced camera system: multifrontal fronts over a nested-dissection tree (k_front.hip) for every GBA.
  //      COVGPU_GBA_DENSE=1: the same code with ONE front = the dense system (equality tests). The dense row-major matrix
  //      Sred remains for the pose graph (k_pgo.hip builds its own block plan on it) and for covgpu_schur (reads it back).
  const char* e_dense = getenv("COVGPU_GBA_DENSE");
  const bool use_nd = (!pgo && allow_arrow) || pgo_nd;
  if (c->sharded && !use_nd) { g_err = "the agent-sharded solve is for GBA problems (covgpu_upload)"; return COVGPU_ERR_INVALID_ARG; }
  const size_t K = P.K;
  RC(dev_upload(c, &P.pose0, p->kf_pose, 7 * K));
  if (p->kf_speed_bias) RC(dev_upload(c, &P.sb0, p->kf_speed_bias, 9 * K)); else { RC(dev_alloc(c, &P.sb0, 9 * K)); HIPCHK(hipMemsetAsync(P.sb0, 0, 9 * K * sizeof(double), c->st)); }
  RC(dev_upload(c, &P.lm0, p->lm_pos, (size_t)3 * P.L));
  RC(dev_alloc(c, &P.pose, 7 * K)); RC(dev_alloc(c, &P.sb, 9 * K)); RC(dev_alloc(c, &P.lm, (size_t)3 * P.L));
  RC(dev_alloc(c, &P.pose_c, 7 * K)); RC(dev_alloc(c, &P.sb_c, 9 * K)); RC(dev_alloc(c, &P.lm_c, (size_t)3 * P.L));
  RC(dev_upload(c, &P.fixed, (const uint8_t*)p->kf_fixed, K));
  RC(dev_upload(c, &P.kf_cam, (const int*)p->kf_cam, K));
  RC(dev_upload(c, &P.cam_extr, p->cam_extr, (size_t)7 * P.A));
  RC(dev_upload(c, &P.cam_intr, p->cam_intr, (size_t)4 * P.A));
  RC(dev_upload(c, &P.cam_dist, p->cam_dist, (size_t)4 * P.A));
  RC(dev_upload(c, &P.cam_dist_type, (const int*)p->cam_dist_type, (size_t)P.A));
  // observation stream: SoA. The interleaved keypoints are split, the landmark index of every observation is filled and the
  // keyframe-major lists are built ON THE DEVICE (k_pairs.hip) — three host loops over O and two more uploads before (12 of the 18 ms of an
  // upload of the 5-agent map)
  std::vector<int> ptr0(1, 0);
  RC(dev_upload(c, &P.lm_obs_ptr, P.L ? (const int*)p->lm_obs_ptr : ptr0.data(), (size_t)P.L + 1));
  RC(dev_upload(c, &P.obs_kf, (const int*)p->obs_kf, (size_t)P.O));
  RC(dev_upload(c, &P.obs_sigma, p->obs_sigma, (size_t)P.O));
  RC(dev_alloc(c, &P.obs_lm, (size_t)P.O)); RC(dev_alloc(c, &P.obs_u, (size_t)P.O)); RC(dev_alloc(c, &P.obs_v, (size_t)P.O));
  RC(dev_alloc(c, &P.kf_obs_ptr, (size_t)P.K + 1)); RC(dev_alloc(c, &P.kf_obs_idx, (size_t)P.O));
  {
    double* d_uv = nullptr; int* d_iota = nullptr;
    HIPCHK(hipMalloc((void**)&d_uv, std::max<size_t>((size_t)2 * P.O, 2) * sizeof(double)));
    if (hipMalloc((void**)&d_iota, std::max<size_t>((size_t)P.O, 1) * sizeof(int)) != hipSuccess) { (void)hipFree(d_uv); g_err = "hipMalloc: out of memory"; return COVGPU_ERR_OUT_OF_MEMORY; }
    hipError_t e = P.O ? hipMemcpyAsync(d_uv, p->obs_uv, (size_t)2 * P.O * sizeof(double), hipMemcpyHostToDevice, c->st) : hipSuccess;
    launch_obs_unpack(P.L, P.O, P.lm_obs_ptr, d_uv, P.obs_lm, P.obs_u, P.obs_v, d_iota, c->st);
    const bool ok = e == hipSuccess && build_kf_lists_device(P.O, P.K, P.obs_kf, d_iota, P.kf_obs_ptr, P.kf_obs_idx, c->st);   // (ends with a stream synchronisation)
    (void)hipStreamSynchronize(c->st);
    (void)hipFree(d_uv); (void)hipFree(d_iota);
    if (!ok) { g_err = "upload: device-side observation lists failed (allocation)"; return COVGPU_ERR_OUT_OF_MEMORY; }
  }
  tm("validate, chains, states, observation stream");
  // keyframe-major observation lists + covisible pair lists (fixed keyframes carry no pose block -> excluded)
  std::vector<int> h_pair_i, h_pair_j;  // kept for the plan below
  {
    // covisible pairs (i > j in chain-major positions, free keyframes only) with, per pair, the observations of every common
    // landmark in landmark order: built on the device (k_pairs.hip: key emission + one stable radix sort + run-length encoding)
    std::vector<int> key(P.K);
    for (int k = 0; k < P.K; ++k) key[k] = p->kf_fixed[k] ? -1 : perm[k];
    int* d_key = nullptr;
    RC(dev_upload(c, &d_key, key.data(), key.size()));
    c->d_pairkey = d_key; c->h_perm = perm;
    PairLists pl;
    if (!build_pairs_device(P.L, P.K, P.lm_obs_ptr, P.obs_kf, d_key, true, c->st, pl)) { g_err = "covisible pair lists: device allocation failed (or more than 2^31 entries)"; return COVGPU_ERR_OUT_OF_MEMORY; }
    for (int* q : {pl.pair_ptr, pl.pair_i, pl.pair_j, pl.pair_oa, pl.pair_ob}) if (q) c->allocs.push_back(q);
    c->alloc_bytes += ((size_t)pl.npairs * 3 + 1 + 2 * pl.nent) * sizeof(int);
    P.npairs = pl.npairs; P.pair_ptr = pl.pair_ptr; P.pair_i = pl.pair_i; P.pair_j = pl.pair_j; P.pair_oa = pl.pair_oa; P.pair_ob = pl.pair_ob;
    std::vector<int> pi(P.npairs), pj(P.npairs);
    if (P.npairs) {
      HIPCHK(hipMemcpyAsync(pi.data(), P.pair_i, (size_t)P.npairs * sizeof(int), hipMemcpyDeviceToHost, c->st));
      HIPCHK(hipMemcpyAsync(pj.data(), P.pair_j, (size_t)P.npairs * sizeof(int), hipMemcpyDeviceToHost, c->st));
    }
    RC(dev_alloc(c, &P.obsZ, (size_t)18 * P.O)); RC(dev_alloc(c, &P.lmRT, (size_t)9 * P.L));
    RC(dev_alloc(c, &P.kobs, (size_t)3 * P.O)); RC(dev_alloc(c, &P.kobs_lm, (size_t)P.O)); RC(dev_alloc(c, &P.obs_zpos, (size_t)P.O));
    launch_kobs_build(P, P.pair_oa, P.pair_ob, pl.nent, c->st);
    RC(dev_alloc(c, &P.cost_part, (size_t)(P.L / 4 + 64)));
    HIPCHK(hipStreamSynchronize(c->st));
    h_pair_i.swap(pi); h_pair_j.swap(pj);
  }
  tm("covisible pair lists");
  // IMU
  RC(dev_upload(c, &P.imu_i, (const int*)p->imu_kf_i, (size_t)P.I));
  RC(dev_upload(c, &P.imu_j, (const int*)p->imu_kf_j, (size_t)P.I));
  RC(dev_upload(c, &P.imu_ptr, P.I ? (const int*)p->imu_sample_ptr : ptr0.data(), (size_t)P.I + 1));
  RC(dev_upload(c, &P.imu_samples, p->imu_samples, (size_t)7 * P.S));
  RC(dev_upload(c, &P.imu_first, p->imu_first, (size_t)6 * P.I));
  {  // per-factor calibration; the options' single set is the fallback for callers that have one IMU model only
    std::vector<double> nz((size_t)5 * P.I);
    for (int f = 0; f < P.I; ++f) {
      double* r = &nz[(size_t)5 * f];
      if (p->imu_noise) std::memcpy(r, p->imu_noise + (size_t)5 * f, 5 * sizeof(double));
      else { r[0] = opt->sigma_a; r[1] = opt->sigma_g; r[2] = opt->sigma_aw; r[3] = opt->sigma_gw; r[4] = opt->gravity; }
      if (!(r[0] > 0 && r[1] > 0 && r[2] > 0 && r[3] > 0) || !(r[4] >= 9.0)) {  // keyframe_base.cpp:51-55 rejects g < 9
        g_err = "invalid problem: IMU noise must be positive and gravity >= 9 (factor " + std::to_string(f) + ")";
        return COVGPU_ERR_INVALID_ARG;
      }
    }
    RC(dev_upload(c, &P.imu_noise, nz.data(), nz.size()));
    HIPCHK(hipStreamSynchronize(c->st));
  }
  RC(dev_alloc(c, &P.pre_delta, (size_t)11 * P.I)); RC(dev_alloc(c, &P.pre_J, (size_t)225 * P.I));
  RC(dev_alloc(c, &P.pre_P, (size_t)225 * P.I)); RC(dev_alloc(c, &P.pre_W, (size_t)225 * P.I));
  RC(dev_alloc(c, &P.pre_bias, (size_t)6 * P.I));
  // edges
  RC(dev_upload(c, &P.edge_i, (const int*)p->edge_i, (size_t)P.E));
  RC(dev_upload(c, &P.edge_j, (const int*)p->edge_j, (size_t)P.E));
  RC(dev_upload(c, &P.edge_meas, p->edge_meas, (size_t)7 * P.E));
  RC(dev_upload(c, &P.edge_sqrt_info, p->edge_sqrt_info, (size_t)36 * P.E));
  RC(dev_upload(c, &P.edge_loss_a, p->edge_loss_a, (size_t)P.E));
  // linear system + step vectors
  RC(dev_upload(c, &P.perm, perm.data(), K)); RC(dev_upload(c, &P.pos_kf, pos_kf.data(), K));
  RC(dev_upload(c, &P.chain_ptr, chain_ptr.data(), chain_ptr.size()));
  RC(dev_upload(c, &P.pos_chain_end, chain_end.data(), K));
  RC(dev_alloc(c, &P.bred, (size_t)P.n));
  RC(dev_alloc(c, &P.bp, (size_t)2 * P.npad));
  const size_t Kv = vi ? K : 0;
  RC(dev_alloc(c, &P.Ad, 81 * Kv)); RC(dev_alloc(c, &P.Ae, 81 * Kv));
  RC(dev_alloc(c, &P.Bp, 54 * Kv)); RC(dev_alloc(c, &P.Bs, 54 * Kv)); RC(dev_alloc(c, &P.Bn, 54 * Kv));
  {
    std::vector<int> cbeg(P.K);
    for (int ch = 0; ch < P.nchains; ++ch) for (int q = chain_ptr[ch]; q < chain_ptr[ch + 1]; ++q) cbeg[q] = chain_ptr[ch];
    RC(dev_upload(c, &P.pos_chain_begin, cbeg.data(), cbeg.size()));
    HIPCHK(hipStreamSynchronize(c->st));
  }
  RC(dev_alloc(c, &P.grad, (size_t)P.N)); RC(dev_alloc(c, &P.hdiag, (size_t)P.N));
  RC(dev_alloc(c, &P.HllInv, (size_t)6 * P.L));
  RC(dev_alloc(c, &P.gn, (size_t)P.N)); RC(dev_alloc(c, &P.step, (size_t)P.N)); RC(dev_alloc(c, &P.vtmp, (size_t)P.N));
  // (sharded solve: the entries of other ranks' unknowns are never written — they must read as 0, not as what hipMalloc left)
  HIPCHK(hipMemsetAsync(P.gn, 0, (size_t)P.N * sizeof(double), c->st)); HIPCHK(hipMemsetAsync(P.step, 0, (size_t)P.N * sizeof(double), c->st));
  HIPCHK(hipMemsetAsync(P.vtmp, 0, (size_t)P.N * sizeof(double), c->st));
  RC(dev_alloc(c, &P.scal, (size_t)SC_COUNT)); RC(dev_alloc(c, &P.flag, (size_t)4));
  RC(dev_alloc(c, &P.tr, (size_t)TR_COUNT));
  P.scal_r = P.scal; P.flag_r = P.flag;  // what the device-side trust region reads (sharded solve: the all-reduced copies, set below)
  // deterministic reductions / scatters
  P.part_imu = 8192; P.part_edge = P.part_imu + ((P.I + 3) / 4) * 4; P.part_vec = P.part_edge + (P.E + 63) / 64;
  P.part_n = P.part_vec + 8192;
  RC(dev_alloc(c, &P.part, (size_t)SC_COUNT * P.part_n));
  HIPCHK(hipMemsetAsync(P.part, 0, (size_t)SC_COUNT * P.part_n * sizeof(double), c->st));
  RC(dev_alloc(c, &P.imuAd, 2 * 81 * Kv)); RC(dev_alloc(c, &P.imuBs, 2 * 54 * Kv));
  RC(dev_alloc(c, &P.imuCd, 3 * 36 * Kv)); RC(dev_alloc(c, &P.imuG, 2 * 30 * Kv));
  RC(dev_alloc(c, &P.edgeOut, (size_t)132 * P.E));
  {
    // keyframe -> incident edges (ascending edge index), unique pose pairs -> edges
    std::vector<int> kptr(P.K + 1, 0), kent(2 * (size_t)P.E);
    for (int e = 0; e < P.E; ++e) {
      if (p->edge_i[e] == p->edge_j[e]) { g_err = "invalid problem: self edge"; return COVGPU_ERR_INVALID_ARG; }
      kptr[p->edge_i[e] + 1]++; kptr[p->edge_j[e] + 1]++;
    }
    for (int k = 0; k < P.K; ++k) kptr[k + 1] += kptr[k];
    { std::vector<int> cur(kptr.begin(), kptr.end() - 1);
      for (int e = 0; e < P.E; ++e) { kent[cur[p->edge_i[e]]++] = 2 * e; kent[cur[p->edge_j[e]]++] = 2 * e + 1; } }
    struct PE { int i, j, ent; };
    std::vector<PE> pe(P.E);
    for (int e = 0; e < P.E; ++e) {
      const int a = perm[p->edge_i[e]], b = perm[p->edge_j[e]];
      pe[e] = (a > b) ? PE{a, b, 2 * e} : PE{b, a, 2 * e + 1};
    }
    std::stable_sort(pe.begin(), pe.end(), [](const PE& x, const PE& y) { return x.i != y.i ? x.i < y.i : x.j < y.j; });
    std::vector<int> eptr, ei, ej, eent(P.E);
    for (int q = 0; q < P.E; ++q) {
      if (q == 0 || pe[q].i != pe[q - 1].i || pe[q].j != pe[q - 1].j) { eptr.push_back(q); ei.push_back(pe[q].i); ej.push_back(pe[q].j); }
      eent[q] = pe[q].ent;
    }
    eptr.push_back(P.E);
    P.nepairs = (int)ei.size();
    RC(dev_upload(c, &P.kf_edge_ptr, kptr.data(), kptr.size())); RC(dev_upload(c, &P.kf_edge_ent, kent.data(), kent.size()));
    RC(dev_upload(c, &P.epair_ptr, eptr.data(), eptr.size())); RC(dev_upload(c, &P.epair_i, ei.data(), ei.size()));
    RC(dev_upload(c, &P.epair_j, ej.data(), ej.size())); RC(dev_upload(c, &P.epair_ent, eent.data(), eent.size()));
    HIPCHK(hipStreamSynchronize(c->st));
    tm("IMU, edges, vectors");
    if (use_nd) {
      NdHostPlan nhp;
      if (c->sharded) {
        // the global plan (covgpu_shard_plan on the FULL problem, the same on every rank), from IR keyframes to this
        // sub-problem's chain positions; this rank holds the fronts of its subtrees and of the replicated top nodes
        if ((int)c->shard_plan->K != P.K || (c->shard_plan->vi != 0) != vi) { g_err = "sharded solve: the problem does not match the shard plan"; return COVGPU_ERR_INVALID_ARG; }
        nhp = *c->shard_plan;
        std::vector<int> to_pos(2 * (size_t)P.K);
        for (int k = 0; k < P.K; ++k) { to_pos[2 * k] = 2 * perm[k]; to_pos[2 * k + 1] = 2 * perm[k] + 1; }
        nd_plan_remap(nhp, to_pos);
      } else {
        const bool one_front = e_dense && e_dense[0] == '1';
        const int leaf = one_front ? 0x3fffffff : nd_leaf_dims(0);
        // The plan of the previous upload serves again if this problem has the same keyframes in the same chain positions and its
        // couplings are a SUBSET of the ones the plan was built for (a tree stays a valid elimination order when couplings
        // disappear: the second round of a GlobalBundleAdjustment call is the first one minus the erased observations).
        std::vector<uint64_t> keys(h_pair_i.size() + ei.size());
        {
          std::vector<uint64_t> ka(h_pair_i.size()), kb(ei.size());
          for (size_t q = 0; q < ka.size(); ++q) ka[q] = ((uint64_t)(uint32_t)h_pair_i[q] << 32) | (uint32_t)h_pair_j[q];
          for (size_t q = 0; q < kb.size(); ++q) kb[q] = ((uint64_t)(uint32_t)ei[q] << 32) | (uint32_t)ej[q];
          if (!std::is_sorted(ka.begin(), ka.end())) std::sort(ka.begin(), ka.end());
          std::merge(ka.begin(), ka.end(), kb.begin(), kb.end(), keys.begin());
          keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
        }
        PlanCache& pc = c->plan_cache;
        const bool hit = pc.valid && pc.K == P.K && pc.vi == vi && pc.leaf == leaf && pc.chain_ptr == chain_ptr && pc.pos_kf == pos_kf &&
                         std::includes(pc.keys.begin(), pc.keys.end(), keys.begin(), keys.end());
        if (hit) nhp = pc.hp;
        else {
          if (!nd_plan_build(P.K, vi, P.nchains, chain_ptr.data(), (int)h_pair_i.size(), h_pair_i.data(), h_pair_j.data(), P.nepairs, ei.data(), ej.data(), leaf, nhp)) {
            g_err = "nested-dissection plan: a coupling joins two branches"; return COVGPU_ERR_INVALID_ARG;
          }
          pc.valid = true; pc.K = P.K; pc.vi = vi; pc.leaf = leaf; pc.chain_ptr = chain_ptr; pc.pos_kf = pos_kf; pc.keys.swap(keys); pc.hp = nhp;
        }
      }
      if (nhp.maxdepth > 64) { g_err = "nested-dissection plan: tree deeper than 64 levels"; return COVGPU_ERR_INVALID_ARG; }
      tm("nested-dissection plan");
      NdDev& nd = c->nd;
      nd_tables(nhp, pos_kf.data(), P.D, c->rank, nd);
      tm("front tables");
      P.nd = 1; P.nd_nnodes = nd.nnodes; P.nd_nlev = (int)nd.lev.size(); P.nd_maxd = nhp.maxdepth;
      nd.ntop = (int)nd.h_top_g.size();
      RC(dev_upload(c, &P.nd_vnode, nd.h_vnode.data(), nd.h_vnode.size())); RC(dev_upload(c, &P.nd_voff, nd.h_voff.data(), nd.h_voff.size()));
      RC(dev_upload(c, &P.nd_vord, nd.h_vord.data(), nd.h_vord.size()));
      if (c->sharded) RC(dev_upload(c, &P.nd_vown, nd.h_vown.data(), nd.h_vown.size()));
      RC(dev_upload(c, &P.nd_ndepth, nd.h_ndepth.data(), nd.h_ndepth.size())); RC(dev_upload(c, &P.nd_nI, nd.h_nI.data(), nd.h_nI.size()));
      RC(dev_upload(c, &P.nd_ntab, nd.h_ntab.data(), nd.h_ntab.size()));
      RC(dev_upload(c, &P.nd_abase, nd.h_abase.data(), nd.h_abase.size())); RC(dev_upload(c, &P.nd_fidx, nd.h_fidx.data(), nd.h_fidx.size()));
      RC(dev_upload(c, &nd.own_dims, nd.h_own_dims.data(), nd.h_own_dims.size())); RC(dev_upload(c, &nd.st_dims, nd.h_st_dims.data(), nd.h_st_dims.size()));
      RC(dev_upload(c, &nd.own_g, nd.h_own_g.data(), nd.h_own_g.size())); RC(dev_upload(c, &nd.st_g, nd.h_st_g.data(), nd.h_st_g.size()));
      RC(dev_upload(c, &nd.gidx, nd.h_gidx.data(), nd.h_gidx.size()));
      RC(dev_upload(c, &nd.tree_fill, nd.h_tree_fill.empty() ? (const int*)nullptr : nd.h_tree_fill.data(), std::max<size_t>(nd.h_tree_fill.size(), 1)));
      RC(dev_upload(c, &nd.cptr, nd.h_cptr.data(), nd.h_cptr.size())); RC(dev_upload(c, &nd.cidx, nd.h_cidx.data(), nd.h_cidx.size()));
      RC(dev_upload(c, &nd.cptr2, nd.h_cptr2.data(), nd.h_cptr2.size())); RC(dev_upload(c, &nd.cidx2, nd.h_cidx2.data(), nd.h_cidx2.size()));
      RC(dev_upload(c, &nd.inv_off, nd.h_inv_off.data(), nd.h_inv_off.size())); RC(dev_upload(c, &nd.inv, nd.h_inv.data(), nd.h_inv.size()));
      RC(dev_upload(c, &nd.rhs_node, nd.h_rhs_node.data(), nd.h_rhs_node.size()));
      RC(dev_upload(c, &nd.ext, nd.h_ext.data(), nd.h_ext.size()));
      auto up_or_zero = [&](int** dst, const std::vector<int>& h, size_t least) { return dev_upload(c, dst, h.empty() ? (const int*)nullptr : h.data(), std::max(h.size(), least)); };
      RC(up_or_zero(&nd.extw, nd.h_extw, 8)); RC(up_or_zero(&nd.extc, nd.h_extc, 6)); RC(up_or_zero(&nd.extc2, nd.h_extc2, 6));
      RC(up_or_zero(&nd.extr, nd.h_extr, kExtRec));
      RC(dev_upload(c, &nd.bb_off, nd.h_bb_off.data(), nd.h_bb_off.size()));
      RC(dev_upload(c, &nd.bb, nd.h_bb.data(), std::max<size_t>(nd.h_bb.size(), 1)));
      RC(dev_upload(c, &nd.top_var, nd.h_top_var.data(), nd.h_top_var.size())); RC(dev_upload(c, &nd.top_r, nd.h_top_r.data(), nd.h_top_r.size()));
      RC(dev_upload(c, &nd.top_g, nd.h_top_g.data(), nd.h_top_g.size()));
      for (NdLevel& L : nd.lev) { RC(dev_upload(c, &L.live, L.live_h.data(), L.live_h.size())); RC(dev_upload(c, &L.plist, L.plist_h.data(), L.plist_h.size())); }
      // one allocation: [subtree fronts | top fronts][top right-hand sides | grad, hdiag of the top unknowns | subtree right-hand
      // sides] — what a sharded solve all-reduces is one contiguous range of it
      RC(dev_alloc(c, &P.nd_M, nd.M_elems + nd.rhs_elems));
      P.nd_rhs = P.nd_M + nd.M_elems;
      RC(dev_alloc(c, &P.nd_Linv, nd.linv_elems));
      RC(dev_alloc(c, &P.nd_dummy, (size_t)64));
      launch_nd_init(P, nd, c->st);
      c->chol.tri_clear();   // the live-tile lists of the bulk updates belong to the previous problem
      HIPCHK(hipStreamSynchronize(c->st));
      tm("table upload, front allocation");
      if (c->sharded) {
        P.shard = 1;
        // weight of every unknown in the trust-region norms: counted by exactly one rank (own subtree: here; top: rank 0)
        std::vector<double> vw((size_t)P.N, 1.0);  // landmarks: all mine (the sub-problem holds this rank's landmarks only)
        for (int k = 0; k < P.K; ++k) {
          const int op = nd.h_vown[2 * perm[k]], os = nd.h_vown[2 * perm[k] + 1];
          const double wp = op == 1 ? 1.0 : (op == 2 && c->rank == 0 ? 1.0 : 0.0), ws = os == 1 ? 1.0 : (os == 2 && c->rank == 0 ? 1.0 : 0.0);
          for (int r = 0; r < 6; ++r) vw[(size_t)P.D * k + r] = wp;
          for (int r = 6; r < P.D; ++r) vw[(size_t)P.D * k + r] = ws;
        }
        RC(dev_upload(c, &P.vw, vw.data(), vw.size()));
        RC(dev_alloc(c, &c->d_red, (size_t)SC_COUNT + 2 * (size_t)c->world));
        static const bool pack_on = getenv("COVGPU_SHARD_PACK") == nullptr || atoi(getenv("COVGPU_SHARD_PACK")) != 0;
        nd.n_top_tiles = (int)(nd.h_top_tiles.size() / 3);
        if (pack_on) {
          RC(dev_upload(c, &nd.top_tiles, nd.h_top_tiles.data(), nd.h_top_tiles.size()));
          RC(dev_alloc(c, &nd.top_pack, (size_t)nd.n_top_tiles * kTile * kTile + nd.rhs_top + 2 * (size_t)nd.ntop));
        }
        RC(dev_alloc(c, &P.scal_r, (size_t)SC_COUNT)); RC(dev_alloc(c, &P.flag_r, (size_t)4));
      }
      HIPCHK(hipStreamSynchronize(c->st));
      if (opt->verbose) {
        int panels = 0;
        for (const NdLevel& L : nd.lev) panels += L.nI / 256;
        std::printf("[covgpu] multifrontal plan: %d fronts in %d levels (%d serial 256-column panels), %.2e flops, fronts %.2f GB%s\n", nd.nnodes,
                    (int)nd.lev.size(), panels, nhp.flops, nd.M_elems * 8e-9, c->sharded ? " (this rank's share)" : "");
      }
    } else {
      RC(dev_alloc(c, &P.Sred, (size_t)P.npad * P.npad));
      RC(dev_alloc(c, &P.Linv, (size_t)(P.npad / kTile) * kTile * kTile));
    }
  }
  if (pgo && P.E && !pgo_nd) {  // block-arrow plan for the pose-graph solve (k_pgo.hip); COVGPU_PGO_DENSE=1 keeps the plain dense solve
    PgoHostPlan hp;
    const char* dense = getenv("COVGPU_PGO_DENSE");
    if (!(dense && dense[0] == '1') && pgo_plan_analyse(P.K, P.E, p->edge_i, p->edge_j, hp)) {
      PgoPlan& plan = c->pgo_plan;
      std::vector<int> idxb;
      for (int kf : hp.border_kf) for (int r = 0; r < 6; ++r) idxb.push_back(6 * kf + r);
      plan.nb = (((int)idxb.size() + kTile - 1) / kTile) * kTile;
      if (plan.nb == 0) plan.nb = kTile;
      idxb.resize(plan.nb, -1);
      RC(dev_upload(c, &plan.idx_b, idxb.data(), idxb.size()));
      RC(dev_alloc(c, &plan.Sb, (size_t)plan.nb * plan.nb)); RC(dev_alloc(c, &plan.rhs_b, (size_t)2 * plan.nb));
      RC(dev_alloc(c, &plan.Linv_b, (size_t)plan.nb * kTile));
      plan.nblk = (int)hp.block_kf.size();
      size_t big = 0;
      for (auto& bk : hp.block_kf) big = std::max(big, bk.size());
      plan.nIpad = (((int)big * 6 + 2 * kTile - 1) / (2 * kTile)) * (2 * kTile);  // whole big panels: tstop is even
      plan.ntot = plan.nIpad + plan.nb;
      std::vector<int> idx((size_t)plan.nblk * plan.ntot, -1);
      for (int a = 0; a < plan.nblk; ++a) {
        int* row = idx.data() + (size_t)a * plan.ntot;
        int q = 0;
        for (int kf : hp.block_kf[a]) for (int r = 0; r < 6; ++r) row[q++] = 6 * kf + r;
        std::copy(idxb.begin(), idxb.end(), row + plan.nIpad);
      }
      RC(dev_upload(c, &plan.idx, idx.data(), idx.size()));
      RC(dev_alloc(c, &plan.M, (size_t)plan.nblk * plan.ntot * plan.ntot)); RC(dev_alloc(c, &plan.rhs, (size_t)plan.nblk * 2 * plan.ntot));
      RC(dev_alloc(c, &plan.Linv, (size_t)plan.nblk * plan.nIpad * kTile));
      plan.active = true;
    }
  }
  HIPCHK(hipMemsetAsync(P.scal, 0, SC_COUNT * sizeof(double), c->st));
  HIPCHK(hipMemsetAsync(P.flag, 0, 4 * sizeof(int), c->st));
  HIPCHK(hipStreamSynchronize(c->st));  // host staging vectors go out of scope
  c->have = true; c->pgo = pgo;
  return COVGPU_OK;
}

static int reset_state(covgpu_context* c) {
  DevProblem& P = c->P;
  HIPCHK(hipMemcpyAsync(P.pose, P.pose0, (size_t)7 * P.K * sizeof(double), hipMemcpyDeviceToDevice, c->st));
  HIPCHK(hipMemcpyAsync(P.sb, P.sb0, (size_t)9 * P.K * sizeof(double), hipMemcpyDeviceToDevice, c->st));
  if (P.L) HIPCHK(hipMemcpyAsync(P.lm, P.lm0, (size_t)3 * P.L * sizeof(double), hipMemcpyDeviceToDevice, c->st));
  return COVGPU_OK;
}

static int read_scalars(covgpu_context* c) {
  HIPCHK(hipMemcpyAsync(c->h_scal, c->P.scal, SC_COUNT * sizeof(double), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipMemcpyAsync(c->h_scal + SC_COUNT, c->P.flag, sizeof(int), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  HIPCHK(hipGetLastError());  // a failed kernel launch anywhere in the batch just drained surfaces here
  return COVGPU_OK;
}
static int chol_failed(covgpu_context* c) { int f; std::memcpy(&f, c->h_scal + SC_COUNT, sizeof(int)); return f; }

// linearise at the current state and form the damped reduced system (A.6): cost, grad, hdiag, Sred, bred
static void enqueue_build(covgpu_context* c, double mu) {
  const DevProblem& P = c->P;
  c->cur_damp = mu;
  if (c->profiling) (void)hipEventRecord(c->ev[0], c->st);
  // the clearing of the fronts' live tiles (1.1 GB of fronts on the 5-agent map) runs on its own stream beside the inertial
  // kernels and the landmark linearisation, which only write per-factor / per-observation records; its first writers
  // (k_kf_reduce ...) wait for it. Every reader of the previous system has finished: each iteration ends with a host sync.
  c->chol.init();
  // Enqueue order = host time: every iteration starts behind a host synchronisation with the chip empty, and a launch is ~6 us of host
  // work. The main stream's own first kernels (the small clears, the landmark linearisation: 0.17 ms, on nobody's results) go out FIRST;
  // the clearing of the fronts (head stream) and the inertial kernels (side stream) follow underneath them. (Before: seven clearing
  // launches ahead of everything, the linearisation started ~0.1 ms into the iteration.)
  // (measured, round 4: clearing the fronts BEHIND the previous linear solve instead — under the trust-region tail — takes 0.11 ms off
  //  this pass and puts 0.12 ms onto the tail and the solve: the 0.65 GB of stores contend with the tail's re-linearisations)
  // (round 6: the landmark linearisation — the head of the pass's critical path — is the FIRST launch of the iteration on the main stream; the small
  //  clears, which it does not depend on, open the side stream instead: their first readers are the inertial kernels there and, on the main
  //  stream, kernels behind the join of the side stream (ev_kf). Before: clears, a fill and a record in front of it, ~12 us.)
  // The side stream's kernels read the estimate (and the preintegration): they follow "the main stream is complete up to here", published by the
  // landmark linearisation's own first thread (no launch in front of it) — or, without landmarks / device flags, by a record.
  hipStream_t side = c->chol.mid;
  {
    DevSignal sig = P.L > 0 ? c->chol.publish_handle(c->chol.ev_zero, c->st, 1) : DevSignal();
    if (sig.flag == nullptr) c->chol.record(c->chol.ev_zero, c->st, 1);
    launch_lm_lin(P, mu, c->st, sig);   // writes per-observation records, per-landmark blocks (incl. the landmark part of grad / hdiag) and cost partials only
  }
  c->chol.wait(side, c->chol.ev_zero);
  launch_zero_system(P, side);
  const bool forked = P.L > 0 && P.npairs > 0;   // (launch_lm_build's condition: its per-keyframe reduction and cost finisher run on `side`)
  if (!forked) { c->chol.record(c->chol.ev_lin, side, 2); c->chol.wait(c->st, c->chol.ev_lin); }   // ... else on the main stream: behind the clears (ev_lin is free: no fork)
  // (the head stream waits for nobody: every reader of the previous system has finished — each iteration ends with a host sync)
  if (P.nd) { launch_nd_zero(P, c->nd, c->chol.head); (void)hipMemsetAsync(P.nd_rhs, 0, c->nd.rhs_elems * sizeof(double), c->chol.head); }   // (the fronts' right-hand sides: was a fill on the chain, in front of the assembly)
  else (void)hipMemsetAsync(P.Sred, 0, (size_t)P.npad * P.npad * sizeof(double), c->chol.head);
  c->chol.record(c->chol.ev_fill, c->chol.head, 3);
  // inertial factors (one wave per factor: latency, not throughput) on the side stream beside the landmark pass; their speed-bias
  // blocks are final before anything of the pose system is touched, the pose-dimension part is gathered after the visual blocks
  launch_imu_build(P, side);
  if (P.vi) {
    launch_imu_gather(P, 1, side);
    launch_finalize_diag(P, mu, 1, side);
  }
  // Between the pair pass and the first factorisation every launch is a link of the chain (~6 us each): the loop edges are linearised
  // on the side stream beside the landmark pass (they depend on the estimate only), and the cost partials are summed there too — behind
  // the last kernel that writes one (the landmark pass's own finisher runs on the side stream when the pass forks).
  launch_edge_build(P, side);
  launch_lm_build(P, mu, c->st, c->chol.ev_fill, side, c->chol.ev_lin, c->chol.ev_kf, &c->chol);
  if (forked) launch_part_finish(P, SC_COST, 1, side);
  c->chol.record(c->chol.ev_kf, side, 4);
  c->chol.wait(c->st, c->chol.ev_kf);
  launch_imu_gather(P, 0, c->st);  // pose-dimension part: adds onto the blocks k_kf_reduce assigned (fixed order: visual, inertial, loop)
  launch_edge_gather(P, c->st);
  launch_finalize_diag(P, mu, P.vi ? 0 : 2, c->st);
  if (!forked) launch_part_finish(P, SC_COST, 1, c->st);
  if (c->profiling) (void)hipEventRecord(c->ev[1], c->st);
}

static void enqueue_solve(covgpu_context* c, double* dst_all) {
  const DevProblem& P = c->P;
  // with profiling on, every bulk trailing-update (SYRK) launch gets its own event pair on its stream so that
  // bench.py can quote the dominant kernel's duration
  if (c->profiling) (void)hipEventRecord(c->ev[2], c->st);
  if (P.nd) {   // GBA: multifrontal solve of the whole system (k_front.hip)
    if (!launch_nd_solve(P, c->nd, dst_all, c->cur_damp, c->st, c->chol) && !c->coll_failed) { c->coll_failed = true; c->coll_err = "scratch allocation of the backward substitution failed"; }
  }
  else launch_pose_graph_solve(P, dst_all, c->st, c->chol, c->pgo_plan.active ? &c->pgo_plan : nullptr);  // pose graph (k_pgo.hip)
  if (c->profiling) (void)hipEventRecord(c->ev[3], c->st);
  launch_lm_backsub(P, dst_all, dst_all, c->st);
}

static void collect_profile(covgpu_context* c, bool built, bool solved) {
  if (solved && !c->profiling) c->chol.collect();  // (prints the COVGPU_TRACE_PANELS marks; nothing else without profiling)
  if (!c->profiling) return;
  float ms = 0;
  if (built && hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) { c->prof.t_build_ms += ms; c->prof.n_build++; }
  if (solved && hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) { c->prof.t_factor_ms += ms; c->prof.n_factor++; }
  if (solved) c->chol.collect();
}

// (no clearing of the partial sums before these: every wave of a reduction kernel stores its slot, the same slots at every launch.
//  Measured and dropped: the inertial / loop-edge kernels on a side stream beside the observation kernel, and the whole Cauchy
//  leg underneath the linear solve — both within the +-1 % run-to-run noise, the event hops cost what the overlap buys.)
static void enqueue_jvp(covgpu_context* c, const double* v_all) {
  const DevProblem& P = c->P;
  launch_obs_jvp(P, v_all, c->st);
  launch_imu_jvp(P, v_all, c->st);
  launch_edge_jvp(P, v_all, c->st);
  launch_part_finish(P, SC_JV2, 1, c->st);
}

static void enqueue_cost_candidate(covgpu_context* c) {
  const DevProblem& P = c->P;
  launch_obs_cost(P, P.pose_c, P.lm_c, c->st);
  launch_imu_cost(P, P.pose_c, P.sb_c, c->st);
  launch_edge_cost(P, P.pose_c, c->st);
  launch_part_finish(P, SC_COST, 1, c->st);
}

// sharded solve: the scalars the device-side trust region is about to read are sums over residuals / unknowns each counted
// by exactly one rank (DevProblem::vw), the gradient max-norm and the factorisation flag are maxima: ONE small all-reduce
// (sum) of [16 scalars | one slot per rank for the max-norm | one slot per rank for the flag] on a scratch copy, enqueued
// on the stream; the reduced values go to P.scal_r / P.flag_r, the rank's own partial sums in P.scal stay untouched.
__global__ void k_shard_scal_pack(DevProblem P, double* buf, int rank, int world) {
  const int t = threadIdx.x;
  if (t < SC_COUNT) buf[t] = (t == SC_GMAX) ? 0.0 : P.scal[t];
  if (t < world) { buf[SC_COUNT + t] = t == rank ? P.scal[SC_GMAX] : 0.0; buf[SC_COUNT + world + t] = t == rank ? (double)P.flag[0] : 0.0; }
}
__global__ void k_shard_scal_unpack(DevProblem P, const double* buf, int world) {
  const int t = threadIdx.x;
  if (t < SC_COUNT && t != SC_GMAX) P.scal_r[t] = buf[t];
  if (t == 0) {
    double gm = 0.0, fl = 0.0;
    for (int r = 0; r < world; ++r) { gm = fmax(gm, buf[SC_COUNT + r]); fl = fmax(fl, buf[SC_COUNT + world + r]); }
    P.scal_r[SC_GMAX] = gm; P.flag_r[0] = fl != 0.0 ? 1 : 0;
  }
}
static void reduce_scalars(covgpu_context* c) {
  if (!c->sharded || !c->reducer) return;
  hipLaunchKernelGGL(k_shard_scal_pack, dim3(1), dim3(64), 0, c->st, c->P, c->d_red, c->rank, c->world);
  if (c->reducer->allreduce(c->d_red, (size_t)SC_COUNT + 2 * (size_t)c->world, 0, c->st) != 0) coll_latch(c);
  hipLaunchKernelGGL(k_shard_scal_unpack, dim3(1), dim3(64), 0, c->st, c->P, (const double*)c->d_red, c->world);
}

// Host wait for the iteration just enqueued. One GPU: hipStreamSynchronize. Sharded: polled, so that a collective that never
// completes (a peer died, gave up before its first collective, or never arrived) ends in an ERROR after COVGPU_COLL_TIMEOUT_S
// (default 300 s) instead of a hang; RCCL's asynchronous communicator errors and the give-up flag of an in-process multi-GPU call
// are seen here too. On any of these the collective is aborted (ncclCommAbort / group abort: the peers return as well).
static int wait_iteration(covgpu_context* c) {
  if (c->sharded && c->reducer) {
    static const double limit = getenv("COVGPU_COLL_TIMEOUT_S") ? std::max(1.0, atof(getenv("COVGPU_COLL_TIMEOUT_S"))) : 300.0;
    RcclReducer* rr = dynamic_cast<RcclReducer*>(c->reducer);
    const auto t0 = std::chrono::steady_clock::now();
    long spins = 0;
    for (;;) {
      const hipError_t q = hipStreamQuery(c->st);
      if (q == hipSuccess) break;
      if (q != hipErrorNotReady) { g_err = std::string("hipStreamQuery: ") + hipGetErrorString(q); return COVGPU_ERR_NO_DEVICE; }
      ++spins;
      // fail checks: after the first 256 busy polls, then every 64 sleeping polls (~4 ms)
      if (spins == 256 || (spins > 256 && (spins & 63) == 0)) {
        const char* why = nullptr;
        if (c->peer_fail && c->peer_fail->load(std::memory_order_relaxed)) why = "another rank of the call gave up";
        else if (rr && rr->async_error() != 0) why = "RCCL reported an asynchronous communicator error";
        else if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) why = "collective timed out (COVGPU_COLL_TIMEOUT_S)";
        if (why) { if (!c->coll_failed) { c->coll_failed = true; c->coll_err = why; } c->reducer->abort(); break; }
      } else if (spins > 256) std::this_thread::sleep_for(std::chrono::microseconds(50));   // (a rank's host thread sleeps between polls instead of burning a core for the whole iteration)
    }
  }
  HIPCHK(hipStreamSynchronize(c->st));
  HIPCHK(hipGetLastError());  // a failed kernel launch anywhere in the batch just drained surfaces here
  if (c->chol.gate_failed()) {   // a device-flag gate between two streams gave up (CholAux::wait): what this iteration computed is not ordered
    g_err = "solve: a device-flag gate between the context's streams timed out (COVGPU_GATE_TIMEOUT_S; COVGPU_GATES=0 selects HIP events)";
    return COVGPU_ERR_GATE_TIMEOUT;
  }
  if (c->coll_failed) {
    if (c->reducer) c->reducer->abort();
    if (c->peer_fail) c->peer_fail->store(1);
    g_err = (c->sharded ? "sharded solve, rank " + std::to_string(c->rank) + ": " : std::string("solve: ")) + c->coll_err;
    return COVGPU_ERR_NO_DEVICE;
  }
  return COVGPU_OK;
}

// Trust-region loop with the step logic on the device (k_dense.hip: k_tr_*): ONE host read-back per iteration — what the
// host needs to enqueue the next one (rebuild or reuse, damping value) and the trace. Sharded solve: the same loop on every
// rank; four collectives per iteration (the top of the tree inside the linear solve, three scalar exchanges), all enqueued on
// the stream.
static int solve_impl_dev(covgpu_context* c, const covgpu_options* opt, covgpu_result* res) {
  if (!c->have) { g_err = "no problem uploaded"; return COVGPU_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  DevProblem& P = c->P;
  const covgpu_options& o = *opt;
  P.reproj_loss_a = o.reproj_loss_a;
  std::memset(res, 0, sizeof(*res));
  c->coll_failed = false; c->coll_err.clear();   // (an aborted collective fails again at once and is latched again)
  const auto t_begin = std::chrono::steady_clock::now();
  c->chol.init();
  if (c->sharded) c->chol.gate_timeout_s = std::max(c->chol.gate_timeout_s, 120.0);   // (a gate may stand behind a collective that waits for a slower rank)
  RC(reset_state(c));
  HIPCHK(hipMemsetAsync(P.flag + 1, 0, sizeof(int), c->st));
  launch_preintegrate(P, c->st);  // R2: repropagate at the initial bias estimate (opt_be.cpp:396)
  double* h = c->h_tr;
  std::memset(h, 0, TR_COUNT * sizeof(double));
  h[TR_RADIUS] = o.initial_radius; h[TR_MU] = 1e-8; h[TR_LMDF] = 2.0; h[TR_FIRST] = 1.0; h[TR_OK] = 1.0;
  HIPCHK(hipMemcpyAsync(P.tr, h, TR_COUNT * sizeof(double), hipMemcpyHostToDevice, c->st));
  HIPCHK(hipStreamSynchronize(c->st));  // (h is reused as the read-back mirror below)
  const TrConsts tc{o.strategy, o.max_radius, o.min_relative_decrease, o.function_tolerance, o.parameter_tolerance, o.gradient_tolerance};
  bool reuse = false, got_initial = false;
  int it = 0, accepted = 0, term = 0;
  // COVGPU_TAIL=0: the round-3 tail (two J*v passes of three kernels, seven partial-sum finishers, four one-thread logic kernels:
  // ~25 dependent launches); default: the fused tail of k_tail.hip (8 launches, one J*v pass for both dogleg directions)
  static const bool fused_tail = getenv("COVGPU_TAIL") == nullptr || atoi(getenv("COVGPU_TAIL")) != 0;
  const bool two = o.strategy == COVGPU_DOGLEG;
  const bool coll = c->sharded && c->reducer != nullptr;   // scalar all-reduce between a finish and the step logic that reads it
  static const bool host_timing = getenv("COVGPU_HOST_TIMING") != nullptr;   // dev aid: host enqueue time | host wait per iteration
  bool use_box = false;
  while (it < o.max_iterations) {
    use_box = false;
    const auto t_enq0 = std::chrono::steady_clock::now();
    if (fused_tail) {
      if (!reuse) {
        const double damp = (o.strategy == COVGPU_LM) ? 1.0 / h[TR_RADIUS] : h[TR_MU];
        enqueue_build(c, damp);
        enqueue_solve(c, P.gn);
        launch_tail_stats(P, c->st);
        launch_tail_jvp(P, two, c->st);
        launch_tail_finish(P, tc, 1, two, !coll, 1, c->st);
        if (coll) { reduce_scalars(c); launch_tail_logic(P, tc, 1, 1, c->st); }
      } else launch_tail_logic(P, tc, 1, 0, c->st);
      launch_tail_apply(P, c->st);
      launch_tail_cost(P, c->st);
      launch_tail_finish(P, tc, 2, two, !coll, 0, c->st);
      if (coll) { reduce_scalars(c); launch_tail_logic(P, tc, 2, 0, c->st); }
      // (one GPU, no profiling events to collect: the state the host needs travels through the pinned mailbox, no D2H copy, no sleeping wait)
      static const bool dev_aids = getenv("COVGPU_TRACE_PANELS") != nullptr || getenv("COVGPU_GATE_LOG") != nullptr;   // (their print-outs read events / device logs behind a synchronisation)
      use_box = c->d_box != nullptr && !coll && !c->profiling && !host_timing && !dev_aids;
      if (use_box) c->box_seq += 1.0;
      launch_tr_accept(P, c->st, use_box ? c->d_box : nullptr, c->box_seq);
    } else {
    if (!reuse) {
      const double damp = (o.strategy == COVGPU_LM) ? 1.0 / h[TR_RADIUS] : h[TR_MU];
      enqueue_build(c, damp);
      enqueue_solve(c, P.gn);
      // (Cauchy direction after the solve: a sharded solve completes gradient and diag(J^T J) of the top unknowns inside it)
      if (o.strategy == COVGPU_DOGLEG) { launch_cauchy_vec(P, c->st); enqueue_jvp(c, P.vtmp); }
      launch_dogleg_stats(P, c->st);
      reduce_scalars(c);
    }
    launch_tr_after_solve(P, tc, reuse ? 0 : 1, c->st);
    launch_combine_step_dev(P, c->st);
    enqueue_jvp(c, P.step);
    launch_xnorm(P, c->st);
    reduce_scalars(c);
    launch_tr_after_model(P, tc, c->st);
    launch_apply_step(P, c->st);
    enqueue_cost_candidate(c);
    reduce_scalars(c);
    launch_tr_decide(P, tc, c->st);
    }
    const auto t_enq1 = std::chrono::steady_clock::now();
    if (use_box) {
      // poll the sequence word (the kernel stores it with release semantics behind the state); every 4096 polls look at the stream: an error, or a stream
      // that drained without the word arriving (never observed), ends in the ordinary path
      volatile double* box = c->h_box;
      bool got = false;
      for (unsigned spins = 1; !got; ++spins) {
        if (box[TR_COUNT] == c->box_seq) { got = true; break; }
        if ((spins & 4095u) == 0) { const hipError_t q = hipStreamQuery(c->st); if (q != hipErrorNotReady) break; }
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      if (got) {
        for (int i = 0; i < TR_COUNT; ++i) h[i] = box[i];
        HIPCHK(hipGetLastError());
        if (c->chol.gate_failed()) { HIPCHK(hipStreamSynchronize(c->st)); g_err = "solve: a device-flag gate between the context's streams timed out"; return COVGPU_ERR_GATE_TIMEOUT; }
      } else {
        HIPCHK(hipMemcpyAsync(h, P.tr, TR_COUNT * sizeof(double), hipMemcpyDeviceToHost, c->st));
        RC(wait_iteration(c));
      }
    } else {
      HIPCHK(hipMemcpyAsync(h, P.tr, TR_COUNT * sizeof(double), hipMemcpyDeviceToHost, c->st));
      RC(wait_iteration(c));
    }
    if (host_timing)
      std::fprintf(stderr, "[covgpu] iteration %d: host enqueue %.0f us, host wait %.0f us\n", it, std::chrono::duration<double>(t_enq1 - t_enq0).count() * 1e6,
                   std::chrono::duration<double>(std::chrono::steady_clock::now() - t_enq1).count() * 1e6);
    collect_profile(c, !reuse, !reuse);
    if (!got_initial) { res->initial_cost = h[TR_INITCOST]; got_initial = true; }
    if (h[TR_RETRY] != 0.0) { reuse = false; continue; }  // factorisation failed: same iteration again with the raised damping
    term = (int)h[TR_TERM];
    if (term == 3 || term == 2) break;
    const int acc = h[TR_ACC] != 0.0;
    accepted += acc;
    if (it < COVGPU_MAX_TRACE) { res->cost_trace[it] = h[TR_COST]; res->radius_trace[it] = h[TR_RADIUS]; res->accepted_trace[it] = acc; }
    if (o.verbose) std::printf("[covgpu] it %2d cost %.9e rho %.3f radius %.3e %s\n", it, h[TR_COST], h[TR_RHO], h[TR_RADIUS], acc ? "ok" : "rej");
    reuse = h[TR_REUSE] != 0.0;
    ++it;
    if (term == 1 || term == 4) break;
  }
  {  // IMU factors whose preintegrated covariance was not positive definite (e.g. zero samples) carry no weight: reported
    int dropped = 0;
    HIPCHK(hipMemcpy(&dropped, P.flag + 1, sizeof(int), hipMemcpyDeviceToHost));
    res->reserved = dropped;
  }
  res->iterations = it; res->accepted = accepted; res->termination = term;
  res->final_cost = h[TR_COST];
  res->t_solve_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
  res->t_linear_solve_s = 0.0;  // (not separable without extra synchronisation: covgpu_get_profile has the phase times)
  return COVGPU_OK;
}

static int download_impl(covgpu_context* c, covgpu_problem* p) {
  if (!c->have) { g_err = "no problem uploaded"; return COVGPU_ERR_INVALID_ARG; }
  const DevProblem& P = c->P;
  HIPCHK(hipMemcpyAsync(p->kf_pose, P.pose, (size_t)7 * P.K * sizeof(double), hipMemcpyDeviceToHost, c->st));
  if (P.vi && p->kf_speed_bias) HIPCHK(hipMemcpyAsync(p->kf_speed_bias, P.sb, (size_t)9 * P.K * sizeof(double), hipMemcpyDeviceToHost, c->st));
  if (P.L) HIPCHK(hipMemcpyAsync(p->lm_pos, P.lm, (size_t)3 * P.L * sizeof(double), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  return COVGPU_OK;
}

extern "C" int covgpu_upload(covgpu_context* c, const covgpu_options* opt, const covgpu_problem* p) { return guarded([&] { return upload_impl(c, opt, p, false); }); }
extern "C" int covgpu_upload_pgo(covgpu_context* c, const covgpu_options* opt, const covgpu_problem* p) { return guarded([&] { return upload_impl(c, opt, p, true); }); }
// A solve whose stream ordering through device flags failed (CholAux::wait timed out: a tool that serialises kernels out of submission order, a
// lost dispatch) is repeated ONCE with HIP events — every solve restarts from the uploaded state (reset_state) — and the context keeps events from then on.
static int solve_impl_dev(covgpu_context* c, const covgpu_options* opt, covgpu_result* res);
static int solve_any(covgpu_context* c, const covgpu_options* opt, covgpu_result* out) {
  int rc = solve_impl_dev(c, opt, out);
  if (rc == COVGPU_ERR_GATE_TIMEOUT) {
    if (c->chol.gate_dead_h[1] == -2)   // (k_panel.hip: pipe_take)
      std::fprintf(stderr, "[covgpu] warning: a hand-over inside the pipelined backward substitution timed out (workgroup %d of front %d of its level); repeating the solve with a launch per tile and HIP events (this context keeps them)\n",
                   c->chol.gate_dead_h[2], c->chol.gate_dead_h[3]);
    else
    std::fprintf(stderr, "[covgpu] warning: device-flag stream ordering timed out (slot %d awaited %d, found %d; %ld signals, %ld gates so far); repeating the solve with HIP events (this context keeps them)\n",
                 c->chol.gate_dead_h[1], c->chol.gate_dead_h[2], c->chol.gate_dead_h[3], c->chol.gate_signals, c->chol.gate_waits);
    if (c->chol.gate_dead_h[1] >= 0 && c->chol.gate_dead_h[1] < (int)c->chol.gate_tag_of_slot.size()) std::fprintf(stderr, "[covgpu]          (tag of that slot: %d)\n", c->chol.gate_tag_of_slot[c->chol.gate_dead_h[1]]);
    (void)hipDeviceSynchronize();
    c->chol.gates_disable();
    rc = solve_impl_dev(c, opt, out);
    if (rc == COVGPU_ERR_GATE_TIMEOUT) { g_err = "solve: stream ordering failed"; rc = COVGPU_ERR_NO_DEVICE; }
  }
  return rc;
}
extern "C" int covgpu_solve_resident(covgpu_context* c, const covgpu_options* opt, covgpu_result* out) { return guarded([&] { return solve_any(c, opt, out); }); }
extern "C" int covgpu_download(covgpu_context* c, covgpu_problem* p) { return guarded([&] { return download_impl(c, p); }); }

static int full_solve(covgpu_context* c, const covgpu_options* opt, covgpu_problem* p, covgpu_result* out, bool pgo) {
  covgpu_result local;
  auto t0 = std::chrono::steady_clock::now();
  RC(upload_impl(c, opt, p, pgo));
  const double t_up = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  RC(solve_any(c, opt, &local));
  t0 = std::chrono::steady_clock::now();
  RC(download_impl(c, p));
  local.t_download_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  local.t_upload_s = t_up;
  if (out) *out = local;
  return COVGPU_OK;
}
extern "C" int covgpu_gba_solve(covgpu_context* c, const covgpu_options* opt, covgpu_problem* p, covgpu_result* out) { return guarded([&] { return full_solve(c, opt, p, out, false); }); }
extern "C" int covgpu_pgo_solve(covgpu_context* c, const covgpu_options* opt, covgpu_problem* p, covgpu_result* out) { return guarded([&] { return full_solve(c, opt, p, out, true); }); }

// ------------------------------------------------------------------------------------------------ both rounds of a GlobalBundleAdjustment call
// optimization_be.cpp:56-618 builds and solves TWO problems: the outlier round (5 iterations, loop edges without loss, :62-265), then —
// after erasing the observations whose loss-corrected residual exceeds th_gba_outlier_global (:270-290) — the main round on the rebuilt
// problem (:296-567), restarted from the SAME map state (the outlier round's estimate is discarded). The second problem is the first
// minus the erased observations, minus the landmarks left with fewer than two (:428-440), with the loop edges' loss switched on (:555)
// and optionally more constant poses (:338-341): nothing a second Map -> IR flatten and a second H2D upload would have to tell the
// device. Here it is derived ON the device from the resident first round (k_pairs.hip: round2_compact_device; pair lists rebuilt by
// the same device code as an upload; the elimination tree stays — couplings only disappear), and only the erase flags travel back for
// the caller's map bookkeeping. p: the FIRST round's problem; on return its poses / speed-bias hold the second round's estimate and
// lm_pos[l] the second round's for every landmark with lm_left[l] >= 2 (the others took no part, as in the reference).
// State of the context afterwards: the resident problem is the SECOND round's (compacted landmark / observation indexing), which the
// caller's covgpu_problem does not describe — so the context is marked as holding no problem when the call returns, successfully or not:
// covgpu_download / covgpu_outlier_pass / covgpu_solve_resident after it fail with "no problem uploaded" instead of scattering into the
// first round's slots, and an error half-way cannot leave a half-switched problem behind. (Everything the caller needs came back through
// p, obs_erase, lm_left and the two results.)
static int gba_two_round_impl(covgpu_context* c, const covgpu_options* opt, covgpu_problem* p, const covgpu_two_round* tr, uint8_t* obs_erase,
                              int32_t* lm_left, int64_t* counts, covgpu_result* round1, covgpu_result* round2);
extern "C" int covgpu_gba_two_round(covgpu_context* c, const covgpu_options* opt, covgpu_problem* p, const covgpu_two_round* tr, uint8_t* obs_erase,
                                    int32_t* lm_left, int64_t* counts, covgpu_result* round1, covgpu_result* round2) {
  const int rc = gba_two_round_impl(c, opt, p, tr, obs_erase, lm_left, counts, round1, round2);
  if (c) { c->have = false; }
  return rc;
}
static int gba_two_round_impl(covgpu_context* c, const covgpu_options* opt, covgpu_problem* p, const covgpu_two_round* tr, uint8_t* obs_erase,
                              int32_t* lm_left, int64_t* counts, covgpu_result* round1, covgpu_result* round2) {
  return guarded([&]() -> int {
    if (!tr || !obs_erase || !lm_left) { g_err = "covgpu_gba_two_round: NULL argument"; return (int)COVGPU_ERR_INVALID_ARG; }
    // (round 6: a sharded context takes the call too — every rank derives the second round of ITS share on its device: its landmarks' outlier decisions,
    //  compaction and pair lists are local, the elimination tree stays, and both solves issue the same collectives on every rank. The caller merges the
    //  ranks' flags and estimates: covgpu_gba_two_round_multi.)
    covgpu_options o1 = *opt;
    o1.max_iterations = tr->round1_iterations > 0 ? tr->round1_iterations : 5;   // :262
    auto t0 = std::chrono::steady_clock::now();
    RC(upload_impl(c, &o1, p, false));
    covgpu_result a, b;
    const double t_up = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    RC(solve_any(c, &o1, &a));
    a.t_upload_s = t_up;
    DevProblem& P = c->P;
    t0 = std::chrono::steady_clock::now();
    // ---- outlier decisions at the resident estimate (:270-290); the flags stay on the device and a copy goes to the caller
    unsigned char* de = nullptr; int* dl = nullptr; unsigned long long* dc = nullptr;
    RC(dev_alloc(c, &de, (size_t)std::max(P.O, 1))); RC(dev_alloc(c, &dl, (size_t)std::max(P.L, 1))); RC(dev_alloc(c, &dc, (size_t)2));
    HIPCHK(hipMemsetAsync(dc, 0, 2 * sizeof(unsigned long long), c->st));
    launch_lm_outliers(P, tr->outlier_threshold, de, dl, dc, c->st);
    unsigned long long hc[2] = {0, 0};
    if (P.O) HIPCHK(hipMemcpyAsync(obs_erase, de, (size_t)P.O, hipMemcpyDeviceToHost, c->st));
    if (P.L) HIPCHK(hipMemcpyAsync(lm_left, dl, sizeof(int) * (size_t)P.L, hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipMemcpyAsync(hc, dc, sizeof(hc), hipMemcpyDeviceToHost, c->st));
    // ---- the second round's observation stream, compacted on the device
    Round2Lists R;
    if (!round2_compact_device(P.L, P.O, P.K, de, dl, P.obs_kf, P.obs_lm, P.obs_u, P.obs_v, P.obs_sigma, P.lm0, c->st, R)) {
      g_err = "covgpu_gba_two_round: device allocation failed while compacting the observation stream"; return (int)COVGPU_ERR_OUT_OF_MEMORY;
    }
    for (void* q : {(void*)R.lm0, (void*)R.obs_u, (void*)R.obs_v, (void*)R.obs_sigma, (void*)R.lm_obs_ptr, (void*)R.lm_old, (void*)R.obs_kf, (void*)R.obs_lm, (void*)R.kf_obs_ptr,
                    (void*)R.kf_obs_idx})
      if (q) c->allocs.push_back(q);
    if (counts) { counts[0] = (int64_t)hc[0]; counts[1] = (int64_t)hc[1]; }   // (round2_compact_device ends with a stream synchronisation)
    const int L1 = P.L;
    if (R.L2 > 0 && R.O2 > 0) {
      P.L = R.L2; P.O = R.O2; P.N = P.n + 3 * P.L;
      P.lm0 = R.lm0; P.lm_obs_ptr = R.lm_obs_ptr; P.obs_kf = R.obs_kf; P.obs_lm = R.obs_lm; P.obs_u = R.obs_u; P.obs_v = R.obs_v; P.obs_sigma = R.obs_sigma;
      P.kf_obs_ptr = R.kf_obs_ptr; P.kf_obs_idx = R.kf_obs_idx;
    } else { P.L = 0; P.O = 0; P.N = P.n; }
    if (tr->kf_fixed_round2 != nullptr) {   // opt.gba_fix_poses_loaded_maps (:338-341): more constant poses in the second round
      HIPCHK(hipMemcpyAsync(P.fixed, tr->kf_fixed_round2, (size_t)P.K, hipMemcpyHostToDevice, c->st));
      std::vector<int> key(P.K);
      for (int k = 0; k < P.K; ++k) key[k] = tr->kf_fixed_round2[k] ? -1 : c->h_perm[k];
      HIPCHK(hipMemcpyAsync(c->d_pairkey, key.data(), sizeof(int) * (size_t)P.K, hipMemcpyHostToDevice, c->st));
      HIPCHK(hipStreamSynchronize(c->st));
    }
    {  // covisible pairs of what is left (same device code as an upload); the elimination tree stays: couplings only disappear
      PairLists pl;
      if (!build_pairs_device(P.L, P.K, P.lm_obs_ptr, P.obs_kf, c->d_pairkey, true, c->st, pl)) { g_err = "covisible pair lists: device allocation failed"; return (int)COVGPU_ERR_OUT_OF_MEMORY; }
      for (int* q : {pl.pair_ptr, pl.pair_i, pl.pair_j, pl.pair_oa, pl.pair_ob}) if (q) c->allocs.push_back(q);
      P.npairs = pl.npairs; P.pair_ptr = pl.pair_ptr; P.pair_i = pl.pair_i; P.pair_j = pl.pair_j; P.pair_oa = pl.pair_oa; P.pair_ob = pl.pair_ob;
      launch_kobs_build(P, P.pair_oa, P.pair_ob, pl.nent, c->st);
    }
    // (the reductions store one partial sum per wave into fixed slots and never clear them: fewer observations = fewer waves, and the
    //  first round's partials beyond them would be summed again)
    HIPCHK(hipMemsetAsync(P.part, 0, (size_t)SC_COUNT * P.part_n * sizeof(double), c->st));
    if (!tr->use_loops_round2) { P.E = 0; P.nepairs = 0; }   // opt.gba_use_map_loop_constraints (:534)
    else if (P.E > 0) {
      std::vector<double> la((size_t)P.E, tr->loop_loss_round2);   // Cauchy on the loop edges in the second round (:555)
      HIPCHK(hipMemcpyAsync(P.edge_loss_a, la.data(), sizeof(double) * (size_t)P.E, hipMemcpyHostToDevice, c->st));
    }
    HIPCHK(hipStreamSynchronize(c->st));
    const double t_prep = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    RC(solve_any(c, opt, &b));
    b.t_upload_s = t_prep;   // (what replaces the second upload: outlier pass, compaction, pair lists)
    // ---- estimate of the second round -> caller's arrays; landmarks through the first-round index of every kept one
    t0 = std::chrono::steady_clock::now();
    HIPCHK(hipMemcpyAsync(p->kf_pose, P.pose, (size_t)7 * P.K * sizeof(double), hipMemcpyDeviceToHost, c->st));
    if (P.vi && p->kf_speed_bias) HIPCHK(hipMemcpyAsync(p->kf_speed_bias, P.sb, (size_t)9 * P.K * sizeof(double), hipMemcpyDeviceToHost, c->st));
    std::vector<double> lm2((size_t)3 * P.L);
    std::vector<int> old((size_t)P.L);
    if (P.L) {
      HIPCHK(hipMemcpyAsync(lm2.data(), P.lm, lm2.size() * sizeof(double), hipMemcpyDeviceToHost, c->st));
      HIPCHK(hipMemcpyAsync(old.data(), R.lm_old, old.size() * sizeof(int), hipMemcpyDeviceToHost, c->st));
    }
    HIPCHK(hipStreamSynchronize(c->st));
    for (int q = 0; q < P.L; ++q) {
      if (old[q] < 0 || old[q] >= L1) { g_err = "covgpu_gba_two_round: landmark map out of range"; return (int)COVGPU_ERR_INVALID_ARG; }
      std::memcpy(p->lm_pos + 3 * (size_t)old[q], lm2.data() + 3 * (size_t)q, 3 * sizeof(double));
    }
    b.t_download_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (round1) *round1 = a;
    if (round2) *round2 = b;
    return (int)COVGPU_OK;
  });
}

// ------------------------------------------------------------------------------------------------ in-process multi-GPU solve
// What a covins_backend process (one process, several GPUs) calls: the whole sharded GlobalBundleAdjustment solve behind one
// C entry point — plan on the full problem, one context and one host thread per rank, the native collective (RCCL when the
// ranks sit on different devices; the in-process group when they share one: virtual ranks, the one-GPU test form), merge.
// Optionally also the outlier decisions of optimization_be.cpp:270-290 at the estimate the solve left resident on each rank.
namespace {
struct SubProblem {
  std::vector<double> lm_pos, obs_uv, obs_sigma, imu_samples, imu_first, imu_noise, edge_meas, edge_info, edge_loss, pose, sb;
  std::vector<int32_t> lm_obs_ptr, obs_kf, imu_i, imu_j, imu_ptr, edge_i, edge_j, lm_id, obs_id;
  covgpu_problem view;
};
void make_sub(const covgpu_problem& p, int r, const int32_t* lm_rank, const int32_t* imu_rank, const int32_t* edge_rank, SubProblem& s) {
  s.pose.assign(p.kf_pose, p.kf_pose + 7 * (size_t)p.num_kf);
  if (p.kf_speed_bias) s.sb.assign(p.kf_speed_bias, p.kf_speed_bias + 9 * (size_t)p.num_kf);
  s.lm_obs_ptr.assign(1, 0);
  for (int l = 0; l < p.num_lm; ++l) {
    if (lm_rank[l] != r) continue;
    s.lm_id.push_back(l);
    s.lm_pos.insert(s.lm_pos.end(), p.lm_pos + 3 * (size_t)l, p.lm_pos + 3 * (size_t)l + 3);
    for (int o = p.lm_obs_ptr[l]; o < p.lm_obs_ptr[l + 1]; ++o) {
      s.obs_id.push_back(o); s.obs_kf.push_back(p.obs_kf[o]);
      s.obs_uv.push_back(p.obs_uv[2 * (size_t)o]); s.obs_uv.push_back(p.obs_uv[2 * (size_t)o + 1]); s.obs_sigma.push_back(p.obs_sigma[o]);
    }
    s.lm_obs_ptr.push_back((int32_t)s.obs_kf.size());
  }
  s.imu_ptr.assign(1, 0);
  for (int f = 0; f < p.num_imu; ++f) {
    if (imu_rank[f] != r) continue;
    s.imu_i.push_back(p.imu_kf_i[f]); s.imu_j.push_back(p.imu_kf_j[f]);
    s.imu_samples.insert(s.imu_samples.end(), p.imu_samples + 7 * (size_t)p.imu_sample_ptr[f], p.imu_samples + 7 * (size_t)p.imu_sample_ptr[f + 1]);
    s.imu_ptr.push_back((int32_t)(s.imu_samples.size() / 7));
    s.imu_first.insert(s.imu_first.end(), p.imu_first + 6 * (size_t)f, p.imu_first + 6 * (size_t)f + 6);
    if (p.imu_noise) s.imu_noise.insert(s.imu_noise.end(), p.imu_noise + 5 * (size_t)f, p.imu_noise + 5 * (size_t)f + 5);
  }
  for (int e = 0; e < p.num_edge; ++e) {
    if (edge_rank[e] != r) continue;
    s.edge_i.push_back(p.edge_i[e]); s.edge_j.push_back(p.edge_j[e]);
    s.edge_meas.insert(s.edge_meas.end(), p.edge_meas + 7 * (size_t)e, p.edge_meas + 7 * (size_t)e + 7);
    s.edge_info.insert(s.edge_info.end(), p.edge_sqrt_info + 36 * (size_t)e, p.edge_sqrt_info + 36 * (size_t)e + 36);
    s.edge_loss.push_back(p.edge_loss_a[e]);
  }
  covgpu_problem& v = s.view;
  v = p;
  v.kf_pose = s.pose.data(); v.kf_speed_bias = p.kf_speed_bias ? s.sb.data() : nullptr;
  v.num_lm = (int32_t)s.lm_id.size(); v.num_obs = (int32_t)s.obs_kf.size(); v.num_imu = (int32_t)s.imu_i.size(); v.num_edge = (int32_t)s.edge_i.size();
  v.num_imu_samples = (int32_t)(s.imu_samples.size() / 7);
  v.lm_pos = s.lm_pos.data(); v.lm_obs_ptr = s.lm_obs_ptr.data(); v.obs_kf = s.obs_kf.data(); v.obs_uv = s.obs_uv.data(); v.obs_sigma = s.obs_sigma.data();
  v.imu_kf_i = s.imu_i.data(); v.imu_kf_j = s.imu_j.data(); v.imu_sample_ptr = s.imu_ptr.data(); v.imu_samples = s.imu_samples.data();
  v.imu_first = s.imu_first.data(); v.imu_noise = p.imu_noise ? s.imu_noise.data() : nullptr;
  v.edge_i = s.edge_i.data(); v.edge_j = s.edge_j.data(); v.edge_meas = s.edge_meas.data(); v.edge_sqrt_info = s.edge_info.data(); v.edge_loss_a = s.edge_loss.data();
}
}  // namespace

// tr == nullptr: ONE solve per rank (+ the outlier decisions at its estimate when obs_erase is given) | tr != nullptr: both rounds of a GlobalBundleAdjustment
// call per rank (gba_two_round_impl on the rank's sharded context; `out1` = the first round's result)
static int gba_multi_impl(const covgpu_options* opt, covgpu_problem* p, covgpu_result* out, int32_t n_ranks, const int32_t* devices,
                          double outlier_threshold, uint8_t* obs_erase, int32_t* lm_left, int64_t* counts, const covgpu_two_round* tr, covgpu_result* out1) {
  return guarded([&] {
    if (n_ranks < 1 || n_ranks > 16 || !devices) { g_err = "covgpu_gba_solve_multi: 1..16 ranks with a device each"; return (int)COVGPU_ERR_INVALID_ARG; }
    std::vector<int32_t> lm_rank(std::max(p->num_lm, 1)), imu_rank(std::max(p->num_imu, 1)), edge_rank(std::max(p->num_edge, 1));
    covgpu_nd_plan* plan = nullptr;
    const int nsub = covgpu_shard_plan(opt, p, n_ranks, &plan, lm_rank.data(), imu_rank.data(), edge_rank.data());
    if (nsub <= 0) { g_err = "covgpu_gba_solve_multi: the problem does not split (" + g_err + ")"; return (int)COVGPU_ERR_INVALID_ARG; }
    struct PlanGuard { covgpu_nd_plan* p; ~PlanGuard() { covgpu_nd_plan_destroy(p); } } plan_guard{plan};   // freed on every exit path
    bool shared_device = false;
    for (int a = 0; a < n_ranks; ++a) for (int b = 0; b < a; ++b) shared_device |= devices[a] == devices[b];
    covgpu_group* grp = nullptr;
    uint8_t uid[128];
    if (shared_device || n_ranks == 1) { RC(covgpu_group_create(n_ranks, &grp)); }
    else { RC(covgpu_rccl_unique_id(uid)); }
    struct GroupGuard { covgpu_group* g; ~GroupGuard() { if (g) covgpu_group_destroy(g); } } grp_guard{grp};
    std::vector<SubProblem> sub(n_ranks);
    for (int r = 0; r < n_ranks; ++r) make_sub(*p, r, lm_rank.data(), imu_rank.data(), edge_rank.data(), sub[r]);
    std::vector<covgpu_result> res(n_ranks), res1(n_ranks);
    std::vector<int> rcs(n_ranks, COVGPU_OK);
    std::vector<std::string> errs(n_ranks);
    std::vector<std::vector<uint8_t>> er(n_ranks);
    std::vector<std::vector<int32_t>> ll(n_ranks);
    std::vector<int64_t> cnt(2 * (size_t)n_ranks, 0);
    // Failure protocol. A rank that fails raises `fail`; the stages that cannot block (context creation, validation + upload of the
    // sub-problem) are separated from the next one by a host barrier after which EVERY rank checks the flag — so no rank enters
    // ncclCommInitRank or its first all-reduce while a peer has already given up. Inside the solve the peers poll the flag while they
    // wait for an iteration (wait_iteration) and abort their communicator; the in-process group is aborted directly.
    std::atomic<int> fail{0};
    struct HostBarrier {
      std::mutex m; std::condition_variable cv; int count = 0, n = 0; long gen = 0;
      void wait() { std::unique_lock<std::mutex> lk(m); const long g0 = gen; if (++count == n) { count = 0; ++gen; cv.notify_all(); } else cv.wait(lk, [&] { return gen != g0; }); }
    } bar;
    bar.n = n_ranks;
    auto work = [&](int r) {
      covgpu_context* c = nullptr;
      covgpu_options o = *opt; o.device = devices[r];
      int rc = COVGPU_OK;
      auto give_up = [&](int code) { if (rc == COVGPU_OK && code != COVGPU_OK) { rc = code; errs[r] = covgpu_last_error(); fail.store(1); if (grp) covgpu_group_abort(grp); } };
      give_up(covgpu_create(&o, &c));                                     // stage 1: a context on the rank's device
      bar.wait();
      if (!fail.load()) give_up(grp ? covgpu_set_shard_group(c, plan, r, grp) : covgpu_set_shard_rccl(c, plan, r, n_ranks, uid));   // stage 2: the collective (ncclCommInitRank: every rank is here)
      if (c) c->peer_fail = &fail;
      bar.wait();
      if (tr != nullptr) {
        // both rounds behind one call on this rank's share (upload, outlier round, device-side second round, solve, download into the share's arrays)
        if (!fail.load()) {
          er[r].assign(sub[r].obs_kf.size() + 1, 0); ll[r].assign(sub[r].lm_id.size() + 1, 0);
          give_up(gba_two_round_impl(c, &o, &sub[r].view, tr, er[r].data(), ll[r].data(), &cnt[2 * (size_t)r], &res1[r], &res[r]));
          if (c) c->have = false;
        }
        rcs[r] = rc;
        if (c) covgpu_destroy(c);
        return;
      }
      const auto t_up0 = std::chrono::steady_clock::now();
      if (!fail.load()) give_up(guarded([&] { return upload_impl(c, &o, &sub[r].view, false); }));  // (guarded: a host exception in a rank's thread would otherwise terminate the process with the peers parked at the barrier) stage 3: validation + H2D of the rank's share (OOM, malformed share)
      const double t_up = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_up0).count();
      bar.wait();
      if (!fail.load()) give_up(guarded([&] { return solve_any(c, &o, &res[r]); }));               // stage 4: the solve (peers of a rank that fails in here: wait_iteration)
      if (!fail.load() && rc == COVGPU_OK) {
        const auto t_dn0 = std::chrono::steady_clock::now();
        give_up(guarded([&] { return download_impl(c, &sub[r].view); }));
        res[r].t_download_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_dn0).count();
        res[r].t_upload_s = t_up;
      }
      if (!fail.load() && rc == COVGPU_OK && obs_erase) {
        give_up(guarded([&] {
          er[r].assign(sub[r].obs_kf.size() + 1, 0); ll[r].assign(sub[r].lm_id.size() + 1, 0);
          return covgpu_outlier_pass(c, outlier_threshold, er[r].data(), ll[r].data(), &cnt[2 * (size_t)r]);
        }));
      }
      rcs[r] = rc;
      if (c) covgpu_destroy(c);
    };
    std::vector<std::thread> th;
    for (int r = 1; r < n_ranks; ++r) th.emplace_back(work, r);
    work(0);
    for (auto& t : th) t.join();
    int rc = COVGPU_OK;
    for (int r = 0; r < n_ranks; ++r) if (rcs[r]) { rc = rcs[r]; g_err = "rank " + std::to_string(r) + ": " + errs[r]; break; }
    if (!rc && fail.load()) { rc = COVGPU_ERR_NO_DEVICE; g_err = "covgpu_gba_solve_multi: a rank gave up"; }
    if (!rc) {
      std::vector<int32_t> pr(p->num_kf), sr(p->num_kf);
      covgpu_nd_plan_owner(plan, pr.data(), sr.data());
      for (int k = 0; k < p->num_kf; ++k) {
        const int a = pr[k] < 0 ? 0 : pr[k], b = sr[k] < 0 ? 0 : sr[k];
        std::memcpy(p->kf_pose + 7 * (size_t)k, sub[a].pose.data() + 7 * (size_t)k, 7 * sizeof(double));
        if (p->kf_speed_bias) std::memcpy(p->kf_speed_bias + 9 * (size_t)k, sub[b].sb.data() + 9 * (size_t)k, 9 * sizeof(double));
      }
      if (counts) { counts[0] = 0; counts[1] = 0; }
      for (int r = 0; r < n_ranks; ++r) {
        for (size_t i = 0; i < sub[r].lm_id.size(); ++i) {
          std::memcpy(p->lm_pos + 3 * (size_t)sub[r].lm_id[i], sub[r].lm_pos.data() + 3 * i, 3 * sizeof(double));
          if (obs_erase && lm_left) lm_left[sub[r].lm_id[i]] = ll[r][i];
        }
        if (obs_erase) for (size_t i = 0; i < sub[r].obs_id.size(); ++i) obs_erase[sub[r].obs_id[i]] = er[r][i];
        if (counts) { counts[0] += cnt[2 * (size_t)r]; counts[1] += cnt[2 * (size_t)r + 1]; }
      }
      auto combine = [&](covgpu_result* dst, const std::vector<covgpu_result>& rs) {
        if (!dst) return;
        // the trust-region trace, costs and counts are identical on every rank (all-reduced scalars): rank 0's. Per-rank fields
        // are combined: IMU factors dropped for a non-PD covariance are counted where the factor lives (sum), timings are the slowest rank's
        *dst = rs[0];
        for (int r = 1; r < n_ranks; ++r) {
          dst->reserved += rs[r].reserved;
          dst->t_upload_s = std::max(dst->t_upload_s, rs[r].t_upload_s); dst->t_download_s = std::max(dst->t_download_s, rs[r].t_download_s);
          dst->t_solve_s = std::max(dst->t_solve_s, rs[r].t_solve_s);
        }
      };
      combine(out, res);
      if (tr != nullptr) combine(out1, res1);
    }
    return rc;
  });
}
extern "C" int covgpu_gba_solve_multi(const covgpu_options* opt, covgpu_problem* p, covgpu_result* out, int32_t n_ranks, const int32_t* devices,
                                      double outlier_threshold, uint8_t* obs_erase, int32_t* lm_left, int64_t* counts) {
  return gba_multi_impl(opt, p, out, n_ranks, devices, outlier_threshold, obs_erase, lm_left, counts, nullptr, nullptr);
}
// Both rounds of a GlobalBundleAdjustment call on n_ranks devices (round 6): ONE flatten and ONE upload per rank, the second round derived on every rank's
// device from its share (covgpu_gba_two_round's scheme); obs_erase / lm_left / counts / the estimates merged as covgpu_gba_solve_multi merges them.
extern "C" int covgpu_gba_two_round_multi(const covgpu_options* opt, covgpu_problem* p, const covgpu_two_round* tr, int32_t n_ranks, const int32_t* devices,
                                          uint8_t* obs_erase, int32_t* lm_left, int64_t* counts, covgpu_result* round1, covgpu_result* round2) {
  if (!tr || !obs_erase || !lm_left) { g_err = "covgpu_gba_two_round_multi: NULL argument"; return COVGPU_ERR_INVALID_ARG; }
  return gba_multi_impl(opt, p, round2, n_ranks, devices, tr->outlier_threshold, obs_erase, lm_left, counts, tr, round1);
}

// ------------------------------------------------------------------------------------------------ test entry points
template <typename T>
static int fetch(covgpu_context* c, T* host, const T* dev, size_t count) {
  if (count) HIPCHK(hipMemcpyAsync(host, dev, count * sizeof(T), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  return COVGPU_OK;
}

extern "C" int covgpu_reprojection_residual_norms(covgpu_context* c, const covgpu_options* opt, const covgpu_problem* p, double* norms) {
  covgpu_options o = *opt; o.visual_only = 1;
  RC(upload_impl(c, &o, p, false)); RC(reset_state(c));
  double* d; RC(dev_alloc(c, &d, (size_t)c->P.O));
  launch_obs_norms(c->P, d, c->st);
  return fetch(c, norms, d, (size_t)c->P.O);
}

// Outlier decisions + landmark bookkeeping at the RESIDENT estimate (the state covgpu_gba_solve / covgpu_solve_resident
// left on the device): no re-upload, only flags and counts come back.
extern "C" int covgpu_outlier_pass(covgpu_context* c, double threshold, uint8_t* obs_erase, int32_t* lm_left, int64_t* counts) {
  if (!c->have || c->pgo) { g_err = "covgpu_outlier_pass needs a resident GBA problem (call covgpu_gba_solve or covgpu_upload + covgpu_solve_resident first)"; return COVGPU_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  const DevProblem& P = c->P;
  unsigned char* de = nullptr; int* dl = nullptr; unsigned long long* dc = nullptr;
  HIPCHK(hipMalloc((void**)&de, (size_t)std::max(P.O, 1))); HIPCHK(hipMalloc((void**)&dl, sizeof(int) * (size_t)std::max(P.L, 1)));
  HIPCHK(hipMalloc((void**)&dc, 2 * sizeof(unsigned long long)));
  HIPCHK(hipMemsetAsync(dc, 0, 2 * sizeof(unsigned long long), c->st));
  launch_lm_outliers(P, threshold, de, dl, dc, c->st);
  unsigned long long hc[2] = {0, 0};
  if (P.O) HIPCHK(hipMemcpyAsync(obs_erase, de, (size_t)P.O, hipMemcpyDeviceToHost, c->st));
  if (P.L) HIPCHK(hipMemcpyAsync(lm_left, dl, sizeof(int) * (size_t)P.L, hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipMemcpyAsync(hc, dc, sizeof(hc), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  (void)hipFree(de); (void)hipFree(dl); (void)hipFree(dc);
  if (counts) { counts[0] = (int64_t)hc[0]; counts[1] = (int64_t)hc[1]; }
  return COVGPU_OK;
}

// Covisibility recount on the RESIDENT problem (Keyframe::UpdateCovisibilityConnections, keyframe_be.cpp:559-608, which
// backend.cpp:164-167 runs over every keyframe after a GBA): weight(i, j) = number of landmarks both keyframes observe; pairs
// with weight >= threshold (sys.covis_thres) come back as (kf_i > kf_j, weight), sorted by (kf_i, kf_j). All keyframes take
// part, constant ones included. The counting runs on the device (k_pairs.hip); *count = pairs found (may exceed capacity).
extern "C" int covgpu_covisibility(covgpu_context* c, int32_t threshold, int64_t capacity, int32_t* kf_i, int32_t* kf_j, int32_t* weight, int64_t* count) {
  return guarded([&]() -> int {
    if (!c->have || c->pgo) { g_err = "covgpu_covisibility needs a resident GBA problem"; return (int)COVGPU_ERR_INVALID_ARG; }
    HIPCHK(hipSetDevice(c->device));
    const DevProblem& P = c->P;
    std::vector<int> iota(P.K);
    for (int k = 0; k < P.K; ++k) iota[k] = k;
    int* d_key = nullptr;
    HIPCHK(hipMalloc((void**)&d_key, sizeof(int) * (size_t)P.K));
    HIPCHK(hipMemcpyAsync(d_key, iota.data(), sizeof(int) * (size_t)P.K, hipMemcpyHostToDevice, c->st));
    PairLists pl;
    const bool ok = build_pairs_device(P.L, P.K, P.lm_obs_ptr, P.obs_kf, d_key, false, c->st, pl);
    (void)hipFree(d_key);
    if (!ok) { g_err = "covgpu_covisibility: device allocation failed"; return (int)COVGPU_ERR_OUT_OF_MEMORY; }
    std::vector<int> pi(pl.npairs), pj(pl.npairs), pp((size_t)pl.npairs + 1);
    if (pl.npairs) {
      HIPCHK(hipMemcpyAsync(pi.data(), pl.pair_i, sizeof(int) * (size_t)pl.npairs, hipMemcpyDeviceToHost, c->st));
      HIPCHK(hipMemcpyAsync(pj.data(), pl.pair_j, sizeof(int) * (size_t)pl.npairs, hipMemcpyDeviceToHost, c->st));
    }
    HIPCHK(hipMemcpyAsync(pp.data(), pl.pair_ptr, sizeof(int) * ((size_t)pl.npairs + 1), hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    for (int* q : {pl.pair_ptr, pl.pair_i, pl.pair_j}) if (q) (void)hipFree(q);
    int64_t n = 0;
    for (int q = 0; q < pl.npairs; ++q) {
      const int w = pp[q + 1] - pp[q];
      if (w < threshold) continue;
      if (n < capacity) { kf_i[n] = pi[q]; kf_j[n] = pj[q]; weight[n] = w; }
      ++n;
    }
    *count = n;
    return (int)COVGPU_OK;
  });
}

extern "C" int covgpu_linearize_reprojection(covgpu_context* c, const covgpu_options* opt, const covgpu_problem* p, double* r, double* Jp,
                                             double* Jl, double* cost) {
  covgpu_options o = *opt; o.visual_only = 1;
  RC(upload_impl(c, &o, p, false)); RC(reset_state(c));
  const size_t O = c->P.O;
  double *dr, *dJp, *dJl, *dc;
  RC(dev_alloc(c, &dr, 2 * O)); RC(dev_alloc(c, &dJp, 12 * O)); RC(dev_alloc(c, &dJl, 6 * O)); RC(dev_alloc(c, &dc, O));
  launch_obs_linearize(c->P, dr, dJp, dJl, dc, c->st);
  RC(fetch(c, r, dr, 2 * O)); RC(fetch(c, Jp, dJp, 12 * O)); RC(fetch(c, Jl, dJl, 6 * O));
  return fetch(c, cost, dc, O);
}

extern "C" int covgpu_preintegrate(covgpu_context* c, const covgpu_options* opt, const covgpu_problem* p, double* delta, double* J, double* Pm) {
  covgpu_options o = *opt; o.visual_only = 0;
  RC(upload_impl(c, &o, p, false)); RC(reset_state(c));
  launch_preintegrate(c->P, c->st);
  const size_t I = c->P.I;
  RC(fetch(c, delta, c->P.pre_delta, 11 * I)); RC(fetch(c, J, c->P.pre_J, 225 * I));
  return fetch(c, Pm, c->P.pre_P, 225 * I);
}

extern "C" int covgpu_linearize_imu(covgpu_context* c, const covgpu_options* opt, const covgpu_problem* p, double* r, double* J) {
  covgpu_options o = *opt; o.visual_only = 0;
  RC(upload_impl(c, &o, p, false)); RC(reset_state(c));
  launch_preintegrate(c->P, c->st);
  const size_t I = c->P.I;
  double *dr, *dJ;
  RC(dev_alloc(c, &dr, 15 * I)); RC(dev_alloc(c, &dJ, 450 * I));
  launch_imu_linearize(c->P, dr, dJ, c->st);
  RC(fetch(c, r, dr, 15 * I));
  return fetch(c, J, dJ, 450 * I);
}

extern "C" int covgpu_linearize_between(covgpu_context* c, const covgpu_options* opt, const covgpu_problem* p, double* r, double* J, double* cost) {
  RC(upload_impl(c, opt, p, true)); RC(reset_state(c));
  const size_t E = c->P.E;
  double *dr, *dJ, *dc;
  RC(dev_alloc(c, &dr, 6 * E)); RC(dev_alloc(c, &dJ, 72 * E)); RC(dev_alloc(c, &dc, E));
  launch_edge_linearize(c->P, dr, dJ, dc, c->st);
  RC(fetch(c, r, dr, 6 * E)); RC(fetch(c, J, dJ, 72 * E));
  return fetch(c, cost, dc, E);
}

static int schur_impl(covgpu_context* c, const covgpu_options* opt, const covgpu_problem* p, bool pgo, double mu, double* S, double* b, double* cost) {
  RC(upload_impl(c, opt, p, pgo, false)); RC(reset_state(c));  // dense form: the test reads C back as one matrix
  launch_preintegrate(c->P, c->st);
  enqueue_build(c, mu);
  RC(read_scalars(c));
  *cost = c->h_scal[SC_COST];
  // assemble the reduced system in IR layout (D rows per keyframe) from its structured parts
  const DevProblem& P = c->P;
  const int n = P.n, npad = P.npad, K = P.K, D = P.D;
  std::vector<double> C((size_t)npad * npad);
  std::vector<int> perm(K), pos_kf(K), cptr(P.nchains + 1);
  RC(fetch(c, C.data(), P.Sred, C.size()));
  RC(fetch(c, perm.data(), P.perm, (size_t)K)); RC(fetch(c, pos_kf.data(), P.pos_kf, (size_t)K));
  RC(fetch(c, cptr.data(), P.chain_ptr, cptr.size()));
  std::fill(S, S + (size_t)n * n, 0.0);
  auto Sat = [&](int r, int cc) -> double& { return S[(size_t)r * n + cc]; };
  for (int a = 0; a < K; ++a)
    for (int bq = 0; bq < K; ++bq) {
      const int ra = 6 * perm[a], rb = 6 * perm[bq];
      for (int r = 0; r < 6; ++r)
        for (int cc = 0; cc < 6; ++cc) {
          const int i = ra + r, j = rb + cc;
          Sat(D * a + r, D * bq + cc) = (j <= i) ? C[(size_t)i * npad + j] : C[(size_t)j * npad + i];
        }
    }
  if (P.vi) {
    std::vector<double> Ad(81 * (size_t)K), Ae(81 * (size_t)K), Bp(54 * (size_t)K), Bs(54 * (size_t)K), Bn(54 * (size_t)K);
    RC(fetch(c, Ad.data(), P.Ad, Ad.size())); RC(fetch(c, Ae.data(), P.Ae, Ae.size()));
    RC(fetch(c, Bp.data(), P.Bp, Bp.size())); RC(fetch(c, Bs.data(), P.Bs, Bs.size())); RC(fetch(c, Bn.data(), P.Bn, Bn.size()));
    std::vector<int> chain_of(K);
    for (int ch = 0; ch < P.nchains; ++ch) for (int q = cptr[ch]; q < cptr[ch + 1]; ++q) chain_of[q] = ch;
    for (int pos = 0; pos < K; ++pos) {
      const int a = pos_kf[pos], ch = chain_of[pos];
      for (int r = 0; r < 9; ++r)
        for (int cc = 0; cc < 9; ++cc) Sat(15 * a + 6 + r, 15 * a + 6 + cc) = Ad[81 * (size_t)pos + 9 * r + cc];
      auto put_b = [&](const std::vector<double>& B, int other_pos) {
        const int o = pos_kf[other_pos];
        for (int r = 0; r < 9; ++r)
          for (int cc = 0; cc < 6; ++cc) {
            const double v = B[54 * (size_t)pos + 6 * r + cc];
            Sat(15 * a + 6 + r, 15 * o + cc) = v; Sat(15 * o + cc, 15 * a + 6 + r) = v;
          }
      };
      put_b(Bs, pos);
      if (pos > cptr[ch]) {
        put_b(Bp, pos - 1);
        const int o = pos_kf[pos - 1];
        for (int r = 0; r < 9; ++r)
          for (int cc = 0; cc < 9; ++cc) {
            const double v = Ae[81 * (size_t)pos + 9 * r + cc];
            Sat(15 * a + 6 + r, 15 * o + 6 + cc) = v; Sat(15 * o + 6 + cc, 15 * a + 6 + r) = v;
          }
      }
      if (pos + 1 < cptr[ch + 1]) put_b(Bn, pos + 1);
    }
  }
  return fetch(c, b, P.bred, (size_t)n);
}
extern "C" int covgpu_schur(covgpu_context* c, const covgpu_options* opt, const covgpu_problem* p, double mu, double* S, double* b, double* cost) {
  return schur_impl(c, opt, p, false, mu, S, b, cost);
}
extern "C" int covgpu_schur_pgo(covgpu_context* c, const covgpu_options* opt, const covgpu_problem* p, double mu, double* S, double* b, double* cost) {
  return schur_impl(c, opt, p, true, mu, S, b, cost);
}

// one damped Gauss-Newton step at the uploaded estimate through the PRODUCT solve path (landmark elimination, multifrontal
// solve of the reduced camera system — sharded if a shard is set — and landmark back-substitution)
extern "C" int covgpu_gn_step(covgpu_context* c, const covgpu_options* opt, const covgpu_problem* p, double mu, double* dx, double* dl, double* cost) {
  RC(upload_impl(c, opt, p, false)); RC(reset_state(c));
  launch_preintegrate(c->P, c->st);
  enqueue_build(c, mu);
  enqueue_solve(c, c->P.gn);
  RC(read_scalars(c));
  if (chol_failed(c)) { g_err = "reduced system is not positive definite"; return COVGPU_ERR_NUMERIC; }
  if (cost) *cost = c->h_scal[SC_COST];
  RC(fetch(c, dx, c->P.gn, (size_t)c->P.n));
  return fetch(c, dl, c->P.gn + c->P.n, (size_t)3 * c->P.L);
}

extern "C" int covgpu_solve_reduced(covgpu_context* c, int32_t n, const double* S, const double* b, double* x) {
  HIPCHK(hipSetDevice(c->device));
  if (n <= 0) { g_err = "n <= 0"; return COVGPU_ERR_INVALID_ARG; }
  const int npad = ((n + kTile - 1) / kTile) * kTile;
  std::vector<double> Sp((size_t)npad * npad, 0.0), bp((size_t)2 * npad, 0.0);
  for (int r = 0; r < npad; ++r) {
    if (r < n) { std::memcpy(&Sp[(size_t)r * npad], S + (size_t)r * n, (size_t)(r + 1) * sizeof(double)); bp[r] = b[r]; }
    else Sp[(size_t)r * npad + r] = 1.0;
  }
  double *dS = nullptr, *db = nullptr, *dL = nullptr; int* df = nullptr;
  HIPCHK(hipMalloc((void**)&dS, Sp.size() * sizeof(double)));
  HIPCHK(hipMalloc((void**)&db, bp.size() * sizeof(double)));
  HIPCHK(hipMalloc((void**)&dL, (size_t)(npad / kTile) * kTile * kTile * sizeof(double)));
  HIPCHK(hipMalloc((void**)&df, 4 * sizeof(int)));
  HIPCHK(hipMemcpyAsync(dS, Sp.data(), Sp.size() * sizeof(double), hipMemcpyHostToDevice, c->st));
  HIPCHK(hipMemcpyAsync(db, bp.data(), bp.size() * sizeof(double), hipMemcpyHostToDevice, c->st));
  HIPCHK(hipMemsetAsync(df, 0, 4 * sizeof(int), c->st));
  dense_cholesky_solve_raw(dS, db, dL, df, npad, c->st, c->chol);
  int flag = 0;
  HIPCHK(hipMemcpyAsync(x, db, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipMemcpyAsync(&flag, df, sizeof(int), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  (void)hipFree(dS); (void)hipFree(db); (void)hipFree(dL); (void)hipFree(df);
  if (flag) { g_err = "reduced system is not positive definite"; return COVGPU_ERR_NUMERIC; }
  return COVGPU_OK;
}

extern "C" int covgpu_relpose_batch(covgpu_context* c, const covgpu_relpose_batch_t* bt, double th_outlier, int32_t min_inliers) {
  HIPCHK(hipSetDevice(c->device));
  if (!bt || bt->num_pairs < 0 || (bt->num_pairs > 0 && (!bt->corr_ptr || !bt->T_ab || !bt->inliers || !bt->cam_a || !bt->cam_b || !bt->dist_type_a || !bt->dist_type_b))) {
    g_err = "covgpu_relpose_batch: NULL array"; return COVGPU_ERR_INVALID_ARG;
  }
  const int B = bt->num_pairs;
  if (B == 0) return COVGPU_OK;
  for (int b = 0; b < B; ++b) if (bt->corr_ptr[b + 1] < bt->corr_ptr[b]) { g_err = "covgpu_relpose_batch: corr_ptr not monotone"; return COVGPU_ERR_INVALID_ARG; }
  const size_t C = (size_t)bt->corr_ptr[B];
  if (C > 0 && (!bt->p_a || !bt->p_b || !bt->kp_a || !bt->kp_b || !bt->sigma_a || !bt->sigma_b || !bt->outlier)) { g_err = "covgpu_relpose_batch: NULL correspondence array"; return COVGPU_ERR_INVALID_ARG; }
  std::vector<void*> tmp;
  auto up = [&](const void* h, size_t bytes, void** d) -> hipError_t {
    hipError_t e = hipMalloc(d, bytes ? bytes : 8);
    if (e != hipSuccess) return e;
    tmp.push_back(*d);
    return bytes ? hipMemcpyAsync(*d, h, bytes, hipMemcpyHostToDevice, c->st) : hipSuccess;
  };
  int *dptr_ = nullptr, *dda = nullptr, *ddb = nullptr, *din = nullptr;
  double *dpB = nullptr, *dpA = nullptr, *dkA = nullptr, *dkB = nullptr, *dsA = nullptr, *dsB = nullptr, *dcA = nullptr, *dcB = nullptr, *dT = nullptr;
  unsigned char* dout = nullptr;
  HIPCHK(up(bt->corr_ptr, sizeof(int) * (B + 1), (void**)&dptr_));
  HIPCHK(up(bt->p_b, 24 * C, (void**)&dpB)); HIPCHK(up(bt->p_a, 24 * C, (void**)&dpA));
  HIPCHK(up(bt->kp_a, 16 * C, (void**)&dkA)); HIPCHK(up(bt->kp_b, 16 * C, (void**)&dkB));
  HIPCHK(up(bt->sigma_a, 8 * C, (void**)&dsA)); HIPCHK(up(bt->sigma_b, 8 * C, (void**)&dsB));
  HIPCHK(up(bt->cam_a, 64 * (size_t)B, (void**)&dcA)); HIPCHK(up(bt->cam_b, 64 * (size_t)B, (void**)&dcB));
  HIPCHK(up(bt->dist_type_a, 4 * (size_t)B, (void**)&dda)); HIPCHK(up(bt->dist_type_b, 4 * (size_t)B, (void**)&ddb));
  HIPCHK(up(bt->T_ab, 56 * (size_t)B, (void**)&dT));
  HIPCHK(hipMalloc((void**)&din, 4 * (size_t)B)); tmp.push_back(din);
  HIPCHK(hipMalloc((void**)&dout, C ? C : 8)); tmp.push_back(dout);
  launch_relpose(B, dptr_, dpB, dpA, dkA, dkB, dsA, dsB, dcA, dda, dcB, ddb, th_outlier, min_inliers, dT, dout, din, c->st);
  HIPCHK(hipMemcpyAsync(bt->T_ab, dT, 56 * (size_t)B, hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipMemcpyAsync(bt->inliers, din, 4 * (size_t)B, hipMemcpyDeviceToHost, c->st));
  if (C) HIPCHK(hipMemcpyAsync(bt->outlier, dout, C, hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  for (void* p : tmp) (void)hipFree(p);
  return COVGPU_OK;
}

extern "C" int covgpu_pgo_reanchor(covgpu_context* c, int32_t K, const double* pose_old, const double* pose_new, double* velocity, int32_t L,
                                   const int32_t* ref_kf, double* lm_pos) {
  HIPCHK(hipSetDevice(c->device));
  if (K < 0 || L < 0 || (K > 0 && (!pose_old || !pose_new)) || (L > 0 && (!ref_kf || !lm_pos))) { g_err = "covgpu_pgo_reanchor: NULL array"; return COVGPU_ERR_INVALID_ARG; }
  for (int l = 0; l < L; ++l) if (ref_kf[l] >= K) { g_err = "covgpu_pgo_reanchor: ref_kf out of range"; return COVGPU_ERR_INVALID_ARG; }
  double *dpo, *dpn, *dv = nullptr, *dl; int* dr;
  std::vector<void*> tmp;
  auto A = [&](void** p, size_t bytes) { hipError_t e = hipMalloc(p, bytes ? bytes : 8); if (e == hipSuccess) tmp.push_back(*p); return e; };
  HIPCHK(A((void**)&dpo, (size_t)7 * K * sizeof(double))); HIPCHK(A((void**)&dpn, (size_t)7 * K * sizeof(double)));
  HIPCHK(A((void**)&dl, (size_t)3 * L * sizeof(double))); HIPCHK(A((void**)&dr, (size_t)L * sizeof(int)));
  HIPCHK(hipMemcpyAsync(dpo, pose_old, (size_t)7 * K * sizeof(double), hipMemcpyHostToDevice, c->st));
  HIPCHK(hipMemcpyAsync(dpn, pose_new, (size_t)7 * K * sizeof(double), hipMemcpyHostToDevice, c->st));
  if (L) {
    HIPCHK(hipMemcpyAsync(dl, lm_pos, (size_t)3 * L * sizeof(double), hipMemcpyHostToDevice, c->st));
    HIPCHK(hipMemcpyAsync(dr, ref_kf, (size_t)L * sizeof(int), hipMemcpyHostToDevice, c->st));
  }
  if (velocity) {
    HIPCHK(A((void**)&dv, (size_t)3 * K * sizeof(double)));
    HIPCHK(hipMemcpyAsync(dv, velocity, (size_t)3 * K * sizeof(double), hipMemcpyHostToDevice, c->st));
  }
  launch_reanchor(K, dpo, dpn, dv, L, dr, dl, c->st);
  if (L) HIPCHK(hipMemcpyAsync(lm_pos, dl, (size_t)3 * L * sizeof(double), hipMemcpyDeviceToHost, c->st));
  if (velocity) HIPCHK(hipMemcpyAsync(velocity, dv, (size_t)3 * K * sizeof(double), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  for (void* p : tmp) (void)hipFree(p);
  return COVGPU_OK;
}

// common.hpp — device-resident problem layout and kernel launch interface of libcovgpu (gfx950 only).
//
// Data layout in HBM:
//   * small gathered tables are array-of-structs (pose [K][7], speed-bias [K][9], landmark [L][3],
//     camera [A][..]): K, A are tiny (<= ~20k rows) and stay L2-resident, each gather is one 24-72 B row;
//   * the observation stream is struct-of-arrays (kf index i32, lm index i32, u, v, sigma f64), grouped by
//     landmark exactly as the IR delivers it, so a wave reads 64 consecutive records fully coalesced;
//   * the reduced camera system is stored STRUCTURED: the pose-pose part C is one dense
//     row-major lower-triangular FP64 matrix of padded order npad (multiple of 128), 6 rows per keyframe in
//     chain-major order perm[]; the speed-bias part is block-tridiagonal along each agent's IMU chain
//     (Ad, Ae: 9x9 per keyframe) and couples to at most three pose blocks (Bp, Bs, Bn: 9x6 per keyframe);
//   * all vectors of the trust-region step (gradient, diag(J^T J), Gauss-Newton step, step, scratch) are
//     length N = n + 3 L with the pose part first, so norms and combinations are single flat kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>

#include <map>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/covgpu.h"

namespace covgpu {

constexpr int kTile = 128;  // panel width / tile edge of the dense reduced-system factorisation

struct DevProblem {
  int K, A, L, O, I, E, S;
  int D;       // reduced dims per keyframe: 15 (VI) or 6
  int n;       // D * K
  int npad;    // n rounded up to a multiple of kTile (leading dimension of Sred)
  int N;       // n + 3 L
  int vi;      // 1: speed-bias blocks + IMU factors active
  int lm_group;  // lanes per landmark of the landmark-major kernels: 4 | 8 | 16 (k_visual.hip; 0 = 16)
  double reproj_loss_a, gravity;

  // state: current, candidate, initial (restart point of covgpu_solve_resident)
  double *pose, *sb, *lm;
  double *pose_c, *sb_c, *lm_c;
  double *pose0, *sb0, *lm0;
  uint8_t* fixed;
  int* kf_cam;
  double *cam_extr, *cam_intr, *cam_dist;
  int* cam_dist_type;

  // observations (landmark-major)
  int *lm_obs_ptr, *obs_kf, *obs_lm;
  double *obs_u, *obs_v, *obs_sigma;

  // keyframe-major view of the observations and covisible-pair lists (static per problem, built at upload)
  int *kf_obs_ptr, *kf_obs_idx;             // [K+1], [O]
  // keyframe-major copies (slot t of kf_obs_idx): what k_kf_reduce reads of an observation, coalesced instead of three scattered 8-byte
  // gathers from the landmark-major stream; obs_zpos = inverse of kf_obs_idx: the slot of observation o — where its Z record lives
  double* kobs;                             // [O][3] u, v, sigma
  int *kobs_lm, *obs_zpos;                  // [O] landmark of slot t | slot of observation o
  int npairs;
  int *pair_ptr, *pair_i, *pair_j;          // [npairs+1], [npairs] chain-major positions, i > j
  int *pair_oa, *pair_ob;                   // [sum] observation of keyframe i / keyframe j of each common landmark
  double *obsZ;                             // [O][18] per-observation Z = (Jp^T Jl) R with Hll^-1 = R R^T: the one record of the landmark elimination (k_visual.hip),
                                            // KEYFRAME-major (slot obs_zpos[o]): the records a covisible pair reads lie in its two keyframes' 60-KB blocks (L2)
  double *lmRT;                             // [L][9] per landmark: lower factor R (6) | t = R^T g_l (3)
  double *cost_part;                        // per-block cost partials

  // IMU factors
  int *imu_i, *imu_j, *imu_ptr;
  double *imu_samples, *imu_first;
  double *imu_noise;  // [I][5] sigma_a sigma_g sigma_aw sigma_gw gravity per factor (keyframe_be.cpp:187-195)
  double *pre_delta;  // [I][11] dp dq dv dt
  double *pre_J;      // [I][225]
  double *pre_P;      // [I][225]
  double *pre_W;      // [I][225] whitening = chol(P)^-1 (lower)
  double *pre_bias;   // [I][6] linearisation ba, bg

  // between factors
  int *edge_i, *edge_j;
  double *edge_meas, *edge_sqrt_info, *edge_loss_a;

  // keyframe ordering of the linear system: IMU chains (one per agent) laid out back to back
  int nchains;
  int max_chain_len;  // keyframes of the longest chain
  int* perm;        // [K]   keyframe -> position in chain-major order (identity when !vi)
  int* pos_kf;      // [K]   position -> keyframe
  int* chain_ptr;   // [nchains+1] positions of each chain
  int* pos_chain_end;  // [K] position -> end position (exclusive) of its chain

  // normal equations (after landmark elimination)
  double* Sred;    // [npad][npad] pose-pose matrix C, lower triangle used, rows 6*perm[k]+r
  double* bred;    // [n] right-hand side in IR layout (D per keyframe); solution is written to gn
  double *Ad, *Ae;          // [K][81] speed-bias diagonal / sub-diagonal (pos, pos-1) blocks, by position
  double *Bp, *Bs, *Bn;     // [K][54] speed-bias(pos) x pose(pos-1 | pos | pos+1) blocks (9x6)
  int* pos_chain_begin;  // [K] position -> first position of its chain
  double* bp;      // [2 npad] pose right-hand side / solution of the dense stage (+ scratch half)
  double* grad;    // [N]  J^T r           (pose part, then landmark part)
  double* hdiag;   // [N]  diag(J^T J)
  double* HllInv;  // [L][6] damped inverse landmark blocks (xx xy xz yy yz zz)
  double* gn;      // [N]  Gauss-Newton (or LM) step
  double* step;    // [N]  trust-region step
  double* vtmp;    // [N]  scratch vector
  double* Linv;    // [npad/kTile][kTile*kTile] inverses of the factor's diagonal blocks
  double* scal;    // [SC_COUNT] device scalars (reductions)
  // deterministic reductions: every wave writes its partial sum to part[slot][index], k_part_finish adds them in a
  // fixed order. Index ranges: [0, 8192) observation streams | part_imu.. IMU waves | part_edge.. edge waves | part_vec..
  double* part;
  int part_n, part_imu, part_edge, part_vec;
  // deterministic IMU / edge accumulation: role-indexed slots filled with plain stores, summed by gather kernels
  double *imuAd, *imuBs, *imuCd, *imuG;   // [2][K][81] | [2][K][54] | [3][K][36] | [2][K][30] (grad 15 | hdiag 15), role 0 = as predecessor, 1 = as successor; imuCd plane 2 = cross block (pos, pos-1)
  double* edgeOut;                         // [E][132]: Hii(36) Hjj(36) Hhi_lo(36) gi(6) gj(6) hdi(6) hdj(6)
  int *kf_edge_ptr, *kf_edge_ent;          // [K+1], [2E] incident (edge*2 + role) per keyframe, ascending
  int nepairs;
  int *epair_ptr, *epair_i, *epair_j, *epair_ent;  // unique pose pairs (chain-major positions i > j) -> edges (edge*2 + transposed)
  int* flag;       // [4]  device flags (Cholesky failure)

  // ---- agent-sharded solve of one map over several GPUs. shard == 0: everything is owned here.
  int shard;       // 1: this context holds one rank's share (its subtrees of the elimination tree + the replicated top; its landmarks, IMU factors, edges)
  double* vw;      // [N] weight of every unknown in the trust-region norms: 1 if this rank counts it (unknowns of its own subtrees,
                   //     its landmarks, the top unknowns on rank 0 only), else 0; nullptr = all 1

  // ---- multifrontal layout of the WHOLE reduced camera system (k_front.hip, nd_plan.hpp). nd == 0: the forms above.
  // Variables: 2 pos = pose block (6), 2 pos + 1 = speed-bias block (9) of chain position pos. Every variable is owned by one
  // node of the nested-dissection tree; a node's front is a dense lower-triangular matrix over [own variables, padded to the
  // level's interior order nI | the ancestor variables its subtree couples to]. The linearisation kernels write straight
  // into the fronts (nd_entry below): the entry between two variables lives in the front of the DEEPER owner.
  int nd, nd_nnodes, nd_nlev, nd_maxd;
  int *nd_vnode, *nd_voff, *nd_vord;   // [2K] owner node (-1: another rank's) | scalar offset inside the owner's own columns | ordinal inside the owner
  int *nd_vown;                        // [2K] 0 another rank's unknown | 1 this rank's | 2 top unknown (replicated; damped after the all-reduce)
  int *nd_ndepth, *nd_nI;              // [nodes] depth (root 0) | padded interior order of the node's level
  long long* nd_ntab;                  // [nodes][2] element offset of the front in nd_M | leading dimension (level order: a level's rows are its batch table)
  int* nd_abase;                       // [nodes][maxd] base into nd_fidx of the ancestor at that depth
  int* nd_fidx;                        // front row of an ancestor's variable (by ordinal), -1: not in this front
  double *nd_M, *nd_rhs, *nd_Linv;     // all fronts | per level [batch][2 ntot] | per level [batch][nI/128][128][128]
  double* nd_dummy;                    // sink for structurally impossible writes (never read)

  // ---- trust-region state on the device (k_dense.hip: k_tr_*): radius, damping, costs, dogleg coefficients, verdicts.
  // The step logic of Ceres' TrustRegionMinimizer / DoglegStrategy / LevenbergMarquardtStrategy runs in
  // one-thread kernels between the vector kernels, so an iteration needs ONE host read-back (at its end) instead of three.
  double* tr;                          // [TR_COUNT]
  double* scal_r; int* flag_r;         // the scalars / factorisation flag the step logic reads: P.scal / P.flag, or (sharded solve) their all-reduced copies
};

enum {
  TR_RADIUS = 0, TR_MU, TR_LMDF, TR_COST, TR_ALPHA, TR_GG, TR_GN2, TR_GDOT, TR_DNORM, TR_CG, TR_CN, TR_MODEL, TR_SN, TR_RHO, TR_COSTNEW,
  TR_OK, TR_VALID, TR_ACC, TR_TERM, TR_FNCONV, TR_RETRY, TR_FIRST, TR_INITCOST, TR_REUSE,
  TR_JA, TR_JB, TR_JC, TR_VV, TR_VG, TR_NN, TR_XN2,   // dot products of the fused tail (k_tail.hip): a rejected step re-derives its coefficients from them
  TR_COUNT = 32
};
static_assert(TR_XN2 < TR_COUNT, "trust-region state slots");
struct TrConsts {  // options the device-side step logic needs
  int strategy;
  double max_radius, min_relative_decrease, function_tolerance, parameter_tolerance, gradient_tolerance;
};

// address of the entry between scalar ra of variable va and scalar rb of variable vb in the multifrontal layout
__device__ __forceinline__ double* nd_entry(const DevProblem& P, int va, int vb, int ra, int rb) {
  const int na = P.nd_vnode[va], nb = P.nd_vnode[vb];
  if (na < 0 || nb < 0) return P.nd_dummy;  // an unknown of another rank's subtree (its residuals never meet this rank's)
  if (na == nb) {
    const int oa = P.nd_voff[va] + ra, ob = P.nd_voff[vb] + rb;
    const int hi = oa > ob ? oa : ob, lo = oa > ob ? ob : oa;
    return P.nd_M + P.nd_ntab[2 * na] + (size_t)hi * (size_t)P.nd_ntab[2 * na + 1] + lo;
  }
  const int da = P.nd_ndepth[na], db = P.nd_ndepth[nb];
  // the deeper owner's front holds it: row = the ancestor variable's border row, column = the own variable's column
  const int no = da > db ? na : nb, vo = da > db ? va : vb, ro = da > db ? ra : rb;
  const int vA = da > db ? vb : va, rA = da > db ? rb : ra, dA = da > db ? db : da;
  const int base = P.nd_abase[(size_t)no * P.nd_maxd + dA];
  const int row = base < 0 ? -1 : P.nd_fidx[base + P.nd_vord[vA]];
  if (row < 0) return P.nd_dummy;
  return P.nd_M + P.nd_ntab[2 * no] + (size_t)(row + rA) * (size_t)P.nd_ntab[2 * no + 1] + (P.nd_voff[vo] + ro);
}

// address of entry (r, c) of the 6x6 pose-pose block (pi, pj), chain positions pi >= pj; for pi == pj only c <= r is stored:
// in the fronts (GBA) or in the dense row-major lower-triangular matrix Sred (pose graph, covgpu_schur)
__device__ __forceinline__ double* c_entry(const DevProblem& P, int pi, int pj, int r, int c) {
  if (P.nd) return nd_entry(P, 2 * pi, 2 * pj, r, c);
  return P.Sred + (size_t)(6 * pi + r) * P.npad + (6 * pj + c);
}

// ---- scalar slots in DevProblem::scal
enum { SC_COST = 0, SC_JV2 = 1, SC_GG = 2, SC_GN2 = 3, SC_GDOT = 4, SC_GMAX = 5, SC_GS = 6, SC_SN2 = 7, SC_XN2 = 8,
       SC_VV = 9, SC_VG = 10, SC_NN = 11, SC_JA = 12, SC_JB = 13, SC_JC = 14,   // fused tail (k_tail.hip): |c|^2, c.n, |n|^2, |J c|^2, |J n|^2, (J c).(J n)
       SC_COUNT = 16 };

struct CholAux;
// a record published by the FIRST thread of the next kernel on the recording stream instead of by a launch of its own (CholAux::publish_handle): at
// that moment everything enqueued before that kernel on its stream is complete, which is all a record says
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) acts on the CURRENT device's copy of the function: once per device, not once per process (a process
// that drives several GPUs — covgpu_gba_solve_multi, one host thread per device — would launch the large-LDS kernels un-prepared on all but the first)
inline bool first_use_on_device(std::atomic<unsigned long long>& seen) {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d > 63) return true;   // (unknown: set the attribute again, it is idempotent)
  const unsigned long long bit = 1ull << d;
  return (seen.fetch_or(bit, std::memory_order_acq_rel) & bit) == 0;
}
struct DevSignal { long long* flag = nullptr; long long seq = 0; };
// ---- launchers (each enqueues on `st`, no synchronisation)
void launch_kobs_build(const DevProblem& P, int* pair_oa, int* pair_ob, size_t nent, hipStream_t st);  // upload: keyframe-major copies, Z slots, pair lists -> Z slots
void launch_lm_lin(const DevProblem& P, double mu, hipStream_t st, DevSignal sig = DevSignal());   // landmark-major linearisation: records, H_ll, g_l, cost partials
void launch_lm_build(const DevProblem& P, double mu, hipStream_t st, hipEvent_t pose_system_cleared = nullptr, hipStream_t side = nullptr,
                     hipEvent_t ev_lin = nullptr, hipEvent_t ev_kf = nullptr, CholAux* ax = nullptr);  // reprojection -> Hll, g, S (Schur), bred, cost
void launch_lm_backsub(const DevProblem& P, const double* dp, double* out_all, hipStream_t st);
void launch_obs_jvp(const DevProblem& P, const double* v_all, hipStream_t st);  // scal[SC_JV2] += sum |J v|^2
void launch_obs_cost(const DevProblem& P, const double* pose, const double* lm, hipStream_t st);  // scal[SC_COST] +=
void launch_obs_linearize(const DevProblem& P, double* r, double* Jp, double* Jl, double* cost, hipStream_t st);
void launch_obs_norms(const DevProblem& P, double* norms, hipStream_t st);
void launch_lm_outliers(const DevProblem& P, double th, unsigned char* erase, int* left, unsigned long long* counts, hipStream_t st);

void launch_preintegrate(const DevProblem& P, hipStream_t st);
void launch_imu_build(const DevProblem& P, hipStream_t st);
void launch_imu_jvp(const DevProblem& P, const double* v_all, hipStream_t st);
void launch_imu_cost(const DevProblem& P, const double* pose, const double* sb, hipStream_t st);
void launch_imu_linearize(const DevProblem& P, double* r, double* J, hipStream_t st);

void launch_edge_build(const DevProblem& P, hipStream_t st);
void launch_edge_jvp(const DevProblem& P, const double* v_all, hipStream_t st);
void launch_edge_cost(const DevProblem& P, const double* pose, hipStream_t st);
void launch_edge_linearize(const DevProblem& P, double* r, double* J, double* cost, hipStream_t st);

void launch_finalize_diag(const DevProblem& P, double mu, int which, hipStream_t st);  // which: 0 pose, 1 speed-bias, 2 both
// deterministic scalar reductions
void launch_part_clear(const DevProblem& P, int slot0, int nslots, hipStream_t st);
void launch_part_finish(const DevProblem& P, int slot0, int nslots, hipStream_t st);
void launch_imu_gather(const DevProblem& P, int which, hipStream_t st);  // which: 0 pose dims, 1 speed-bias dims, 2 both
void launch_edge_gather(const DevProblem& P, hipStream_t st);
// structured solve of the damped reduced system: speed-bias chains -> dense pose system -> back-substitution.
// Solution (IR layout, D per keyframe) is written to dst[0..n).
struct PgoPlan;
// pose graph: the system assembled in P.Sred / P.bred -> dst (IR layout); block-arrow elimination (k_pgo.hip) or plain dense Cholesky
void launch_pose_graph_solve(const DevProblem& P, double* dst, hipStream_t st, CholAux& ax, PgoPlan* pgo);
void launch_zero_system(const DevProblem& P, hipStream_t st);       // every small per-iteration buffer, one launch
// per-context resources of the dense factorisation: auxiliary stream for the look-ahead, ordering events, and
// (profiling only) one timed event pair around every bulk trailing-update launch
struct CholAux {
  hipStream_t aux = nullptr, mid = nullptr, head = nullptr;
  hipEvent_t ev_fill = nullptr;  // pose system cleared (head stream, beside the linearisation)
  hipEvent_t ev_xb = nullptr;    // multifrontal look-ahead: second half of a level's extend-add done (bulk stream)
  int* bwd_cnt = nullptr;        // ticket counters of k_bwd_front (65536, zero between launches)
  double* bwd_scr = nullptr; size_t bwd_scr_elems = 0;   // its scratch (grown on demand by launch_nd_solve)
  double* bwd_pipe = nullptr; size_t bwd_pipe_elems = 0; // k_bwd_pipe's scratch, filled with its "empty" word (grown on demand by launch_nd_solve)
  bool pipe_broken = false;       // a pipeline workgroup waited too long (gate_failed): launch per tile for the life of the context
  hipEvent_t ev_xa = nullptr;    // ... the chain's stream has enqueued the level below completely
  hipEvent_t ev_zero = nullptr;  // per-iteration buffers cleared (main stream): the side stream's inertial kernels follow
  hipEvent_t ev_lin = nullptr, ev_kf = nullptr;  // landmark linearisation done (main stream) | per-keyframe reduction done (side stream): launch_lm_build
  std::vector<hipEvent_t> ev, prof_ev, panel_ev;  // panel_ev: start of every big panel on the main stream (COVGPU_TRACE_PANELS=1)
  std::vector<double> prof_flops;
  // (profiling only) the same for every k_potrf_panel launch — the serial chain, the kernel with the largest share of GPU time
  std::vector<hipEvent_t> prof_ev2;
  std::vector<double> prof_flops2;
  double potrf_ms = 0, potrf_flops = 0;
  long n_potrf = 0;
  // per big panel of the batched (arrow) factorisation: device list of the LIVE (batch, ti, tj) tiles of its bulk update,
  // interleaved so that list position p runs on XCD p % 8 and every XCD gets the same number of tiles (k_chol.hip)
  struct TriCache {
    std::vector<int*> list; std::vector<int> count; int key = -1;
    std::vector<int*> listC; std::vector<int> countC;   // the bulk update's tiles in the next-but-one panel's two tile columns (launched first, k_chol.hip: eA)
    int *listA = nullptr, *listB = nullptr; int countA = 0, countB = 0;   // last panel's update split at DenseBatch::split_ta
    void clear();
  };
  TriCache tri0;                  // block-arrow batches of the pose-graph solve (k_pgo.hip)
  std::vector<TriCache> tri_lev;  // one per level of the multifrontal solve (k_front.hip), selected by DenseBatch::tri_slot
  void tri_clear();
  // multi-GPU: sum `n` device doubles over all ranks, in place (solver.hip installs it when a shard is set; nullptr = single GPU)
  // (stream-ordered: the call enqueues the collective on `st`; no host synchronisation is implied)
  void (*reduce)(void* ctx, double* dev, size_t n, int op, hipStream_t st) = nullptr;
  void* reduce_ctx = nullptr;
  int panel_n = 0;
  std::vector<int> panel_tag;
  // dev aid (COVGPU_TRACE_PANELS=1): timestamp on stream s, printed by collect() relative to the first mark of the solve. Tags:
  // >= 0 start of big panel `tag` of a factorisation | -1 solve begins | -2 fronts assembled | -3 extend-add done | -4 level factored |
  // -5 level back-substituted | -6 solve ends
  void mark(hipStream_t s, int tag);
  bool profile = false;
  double syrk_ms = 0, syrk_flops = 0;
  long n_syrk = 0;
  void init();
  void destroy();
  void collect();
  // ---- ordering between this context's streams through DEVICE FLAGS instead of HIP events (round 6; tools/gate_probe.hip:
  // a cross-stream dependency through an event costs ~14 us and holds the recording stream's next kernel, through a flag ~5 us and 1.3):
  //   record(e, s)  a one-thread kernel on s publishes a fresh sequence number in e's slot (the producer's data was released by its own end of kernel)
  //   wait(s, e..)  a one-wave kernel on s polls the slots for the numbers their LAST record carried (HIP's semantics: the wait captures the record
  //                 that precedes it in host order); the consumer kernel behind it starts at an ordinary dependent boundary and acquires at its start
  // A gate is always enqueued after its signal: the earliest unfinished packet in host order never waits for a later one — no deadlock whatever
  // the streams' mapping onto hardware queues. A gate that nevertheless waits longer than gate_timeout_s (a tool that serialises kernels out of
  // order, a lost dispatch) raises gate_dead: every later gate returns at once, the solve reports an error and solver.hip repeats it with
  // events. gates_on == false (COVGPU_GATES=0, or after such a failure): plain hipEventRecord / hipStreamWaitEvent.
  static constexpr int kGateSlots = 16384;
  bool gates_on = false, gates_broken = false;
  long long* gate_flags = nullptr;       // [kGateSlots] device
  int* gate_dead = nullptr;              // device: a gate gave up
  int* gate_dead_h = nullptr;            // pinned host mirror of it (written by the gate itself; read after the host synchronisation)
  // A slot belongs to ONE (event, recording stream) pair: the signal is a plain store, and the numbers of a slot must arrive in order — the panel
  // events are shared between the levels of a solve (k_chol.hip indexes one array by role and panel), so one event is recorded on the chain's
  // stream at one level and on a side stream at the next; two streams storing into one slot can overtake each other and leave the OLDER number
  // (round 6: the pose-graph solve hung on exactly that). gate_slot: event -> slot of its LAST record (what a wait polls).
  std::unordered_map<hipEvent_t, int> gate_slot;
  std::map<std::pair<hipEvent_t, hipStream_t>, int> gate_slot_es;   // (event, stream) -> slot
  std::vector<long long> gate_seq;       // per slot: sequence number of the last record (0: never recorded)
  long long gate_counter = 0;
  double gate_timeout_s = 3.0;
  long gate_signals = 0, gate_waits = 0; // launches issued (statistics)
  void record(hipEvent_t e, hipStream_t s, int tag = 0);
  void wait(hipStream_t s, hipEvent_t e0, hipEvent_t e1 = nullptr, hipEvent_t e2 = nullptr, hipEvent_t e3 = nullptr);
  // records and waits that stand side by side on one stream as ONE launch (every launch on the panel chain's stream is ~3.5 us under load): the
  // kernel publishes r0 / r1 first, then polls w0 .. w3 — the same order as record(r0); record(r1); wait(w0 ..)
  DevSignal publish_handle(hipEvent_t e, hipStream_t s, int tag = 0);
  void record_handle(DevSignal d, hipStream_t s);   // publishes a handle by a launch of its own after all (the kernel it was meant for is not launched)   // gates on: the handle for the next kernel of the recording stream (flag == nullptr: record(e, s) instead)
  void sync(hipStream_t s, hipEvent_t r0, int tag0, hipEvent_t r1, int tag1, hipEvent_t w0 = nullptr, hipEvent_t w1 = nullptr, hipEvent_t w2 = nullptr,
            hipEvent_t w3 = nullptr);
  int gate_slot_of(hipEvent_t e, hipStream_t s);   // the slot of (e, s), created on demand; it becomes e's current slot
  // dev aid (COVGPU_GATE_LOG=1): every signal / gate stamps wall_clock64 (100 MHz) into a device log — an un-profiled timeline of the streams'
  // hand-overs, printed by collect(): "S<tag>@t" a signal of record(.., tag), "G<tag of the first awaited record>@t_start+wait" a gate
  long long* gate_log = nullptr; int gate_log_n = 0;
  std::vector<int> gate_log_tag; std::vector<char> gate_log_kind; std::vector<int> gate_tag_of_slot;
  static constexpr int kGateLogMax = 16384;
  bool gate_failed() const { return gate_dead_h != nullptr && *(volatile int*)gate_dead_h != 0; }
  void gates_disable();                  // after a failure: back to events for the life of the context (the flags are reset)
};
// dense SPD solve of Sred x = bred in place (lower Cholesky on the FP64 MFMA path); flag[0] != 0 on failure
// batched form: n independent systems of identical shape; sM / sL / sR = elements between consecutive matrices, Linv
// sets and right-hand sides. The same launches serve all of them (one more grid dimension).
// multifrontal fronts: the backward substitution reads the given (ancestor) unknowns straight from the solution vector and
// writes the front's own unknowns there — no separate gather / scatter launches (k_panel.hip: k_bwd_given, k_bwd_step_sub)
struct BwdXfer {
  const int* gidx = nullptr;                   // solution index of every own / border scalar of every front (nullptr: disabled)
  const int *own_g = nullptr, *st_g = nullptr, *own_dims = nullptr, *st_dims = nullptr;   // per front (level order)
  double* x = nullptr;                         // solution vector
  int first = 0;                               // first front of the batch
};
struct DenseBatch {
  int n = 0; size_t sM = 0, sL = 0, sR = 0;
  const int* live = nullptr; int tI = 0; const int* live_h = nullptr;  // live / tI: see GemmArgs (k_chol.hip)
  const long long* tab = nullptr;  // per-matrix (element offset, leading dimension): fronts of unequal order in one batch (GemmArgs::btab)
  int tri_slot = -1;               // which CholAux::tri_lev entry caches the live-tile lists of this batch's bulk updates
  BwdXfer xfer;                    // dense_backward_solve only
  // Look-ahead ACROSS levels of the multifrontal solve (k_front.hip): split_ta > 0 splits the trailing update of the batch's last
  // panel — border tile rows < split_ta (what the parents' FIRST panel receives) on the chain's stream, the rest on the bulk
  // stream, which the caller continues with the second half of the extend-add while the parents' first panel is factored.
  // pre_trsm: event the first panel's substitutions wait for (the caller's second-half extend-add of THIS batch's fronts).
  int split_ta = 0;
  hipEvent_t pre_trsm = nullptr;
  int* bwd_cnt = nullptr;          // [n] zeroed ticket counters: whole-front backward substitution in one launch (k_panel.hip: k_bwd_front)
  double* bwd_scr = nullptr;       // its scratch: [n][interior tiles <= 4][row chunks][128]
  double* bwd_pipe = nullptr;      // k_bwd_pipe's sentinel-filled scratch (fronts of bwd_pipe_min_tiles() interior tiles and more); nullptr: launch per tile
  int *pipe_dead = nullptr, *pipe_dead_h = nullptr; double pipe_timeout_s = 3.0;   // CholAux::gate_dead / gate_dead_h / gate_timeout_s
  const int* own_dims_h = nullptr; // host copy of own_dims (flop accounting of the profiled run)
  const int* own_dims = nullptr;   // device, [n]: real interior order of every matrix of the batch — substitutions and rank updates stop at a
                                   // front's OWN last real column (own_max is the batch's: levels mix fronts of 9 .. 250 unknowns)
  // border x border 128-tiles of a front that NO child contributes to are not cleared per iteration; the first panel's trailing update
  // then starts them from zero instead of reading them (GemmArgs::beta0). beta0_off[front of the batch] -> first entry of its lower-triangular
  // tile map in beta0 (1: some child adds into the tile — it is cleared and read as before)
  const int* beta0_off = nullptr; const int* beta0 = nullptr;
  const int* plist = nullptr; const int* pbig_h = nullptr; const int* psmall_h = nullptr;   // NdLevel::plist / pbig / psmall
  int own_max = 0;                 // > 0: largest real interior order over the batch — columns beyond it are identity padding in EVERY
                                   // matrix (L = I, block inverses = I, y = 0 already in place), so the panel kernel factors only the
                                   // 16-column blocks that hold a real column and skips all-padding panels
};
void dense_cholesky_solve_raw(double* S, double* b, double* Linv, int* flag, int npad, hipStream_t st, CholAux& ax, int tstop = -1,
                              bool solve = true, DenseBatch bt = DenseBatch());  // tstop >= 0 (even): eliminate tile columns [0, tstop) only
int bwd_pipe_min_tiles();    // fronts of at least this many interior tiles: backward substitution as one pipelined launch (k_chol.hip)
int bwd_front_max_tiles();   // fronts of at most this many interior tiles: backward substitution in one launch (k_chol.hip)
void dense_backward_solve(double* S, double* b, double* Linv, int npad, hipStream_t st, int tfact, int tend, DenseBatch bt = DenseBatch());
// 256-column panel chain (k_panel.hip): with it the Linv buffer holds, per tile, the eight 16x16 diagonal-block inverses
// instead of the 128x128 inverse. COVGPU_PANEL=0 selects the round-2a chain (two 128-column potrf + inverse per panel).
bool launch_potrf_panel(double* S, size_t ld, int t0, int w, double* Linv, int* flag, double* b, int npad, int nbt, size_t sM, size_t sL, size_t sR,
                        hipStream_t st, const long long* btab = nullptr, int nb = -1, const int* own = nullptr, const int* list = nullptr, int n_big = 0,
                        int n_small = 0, DevSignal sa = DevSignal(), DevSignal sb = DevSignal());   // returns false if nothing was launched   // nb: 16-column blocks to factor (-1: the whole panel); own / list / n_big / n_small: k_panel.hip (round 6)
void launch_bwd_given(const double* S, size_t ld, int r0, int r1, double* y, double* x, int ncol, int nbt, size_t sM, size_t sR, hipStream_t st,
                      const long long* btab, const int* live, int tI, BwdXfer xf = BwdXfer());
void launch_bwd_front(const double* S, int tI, int ntiles, int nchunk, double* y, const double* Linv, int nbt, size_t sL, size_t sR, hipStream_t st,
                      const long long* btab, const int* live, BwdXfer xf, int* cnt, double* scr);
// fronts of many interior tiles: the whole backward substitution as one launch of cooperating workgroups (k_panel.hip: k_bwd_pipe). pipe: sentinel-filled
// scratch of at least nbt * ntiles * (nchunk + 1) * 128 doubles (launch_pipe_fill once; the kernel leaves it as it found it)
void launch_bwd_pipe(const double* S, int tI, int ntiles, int nchunk, double* y, const double* Linv, int nbt, size_t sL, size_t sR, hipStream_t st,
                     const long long* btab, const int* live, BwdXfer xf, double* pipe, int* dead, int* dead_h, double timeout_s);
void launch_pipe_fill(double* buf, size_t n, hipStream_t st);
// several (bottom) levels of the tree in ONE launch (k_panel.hip: k_bwd_tree); lev[0] is the HIGHEST of them. pipe: as above, at least the sum over the
// levels of nbt * T * (nchunk + 1) * 128 doubles
constexpr int kBwdTreeMax = 8;
constexpr int kPipeChunk = 128;              // border rows per helper workgroup of the pipelined backward substitution
struct BwdTreeLevel {
  int nbt, T, nchunk, tI, first;             // fronts | interior tiles (max) | 256-row chunks of the border (max) | padded interior tiles | first node
  double* y; size_t bsR; const double* Dinv; size_t bsL; const long long* btab; const int* live;
  size_t scr_off, xpub_off;                  // (filled in by the launcher)
};
void launch_bwd_tree(const double* M, const BwdTreeLevel* lev, int nlev, BwdXfer xf, double* pipe, int* dead, int* dead_h, double timeout_s, hipStream_t st,
                     bool form64 = false);   // form64: the tile workgroups of k_bwd_pipe64 (top levels: at most 128 of them in the launch)
unsigned long long pipe_empty_word();        // the "empty" word of the hand-over slots (k_nd_assemble fills the solution vector's merged entries with it)
void launch_trsm_sub(double* S, size_t ld, int t0, int w, int r0, int r1, const double* Linv, double* b, int npad, int nbt, size_t sM, size_t sL,
                     size_t sR, const int* live, int tI, hipStream_t st, bool chain = false, const long long* btab = nullptr, int nb = -1, const int* own = nullptr);
void launch_bwd_step_sub(const double* S, size_t ld, int p, const double* Linv_p, double* y, double* x, int ncol, int nblocks, int nbt, size_t sM,
                         size_t sL, size_t sR, hipStream_t st, const long long* btab = nullptr, const int* live = nullptr, int tI = 0,
                         BwdXfer xf = BwdXfer());  // xf.gidx != nullptr (pass it with p == 0 only): the front's own unknowns go to the solution vector

// ---- block-arrow pose-graph solve (k_pgo.hip)
struct PgoHostPlan { std::vector<std::vector<int>> block_kf; std::vector<int> border_kf; };
bool pgo_plan_analyse(int K, int E, const int* ei, const int* ej, PgoHostPlan& out);
struct PgoPlan {
  bool active = false;
  int nblk = 0, nIpad = 0, ntot = 0;  // blocks; padded interior order (multiple of 256, the largest block's); nIpad + nb
  int nb = 0;                          // padded border order (multiple of 128)
  int* idx = nullptr;                  // [nblk][ntot] row of the global system, -1 = padding
  double *M = nullptr, *rhs = nullptr, *Linv = nullptr;  // [nblk][ntot][ntot] | [nblk][2 ntot] | [nblk][nIpad/128][128][128]
  int* idx_b = nullptr;
  double *Sb = nullptr, *rhs_b = nullptr, *Linv_b = nullptr;
};
void launch_pgo_block_solve(const DevProblem& P, PgoPlan& plan, hipStream_t st, CholAux& ax);

// ---- covisible keyframe pairs on the device (k_pairs.hip)
struct PairLists {
  int npairs = 0; size_t nent = 0;
  int *pair_ptr = nullptr, *pair_i = nullptr, *pair_j = nullptr;   // [npairs + 1] | [npairs] row keyframe (i > j) | column keyframe, sorted by (i, j)
  int *pair_oa = nullptr, *pair_ob = nullptr;                      // [nent] observation of keyframe i / of keyframe j of every common landmark, landmark order
};
bool build_pairs_device(int L, int K, const int* d_lm_obs_ptr, const int* d_obs_kf, const int* d_key_of_kf, bool want_obs, hipStream_t st, PairLists& out);
// upload: observation stream unpacked and the keyframe-major lists built on the device (k_pairs.hip)
void launch_obs_unpack(int L, int O, const int* lm_obs_ptr, const double* uv, int* obs_lm, double* u, double* v, int* iota, hipStream_t st);
bool build_kf_lists_device(int O, int K, const int* d_obs_kf, const int* d_iota, int* kf_obs_ptr, int* kf_obs_idx, hipStream_t st);
// second round of a GlobalBundleAdjustment call (optimization_be.cpp:296-557) from the resident first round: the compacted observation
// stream (k_pairs.hip: round2_compact_device). lm_old[q] = first-round index of second-round landmark q.
struct Round2Lists {
  int L2 = 0, O2 = 0;
  double *lm0 = nullptr, *obs_u = nullptr, *obs_v = nullptr, *obs_sigma = nullptr;
  int *lm_obs_ptr = nullptr, *lm_old = nullptr, *obs_kf = nullptr, *obs_lm = nullptr, *kf_obs_ptr = nullptr, *kf_obs_idx = nullptr;
  void free_all() {
    for (void* q : {(void*)lm0, (void*)obs_u, (void*)obs_v, (void*)obs_sigma, (void*)lm_obs_ptr, (void*)lm_old, (void*)obs_kf, (void*)obs_lm, (void*)kf_obs_ptr, (void*)kf_obs_idx})
      if (q) (void)hipFree(q);
  }
};
bool round2_compact_device(int L, int O, int K, const unsigned char* d_erase, const int* d_left, const int* d_obs_kf, const int* d_obs_lm, const double* d_u,
                           const double* d_v, const double* d_sigma, const double* d_lm0, hipStream_t st, Round2Lists& out);

// ---- multifrontal solve of the whole reduced camera system (k_front.hip)
struct NdHostPlan;
constexpr int kExtRec = 144;   // ints per extend-add record: 8 header | 8 child | 64 + 64 row maps (576 bytes)
struct NdLevel {
  int n = 0, nI = 0, ntot = 0;          // fronts in the batch (0: nothing of this level on this rank) | padded interior order | largest front order
  int first = 0;                          // first node (level order) = row of nd_ntab where this level's batch table starts
  size_t rhs_off = 0, linv_off = 0;       // element offsets of the level's right-hand sides / block inverses
  int* live = nullptr;                    // [n][2] device: real interior tiles | real border tiles (GemmArgs::live)
  std::vector<int> live_h;
  int own_max = 0;                        // largest real interior order of the batch (DenseBatch::own_max)
  // k_potrf_panel's front lists (round 6): the level's fronts (index in the batch) by DEcreasing interior order; per 256-column panel P the first
  // pbig[P] of them have more than 128 real columns in that panel (sixteen-wave form), the next psmall[P] have 1 .. 128 (four-wave form), the rest none
  int* plist = nullptr; std::vector<int> plist_h, pbig, psmall;
  int ext_first = 0, ext_count = 0;       // extend-add work list (NdDev::ext) for the SUBTREE children of this level's fronts
  int ext_countA = 0;                     // ... of which the first ext_countA tiles lie in the fronts' first 256 rows (first panel)
  int split_ta = 0;                       // border tiles (128) of this level's fronts that reach their parents' first 256 columns (max)
  int ext2_first = 0, ext2_count = 0, ext2_countA = 0;   // ... for their TOP children (top levels of a sharded solve only)
};
struct NdDev {
  bool active = false;
  int nnodes = 0;                          // local fronts
  int top_lev0 = 0;                        // first top level (== lev.size(): none — single GPU)
  std::vector<NdLevel> lev;
  // device tables of the assembly kernels (owned by the context's allocation list)
  int *own_dims = nullptr, *st_dims = nullptr, *own_g = nullptr, *st_g = nullptr;  // [nodes] sizes | offsets into gidx
  int *gidx = nullptr;                     // solution index D kf + component of every own / border scalar of every node
  // bottom levels whose backward substitution runs as ONE launch (k_bwd_tree): levels [0, tree_levels) (0: none) and the solution indices of their
  // fronts' own unknowns (filled with the hand-over's "empty" word at the start of a solve)
  int tree_levels = 0; int* tree_fill = nullptr; std::vector<int> h_tree_fill;
  int tree_top0 = 0;                       // ... and the TOP levels [tree_top0, levels) as one launch of the 64x64-inverse form (== levels: none)
  int *cptr = nullptr, *cidx = nullptr;    // [nodes + 1], children that are subtree nodes (all children on a single GPU)
  int *cptr2 = nullptr, *cidx2 = nullptr;  // [nodes + 1], children that are top nodes
  int *inv_off = nullptr, *inv = nullptr;  // [nodes] offset of the node's map parent front row -> own front row (-1: none)
  int *rhs_node = nullptr;                 // [nodes] element offset of the node's right-hand side in nd_rhs
  int *ext = nullptr;                      // extend-add work lists: (node, tile row, tile column) of 64x64 tiles
  // the same lists FLATTENED for the kernel (round 5): per tile 8 ints {front offset lo, hi | leading dimension | right-hand side offset | tile row |
  // tile column | first, last+1 child entry}, per child entry 6 ints {front offset lo, hi | leading dimension | right-hand side offset | offset of its
  // row map | 0} — a workgroup reaches a child's entries through three dependent loads (tile, child, row map) instead of six (tile, cptr, cidx,
  // inv_off / ntab, row map)
  int *extw = nullptr, *extc = nullptr, *extc2 = nullptr;
  std::vector<int> h_extw, h_extc, h_extc2, h_ext_kind;
  // ... and as records of kExtRec ints (k_nd_extend_rec): [tiles] first records, then the overflow records of tiles with several contributing children
  int* extr = nullptr; size_t ext_over = 0; std::vector<int> h_extr;
  int *bb_off = nullptr, *bb = nullptr;    // per node: offset of its lower-triangular map over BORDER 128-tiles in bb (1: a child contributes to the tile; see DenseBatch::beta0)
  int *top_var = nullptr, *top_r = nullptr, *top_g = nullptr;  // per scalar unknown of the top nodes: variable | component | solution index
  int ntop = 0;
  // sharded solve: what the ranks exchange of the top fronts = their LIVE LOWER-TRIANGULAR 128x128 tiles, packed (the fronts are stored
  // as full squares: the upper triangle and the dead tiles are never read). top_tiles: (node, tile row, tile column) triples;
  // top_pack: [n_top_tiles][128][128 tiles | top right-hand sides | grad, hdiag of the top unknowns] — one all-reduce
  int* top_tiles = nullptr; int n_top_tiles = 0; double* top_pack = nullptr;
  std::vector<int> h_top_tiles;
  // sharded solve: [top fronts | top right-hand sides | grad, hdiag of the top unknowns] is ONE contiguous range of the buffer
  size_t M_sub = 0, rhs_top = 0, gh_off = 0;   // elements of the subtree fronts | of the top levels' right-hand sides | offset of [grad | hdiag] in nd_rhs
  // host staging of the tables (filled by nd_tables, uploaded by solver.hip)
  std::vector<int> h_vnode, h_voff, h_vord, h_vown, h_ndepth, h_nI, h_abase, h_fidx, h_own_dims, h_st_dims, h_own_g, h_st_g, h_gidx, h_cptr, h_cidx, h_cptr2,
      h_cidx2, h_inv_off, h_inv, h_rhs_node, h_ext, h_top_var, h_top_r, h_top_g, h_bb_off, h_bb;
  std::vector<long long> h_ntab;
  size_t M_elems = 0, rhs_elems = 0, linv_elems = 0;
  double plan_flops = 0;                   // flops of one factorisation of the plan's fronts (NdHostPlan::flops: dense count on the real sizes)
};
void nd_tables(const NdHostPlan& hp, const int* pos_kf, int D, int rank, NdDev& dev);  // host tables + level shapes from the plan
void launch_nd_init(const DevProblem& P, const NdDev& nd, hipStream_t st);    // once per upload: identity block inverses for the padding columns
void launch_nd_zero(const DevProblem& P, const NdDev& nd, hipStream_t st);    // per iteration: clear the live tiles, identity on interior padding
bool launch_nd_solve(const DevProblem& P, NdDev& nd, double* dst, double mu, hipStream_t st, CholAux& ax);  // false: scratch allocation failed, nothing enqueued. damped system in the fronts + bred -> dst (IR layout)

// fused trust-region tail (k_tail.hip): T1 stats, T2 J*v of (c, n) over all residual families, T3 / T6 finish (+ step logic in the
// last workgroup), T4 candidate, T5 candidate cost
void launch_tail_stats(const DevProblem& P, hipStream_t st);
void launch_tail_jvp(const DevProblem& P, bool two, hipStream_t st);
void launch_tail_cost(const DevProblem& P, hipStream_t st);
void launch_tail_finish(const DevProblem& P, TrConsts tc, int stage, bool two, bool with_logic, int fresh, hipStream_t st);
void launch_tail_logic(const DevProblem& P, TrConsts tc, int stage, int fresh, hipStream_t st);
void launch_tail_apply(const DevProblem& P, hipStream_t st);

void launch_dogleg_stats(const DevProblem& P, hipStream_t st);  // GG, GN2, GDOT, GMAX from grad/hdiag/gn
void launch_cauchy_vec(const DevProblem& P, hipStream_t st);    // vtmp = grad / d^2
void launch_combine_step(const DevProblem& P, double cg, double cn, hipStream_t st);  // step = cg*grad/d^2 + cn*gn ; GS, SN2
// device-side trust region (one-thread kernels on P.tr, see DevProblem::tr)
void launch_tr_after_solve(const DevProblem& P, TrConsts tc, int fresh, hipStream_t st);  // verdict on the linear solve, dogleg coefficients (fresh = 0: rejected dogleg step, same linearisation)
void launch_combine_step_dev(const DevProblem& P, hipStream_t st);                          // combine_step with the coefficients in P.tr
void launch_tr_after_model(const DevProblem& P, TrConsts tc, hipStream_t st);               // model decrease, step norm, validity, parameter tolerance
void launch_tr_decide(const DevProblem& P, TrConsts tc, hipStream_t st);                    // rho test, radius / damping update, accept: x = candidate
void launch_apply_step(const DevProblem& P, hipStream_t st);    // candidate = x (+) step
void launch_accept(const DevProblem& P, hipStream_t st);        // x = candidate
void launch_tr_accept(const DevProblem& P, hipStream_t st, double* box = nullptr, double seq = 0.0);     // x = candidate if TR_ACC (device-side decision); box: k_dense.hip
void launch_xnorm(const DevProblem& P, hipStream_t st);         // XN2
void launch_relpose(int num, const int* ptr, const double* pB, const double* pA, const double* kpA, const double* kpB, const double* sigA,
                    const double* sigB, const double* camA, const int* distA, const double* camB, const int* distB, double th, int min_inliers,
                    double* T, unsigned char* outlier, int* inliers, hipStream_t st);
void launch_reanchor(int K, const double* pose_old, const double* pose_new, double* vel, int L, const int* ref, double* lm,
                     hipStream_t st);

}  // namespace covgpu

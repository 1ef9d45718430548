// optimization_gpu.hpp — header-only C++ facade: covins::Optimization::{GlobalBundleAdjustment,
// PoseGraphOptimization} re-implemented on top of the C ABI of libcovgpu (include/covgpu.h).
//
// Drop-in boundary. The reference declares, in covins_backend/include/covins/covins_backend/
// optimization_be.hpp:38-51,
//     static auto GlobalBundleAdjustment(MapPtr map, int interations_limit, double time_limit,
//                                        bool visual_only = false, bool outlier_removal = true,
//                                        bool estimate_bias = false) -> void;
//     static auto PoseGraphOptimization(MapPtr map, PoseMap corrected_poses) -> void;
// and callers link those symbols directly (backend.cpp:141-156, placerec_be.cpp:327, placerec_gen_be.cpp:250).
// `covins_gpu::OptimizationT<Types>` keeps both signatures verbatim. It walks Map / Keyframe / Landmark with the
// very accessors optimization_be.cpp uses (GetKeyframesVec, IsInvalid, UpdateCeresFromState-equivalent reads,
// GetObservations, GetPredecessor, GetLoopConstraints, ...), applies the same gating rules and constants, and
// replaces each `ceres::Problem` + `ceres::Solve` by one flat `covgpu_problem` + covgpu_gba_solve /
// covgpu_pgo_solve. Write-back and map maintenance follow optimization_be.cpp:572-614 and :1037-1083.
//
// The header is Eigen-free on purpose (no Eigen in this build image): 4x4 transforms, 3-vectors and 6x6
// matrices are only touched through operator()(r,c) / operator[](i), which Eigen types and the test stand-ins
// (tests/cpp/standin_map.hpp) both provide. `Types` names the map classes and the few operations whose
// spelling differs between the real COVINS classes and a stand-in (see INTEGRATION.md for the COVINS binding).
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <unordered_map>
#include <memory>
#include <set>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../covgpu.h"

namespace covins_gpu {

// covins_params::opt / sys / placerec values read on this path (config/config_backend.yaml:8,115-140;
// config_backend.hpp:180-207). A plain struct instead of static-init globals.
struct Params {
  int gba_use_map_loop_constraints = 1;
  double th_gba_outlier_global = 0.92;
  int gba_fix_poses_loaded_maps = 0;
  int pgo_iteration_limit = 10;
  int use_nbr_kfs = 1;
  int use_robust_loss = 1;
  double robust_loss_th = 0.5;
  int pgo_fix_kfs_after_gba = 1;
  int pgo_fix_poses_loaded_maps = 0;
  double wt_kf_r = 10.0, wt_kf_t = 1.0, wt_kf_n1 = 10.0, wt_kf_n23 = 2.0, wt_kf_n45 = 3.0;
  double th_outlier_align = 1.3;   // opt.th_outlier_align (config_backend.yaml:117), OptimizeRelativePose
  std::string placerec_type = "COVINS";
  int strategy = COVGPU_DOGLEG;  // the reference runs DOGLEG (optimization_be.cpp:261,564,1028)
  int device = 0;
  // multi-GPU GlobalBundleAdjustment inside this one process (covgpu_gba_solve_multi): ranks > 1 shards the ONE map by sub-map
  // over `ranks` contexts; rank r runs on HIP device devices[r] (empty: devices 0 .. ranks-1). Ranks on different devices talk
  // over RCCL; ranks that share a device (a one-GPU box: the test form) over the library's in-process group.
  int n_gpus = 1;
  std::vector<int> devices;
  int flatten_threads = 0;  // host threads of the Map -> IR walk over landmarks; 0 = sys.threads_server-like default (hardware, <= 16)
  // GlobalBundleAdjustment with outlier removal: derive the second round's problem on the device from the resident first round
  // (covgpu_gba_two_round: ONE Map -> IR walk and ONE upload per call); 0: the reference's literal sequence — walk, solve, erase, walk, solve
  int device_second_round = 1;
  // Opt-in (0: the call ends with the reference's own map->Clean(), optimization_be.cpp:614). 1: the end of the two-round call reproduces what
  // Map::Clean's map check does (map_be.cpp:698-717: every landmark of the map whose observation map holds fewer than two entries is erased)
  // from counts the call already has — the Map -> IR walk visited every landmark's observations once, the device returned how many of them
  // each landmark kept — instead of copying every landmark's observation map AGAIN to read its size (50 ms of a 180 ms call on the 5-agent
  // map). NOT reproduced: Clean's second loop over every keyframe's landmark vector (:719-729), which only finds landmarks that a keyframe
  // references but the map does not list — none in a consistent map. tests/test_facade.py compares the two ends.
  int device_clean = 0;
  // (IMU noise and gravity are NOT parameters: every IMU factor carries its keyframe's own VICalibration values,
  //  Types::imu_calib, as the reference's per-keyframe preintegrators do — keyframe_be.cpp:187-195.)
};

namespace detail {

// f(keyframe, feature index) for every observation of a landmark: through Types::visit_observations(landmark, f) when the binding
// offers it (a walk under the landmark's lock, no copy), otherwise over the copy Landmark::GetObservations() returns — a
// std::map of smart pointers: one allocation and one atomic reference-count round trip per observation, 60 % of the walk.
template <class Types, class L, class F>
inline auto visit_observations(const L& lm, F&& f, int) -> decltype(Types::visit_observations(lm, f), void()) { Types::visit_observations(lm, f); }
template <class Types, class L, class F>
inline void visit_observations(const L& lm, F&& f, long) { const auto obs = lm.GetObservations(); for (auto& m : obs) f(m.first, m.second); }

inline void fatal(const char* msg) {  // the reference prints COUTFATAL and exit(-1) (e.g. optimization_be.cpp:113-114)
  std::fprintf(stderr, "[covins_gpu] FATAL: %s\n", msg);
  std::exit(-1);
}

// rotation matrix (rows/cols 0..2 of any (r,c)-indexable) -> Hamilton quaternion x,y,z,w (what Eigen::Quaterniond(R) yields)
template <class M>
inline void rot_to_quat(const M& T, double* q) {
  const double m00 = T(0, 0), m11 = T(1, 1), m22 = T(2, 2), tr = m00 + m11 + m22;
  double x, y, z, w;
  if (tr > 0) {
    const double s = std::sqrt(tr + 1.0) * 2;
    w = 0.25 * s; x = (T(2, 1) - T(1, 2)) / s; y = (T(0, 2) - T(2, 0)) / s; z = (T(1, 0) - T(0, 1)) / s;
  } else if (m00 > m11 && m00 > m22) {
    const double s = std::sqrt(1.0 + m00 - m11 - m22) * 2;
    w = (T(2, 1) - T(1, 2)) / s; x = 0.25 * s; y = (T(0, 1) + T(1, 0)) / s; z = (T(0, 2) + T(2, 0)) / s;
  } else if (m11 > m22) {
    const double s = std::sqrt(1.0 + m11 - m00 - m22) * 2;
    w = (T(0, 2) - T(2, 0)) / s; x = (T(0, 1) + T(1, 0)) / s; y = 0.25 * s; z = (T(1, 2) + T(2, 1)) / s;
  } else {
    const double s = std::sqrt(1.0 + m22 - m00 - m11) * 2;
    w = (T(1, 0) - T(0, 1)) / s; x = (T(0, 2) + T(2, 0)) / s; y = (T(1, 2) + T(2, 1)) / s; z = 0.25 * s;
  }
  const double n = std::sqrt(x * x + y * y + z * z + w * w);
  q[0] = x / n; q[1] = y / n; q[2] = z / n; q[3] = w / n;
}
// pose block [qx qy qz qw px py pz] from a 4x4 transform (keyframe_base.cpp:486-499)
template <class M>
inline void transform_to_pose(const M& T, double* p) {
  rot_to_quat(T, p);
  p[4] = T(0, 3); p[5] = T(1, 3); p[6] = T(2, 3);
}
// Utils::Ceres2Transform (utils_base.cpp:28-43): normalise q, fill a 4x4
template <class M>
inline void pose_to_transform(const double* p, M& T) {
  double x = p[0], y = p[1], z = p[2], w = p[3];
  const double n = std::sqrt(x * x + y * y + z * z + w * w);
  x /= n; y /= n; z /= n; w /= n;
  T(0, 0) = 1 - 2 * (y * y + z * z); T(0, 1) = 2 * (x * y - w * z);     T(0, 2) = 2 * (x * z + w * y);     T(0, 3) = p[4];
  T(1, 0) = 2 * (x * y + w * z);     T(1, 1) = 1 - 2 * (x * x + z * z); T(1, 2) = 2 * (y * z - w * x);     T(1, 3) = p[5];
  T(2, 0) = 2 * (x * z - w * y);     T(2, 1) = 2 * (y * z + w * x);     T(2, 2) = 1 - 2 * (x * x + y * y); T(2, 3) = p[6];
  T(3, 0) = 0; T(3, 1) = 0; T(3, 2) = 0; T(3, 3) = 1;
}
// [q,t] of Ta^-1 Tb for two 4x4 transforms (optimization_be.cpp:956-958, 1006-1008)
template <class M>
inline void relative_pose(const M& Ta, const M& Tb, double* out7) {
  double R[3][3], t[3];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) {
      R[r][c] = 0;
      for (int k = 0; k < 3; ++k) R[r][c] += Ta(k, r) * Tb(k, c);
    }
    t[r] = 0;
    for (int k = 0; k < 3; ++k) t[r] += Ta(k, r) * (Tb(k, 3) - Ta(k, 3));
  }
  struct V { double (*R)[3]; double operator()(int r, int c) const { return R[r][c]; } } v{R};
  rot_to_quat(v, out7);
  out7[4] = t[0]; out7[5] = t[1]; out7[6] = t[2];
}
// upper-triangular chol(A^-1)^T of a 6x6 SPD matrix given row-major (optimization_be.cpp:922-923)
inline void sqrt_info_from_cov(const double* cov, double* S) {
  double A[6][12];
  for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { A[r][c] = cov[6 * r + c]; A[r][6 + c] = (r == c); }
  for (int c = 0; c < 6; ++c) {  // Gauss-Jordan with partial pivoting
    int p = c;
    for (int r = c + 1; r < 6; ++r) if (std::fabs(A[r][c]) > std::fabs(A[p][c])) p = r;
    for (int k = 0; k < 12; ++k) std::swap(A[c][k], A[p][k]);
    const double d = A[c][c];
    for (int k = 0; k < 12; ++k) A[c][k] /= d;
    for (int r = 0; r < 6; ++r) if (r != c) { const double f = A[r][c]; for (int k = 0; k < 12; ++k) A[r][k] -= f * A[c][k]; }
  }
  double L[6][6] = {};
  for (int j = 0; j < 6; ++j) {
    double d = A[j][6 + j];
    for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
    L[j][j] = std::sqrt(d);
    for (int i = j + 1; i < 6; ++i) {
      double s = A[i][6 + j];
      for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
      L[i][j] = s / L[j][j];
    }
  }
  for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) S[6 * r + c] = L[c][r];
}

// C = A B for 4x4 transforms; p_cam = (T_w_c)^-1 p_w
template <class M>
inline void mat_mul(const M& A, const M& B, M& C) {
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { double s = 0; for (int k = 0; k < 4; ++k) s += A(r, k) * B(k, c); C(r, c) = s; }
}
template <class M, class V>
inline void to_camera(const M& Twc, const V& pw, double* out3) {
  for (int r = 0; r < 3; ++r) { double s = 0; for (int k = 0; k < 3; ++k) s += Twc(k, r) * (pw[k] - Twc(k, 3)); out3[r] = s; }
}

struct Flat {  // owning storage behind one covgpu_problem
  std::vector<double> pose, sb, cam_extr, cam_intr, cam_dist, lm, uv, sigma, samples, first, noise, meas, info, loss;
  std::vector<uint8_t> fixed;
  std::vector<int32_t> kf_cam, cam_type, obs_ptr, obs_kf, imu_i, imu_j, imu_ptr, ei, ej;
  covgpu_problem view() {
    covgpu_problem p{};
    p.num_kf = (int32_t)fixed.size(); p.num_cam = (int32_t)cam_type.size(); p.num_lm = (int32_t)(lm.size() / 3);
    p.num_obs = (int32_t)obs_kf.size(); p.num_imu = (int32_t)imu_i.size(); p.num_edge = (int32_t)ei.size();
    p.num_imu_samples = (int32_t)(samples.size() / 7);
    if (obs_ptr.empty()) obs_ptr.push_back(0);
    if (imu_ptr.empty()) imu_ptr.push_back(0);
    p.kf_pose = pose.data(); p.kf_speed_bias = sb.data(); p.kf_fixed = fixed.data(); p.kf_cam = kf_cam.data();
    p.cam_extr = cam_extr.data(); p.cam_intr = cam_intr.data(); p.cam_dist = cam_dist.data(); p.cam_dist_type = cam_type.data();
    p.lm_pos = lm.data(); p.lm_obs_ptr = obs_ptr.data(); p.obs_kf = obs_kf.data(); p.obs_uv = uv.data(); p.obs_sigma = sigma.data();
    p.imu_kf_i = imu_i.data(); p.imu_kf_j = imu_j.data(); p.imu_sample_ptr = imu_ptr.data(); p.imu_samples = samples.data();
    p.imu_first = first.data(); p.imu_noise = noise.data();
    p.edge_i = ei.data(); p.edge_j = ej.data(); p.edge_meas = meas.data(); p.edge_sqrt_info = info.data(); p.edge_loss_a = loss.data();
    return p;
  }
};

}  // namespace detail

// `Types` must provide (see tests/cpp/standin_map.hpp and INTEGRATION.md):
//   typedefs  Map, Keyframe, Landmark, TransformType, Vector3Type
//   static bool camera(const Keyframe&, double intr[4], double dist[4], int* dist_type);   false = unknown model
//   static int  imu_count(const Keyframe&);
//   static void imu_sample(const Keyframe&, int i, double* dt, double acc[3], double gyr[3]);
//   static void imu_first(const Keyframe&, double acc0[3], double gyr0[3]);
//   static void imu_calib(const Keyframe&, double out5[5]);   sigma_a_c, sigma_g_c, sigma_aw_c, sigma_gw_c, g of the
//                                                             keyframe's VICalibration (keyframe_be.cpp:187-195)
template <class Types>
class OptimizationT {
 public:
  using Map = typename Types::Map;
  using Keyframe = typename Types::Keyframe;
  using Landmark = typename Types::Landmark;
  using MapPtr = std::shared_ptr<Map>;
  using KeyframePtr = std::shared_ptr<Keyframe>;
  using LandmarkPtr = std::shared_ptr<Landmark>;
  using TransformType = typename Types::TransformType;
  using Vector3Type = typename Types::Vector3Type;
  using idpair = std::pair<size_t, size_t>;
  using PoseMap = std::map<idpair, TransformType>;

  OptimizationT() = delete;  // static-only, like the reference class (optimization_be.hpp:36)

  static Params& params() { static Params p; return p; }

  // row k of the IR = kfs[k], landmark l = lms[l]; observation i of the IR (landmark-major, Flat::obs_ptr) = feature obs_feat[i] of
  // keyframe kfs[Flat::obs_kf[i]]. (No smart-pointer copies per observation: 0.87 M observations x four atomic reference-count
  // updates on the keyframes' control blocks, contended between the walking threads, were most of a 0.28 s walk.)
  struct Index {
    std::vector<KeyframePtr> kfs; std::vector<LandmarkPtr> lms; std::vector<size_t> obs_feat;
    // for Params::device_clean: per IR landmark the entries of its observation map that are NOT in the IR (invalid / null keyframes), and the
    // valid landmarks the gating left out whose observation map holds fewer than two entries (what Map::Clean erases whatever the solve does)
    std::vector<int32_t> lm_extra; std::vector<LandmarkPtr> short_lms;
  };

  // Map -> IR for one GBA round (optimization_be.cpp:74-254 round 1, :308-557 round 2)
  static void FlattenGBA(const MapPtr& map, bool visual_only, bool round2, detail::Flat& f, Index& ix) {
    const Params& prm = params();
    const bool tim = std::getenv("COVGPU_FLATTEN_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
      if (!tim) return;
      const auto t = std::chrono::steady_clock::now();
      std::fprintf(stderr, "[covins_gpu] flatten %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
      t_last = t;
    };
    auto keyframes = map->GetKeyframesVec();
    auto landmarks = map->GetLandmarksVec();
    lap("map vectors");
    std::unordered_map<const Keyframe*, int32_t> row;
    row.reserve(2 * keyframes.size() + 16);
    std::map<size_t, std::vector<int32_t>> cams_of_client;
    for (auto& kf : keyframes) {
      if (kf->IsInvalid()) continue;
      const int32_t k = (int32_t)ix.kfs.size();
      row[kf.get()] = k; ix.kfs.push_back(kf);
      double p7[7];
      detail::transform_to_pose(kf->GetPoseTws(), p7);  // UpdateCeresFromState (keyframe_base.cpp:486-499)
      f.pose.insert(f.pose.end(), p7, p7 + 7);
      const Vector3Type v = kf->GetStateVelocity();
      Vector3Type ba, bg;
      kf->GetStateBias(ba, bg);
      for (int i = 0; i < 3; ++i) f.sb.push_back(v[i]);
      for (int i = 0; i < 3; ++i) f.sb.push_back(ba[i]);
      for (int i = 0; i < 3; ++i) f.sb.push_back(bg[i]);
      bool fixed = (kf->id_.first == 0 && kf->id_.second == map->id_map_);                       // :88-89, 329-331
      if (round2 && kf->is_loaded_ && prm.gba_fix_poses_loaded_maps) fixed = true;                // :338-341
      f.fixed.push_back(fixed ? 1 : 0);
      // Extrinsics, intrinsics and distortion are constant parameter blocks OF THIS KEYFRAME (:336,349,352:
      // kf->ceres_extrinsics_ / camera_ of every keyframe). Identical rows are shared: normally one per agent, but a
      // keyframe whose calibration differs from its agent's earlier ones gets its own camera row.
      double intr[4], dist[4], e7[7]; int dt = 0;
      if (!Types::camera(*kf, intr, dist, &dt)) detail::fatal("Unknown projection / distortion type.");  // :112-115, 201-204
      detail::transform_to_pose(kf->GetStateExtrinsics(), e7);
      std::vector<int32_t>& cands = cams_of_client[kf->id_.second];
      int32_t cam = -1;
      for (int32_t c : cands) {
        bool same = f.cam_type[c] == dt;
        for (int i = 0; same && i < 7; ++i) same = f.cam_extr[7 * c + i] == e7[i];
        for (int i = 0; same && i < 4; ++i) same = f.cam_intr[4 * c + i] == intr[i] && f.cam_dist[4 * c + i] == dist[i];
        if (same) { cam = c; break; }
      }
      if (cam < 0) {
        cam = (int32_t)f.cam_type.size(); cands.push_back(cam);
        f.cam_extr.insert(f.cam_extr.end(), e7, e7 + 7); f.cam_intr.insert(f.cam_intr.end(), intr, intr + 4);
        f.cam_dist.insert(f.cam_dist.end(), dist, dist + 4); f.cam_type.push_back(dt);
      }
      f.kf_cam.push_back(cam);
    }
    lap("keyframes");
    // IMU factors (:117-144 / :367-420)
    f.imu_ptr.assign(1, 0);
    if (!visual_only)
      for (auto& kf : ix.kfs) {
        KeyframePtr pred = kf->GetPredecessor();
        if (!pred || pred->IsInvalid()) {
          if (kf->id_.first != 0) detail::fatal("keyframe without predecessor");  // :121-124, 371-374
          continue;
        }
        const int n = Types::imu_count(*kf);
        if (round2 && n == 0) continue;  // "0 IMU measurements - skip IMU factor" (:382-385)
        f.imu_i.push_back(row.at(pred.get())); f.imu_j.push_back(row.at(kf.get()));
        double a0[3], g0[3];
        Types::imu_first(*kf, a0, g0);
        f.first.insert(f.first.end(), a0, a0 + 3); f.first.insert(f.first.end(), g0, g0 + 3);
        double nz[5];
        Types::imu_calib(*kf, nz);  // the preintegrator of THIS keyframe was built from its own calibration (keyframe_be.cpp:187-195)
        if (!(nz[0] > 0 && nz[1] > 0 && nz[2] > 0 && nz[3] > 0) || nz[4] < 9.0) detail::fatal("IMU calibration unset (sigma <= 0 or g < 9)");  // keyframe_base.cpp:51-55
        f.noise.insert(f.noise.end(), nz, nz + 5);
        for (int i = 0; i < n; ++i) {
          double dt, a[3], g[3];
          Types::imu_sample(*kf, i, &dt, a, g);
          if (dt == 0.0) continue;  // "dt: 0 -- skip measurement" (keyframe_be.cpp:199-202): the buffer may hold the initial reading
          f.samples.push_back(dt); f.samples.insert(f.samples.end(), a, a + 3); f.samples.insert(f.samples.end(), g, g + 3);
        }
        f.imu_ptr.push_back((int32_t)(f.samples.size() / 7));
      }
    // landmarks + observations (:147-236 / :425-530). The walk is O(#observations) of pointer chasing and std::map
    // copies (GetObservations), the dominant host cost of a call once the solve runs on the GPU: landmark ranges go to
    // host threads, each filling its own Flat/Index slice; the slices are concatenated in landmark order, so the IR is
    // the same for any thread count. (The map is exclusively checked out for the call, backend.cpp:134; the accessors
    // take the per-object mutexes.)
    lap("IMU factors");
    const size_t th_min_observations = 2;
    int nth = prm.flatten_threads > 0 ? prm.flatten_threads : (int)std::thread::hardware_concurrency();
    nth = std::max(1, std::min(nth, 16));
    if (landmarks.size() < 4096) nth = 1;
    struct Slice { std::vector<double> lm, uv, sigma; std::vector<int32_t> obs_kf, nobs, extra; std::vector<LandmarkPtr> lms, shorts; std::vector<size_t> feat; };
    std::vector<Slice> slices(nth);
    auto walk = [&](int t) {
      Slice sl;   // (thread-local while it grows: the vectors' end pointers of neighbouring slices[] entries share cache lines)
      const size_t l0 = landmarks.size() * (size_t)t / nth, l1 = landmarks.size() * (size_t)(t + 1) / nth;
      sl.lm.reserve(3 * (l1 - l0)); sl.lms.reserve(l1 - l0); sl.nobs.reserve(l1 - l0);
      sl.obs_kf.reserve(12 * (l1 - l0)); sl.uv.reserve(24 * (l1 - l0)); sl.sigma.reserve(12 * (l1 - l0)); sl.feat.reserve(12 * (l1 - l0));
      for (size_t li = l0; li < l1; ++li) {
        const LandmarkPtr& lm = landmarks[li];
        if (lm->IsInvalid()) continue;
        // (emitted straight into the slice and rolled back if fewer than two valid observations remain: :150-160, 428-440)
        const size_t o_mark = sl.obs_kf.size();
        int32_t n = 0, total = 0;
        detail::visit_observations<Types>(*lm, [&](const KeyframePtr& kfx, size_t feat) {
          ++total;
          if (!kfx || kfx->IsInvalid()) return;
          sl.obs_kf.push_back(row.at(kfx.get()));
          sl.uv.push_back((double)kfx->keypoints_distorted_[feat][0]);  // float -> double (utils_base.hpp:76-80)
          sl.uv.push_back((double)kfx->keypoints_distorted_[feat][1]);
          sl.sigma.push_back(((double)kfx->keypoints_aors_[feat][1] + 1) * 2.0);  // :184, 478
          sl.feat.push_back(feat);
          ++n;
        }, 0);
        if ((size_t)n < th_min_observations) {
          sl.obs_kf.resize(o_mark); sl.uv.resize(2 * o_mark); sl.sigma.resize(o_mark); sl.feat.resize(o_mark);
          if (total < 2) sl.shorts.push_back(lm);
          continue;
        }
        const Vector3Type pw = lm->GetWorldPos();
        for (int i = 0; i < 3; ++i) sl.lm.push_back(pw[i]);
        sl.lms.push_back(lm);
        sl.nobs.push_back(n);
        sl.extra.push_back(total - n);
      }
      slices[t] = std::move(sl);
    };
    auto in_threads = [&](const std::function<void(int)>& fn) {
      std::vector<std::thread> th;
      for (int t = 1; t < nth; ++t) th.emplace_back(fn, t);
      fn(0);
      for (auto& x : th) x.join();
    };
    in_threads(walk);
    lap("landmark walk (threads)");
    // slices -> IR at their offsets (prefix sums), again in the threads: 35 MB of copies
    std::vector<size_t> lm0(nth + 1, 0), ob0(nth + 1, 0);
    for (int t = 0; t < nth; ++t) { lm0[t + 1] = lm0[t] + slices[t].lms.size(); ob0[t + 1] = ob0[t] + slices[t].obs_kf.size(); }
    f.lm.resize(3 * lm0[nth]); f.obs_ptr.resize(lm0[nth] + 1); ix.lms.resize(lm0[nth]); ix.lm_extra.resize(lm0[nth]);
    for (int t = 0; t < nth; ++t) ix.short_lms.insert(ix.short_lms.end(), slices[t].shorts.begin(), slices[t].shorts.end());
    f.uv.resize(2 * ob0[nth]); f.sigma.resize(ob0[nth]); f.obs_kf.resize(ob0[nth]); ix.obs_feat.resize(ob0[nth]);
    f.obs_ptr[0] = 0;
    in_threads([&](int t) {
      Slice& sl = slices[t];
      std::copy(sl.lm.begin(), sl.lm.end(), f.lm.begin() + 3 * lm0[t]);
      std::copy(sl.uv.begin(), sl.uv.end(), f.uv.begin() + 2 * ob0[t]);
      std::copy(sl.sigma.begin(), sl.sigma.end(), f.sigma.begin() + ob0[t]);
      std::copy(sl.obs_kf.begin(), sl.obs_kf.end(), f.obs_kf.begin() + ob0[t]);
      std::copy(sl.feat.begin(), sl.feat.end(), ix.obs_feat.begin() + ob0[t]);
      std::move(sl.lms.begin(), sl.lms.end(), ix.lms.begin() + lm0[t]);
      std::copy(sl.extra.begin(), sl.extra.end(), ix.lm_extra.begin() + lm0[t]);
      int32_t at = (int32_t)ob0[t];
      for (size_t q = 0; q < sl.nobs.size(); ++q) { at += sl.nobs[q]; f.obs_ptr[lm0[t] + q + 1] = at; }
      sl = Slice();
    });
    lap("slices -> IR");
    // loop edges (:238-254 / :534-557): sqrt_info = diag(100 I3, 1e4 I3); loss only in round 2
    if (!round2 || prm.gba_use_map_loop_constraints)
      for (auto& lc : map->GetLoopConstraints()) {
        auto a = row.find(lc.kf1.get()), b = row.find(lc.kf2.get());
        if (a == row.end() || b == row.end()) { std::fprintf(stderr, "[covins_gpu] Loop KF missing -- skip loop\n"); continue; }  // :546-549
        double m7[7];
        detail::transform_to_pose(lc.T_s1_s2, m7);
        f.ei.push_back(a->second); f.ej.push_back(b->second); f.meas.insert(f.meas.end(), m7, m7 + 7);
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) f.info.push_back(r == c ? (r < 3 ? 100.0 : 1e4) : 0.0);
        f.loss.push_back(round2 ? 1.0 : 0.0);
      }
  }

  static covgpu_options Options(int max_it, bool visual_only) {
    const Params& prm = params();
    covgpu_options o;
    covgpu_default_options(&o);
    o.strategy = prm.strategy; o.max_iterations = max_it; o.visual_only = visual_only ? 1 : 0; o.device = prm.device;
    return o;
  }

  // One context per calling thread and device, created on first use and kept (streams, pinned buffers, the device context):
  // the reference calls these functions per loop closure (placerec_be.cpp:327) and per loop candidate (OptimizeRelativePose) —
  // creating and destroying a context for every call cost more than the single-pair solve it wrapped.
  struct ContextHolder {
    covgpu_context* ctx = nullptr; int device = -1;
    ~ContextHolder() { if (ctx) covgpu_destroy(ctx); }
  };
  static ContextHolder& Holder() {
    static thread_local ContextHolder h;
    return h;
  }
  // Explicit release of the calling thread's cached context (streams, pinned buffers, the last problem's device buffers): for callers
  // that do not want to rely on the thread_local destructor at thread / process teardown, or want the HBM back between calls.
  static void Shutdown() {
    ContextHolder& h = Holder();
    if (h.ctx != nullptr) { covgpu_destroy(h.ctx); h.ctx = nullptr; h.device = -1; }
  }
  static covgpu_context* Context() {
    ContextHolder& h = Holder();
    const int dev = params().device;
    if (h.ctx != nullptr && h.device != dev) { covgpu_destroy(h.ctx); h.ctx = nullptr; }
    if (h.ctx == nullptr) {
      covgpu_options o = Options(1, false);
      if (covgpu_create(&o, &h.ctx) != COVGPU_OK) detail::fatal(covgpu_last_error());
      h.device = dev;
    }
    return h.ctx;
  }
  // wall milliseconds of the stages of the calling thread's last GlobalBundleAdjustment call (measurement hook; no effect on the call)
  static std::vector<std::pair<std::string, double>>& last_stages() {
    static thread_local std::vector<std::pair<std::string, double>> v;
    return v;
  }
  // one GBA solve (+ optionally the outlier decisions at its estimate): one GPU through the thread's context, or sharded
  static void SolveGBA(covgpu_context* ctx, covgpu_options& o, covgpu_problem& p, covgpu_result& r, std::vector<uint8_t>* erase,
                       std::vector<int32_t>* lm_left, int64_t* counts) {
    const Params& prm = params();
    if (prm.n_gpus > 1) {
      std::vector<int32_t> dev(prm.n_gpus);
      for (int i = 0; i < prm.n_gpus; ++i) dev[i] = i < (int)prm.devices.size() ? prm.devices[i] : i;
      if (covgpu_gba_solve_multi(&o, &p, &r, prm.n_gpus, dev.data(), prm.th_gba_outlier_global, erase ? erase->data() : nullptr,
                                 lm_left ? lm_left->data() : nullptr, counts) != COVGPU_OK)
        detail::fatal(covgpu_last_error());
      return;
    }
    if (covgpu_gba_solve(ctx, &o, &p, &r) != COVGPU_OK) detail::fatal(covgpu_last_error());
    // problem.Evaluate + threshold (:270-289) on the device at the estimate the solve left resident: flags come back
    if (erase && covgpu_outlier_pass(ctx, prm.th_gba_outlier_global, erase->data(), lm_left->data(), counts) != COVGPU_OK) detail::fatal(covgpu_last_error());
  }

  // ---- optimization_be.cpp:56-618
  static auto GlobalBundleAdjustment(MapPtr map, int interations_limit, double time_limit, bool visual_only = false,
                                     bool outlier_removal = true, bool estimate_bias = false) -> void {
    (void)time_limit; (void)estimate_bias;  // never read by the reference either
    std::printf("+++ GBA: Start +++\n");
    const bool tim = std::getenv("COVGPU_FLATTEN_TIMING") != nullptr;   // per-stage wall times of the call on stderr
    auto t_last = std::chrono::steady_clock::now();
    last_stages().clear();
    auto lap = [&](const char* what) {   // (two clock reads per stage: kept for last_stages(), printed on request)
      const auto t = std::chrono::steady_clock::now();
      const double ms = std::chrono::duration<double, std::milli>(t - t_last).count();
      last_stages().emplace_back(what, ms);
      if (tim) std::fprintf(stderr, "[covins_gpu] GBA call %-34s %7.2f ms\n", what, ms);
      t_last = t;
    };
    covgpu_context* ctx = params().n_gpus > 1 ? nullptr : Context();   // (the sharded solve creates its own contexts, one per device)
    lap("context");
    if (outlier_removal && params().device_second_round) {
      // Both rounds behind one call. The first round's problem is walked once (:80-254); the outlier round, the erase decisions
      // (:270-290) and the rebuilt second-round problem (:296-557) stay on the device; the map is brought to the state the reference
      // leaves: observations erased (:281-289), the second round's estimate written back (:572-609), Map::Clean (:614).
      const Params& prm = params();
      detail::Flat f; Index ix;
      FlattenGBA(map, visual_only, false, f, ix);
      lap("Map -> IR (once)");
      std::vector<uint8_t> fixed2;
      if (prm.gba_fix_poses_loaded_maps) {   // :338-341
        fixed2.assign(f.fixed.begin(), f.fixed.end());
        for (size_t k = 0; k < ix.kfs.size(); ++k) if (ix.kfs[k]->is_loaded_) fixed2[k] = 1;
      }
      covgpu_two_round tr;
      tr.outlier_threshold = prm.th_gba_outlier_global; tr.round1_iterations = 5; tr.use_loops_round2 = prm.gba_use_map_loop_constraints ? 1 : 0;
      tr.loop_loss_round2 = 1.0; tr.kf_fixed_round2 = fixed2.empty() ? nullptr : fixed2.data();
      covgpu_problem p = f.view();
      covgpu_options o = Options(interations_limit, visual_only);
      covgpu_result r1, r2;
      std::vector<uint8_t> erase(f.obs_kf.size() + 1);
      std::vector<int32_t> lm_left(ix.lms.size() + 1);
      int64_t counts[2] = {0, 0};
      if (prm.n_gpus > 1) {   // (round 6) the sharded route: one upload per rank, the second round derived on every rank's device from its share
        std::vector<int32_t> dev(prm.n_gpus);
        for (int i = 0; i < prm.n_gpus; ++i) dev[i] = i < (int)prm.devices.size() ? prm.devices[i] : i;
        if (covgpu_gba_two_round_multi(&o, &p, &tr, prm.n_gpus, dev.data(), erase.data(), lm_left.data(), counts, &r1, &r2) != COVGPU_OK) detail::fatal(covgpu_last_error());
      } else if (covgpu_gba_two_round(ctx, &o, &p, &tr, erase.data(), lm_left.data(), counts, &r1, &r2) != COVGPU_OK) detail::fatal(covgpu_last_error());
      lap("upload + both rounds on the device");
      size_t num_bad = 0, lms2 = 0;
      for (size_t l = 0; l < ix.lms.size(); ++l) {
        for (int32_t i = f.obs_ptr[l]; i < f.obs_ptr[l + 1]; ++i)
          if (erase[i]) {  // :281-289
            const KeyframePtr& kf = ix.kfs[f.obs_kf[i]];
            kf->EraseLandmark(ix.obs_feat[i]);
            ix.lms[l]->EraseObservation(kf);
            ++num_bad;
          }
        lms2 += lm_left[l] >= 2 ? 1 : 0;
      }
      std::printf("--> GBA removed %zu of %zu observations\n", num_bad, f.obs_kf.size() * 2);
      std::printf("--> KFs: %zu\n--> LMs: %zu\n", ix.kfs.size(), lms2);
      lap("erase observations");
      if (r2.termination == 4) std::fprintf(stderr, "[covins_gpu] GBA: linear solve failed, keeping the last accepted estimate (as ceres::Solve would)\n");
      if (r2.reserved > 0) std::fprintf(stderr, "[covins_gpu] GBA: %d IMU factors without a positive definite covariance carry no weight\n", r2.reserved);
      for (size_t k = 0; k < ix.kfs.size(); ++k) {  // :572-595
        KeyframePtr& kf = ix.kfs[k];
        TransformType T;
        detail::pose_to_transform(&f.pose[7 * k], T);
        kf->SetPoseTws(T);
        kf->SetPoseOptimized();
        if (!visual_only) {
          const double* s = &f.sb[9 * k];
          Vector3Type vel, bA, bG;
          for (int i = 0; i < 3; ++i) { vel[i] = s[i]; bA[i] = s[3 + i]; bG[i] = s[6 + i]; }
          kf->SetStateBias(bA, bG);
          kf->SetStateVelocity(vel);
          kf->SetVelBiasOptimized();
        }
        kf->is_gba_optimized_ = true;
      }
      for (size_t l = 0; l < ix.lms.size(); ++l) {  // :598-609 — the landmarks of the SECOND round (:428-440: two observations left)
        if (lm_left[l] < 2) continue;
        Vector3Type pw;
        for (int i = 0; i < 3; ++i) pw[i] = f.lm[3 * l + i];
        ix.lms[l]->SetWorldPos(pw);
        ix.lms[l]->SetOptimized();
        ix.lms[l]->is_gba_optimized_ = true;
      }
      lap("write-back");
      std::printf("--> Clean Map\n");
      if (prm.device_clean) {
        // Map::Clean's map check (map_be.cpp:698-717) from the counts at hand: entries left = kept by the device + entries outside the IR
        size_t removed = 0;
        for (size_t l = 0; l < ix.lms.size(); ++l) if (lm_left[l] + ix.lm_extra[l] < 2) { map->EraseLandmark(ix.lms[l]); ++removed; }
        for (auto& lm : ix.short_lms) { map->EraseLandmark(lm); ++removed; }
        std::printf("----> Done: Removed %zu Landmarks (counts of the call; Params::device_clean)\n", removed);
        lap("Map::Clean (from the call's counts)");
      } else {
        map->Clean();  // :614
        lap("Map::Clean");
      }
      std::printf("--> done.\n+++ GBA: End +++\n");
      return;
    }
    if (outlier_removal) {  // first round (:62-293)
      detail::Flat f; Index ix;
      FlattenGBA(map, visual_only, false, f, ix);
      lap("round 1: Map -> IR");
      covgpu_problem p = f.view();
      covgpu_options o = Options(5, visual_only);  // max_num_iterations = 5 (:262)
      covgpu_result r;
      std::vector<uint8_t> erase(f.obs_kf.size() + 1);
      std::vector<int32_t> lm_left(ix.lms.size() + 1);
      int64_t counts[2] = {0, 0};
      SolveGBA(ctx, o, p, r, &erase, &lm_left, counts);
      lap("round 1: upload + solve + outlier pass");
      size_t num_bad = 0;
      for (size_t l = 0; l < ix.lms.size(); ++l)
        for (int32_t i = f.obs_ptr[l]; i < f.obs_ptr[l + 1]; ++i)
          if (erase[i]) {  // :281-289
            const KeyframePtr& kf = ix.kfs[f.obs_kf[i]];
            kf->EraseLandmark(ix.obs_feat[i]);
            ix.lms[l]->EraseObservation(kf);
            ++num_bad;
          }
      std::printf("--> GBA removed %zu of %zu observations\n", num_bad, f.obs_kf.size() * 2);
      lap("round 1: erase observations");
    }
    lap("round 1: release IR");
    {  // second round (:296-610)
      detail::Flat f; Index ix;
      FlattenGBA(map, visual_only, true, f, ix);
      lap("round 2: Map -> IR");
      std::printf("--> KFs: %zu\n--> LMs: %zu\n", ix.kfs.size(), ix.lms.size());
      covgpu_problem p = f.view();
      covgpu_options o = Options(interations_limit, visual_only);
      covgpu_result r;
      SolveGBA(ctx, o, p, r, nullptr, nullptr, nullptr);
      lap("round 2: upload + solve");
      if (r.termination == 4) std::fprintf(stderr, "[covins_gpu] GBA: linear solve failed, keeping the last accepted estimate (as ceres::Solve would)\n");
      if (r.reserved > 0) std::fprintf(stderr, "[covins_gpu] GBA: %d IMU factors without a positive definite covariance carry no weight\n", r.reserved);
      for (size_t k = 0; k < ix.kfs.size(); ++k) {  // :572-595
        KeyframePtr& kf = ix.kfs[k];
        TransformType T;
        detail::pose_to_transform(&f.pose[7 * k], T);
        kf->SetPoseTws(T);
        kf->SetPoseOptimized();
        if (!visual_only) {
          const double* s = &f.sb[9 * k];
          Vector3Type vel, bA, bG;
          for (int i = 0; i < 3; ++i) { vel[i] = s[i]; bA[i] = s[3 + i]; bG[i] = s[6 + i]; }
          kf->SetStateBias(bA, bG);
          kf->SetStateVelocity(vel);
          kf->SetVelBiasOptimized();
        }
        kf->is_gba_optimized_ = true;
      }
      for (size_t l = 0; l < ix.lms.size(); ++l) {  // :598-609
        Vector3Type pw;
        for (int i = 0; i < 3; ++i) pw[i] = f.lm[3 * l + i];
        ix.lms[l]->SetWorldPos(pw);
        ix.lms[l]->SetOptimized();
        ix.lms[l]->is_gba_optimized_ = true;
      }
      lap("round 2: write-back");
    }
    lap("round 2: release IR");
    std::printf("--> Clean Map\n");
    map->Clean();  // :614
    lap("Map::Clean");
    std::printf("--> done.\n+++ GBA: End +++\n");
  }

  // ---- optimization_be.cpp:833-1086
  static auto PoseGraphOptimization(MapPtr map, PoseMap corrected_poses) -> void {
    const Params& prm = params();
    const bool tim = std::getenv("COVGPU_FLATTEN_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
      if (!tim) return;
      const auto t = std::chrono::steady_clock::now();
      std::fprintf(stderr, "[covins_gpu] flatten %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
      t_last = t;
    };
    auto keyframes = map->GetKeyframesVec();
    auto landmarks = map->GetLandmarksVec();
    detail::Flat f;
    std::vector<KeyframePtr> kfs;
    std::map<Keyframe*, int32_t> row;
    for (auto& kf : keyframes) {  // :850-882
      if (kf->IsInvalid()) continue;
      row[kf.get()] = (int32_t)kfs.size(); kfs.push_back(kf);
      double p7[7];
      auto mit = corrected_poses.find(kf->id_);
      if (mit != corrected_poses.end()) detail::transform_to_pose(mit->second, p7);
      else detail::transform_to_pose(kf->GetPoseTws(), p7);
      f.pose.insert(f.pose.end(), p7, p7 + 7);
      for (int i = 0; i < 9; ++i) f.sb.push_back(0.0);
      bool fixed = (kf->id_.first == 0 && kf->id_.second == map->id_map_);
      if (kf->is_gba_optimized_ && prm.pgo_fix_kfs_after_gba) fixed = true;
      else if (kf->is_loaded_ && prm.pgo_fix_poses_loaded_maps) fixed = true;
      f.fixed.push_back(fixed ? 1 : 0);
      f.kf_cam.push_back(0);
    }
    f.cam_type.push_back(0);
    f.cam_extr.assign(7, 0.0); f.cam_extr[3] = 1.0; f.cam_intr.assign(4, 1.0); f.cam_dist.assign(4, 0.0);
    double W1[36] = {}, W23[36] = {}, W45[36] = {};  // :889-902
    for (int i = 0; i < 6; ++i) {
      W1[7 * i] = (i < 3 ? prm.wt_kf_r : prm.wt_kf_t) * prm.wt_kf_n1;
      W23[7 * i] = W1[7 * i] / prm.wt_kf_n23; W45[7 * i] = W1[7 * i] / prm.wt_kf_n45;
    }
    auto add_edge = [&](int32_t a, int32_t b, const double* m7, const double* S, double loss) {
      f.ei.push_back(a); f.ej.push_back(b); f.meas.insert(f.meas.end(), m7, m7 + 7); f.info.insert(f.info.end(), S, S + 36); f.loss.push_back(loss);
    };
    for (auto& lc : map->GetLoopConstraints()) {  // :912-944
      double S[36], m7[7];
      if (prm.placerec_type == "COVINS") for (int i = 0; i < 36; ++i) S[i] = W1[i];
      else {
        double cov[36];
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) cov[6 * r + c] = lc.cov_mat(r, c);
        detail::sqrt_info_from_cov(cov, S);
      }
      detail::transform_to_pose(lc.T_s1_s2, m7);
      auto a = row.find(lc.kf1.get()), b = row.find(lc.kf2.get());  // an invalidated loop keyframe must not take the server down
      if (a == row.end() || b == row.end()) { std::fprintf(stderr, "[covins_gpu] PGO: loop KF missing -- skip loop\n"); continue; }
      add_edge(a->second, b->second, m7, S, prm.use_robust_loss ? prm.robust_loss_th : 0.0);
    }
    std::set<std::pair<Keyframe*, Keyframe*>> inserted;
    for (auto& kf : kfs) {  // successor edges from the VIO poses (:947-972)
      KeyframePtr succ = kf->GetSuccessor();
      if (!succ) continue;
      auto sr = row.find(succ.get());
      if (sr == row.end()) continue;  // successor invalid / not in the map: no parameter block to tie to
      if (!inserted.insert({kf.get(), succ.get()}).second) continue;
      double m7[7];
      detail::relative_pose(kf->GetPoseTws_vio(), succ->GetPoseTws_vio(), m7);
      add_edge(row.at(kf.get()), sr->second, m7, W1, 0.0);
    }
    if (prm.use_nbr_kfs)  // five previous neighbours (:976-1021)
      for (auto& kf : kfs) {
        std::vector<KeyframePtr> connections;
        KeyframePtr temp = kf;
        for (int j = 1; j < 6; ++j)
          if (int(kf->id_.first) - j > 0) {
            temp = temp ? temp->GetPredecessor() : KeyframePtr();  // a chain shorter than id_.first suggests (culled keyframes) ends here
            if (!temp) break;
            connections.push_back(temp);
          }
        size_t k = 0;
        for (auto& kfc : connections) {
          k++;
          const double* S = (k <= 1) ? W1 : (k <= 3 ? W23 : W45);
          auto cr = row.find(kfc.get());
          if (cr == row.end()) continue;
          if (!inserted.insert({kf.get(), kfc.get()}).second) continue;
          double m7[7];
          detail::relative_pose(kf->GetPoseTws_vio(), kfc->GetPoseTws_vio(), m7);
          add_edge(row.at(kf.get()), cr->second, m7, S, 0.0);
        }
      }
    covgpu_problem p = f.view();
    covgpu_options o = Options(prm.pgo_iteration_limit, false);
    covgpu_context* ctx = Context();
    covgpu_result r;
    if (covgpu_pgo_solve(ctx, &o, &p, &r) != COVGPU_OK) detail::fatal(covgpu_last_error());
    // recover (:1033-1083): poses, velocity rotation, landmark re-anchoring on the device
    std::vector<double> pose_old(7 * kfs.size()), vel(3 * kfs.size());
    for (size_t k = 0; k < kfs.size(); ++k) {
      detail::transform_to_pose(kfs[k]->GetPoseTws(), &pose_old[7 * k]);
      const Vector3Type v = kfs[k]->GetStateVelocity();
      for (int i = 0; i < 3; ++i) vel[3 * k + i] = v[i];
    }
    std::vector<LandmarkPtr> lms;
    std::vector<int32_t> ref;
    std::vector<double> lmp;
    for (auto& lm : landmarks) {
      if (lm->IsInvalid()) continue;
      KeyframePtr kf_ref = lm->GetReferenceKeyframe();
      if (!kf_ref) { if (!lm->GetObservations().empty()) map->EraseLandmark(lm); continue; }  // :1059-1065
      auto it = row.find(kf_ref.get());
      if (it == row.end()) { map->EraseLandmark(lm); continue; }                              // :1069-1073
      const Vector3Type pw = lm->GetWorldPos();
      lms.push_back(lm); ref.push_back(it->second);
      for (int i = 0; i < 3; ++i) lmp.push_back(pw[i]);
    }
    if (covgpu_pgo_reanchor(ctx, (int32_t)kfs.size(), pose_old.data(), f.pose.data(), vel.data(), (int32_t)lms.size(), ref.data(),
                            lmp.data()) != COVGPU_OK)
      detail::fatal(covgpu_last_error());
    for (size_t k = 0; k < kfs.size(); ++k) {
      TransformType T;
      detail::pose_to_transform(&f.pose[7 * k], T);
      kfs[k]->SetPoseTws(T);  // UpdateFromCeres (:1045)
      Vector3Type v;
      for (int i = 0; i < 3; ++i) v[i] = vel[3 * k + i];
      kfs[k]->SetStateVelocity(v);
      kfs[k]->SetPoseOptimized();
    }
    for (size_t l = 0; l < lms.size(); ++l) {
      Vector3Type pw;
      for (int i = 0; i < 3; ++i) pw[i] = lmp[3 * l + i];
      lms[l]->SetWorldPos(pw);
      lms[l]->SetOptimized();
    }
    std::printf("--> PGO END \n");
  }

  // ---- optimization_be.cpp:620-831, batched: one entry per loop candidate (placerec_be.cpp:116-165 verifies them one by
  //      one; a batch goes to the device in ONE launch, covgpu_relpose_batch). Semantics per entry are the reference's:
  //      matches1[i] is reset for removed correspondences, T12 is updated unless fewer than 12 inliers are left, the
  //      return value is the inlier count (0 = rejected).
  using LandmarkVector = std::vector<LandmarkPtr>;
  struct RelPoseJob { KeyframePtr kf1, kf2; LandmarkVector* matches1; TransformType* T12; int result = 0; };
  static void OptimizeRelativePoseBatch(std::vector<RelPoseJob>& jobs) {
    const size_t B = jobs.size();
    if (B == 0) return;
    std::vector<int32_t> ptr(1, 0), dA(B), dB(B), inl(B);
    std::vector<double> pB, pA, kA, kB, sA, sB, camA(8 * B), camB(8 * B), T(7 * B);
    std::vector<std::vector<int>> index(B);   // correspondence -> position in matches1
    for (size_t b = 0; b < B; ++b) {
      RelPoseJob& j = jobs[b];
      int da = 0, db = 0;
      if (!Types::camera(*j.kf1, &camA[8 * b], &camA[8 * b + 4], &da) || !Types::camera(*j.kf2, &camB[8 * b], &camB[8 * b + 4], &db))
        detail::fatal("Unknown projection / distortion type.");  // :668-671, 695-697
      dA[b] = da; dB[b] = db;
      detail::transform_to_pose(*j.T12, &T[7 * b]);
      // TcwA = (Tws1 Tsc1)^-1, TcwB = (Tws2 Tsc1)^-1 — the reference uses kf1's extrinsics for both (:640-641)
      TransformType TwcA, TwcB;
      detail::mat_mul(j.kf1->GetPoseTws(), j.kf1->GetStateExtrinsics(), TwcA);
      detail::mat_mul(j.kf2->GetPoseTws(), j.kf1->GetStateExtrinsics(), TwcB);
      const auto lmsA = j.kf1->GetLandmarks();
      const int N = (int)j.matches1->size();
      for (int i = 0; i < N; ++i) {
        const LandmarkPtr& mb = (*j.matches1)[i];
        if (!mb) continue;
        const LandmarkPtr ma = i < (int)lmsA.size() ? lmsA[i] : LandmarkPtr();
        const int iB = mb->GetFeatureIndex(j.kf2);
        if (!ma || ma->IsInvalid() || mb->IsInvalid() || iB < 0) continue;        // :648-651
        double a3[3], b3[3];
        detail::to_camera(TwcA, ma->GetWorldPos(), a3); detail::to_camera(TwcB, mb->GetWorldPos(), b3);   // :652-656
        pA.insert(pA.end(), a3, a3 + 3); pB.insert(pB.end(), b3, b3 + 3);
        kA.push_back((double)j.kf1->keypoints_distorted_[i][0]); kA.push_back((double)j.kf1->keypoints_distorted_[i][1]);
        kB.push_back((double)j.kf2->keypoints_distorted_[iB][0]); kB.push_back((double)j.kf2->keypoints_distorted_[iB][1]);
        sA.push_back(((double)j.kf1->keypoints_aors_[i][1] + 1) * 2.0); sB.push_back(((double)j.kf2->keypoints_aors_[iB][1] + 1) * 2.0);   // :658, 717
        index[b].push_back(i);
      }
      ptr.push_back((int32_t)sA.size());
    }
    std::vector<uint8_t> out(sA.size() + 1);
    covgpu_relpose_batch_t bt{};
    bt.num_pairs = (int32_t)B; bt.corr_ptr = ptr.data(); bt.p_b = pB.data(); bt.p_a = pA.data(); bt.kp_a = kA.data(); bt.kp_b = kB.data();
    bt.sigma_a = sA.data(); bt.sigma_b = sB.data(); bt.cam_a = camA.data(); bt.cam_b = camB.data(); bt.dist_type_a = dA.data(); bt.dist_type_b = dB.data();
    bt.T_ab = T.data(); bt.outlier = out.data(); bt.inliers = inl.data();
    covgpu_context* ctx = Context();
    if (covgpu_relpose_batch(ctx, &bt, params().th_outlier_align, 12) != COVGPU_OK) detail::fatal(covgpu_last_error());
    for (size_t b = 0; b < B; ++b) {
      RelPoseJob& j = jobs[b];
      for (size_t c = 0; c < index[b].size(); ++c)
        if (out[ptr[b] + c]) (*j.matches1)[index[b][c]].reset();   // "matches1[i] = NULL" (:807; the reference indexes by residual, a quirk not copied)
      j.result = inl[b];
      if (inl[b] > 0) detail::pose_to_transform(&T[7 * b], *j.T12);  // :820
    }
  }
  // same signature as the reference (th2 is unused there too)
  static auto OptimizeRelativePose(KeyframePtr kf1, KeyframePtr kf2, LandmarkVector& matches1, TransformType& T12, const double th2) -> int {
    (void)th2;
    std::vector<RelPoseJob> jobs(1);
    jobs[0].kf1 = kf1; jobs[0].kf2 = kf2; jobs[0].matches1 = &matches1; jobs[0].T12 = &T12;
    OptimizeRelativePoseBatch(jobs);
    return jobs[0].result;
  }
};

}  // namespace covins_gpu

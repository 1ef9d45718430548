// standin_map.hpp — minimal stand-ins for covins::Map / Keyframe / Landmark with the SAME member names the
// optimisation path touches (SURVEY.md §8b "Inputs read" / "Outputs written"). The real classes cannot be compiled
// in this image (they need Eigen 3.3.4, OpenCV, aslam_cv2, robopt_open, DBoW2); these exist so that the C++ facade
// include/covins_gpu/optimization_gpu.hpp is compiled and exercised by tests/.
#pragma once
#include <array>
#include <map>
#include <mutex>
#include <memory>
#include <utility>
#include <vector>

namespace standin {

struct Mat4 {
  double m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  double& operator()(int r, int c) { return m[4 * r + c]; }
  double operator()(int r, int c) const { return m[4 * r + c]; }
};
struct Mat6 {
  double m[36] = {};
  Mat6() { for (int i = 0; i < 6; ++i) m[7 * i] = 1.0; }
  double& operator()(int r, int c) { return m[6 * r + c]; }
  double operator()(int r, int c) const { return m[6 * r + c]; }
};
struct Vec3 {
  double v[3] = {0, 0, 0};
  double& operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
};

class Keyframe;
class Landmark;
using KeyframePtr = std::shared_ptr<Keyframe>;
using LandmarkPtr = std::shared_ptr<Landmark>;
using idpair = std::pair<size_t, size_t>;

struct Camera { double intr[4]; double dist[4]; int dist_type; bool known = true; };

class Keyframe : public std::enable_shared_from_this<Keyframe> {
 public:
  idpair id_;
  bool is_loaded_ = false, is_gba_optimized_ = false;
  std::vector<std::array<float, 2>> keypoints_distorted_;
  std::vector<std::array<float, 4>> keypoints_aors_;  // angle, octave, response, size
  std::vector<LandmarkPtr> landmarks_;
  Camera camera_;
  // raw IMU between predecessor and this keyframe (what robopt::imu::PreintegrationBase buffers)
  std::vector<std::array<double, 7>> imu_;  // dt, acc, gyr
  double acc0_[3] = {0, 0, 0}, gyr0_[3] = {0, 0, 0};
  double imu_calib_[5] = {0, 0, 0, 0, 0};  // VICalibration: sigma_a_c sigma_g_c sigma_aw_c sigma_gw_c g (typedefs_base.hpp:333-340)

  bool IsInvalid() const { return invalid_; }
  void SetInvalid() { invalid_ = true; }
  Mat4 GetPoseTws() const { return T_w_s_; }
  Mat4 GetPoseTws_vio() const { return T_w_s_vio_; }
  Mat4 GetStateExtrinsics() const { return T_s_c_; }
  Vec3 GetStateVelocity() const { return vel_; }
  void GetStateBias(Vec3& ba, Vec3& bg) const { ba = ba_; bg = bg_; }
  void SetPoseTws(Mat4 T, bool = true) { T_w_s_ = T; }
  void SetPoseTws_vio(Mat4 T, bool = true) { T_w_s_vio_ = T; }
  void SetStateExtrinsics(Mat4 T) { T_s_c_ = T; }
  void SetStateBias(Vec3 ba, Vec3 bg) { ba_ = ba; bg_ = bg; }
  void SetStateVelocity(Vec3 v) { vel_ = v; }
  void SetPoseOptimized() { pose_optimized_ = true; }
  void SetVelBiasOptimized() { vel_bias_optimized_ = true; }
  KeyframePtr GetPredecessor() const { return pred_.lock(); }
  KeyframePtr GetSuccessor() const { return succ_.lock(); }
  void EraseLandmark(size_t index) { if (index < landmarks_.size()) landmarks_[index].reset(); }
  std::vector<LandmarkPtr> GetLandmarks() const { return landmarks_; }

  std::weak_ptr<Keyframe> pred_, succ_;
  bool pose_optimized_ = false, vel_bias_optimized_ = false;

 private:
  bool invalid_ = false;
  Mat4 T_w_s_, T_w_s_vio_, T_s_c_;
  Vec3 vel_, ba_, bg_;
};

class Landmark {
 public:
  using KfObservations = std::map<KeyframePtr, size_t>;
  idpair id_;
  bool is_gba_optimized_ = false, optimized_ = false;
  bool IsInvalid() const { return invalid_; }
  void SetInvalid() { invalid_ = true; }
  Vec3 GetWorldPos() const { return pos_w_; }
  void SetWorldPos(Vec3 p) { pos_w_ = p; }
  KfObservations GetObservations() const { std::unique_lock<std::mutex> lock(mtx_obs_); return observations_; }  // landmark_base.cpp:84-87
  // the three-line addition INTEGRATION.md proposes for LandmarkBase: walk the observations under the lock without copying the map
  template <class F> void VisitObservations(F&& f) const { std::unique_lock<std::mutex> lock(mtx_obs_); for (auto& o : observations_) f(o.first, o.second); }
  void AddObservation(KeyframePtr kf, size_t idx) { observations_[kf] = idx; }
  void EraseObservation(KeyframePtr kf) { observations_.erase(kf); }
  int GetFeatureIndex(KeyframePtr kf) const { auto it = observations_.find(kf); return it == observations_.end() ? -1 : (int)it->second; }
  KeyframePtr GetReferenceKeyframe() const { return ref_.lock(); }
  void SetReferenceKeyframe(KeyframePtr kf) { ref_ = kf; }
  void SetOptimized() { optimized_ = true; }

 private:
  bool invalid_ = false;
  Vec3 pos_w_;
  KfObservations observations_;
  mutable std::mutex mtx_obs_;
  std::weak_ptr<Keyframe> ref_;
};

struct LoopConstraint {  // typedefs_base.hpp:264-277
  KeyframePtr kf1, kf2;
  Mat4 T_s1_s2;
  Mat6 cov_mat;
};

class Map {
 public:
  using LoopVector = std::vector<LoopConstraint>;
  size_t id_map_ = 0;
  std::map<idpair, KeyframePtr> keyframes_;  // sorted by (id, client) like typedefs_base.hpp:178
  std::map<idpair, LandmarkPtr> landmarks_;
  LoopVector loops_;
  std::vector<KeyframePtr> GetKeyframesVec() const { std::vector<KeyframePtr> v; for (auto& p : keyframes_) v.push_back(p.second); return v; }
  std::vector<LandmarkPtr> GetLandmarksVec() const { std::vector<LandmarkPtr> v; for (auto& p : landmarks_) v.push_back(p.second); return v; }
  LoopVector GetLoopConstraints() const { return loops_; }
  void EraseLandmark(LandmarkPtr lm) { lm->SetInvalid(); }
  void Clean() {  // Map::Clean (map_be.cpp:448-454, 698-743): drop landmarks with < 2 observations
    for (auto& p : landmarks_) if (!p.second->IsInvalid() && p.second->GetObservations().size() < 2) p.second->SetInvalid();
  }
};

struct TypesBase {
  using Map = standin::Map;
  using Keyframe = standin::Keyframe;
  using Landmark = standin::Landmark;
  using TransformType = Mat4;
  using Vector3Type = Vec3;
  static bool camera(const Keyframe& kf, double intr[4], double dist[4], int* dist_type) {
    if (!kf.camera_.known) return false;
    for (int i = 0; i < 4; ++i) { intr[i] = kf.camera_.intr[i]; dist[i] = kf.camera_.dist[i]; }
    *dist_type = kf.camera_.dist_type;
    return true;
  }
  static int imu_count(const Keyframe& kf) { return (int)kf.imu_.size(); }
  static void imu_sample(const Keyframe& kf, int i, double* dt, double acc[3], double gyr[3]) {
    *dt = kf.imu_[i][0];
    for (int k = 0; k < 3; ++k) { acc[k] = kf.imu_[i][1 + k]; gyr[k] = kf.imu_[i][4 + k]; }
  }
  static void imu_first(const Keyframe& kf, double acc0[3], double gyr0[3]) {
    for (int k = 0; k < 3; ++k) { acc0[k] = kf.acc0_[k]; gyr0[k] = kf.gyr0_[k]; }
  }
  static void imu_calib(const Keyframe& kf, double out5[5]) { for (int k = 0; k < 5; ++k) out5[k] = kf.imu_calib_[k]; }
};
// with the optional observation visitor (optimization_gpu.hpp: detail::visit_observations); TypesBase exercises the fallback
// through Landmark::GetObservations() — what an unmodified LandmarkBase offers
struct Types : TypesBase {
  template <class F> static void visit_observations(const Landmark& lm, F&& f) { lm.VisitObservations(f); }
};

}  // namespace standin

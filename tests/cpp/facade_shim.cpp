// facade_shim.cpp — C entry points that let tests/ drive the C++ facade (include/covins_gpu/optimization_gpu.hpp)
// on a stand-in map built from the same arrays as covins_amd.mapdata.SlamMap.
#include <cstring>

#include "../../include/covins_gpu/optimization_gpu.hpp"
#include "standin_map.hpp"

using namespace standin;
using Opt = covins_gpu::OptimizationT<standin::Types>;
using OptCopy = covins_gpu::OptimizationT<standin::TypesBase>;   // no observation visitor: the walk goes through Landmark::GetObservations()
static int g_use_visitor = 1;

struct Handle {
  std::shared_ptr<Map> map;
  std::vector<KeyframePtr> kfs;
  std::vector<LandmarkPtr> lms;
};

static Mat4 pose_to_mat(const double* p) { Mat4 T; covins_gpu::detail::pose_to_transform(p, T); return T; }

extern "C" {

Handle* shim_build(int K, const int* kf_id, const int* kf_client, const unsigned char* kf_invalid, const unsigned char* kf_loaded,
                   const unsigned char* kf_gba, const double* pose, const double* pose_vio, const double* vel, const double* ba,
                   const double* bg, const int* pred, const int* succ, const int* kf_cam, const double* cam_extr, const double* cam_intr,
                   const double* cam_dist, const int* cam_type, const double* cam_imu_calib, const long* imu_ptr, const double* imu_samples, const double* imu_first,
                   int L, const double* lm_pos, const unsigned char* lm_invalid, const int* lm_ref, const int* lm_obs_ptr, const int* obs_kf,
                   const float* obs_uv, const int* obs_octave, int NL, const int* loop_kf1, const int* loop_kf2, const double* loop_T,
                   const double* loop_cov, int id_map) {
  Handle* h = new Handle();
  h->map = std::make_shared<Map>();
  h->map->id_map_ = (size_t)id_map;
  for (int k = 0; k < K; ++k) {
    auto kf = std::make_shared<Keyframe>();
    kf->id_ = {(size_t)kf_id[k], (size_t)kf_client[k]};
    kf->is_loaded_ = kf_loaded[k]; kf->is_gba_optimized_ = kf_gba[k];
    if (kf_invalid[k]) kf->SetInvalid();
    kf->SetPoseTws(pose_to_mat(pose + 7 * k)); kf->SetPoseTws_vio(pose_to_mat(pose_vio + 7 * k));
    const int c = kf_cam[k];
    kf->SetStateExtrinsics(pose_to_mat(cam_extr + 7 * c));
    for (int i = 0; i < 4; ++i) { kf->camera_.intr[i] = cam_intr[4 * c + i]; kf->camera_.dist[i] = cam_dist[4 * c + i]; }
    kf->camera_.dist_type = cam_type[c];
    for (int i = 0; i < 5; ++i) kf->imu_calib_[i] = cam_imu_calib[5 * c + i];
    Vec3 v, a, g;
    for (int i = 0; i < 3; ++i) { v[i] = vel[3 * k + i]; a[i] = ba[3 * k + i]; g[i] = bg[3 * k + i]; }
    kf->SetStateVelocity(v); kf->SetStateBias(a, g);
    for (long s = imu_ptr[k]; s < imu_ptr[k + 1]; ++s) {
      std::array<double, 7> smp;
      for (int i = 0; i < 7; ++i) smp[i] = imu_samples[7 * s + i];
      kf->imu_.push_back(smp);
    }
    for (int i = 0; i < 3; ++i) { kf->acc0_[i] = imu_first[6 * k + i]; kf->gyr0_[i] = imu_first[6 * k + 3 + i]; }
    h->kfs.push_back(kf);
    h->map->keyframes_[kf->id_] = kf;
  }
  for (int k = 0; k < K; ++k) {
    if (pred[k] >= 0) h->kfs[k]->pred_ = h->kfs[pred[k]];
    if (succ[k] >= 0) h->kfs[k]->succ_ = h->kfs[succ[k]];
  }
  for (int l = 0; l < L; ++l) {
    auto lm = std::make_shared<Landmark>();
    lm->id_ = {(size_t)l, 0};
    Vec3 p; for (int i = 0; i < 3; ++i) p[i] = lm_pos[3 * l + i];
    lm->SetWorldPos(p);
    if (lm_invalid[l]) lm->SetInvalid();
    if (lm_ref[l] >= 0) lm->SetReferenceKeyframe(h->kfs[lm_ref[l]]);
    for (int o = lm_obs_ptr[l]; o < lm_obs_ptr[l + 1]; ++o) {
      KeyframePtr kf = h->kfs[obs_kf[o]];
      const size_t idx = kf->keypoints_distorted_.size();
      kf->keypoints_distorted_.push_back({obs_uv[2 * o], obs_uv[2 * o + 1]});
      kf->keypoints_aors_.push_back({0.f, (float)obs_octave[o], 0.f, 0.f});
      kf->landmarks_.push_back(lm);
      lm->AddObservation(kf, idx);
    }
    h->lms.push_back(lm);
    h->map->landmarks_[lm->id_] = lm;
  }
  for (int i = 0; i < NL; ++i) {
    LoopConstraint lc;
    lc.kf1 = h->kfs[loop_kf1[i]]; lc.kf2 = h->kfs[loop_kf2[i]];
    lc.T_s1_s2 = pose_to_mat(loop_T + 7 * i);
    for (int q = 0; q < 36; ++q) lc.cov_mat.m[q] = loop_cov[36 * i + q];
    h->map->loops_.push_back(lc);
  }
  return h;
}

void shim_free(Handle* h) { delete h; }

// flattening only (no GPU): sizes, then arrays on a second call
int shim_flatten_gba(Handle* h, int visual_only, int round2, int* sizes /* K L O I E S */, double* pose, unsigned char* fixed, double* lm,
                     int* obs_ptr, int* obs_kf, double* uv, double* sigma, int* imu_i, int* imu_j, int* ei, int* ej, double* loss, double* noise, int* kf_cam,
                     int* ncam) {
  covins_gpu::detail::Flat f;
  if (g_use_visitor) { Opt::Index ix; Opt::FlattenGBA(h->map, visual_only != 0, round2 != 0, f, ix); }
  else { OptCopy::Index ix; OptCopy::FlattenGBA(h->map, visual_only != 0, round2 != 0, f, ix); }
  covgpu_problem p = f.view();
  sizes[0] = p.num_kf; sizes[1] = p.num_lm; sizes[2] = p.num_obs; sizes[3] = p.num_imu; sizes[4] = p.num_edge; sizes[5] = p.num_imu_samples;
  if (ncam) *ncam = p.num_cam;
  if (!pose) return 0;
  if (noise) std::memcpy(noise, f.noise.data(), f.noise.size() * 8);
  if (kf_cam) std::memcpy(kf_cam, f.kf_cam.data(), f.kf_cam.size() * 4);
  std::memcpy(pose, f.pose.data(), f.pose.size() * 8); std::memcpy(fixed, f.fixed.data(), f.fixed.size());
  std::memcpy(lm, f.lm.data(), f.lm.size() * 8); std::memcpy(obs_ptr, f.obs_ptr.data(), f.obs_ptr.size() * 4);
  std::memcpy(obs_kf, f.obs_kf.data(), f.obs_kf.size() * 4); std::memcpy(uv, f.uv.data(), f.uv.size() * 8);
  std::memcpy(sigma, f.sigma.data(), f.sigma.size() * 8);
  std::memcpy(imu_i, f.imu_i.data(), f.imu_i.size() * 4); std::memcpy(imu_j, f.imu_j.data(), f.imu_j.size() * 4);
  std::memcpy(ei, f.ei.data(), f.ei.size() * 4); std::memcpy(ej, f.ej.data(), f.ej.size() * 4); std::memcpy(loss, f.loss.data(), f.loss.size() * 8);
  return 0;
}

void shim_use_visitor(int on) { g_use_visitor = on; }
void shim_set_flatten_threads(int n) { Opt::params().flatten_threads = n; OptCopy::params().flatten_threads = n; }

void shim_gba(Handle* h, int iterations, int visual_only, int outlier_removal) {
  Opt::GlobalBundleAdjustment(h->map, iterations, -1.0, visual_only != 0, outlier_removal != 0, false);
}

// corrected poses: n entries (kf index, pose7)
void shim_pgo(Handle* h, int n, const int* kf_idx, const double* poses) {
  Opt::PoseMap cp;
  for (int i = 0; i < n; ++i) cp[h->kfs[kf_idx[i]]->id_] = pose_to_mat(poses + 7 * i);
  Opt::PoseGraphOptimization(h->map, cp);
}

void shim_set_params(int strategy, const char* placerec_type) {
  Opt::params().strategy = strategy;
  Opt::params().placerec_type = placerec_type;
}
// n > 1: GlobalBundleAdjustment shards the map over n in-process ranks, all on HIP device `device` (virtual ranks: the one-GPU form)
void shim_set_gpus(int n, int device) { Opt::params().n_gpus = n; Opt::params().devices.assign((size_t)(n > 1 ? n : 0), device); }
// stages of the last GlobalBundleAdjustment call of this thread: fills name (<= 63 chars each, 64-byte slots) / ms, returns the count
int shim_last_stages(char* names, double* ms, int cap) {
  int n = 0;
  for (auto& st : Opt::last_stages()) {
    if (n >= cap) break;
    std::snprintf(names + 64 * n, 64, "%s", st.first.c_str()); ms[n] = st.second; ++n;
  }
  return n;
}
void shim_set_device_second_round(int on) { Opt::params().device_second_round = on; }
void shim_set_device_clean(int on) { Opt::params().device_clean = on; }   // Params::device_clean: the call's own counts instead of map->Clean()
void shim_shutdown() { Opt::Shutdown(); }   // releases the calling thread's cached context; the next call creates a new one
void shim_set_invalid(Handle* h, int kf) { h->kfs[kf]->SetInvalid(); }

// OptimizeRelativePose between two keyframes of the map: matches1[i] = the landmark of kf2 matched to feature i of kf1 (here:
// the SAME landmark when both observe it), T12 in/out as pose7. Returns the inlier count; removed[] flags per feature of kf1.
int shim_relpose(Handle* h, int kf1, int kf2, double* T12_pose7, unsigned char* removed) {
  KeyframePtr a = h->kfs[kf1], b = h->kfs[kf2];
  std::vector<LandmarkPtr> matches(a->landmarks_.size());
  for (size_t i = 0; i < matches.size(); ++i)
    if (a->landmarks_[i] && a->landmarks_[i]->GetFeatureIndex(b) >= 0) matches[i] = a->landmarks_[i];
  std::vector<LandmarkPtr> before = matches;
  Mat4 T = pose_to_mat(T12_pose7);
  const int n = Opt::OptimizeRelativePose(a, b, matches, T, 0.0);
  covins_gpu::detail::transform_to_pose(T, T12_pose7);
  for (size_t i = 0; i < matches.size(); ++i) removed[i] = (before[i] && !matches[i]) ? 1 : 0;
  return n;
}

void shim_get_state(Handle* h, double* pose, double* vel, double* ba, double* bg, unsigned char* kf_gba, double* lm, unsigned char* lm_invalid,
                    int* lm_nobs) {
  for (size_t k = 0; k < h->kfs.size(); ++k) {
    covins_gpu::detail::transform_to_pose(h->kfs[k]->GetPoseTws(), pose + 7 * k);
    Vec3 v = h->kfs[k]->GetStateVelocity(), a, g;
    h->kfs[k]->GetStateBias(a, g);
    for (int i = 0; i < 3; ++i) { vel[3 * k + i] = v[i]; ba[3 * k + i] = a[i]; bg[3 * k + i] = g[i]; }
    kf_gba[k] = h->kfs[k]->is_gba_optimized_;
  }
  for (size_t l = 0; l < h->lms.size(); ++l) {
    Vec3 p = h->lms[l]->GetWorldPos();
    for (int i = 0; i < 3; ++i) lm[3 * l + i] = p[i];
    lm_invalid[l] = h->lms[l]->IsInvalid();
    lm_nobs[l] = (int)h->lms[l]->GetObservations().size();
  }
}

}  // extern "C"

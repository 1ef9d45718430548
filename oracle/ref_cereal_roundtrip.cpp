// ref_cereal_roundtrip — TEST INFRASTRUCTURE ONLY: pins covins_amd/mapio.py to the REFERENCE's own serialisation code.
//
// Compiled (oracle/Makefile, target _ref/cereal_roundtrip) against the reference's headers and sources WHERE THEY LIE:
//   /root/reference/covins_comm/include/covins/covins_base/msgs/msg_keyframe.hpp   MsgKeyframe::save/load (:129-203), the Eigen /
//                                                                                  cv::Mat cereal savers (:207-285), PreintegrationData
//   /root/reference/covins_comm/include/covins/covins_base/msgs/msg_landmark.hpp   MsgLandmark::save/load (:66-97)
//   /root/reference/covins_comm/include/covins/covins_base/typedefs_base.hpp       VICalibration::serialize (:376-380), idpair, containers
//   /root/reference/covins_comm/src/covins_base/msgs/msg_{keyframe,landmark}.cpp   constructors
//   /root/reference/covins_comm/thirdparty/cereal                                  cereal::Binary{Input,Output}Archive
//   /root/reference/covins_backend/include/covins/covins_backend/map_be.hpp:126-136   struct MsgMap (extracted by the Makefile into
//                                                                                  _ref/msg_map.inc: map_be.hpp itself pulls in the back-end)
// Eigen and OpenCV are not installed: oracle/standin/ supplies a storage-only Eigen::Matrix and cv::Mat with the members those
// headers use. No reference source is copied into this repository; outputs go to oracle/_ref/ only.
//
// What it does = Map::LoadFromFile followed by Map::SaveToFile at the archive level (map_be.cpp:508-696, :813-922):
//   for every keyframes/keyframesN.txt, mappoints/mappointsN.txt and mapdata.txt under <in>: load it with the reference's
//   load() through cereal::BinaryInputArchive (MsgKeyframe(true) / MsgLandmark(true) / MsgMap), print what the reference decoded,
//   save it with the reference's save() through cereal::BinaryOutputArchive into <out>.
// usage: cereal_roundtrip <in_dir> <out_dir>   (the summary goes to stdout as JSON)
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "covins_base/msgs/msg_keyframe.hpp"
#include "covins_base/msgs/msg_landmark.hpp"

namespace covins {
#include "_ref/msg_map.inc"
}

static std::string slurp(const std::string& p) {
  std::ifstream f(p, std::ios::binary);
  std::stringstream ss; ss << f.rdbuf();
  return ss.str();
}
static void spit(const std::string& p, const std::string& s) { std::ofstream f(p, std::ios::binary); f << s; }
static std::vector<std::string> list(const std::string& d) {
  std::vector<std::string> out;
  if (DIR* dir = opendir(d.c_str())) {
    while (dirent* e = readdir(dir)) if (e->d_name[0] != '.') out.push_back(e->d_name);
    closedir(dir);
  }
  std::sort(out.begin(), out.end());
  return out;
}
template <class M> static std::string mat(const M& m) {
  std::ostringstream o; o << std::setprecision(17) << "[";
  for (int i = 0; i < m.rows(); ++i) for (int j = 0; j < m.cols(); ++j) o << (i + j ? "," : "") << m(i, j);
  o << "]";
  return o.str();
}

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s <in_dir> <out_dir>\n", argv[0]); return 2; }
  const std::string in = argv[1], out = argv[2];
  mkdir(out.c_str(), 0777); mkdir((out + "/keyframes").c_str(), 0777); mkdir((out + "/mappoints").c_str(), 0777);
  std::cout << std::setprecision(17) << "{\n \"keyframes\": {";
  bool first = true;
  for (const std::string& f : list(in + "/keyframes")) {
    covins::MsgKeyframe msg(true);                       // map_be.cpp:560: MsgKeyframe msg(true)
    { std::stringstream ss(slurp(in + "/keyframes/" + f)); cereal::BinaryInputArchive ia(ss); ia(msg); }
    std::stringstream os; { cereal::BinaryOutputArchive oa(os); oa(msg); }   // map_be.cpp:868-870
    spit(out + "/keyframes/" + f, os.str());
    std::cout << (first ? "" : ",") << "\n  \"" << f << "\": {\"id\": [" << msg.id.first << "," << msg.id.second << "], \"timestamp\": " << msg.timestamp
              << ", \"T_w_s\": " << mat(msg.T_w_s) << ", \"T_s_c\": " << mat(msg.T_s_c) << ", \"velocity\": " << mat(msg.velocity)
              << ", \"bias_accel\": " << mat(msg.bias_accel) << ", \"bias_gyro\": " << mat(msg.bias_gyro)
              << ", \"intrinsics\": " << mat(msg.calibration.intrinsics) << ", \"dist_coeffs\": " << mat(msg.calibration.dist_coeffs)
              << ", \"cam_model\": " << (int)msg.calibration.cam_model << ", \"dist_model\": " << (int)msg.calibration.dist_model
              << ", \"sigma_a_c\": " << msg.calibration.sigma_a_c << ", \"sigma_gw_c\": " << msg.calibration.sigma_gw_c << ", \"g\": " << msg.calibration.g
              << ", \"n_keypoints\": " << msg.keypoints_distorted.size() << ", \"n_imu\": " << msg.preintegration.dt.size()
              << ", \"n_landmarks\": " << msg.landmarks.size() << ", \"desc_rows\": " << msg.descriptors.rows << ", \"desc_cols\": " << msg.descriptors.cols
              << ", \"pred\": [" << msg.id_predecessor.first << "," << msg.id_predecessor.second << "], \"succ\": [" << msg.id_successor.first << ","
              << msg.id_successor.second << "]"
              << ", \"kp0\": " << (msg.keypoints_distorted.empty() ? std::string("[]") : mat(msg.keypoints_distorted[0]))
              << ", \"dt_sum\": " << [&] { double s = 0; for (double d : msg.preintegration.dt) s += d; return s; }() << "}";
    first = false;
  }
  std::cout << "\n },\n \"mappoints\": {";
  first = true;
  for (const std::string& f : list(in + "/mappoints")) {
    covins::MsgLandmark msg(true);                       // map_be.cpp:627: MsgLandmark msg(true)
    { std::stringstream ss(slurp(in + "/mappoints/" + f)); cereal::BinaryInputArchive ia(ss); ia(msg); }
    std::stringstream os; { cereal::BinaryOutputArchive oa(os); oa(msg); }   // map_be.cpp:892-894
    spit(out + "/mappoints/" + f, os.str());
    std::cout << (first ? "" : ",") << "\n  \"" << f << "\": {\"id\": [" << msg.id.first << "," << msg.id.second << "], \"pos_w\": " << mat(msg.pos_w)
              << ", \"n_obs\": " << msg.observations.size() << ", \"ref\": [" << msg.id_reference.first << "," << msg.id_reference.second << "], \"obs\": [";
    bool f2 = true;
    for (const auto& o : msg.observations) { std::cout << (f2 ? "" : ",") << "[" << o.first.first << "," << o.first.second << "," << o.second << "]"; f2 = false; }
    std::cout << "]}";
    first = false;
  }
  std::cout << "\n },\n";
  {
    covins::MsgMap msg;                                  // map_be.cpp:520: MsgMap map_msg
    { std::stringstream ss(slurp(in + "/mapdata.txt")); cereal::BinaryInputArchive ia(ss); ia(msg); }
    std::stringstream os; { cereal::BinaryOutputArchive oa(os); oa(msg); }   // map_be.cpp:911-913
    spit(out + "/mapdata.txt", os.str());
    std::cout << " \"mapdata\": {\"id_map\": " << msg.id_map << ", \"n_loops\": " << msg.keyframes1.size() << ", \"kf1\": [";
    for (size_t i = 0; i < msg.keyframes1.size(); ++i) std::cout << (i ? "," : "") << "[" << msg.keyframes1[i].first << "," << msg.keyframes1[i].second << "]";
    std::cout << "], \"T12_0\": " << (msg.transforms12.empty() ? std::string("[]") : mat(msg.transforms12[0]))
              << ", \"cov_0\": " << (msg.cov.empty() ? std::string("[]") : mat(msg.cov[0])) << "}\n}\n";
  }
  return 0;
}

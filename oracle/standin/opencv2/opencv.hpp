// Stand-in for <opencv2/opencv.hpp> — TEST INFRASTRUCTURE ONLY (oracle/Makefile: _ref/cereal_fixture).
// OpenCV is not installed in this image. cv::Mat here is a plain byte matrix with the members the reference's cereal saver
// reads (msg_keyframe.hpp:237-283: rows, cols, type(), isContinuous(), elemSize(), ptr()); cv::FileStorage only has to exist for
// the yaml helpers of typedefs_base.hpp:73-100 to compile (they are never called).
#pragma once
#include <string>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_64F 6
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)

namespace cv {
class Mat {
 public:
  int rows = 0, cols = 0;
  Mat() = default;
  Mat(int r, int c, int t) { create(r, c, t); }
  void create(int r, int c, int t) { rows = r; cols = c; type_ = t; d_.assign((size_t)r * c * elemSize(), 0); }
  int type() const { return type_; }
  bool isContinuous() const { return true; }
  size_t elemSize() const {
    static const int depth_bytes[8] = {1, 1, 2, 2, 4, 4, 8, 2};
    return (size_t)depth_bytes[type_ & 7] * ((type_ >> 3) + 1);
  }
  unsigned char* ptr(int i = 0) { return d_.data() + (size_t)i * cols * elemSize(); }
  const unsigned char* ptr(int i = 0) const { return d_.data() + (size_t)i * cols * elemSize(); }

 private:
  int type_ = 0;
  std::vector<unsigned char> d_;
};

struct FileNode {
  operator double() const { return 0.0; }
  operator std::string() const { return std::string(); }
};
class FileStorage {
 public:
  enum { READ = 0 };
  FileStorage(const std::string&, int) {}
  bool isOpened() const { return false; }
  FileNode operator[](const std::string&) const { return FileNode(); }
};
}  // namespace cv

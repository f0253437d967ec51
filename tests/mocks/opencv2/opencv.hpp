// declaration-only stand-in for the OpenCV calls of ros/point_cloud_node.cpp (see tests/mocks/README.md)
#pragma once
#include <cstdint>
#include <string>
#include <vector>
typedef unsigned char uchar;
#define CV_8UC1 0
#define CV_32F 5
#define CV_64F 6
namespace cv {
struct Size { int width, height; Size(); Size(int w, int h); };
struct Rect { int x, y, width, height; Rect(); Rect(int x, int y, int w, int h); };
struct Vec3d { double val[3]; double& operator[](int i); };
class Mat {
 public:
  Mat(); Mat(int rows, int cols, int type);
  explicit Mat(const std::vector<uchar>& v, bool copy = false);
  explicit Mat(const Vec3d& v, bool copy = true);
  uchar* data; int rows, cols;
  bool empty() const;
  Mat clone() const;
  Mat operator()(const Rect& roi) const;
  template <class T> T* ptr(int row = 0);
  template <class T> const T* ptr(int row = 0) const;
};
class FileNode { public: FileNode(); };
void operator>>(const FileNode& n, Mat& m);
void operator>>(const FileNode& n, Vec3d& v);
class FileStorage {
 public:
  enum { READ = 0, WRITE = 1 };
  FileStorage(const std::string& path, int flags);
  FileNode operator[](const char* name) const;
  bool isOpened() const;
};
enum { IMREAD_GRAYSCALE = 0, INTER_LINEAR = 1, CALIB_ZERO_DISPARITY = 1024 };
Mat imdecode(const Mat& buf, int flags);
void remap(const Mat& src, Mat& dst, const Mat& map1, const Mat& map2, int interpolation, int border_mode = 0);
void stereoRectify(const Mat& K1, const Mat& D1, const Mat& K2, const Mat& D2, Size image_size, const Mat& R, const Mat& T, Mat& R1, Mat& R2, Mat& P1, Mat& P2,
                   Mat& Q, int flags = CALIB_ZERO_DISPARITY, double alpha = -1, Size new_image_size = Size(), Rect* roi1 = nullptr, Rect* roi2 = nullptr);
void initUndistortRectifyMap(const Mat& K, const Mat& D, const Mat& R, const Mat& P, Size size, int m1type, Mat& map1, Mat& map2);
}  // namespace cv

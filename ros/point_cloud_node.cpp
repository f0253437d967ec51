// ros/point_cloud_node.cpp — the `point_cloud` node of jackal_nav over libjn_stereo.so (seam B3).
//
// Built only where catkin, OpenCV and popt exist (ros/CMakeLists.txt); NOT built or tested in the development image,
// which has none of them.  It keeps the node surface of the reference's src/obstacle_avoidance/point_cloud.cpp so that
// the existing launch files, `navigate` and rviz configurations work unchanged:
//   node name      jackal_obstacle_avoidance                                   (point_cloud.cpp:499)
//   subscribes     /webcam/left/image_raw/compressed, /webcam/right/image_raw/compressed   (:567-568, queue 1)
//   publishes      /webcam/left/depth_map (mono8), /webcam/left/obstacle_scan (LaserScan),
//                  /webcam/left/point_cloud (PointCloud, with -g), /jackal/time_log (with -l),
//                  visualization_marker (advertised only)                        (:520-528)
//   flags          -h img-height  -c calib-file  -l logging  -g gen-pcl  -m calib-extrinsic  -d/-p/-s time files   (:502-514)
//   frame ids      "jackal" for scan and cloud (:277, :315)
// What runs where: JPEG decode stays cv::imdecode (:436, :478); rectification maps come from cv::stereoRectify /
// cv::initUndistortRectifyMap exactly as in the reference (:543-554) so that the rectified pair is bit-identical to
// the existing deployment's; everything after the rectified pair — ELAS, u8 map, LUT, reprojection, 90-bin scan,
// point cloud — runs on the GPU through the C ABI of include/jn_stereo.h.
// The -m flag (dynamic_reconfigure of the camera->robot extrinsics, :305-311, :560-564) is accepted and updates XR / XT
// through a plain parameter poll instead of a dynamic_reconfigure server, to keep this target free of generated code.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include <ros/ros.h>
#include <cv_bridge/cv_bridge.h>
#include <image_transport/image_transport.h>
#include <sensor_msgs/CompressedImage.h>
#include <sensor_msgs/LaserScan.h>
#include <sensor_msgs/PointCloud.h>
#include <sensor_msgs/image_encodings.h>
#include <visualization_msgs/Marker.h>
#include <opencv2/opencv.hpp>
#include <popt.h>

#include "jackal_nav/JackalTimeLog.h"
#include "jn_stereo.h"

namespace {

struct Options {
  int crop_height = 180;            // -h (point_cloud.cpp:53, :503)
  char* calib_file = nullptr;       // -c
  int logging = 0, gen_pcl = 0, calib_extrinsic = 0;
  char *dmap_file = nullptr, *pcl_file = nullptr, *scan_file = nullptr;
};

class PointCloudNode {
 public:
  PointCloudNode(ros::NodeHandle& nh, const Options& opt) : nh_(nh), opt_(opt), it_(nh) {
    depth_pub_ = it_.advertise("/webcam/left/depth_map", 1);
    cloud_pub_ = nh_.advertise<sensor_msgs::PointCloud>("/webcam/left/point_cloud", 1);
    scan_pub_ = nh_.advertise<sensor_msgs::LaserScan>("/webcam/left/obstacle_scan", 1);
    marker_pub_ = nh_.advertise<visualization_msgs::Marker>("visualization_marker", 1);
    if (opt_.logging) log_pub_ = nh_.advertise<jackal_nav::JackalTimeLog>("/jackal/time_log", 1);
    load_calibration();
    setup_gpu();
    left_sub_ = nh_.subscribe("/webcam/left/image_raw/compressed", 1, &PointCloudNode::on_left, this);
    right_sub_ = nh_.subscribe("/webcam/right/image_raw/compressed", 1, &PointCloudNode::on_right, this);
  }
  ~PointCloudNode() {
    jn_elas_destroy(elas_);
    for (void* p : {(void*)d_left_, (void*)d_right_, (void*)d_lut_, (void*)d_u8_, (void*)d_D1_, (void*)d_D2_, (void*)d_bins_, (void*)d_meta_, (void*)d_xyz_})
      if (p) jn_device_free(0, p);
  }

 private:
  static constexpr int kW = 320, kH = 180;          // rawimsize (point_cloud.cpp:49-50)
  ros::NodeHandle& nh_;
  Options opt_;
  image_transport::ImageTransport it_;
  image_transport::Publisher depth_pub_;
  ros::Publisher cloud_pub_, scan_pub_, marker_pub_, log_pub_;
  ros::Subscriber left_sub_, right_sub_;
  cv::Mat lmapx_, lmapy_, rmapx_, rmapy_, right_rect_;
  jn_scan_params sp_;
  jn_elas* elas_ = nullptr;
  uint8_t *d_left_ = nullptr, *d_right_ = nullptr, *d_lut_ = nullptr, *d_u8_ = nullptr;
  float *d_D1_ = nullptr, *d_D2_ = nullptr, *d_xyz_ = nullptr;
  double *d_bins_ = nullptr, *d_meta_ = nullptr;
  uint32_t seq_ = 0;

  void load_calibration() {
    cv::Mat K1, K2, D1, D2, R, XR, XT, R1, R2, P1, P2, Q;
    cv::Vec3d T;
    cv::FileStorage fs(opt_.calib_file, cv::FileStorage::READ);
    fs["K1"] >> K1; fs["K2"] >> K2; fs["D1"] >> D1; fs["D2"] >> D2; fs["R"] >> R; fs["T"] >> T; fs["XR"] >> XR; fs["XT"] >> XT;
    const cv::Size calib_size(640, 360), raw(kW, kH);             // point_cloud.cpp:38, :540
    cv::stereoRectify(K1, D1, K2, D2, calib_size, R, cv::Mat(T), R1, R2, P1, P2, Q, cv::CALIB_ZERO_DISPARITY, 0, raw);
    cv::initUndistortRectifyMap(K1, D1, R1, P1, raw, CV_32F, lmapx_, lmapy_);
    cv::initUndistortRectifyMap(K2, D2, R2, P2, raw, CV_32F, rmapx_, rmapy_);
    jn_scan_params_default(&sp_, kW, opt_.crop_height);
    std::memcpy(sp_.Q, Q.ptr<double>(), 16 * sizeof(double));
    std::memcpy(sp_.XR, XR.ptr<double>(), 9 * sizeof(double));
    std::memcpy(sp_.XT, XT.ptr<double>(), 3 * sizeof(double));
  }

  void setup_gpu() {
    const int W = kW, H = opt_.crop_height;
    jn_elas_params p;
    jn_elas_params_default(&p, JN_SETTING_ROBOTICS);
    p.postprocess_only_left = 1;                                   // point_cloud.cpp:416-417
    if (jn_elas_create(&p, W, H, 1, 0, 8, 1, &elas_) != JN_OK) { ROS_FATAL("jn_elas_create failed: no usable MI355X"); ros::shutdown(); return; }
    const int64_t px = (int64_t)W * H;
    jn_device_malloc(0, px, (void**)&d_left_); jn_device_malloc(0, px, (void**)&d_right_);
    jn_device_malloc(0, 2 * px, (void**)&d_lut_); jn_device_malloc(0, px, (void**)&d_u8_);
    jn_device_malloc(0, 4 * px, (void**)&d_D1_); jn_device_malloc(0, 4 * px, (void**)&d_D2_);
    jn_device_malloc(0, sizeof(double) * sp_.bins, (void**)&d_bins_); jn_device_malloc(0, sizeof(double) * 4, (void**)&d_meta_);
    if (opt_.gen_pcl) jn_device_malloc(0, 12 * px, (void**)&d_xyz_);
    else jn_build_valid_disp_lut(0, &sp_, W, H, d_lut_);         // cacheDisparityValues(), :557-558
  }

  cv::Mat rectify(const sensor_msgs::CompressedImageConstPtr& msg, const cv::Mat& mapx, const cv::Mat& mapy) const {
    cv::Mat grey = cv::imdecode(cv::Mat(msg->data), cv::IMREAD_GRAYSCALE), rect;
    if (grey.empty()) return rect;
    cv::remap(grey, rect, mapx, mapy, cv::INTER_LINEAR);
    return rect(cv::Rect(0, 0, kW, opt_.crop_height)).clone();
  }

  void on_right(const sensor_msgs::CompressedImageConstPtr& msg) { right_rect_ = rectify(msg, rmapx_, rmapy_); }   // :473-490

  void poll_extrinsics() {                                         // -m: PHI_X/Y/Z (degrees) and TRANS_X/Y/Z as ROS parameters
    double phi[3] = {0, 0, 0}, tr[3] = {sp_.XT[0], sp_.XT[1], sp_.XT[2]};
    const char* pn[3] = {"PHI_X", "PHI_Y", "PHI_Z"}; const char* tn[3] = {"TRANS_X", "TRANS_Y", "TRANS_Z"};
    bool any = false;
    for (int i = 0; i < 3; i++) { any |= nh_.getParam(pn[i], phi[i]); any |= nh_.getParam(tn[i], tr[i]); }
    if (!any) return;
    const double d2r = 3.1415 / 180.0;
    const double cx = std::cos(phi[0] * d2r), sx = std::sin(phi[0] * d2r), cy = std::cos(phi[1] * d2r), sy = std::sin(phi[1] * d2r),
                 cz = std::cos(phi[2] * d2r), sz = std::sin(phi[2] * d2r);
    const double Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1}, Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy}, Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
    double zy[9], zyx[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { zy[3 * r + c] = 0; for (int k = 0; k < 3; k++) zy[3 * r + c] += Rz[3 * r + k] * Ry[3 * k + c]; }
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { zyx[3 * r + c] = 0; for (int k = 0; k < 3; k++) zyx[3 * r + c] += zy[3 * r + k] * Rx[3 * k + c]; }
    std::memcpy(sp_.XR, zyx, sizeof zyx); std::memcpy(sp_.XT, tr, sizeof tr);
    if (!opt_.gen_pcl) jn_build_valid_disp_lut(0, &sp_, kW, opt_.crop_height, d_lut_);
  }

  void on_left(const sensor_msgs::CompressedImageConstPtr& msg) {   // :431-471
    const int W = kW, H = opt_.crop_height;
    const cv::Mat left = rectify(msg, lmapx_, lmapy_);
    if (left.empty() || right_rect_.empty()) return;               // the reference publishes the left image as "dmap" here (:407-408); nothing useful to scan
    if (opt_.calib_extrinsic) poll_extrinsics();
    const ros::WallTime t0 = ros::WallTime::now();
    jn_memcpy_h2d(0, d_left_, left.data, (int64_t)W * H); jn_memcpy_h2d(0, d_right_, right_rect_.data, (int64_t)W * H);
    jn_memcpy_h2d(0, d_D1_, std::vector<float>((size_t)W * H, 0.f).data(), 4 * (int64_t)W * H);   // Mat::zeros of :413: a failed frame publishes zeros
    int32_t st = JN_OK;
    jn_elas_process_batch(elas_, 1, d_left_, d_right_, W, (int64_t)W * H, d_D1_, d_D2_, &st);
    if (st != JN_OK) std::printf("ERROR: Need at least 3 support points!\n");   // elas.cpp:66-71
    const ros::WallTime t1 = ros::WallTime::now();
    std::vector<double> bins(sp_.bins), meta(4);
    sensor_msgs::PointCloud cloud;
    if (opt_.gen_pcl) {                                            // :321-404
      jn_disparity_to_u8(0, d_D1_, d_u8_, (int64_t)W * H);
      int64_t count = 0;
      jn_point_cloud(0, &sp_, d_u8_, W, H, d_xyz_, &count);
      std::vector<float> xyz((size_t)count * 3);
      if (count) jn_memcpy_d2h(0, xyz.data(), d_xyz_, 12 * count);
      cloud.header.frame_id = "jackal"; cloud.header.stamp = ros::Time::now();
      cloud.points.resize(count);
      for (int64_t i = 0; i < count; i++) { cloud.points[i].x = xyz[3 * i]; cloud.points[i].y = xyz[3 * i + 1]; cloud.points[i].z = xyz[3 * i + 2]; }
      cloud_pub_.publish(cloud);
      jn_obstacle_scan_cloud(0, &sp_, 1, d_u8_, W, H, d_bins_, d_meta_);
    } else {
      jn_disparity_scan(0, &sp_, 1, d_D1_, d_lut_, W, H, d_u8_, d_bins_, d_meta_);      // :422 + :213-296 in one pass
    }
    const ros::WallTime t2 = ros::WallTime::now();
    jn_memcpy_d2h(0, bins.data(), d_bins_, sizeof(double) * sp_.bins); jn_memcpy_d2h(0, meta.data(), d_meta_, sizeof(double) * 4);
    cv::Mat depth(H, W, CV_8UC1);
    jn_memcpy_d2h(0, depth.data, d_u8_, (int64_t)W * H);
    depth_pub_.publish(cv_bridge::CvImage(std_msgs::Header(), sensor_msgs::image_encodings::MONO8, depth).toImageMsg());   // :425-426

    sensor_msgs::LaserScan scan;                                    // :271-283
    std::vector<float> ranges(sp_.bins);
    const int nr = jn_compact_ranges(bins.data(), sp_.bins, ranges.data());
    scan.header.seq = seq_; scan.header.frame_id = "jackal"; scan.header.stamp = ros::Time::now();
    scan.angle_min = (float)meta[0]; scan.angle_max = (float)meta[1]; scan.range_min = (float)meta[2]; scan.range_max = (float)meta[3];
    scan.angle_increment = 3.1415f / 180.f; scan.scan_time = 0.001f; scan.time_increment = 0.1f;
    scan.ranges.assign(ranges.begin(), ranges.begin() + nr);
    scan_pub_.publish(scan);

    if (opt_.logging) {                                             // :446-462, msg/JackalTimeLog.msg
      jackal_nav::JackalTimeLog log;
      log.header.seq = seq_; log.header.stamp = ros::Time::now();
      log.dmap_time = (float)(t1 - t0).toSec();
      (opt_.gen_pcl ? log.pcl_time : log.obstacle_scan_time) = (float)(t2 - t1).toSec();
      log_pub_.publish(log);
      auto append = [](const char* path, double v) { if (path) { std::ofstream f(path, std::ios::app); f << v << "\n"; } };
      append(opt_.dmap_file, log.dmap_time); append(opt_.gen_pcl ? opt_.pcl_file : opt_.scan_file, (t2 - t1).toSec());
    }
    seq_++;
  }
};

}  // namespace

int main(int argc, char** argv) {
  ros::init(argc, argv, "jackal_obstacle_avoidance");
  ros::NodeHandle nh;
  Options o;
  struct poptOption table[] = {
      {"img-height", 'h', POPT_ARG_INT, &o.crop_height, 0, "Image height", "NUM"},
      {"calib-file", 'c', POPT_ARG_STRING, &o.calib_file, 0, "Stereo calibration file", "STR"},
      {"logging", 'l', POPT_ARG_NONE, &o.logging, 0, "Log pipeline time", "NONE"},
      {"gen-pcl", 'g', POPT_ARG_NONE, &o.gen_pcl, 0, "Generate PCL", "NONE"},
      {"calib-extrinsic", 'm', POPT_ARG_NONE, &o.calib_extrinsic, 0, "Calibrate extrinsics between left camera and robot", "NONE"},
      {"dmap-file", 'd', POPT_ARG_STRING, &o.dmap_file, 0, "DMAP time file", "STR"},
      {"pcl-file", 'p', POPT_ARG_STRING, &o.pcl_file, 0, "PCL time file", "STR"},
      {"scan-file", 's', POPT_ARG_STRING, &o.scan_file, 0, "Scan time file", "STR"},
      POPT_AUTOHELP{nullptr, 0, 0, nullptr, 0, nullptr, nullptr}};
  poptContext ctx = poptGetContext(nullptr, argc, const_cast<const char**>(argv), table, 0);
  while (poptGetNextOpt(ctx) >= 0) {}
  poptFreeContext(ctx);
  if (!o.calib_file) { std::fprintf(stderr, "point_cloud: -c <calibration yml> is required\n"); return 2; }
  PointCloudNode node(nh, o);
  ros::spin();
  return 0;
}

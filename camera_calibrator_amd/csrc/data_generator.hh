// data_generator.hh -- synthetic 2D/3D correspondences with the surface of the reference's
// calibrator::DataGenerator (src/data_generator.hh:14-48), without OpenCV: cv::projectPoints with zero
// rvec/tvec (data_generator.cpp:20) is the radial-tangential model evaluated in double and stored as
// float. Harness code: used by bench.py and the tests to produce inputs, never by the solver.
#pragma once
#include <random>

#include "types.hh"

namespace calibrator {

struct GeneratedData {
  Points2D image;
  Points3D world;
};

class DataGenerator {
 public:
  DataGenerator(int img_width, int img_height);
  void SetK(const Matrix3& K);
  Matrix3 GetK() const { return K_; }
  int GetWidth() const { return width_; }
  int GetHeight() const { return height_; }
  DynamicVector GetDistortion() const { return dist_; }
  void SetDistortion(const DynamicVector& dist);
  void SetNoiseInPixels(const float noise);
  GeneratedData GetDistortedPoints(const int num_p = 100);
  GeneratedData GetDistortedPointsPlanar(const int num_p = 100);
  /// candidates thrown away because they fell outside the image (each still consumed random draws)
  long long Rejected() const { return rejected_; }

 private:
  Point3D GetRandomPixel();
  Point3D GetRandom3DPointVisibleToCamera(const Matrix3& K_inv);
  Plane GetRandomPlane(const Matrix3& K_inv);
  bool ProjectAndDistort(const Point3D& p, Point2D* out);

  int width_{0};
  int height_{0};
  float min_distance_{0.2f};
  float max_distance_{1.0f};
  float noise_in_pixels_{0.0f};
  long long rejected_{0};
  Matrix3 K_{Matrix3::Identity()};
  DynamicVector dist_{DynamicVector::Zero(5)};
  std::mt19937 gen_{0};
  std::uniform_real_distribution<float> rand_w_;
  std::uniform_real_distribution<float> rand_h_;
  std::uniform_real_distribution<float> rand_dist_;
  std::uniform_real_distribution<float> rand_pixel_;
};

}  // namespace calibrator

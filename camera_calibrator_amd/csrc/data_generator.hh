// data_generator.hh -- synthetic 2D/3D correspondences. Public surface = the reference's
// calibrator::DataGenerator (src/data_generator.hh:14-33: constructor, SetK/GetK, GetWidth/GetHeight,
// SetDistortion/GetDistortion, SetNoiseInPixels, GetDistortedPoints, GetDistortedPointsPlanar), without
// OpenCV: cv::projectPoints with zero rvec/tvec (data_generator.cpp:20) is the radial-tangential model
// evaluated in double and stored as float. Harness code: used by bench.py and the tests to produce
// inputs, never by the solver.
#pragma once
#include <random>

#include "types.hh"

namespace calibrator {

/// One generated view: image points (pixels) and the matching 3-D points.
struct GeneratedData {
  Points2D image;
  Points3D world;
};

class DataGenerator {
 public:
  DataGenerator(int img_width, int img_height);

  // ---- camera model of the synthetic camera
  void SetK(const Matrix3& K);
  void SetDistortion(const DynamicVector& dist);   ///< k1 k2 p1 p2 k3
  void SetNoiseInPixels(const float noise);        ///< uniform in [-noise, noise] on both pixel axes
  Matrix3 GetK() const { return cam_.K; }
  DynamicVector GetDistortion() const { return cam_.dist; }
  int GetWidth() const { return cam_.width; }
  int GetHeight() const { return cam_.height; }

  // ---- sampling: num_p points that project inside the image (candidates that do not are redrawn)
  GeneratedData GetDistortedPointsPlanar(const int num_p = 100);   ///< points on one random plane, world z = 0
  GeneratedData GetDistortedPoints(const int num_p = 100);         ///< points anywhere in the frustum

  /// candidates thrown away because they fell outside the image (each still consumed random draws)
  long long Rejected() const { return rejected_; }

  /// Everything the sampling routines in data_generator.cpp need (they are free functions there).
  struct Camera {
    int width = 0, height = 0;
    Matrix3 K = Matrix3::Identity();
    DynamicVector dist = DynamicVector::Zero(5);
  };
  struct Draws {  // one engine, four distributions, consumed in the reference's order
    std::mt19937 engine{0};
    std::uniform_real_distribution<float> u, v, depth{0.2f, 1.0f}, pixel_noise{0.0f, 0.0f};
  };

 private:
  Camera cam_;
  Draws rnd_;
  long long rejected_ = 0;
};

}  // namespace calibrator

/*
 * cc_harness.h -- C ABI of the synthetic-input harness: the reference's calibrator::DataGenerator
 * (/root/reference/src/data_generator.hh:14-48, data_generator.cpp:10-184) without OpenCV. Used by
 * bench.py and the tests to create inputs; it is host code and not part of the solver path.
 * std::mt19937 seed 0 and std::uniform_real_distribution<float> as in the reference
 * (data_generator.hh:43-47); one cc_generator_planar call = one GetDistortedPointsPlanar call
 * (= one frame of the reference's test, src/test_calibrator.cpp:52-60).
 */
#ifndef CC_HARNESS_H
#define CC_HARNESS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct cc_generator cc_generator;

cc_generator* cc_generator_create(int32_t width, int32_t height);
void cc_generator_destroy(cc_generator* g);
void cc_generator_set_k(cc_generator* g, const float* K9);            /* row-major 3x3 */
void cc_generator_set_distortion(cc_generator* g, const float* dist5); /* k1 k2 p1 p2 k3 */
void cc_generator_set_noise(cc_generator* g, float noise_in_pixels);
/* fill uv[2*num_p], xyz[3*num_p]; return the number of points written */
int64_t cc_generator_planar(cc_generator* g, int32_t num_p, float* uv, float* xyz);
int64_t cc_generator_points(cc_generator* g, int32_t num_p, float* uv, float* xyz);

/* Synthetic rig scenario: the reference's own rig test (/root/reference/src/test_extrinsics_calibrator.cpp:48-134)
 * at any size. Camera 0 is the rig frame (the test freezes it), the others sit U(-0.03, 0.03) off in x and y;
 * every frame looks at the origin from U(0.3, 1.0)^3 and holds `pts` world points of U(-0.2, 0.2)^3 seen by every
 * camera. cam_T (what AddCameraTRig gets: 5 mm / 0.1 deg off for cameras >= 1), cam_T_true and frame_T (what
 * AddObservationFrame gets: 20 mm / 1 deg off) are 16 floats each, column-major like the JSON wire format
 * (extrinsics_calibrator.cpp:271); world_xyz 3 floats per world point (+-1 mm noise); observations in the test's
 * AddObservation order (frame, point, camera), normalised coordinates with +-2/500 noise: cams*frames*pts entries
 * of obs_cam / obs_world / obs_uv (2 floats). std::mt19937{seed}. */
void cc_rig_scenario(int32_t cams, int32_t frames, int32_t pts, uint32_t seed, float* cam_T, float* cam_T_true,
                     float* frame_T, float* world_xyz, uint32_t* obs_cam, uint64_t* obs_world, float* obs_uv);
/* Affine3f (16 floats, column-major) -> the fp64 quaternion (w x y z) and translation that
 * ExtrinsicsCalibrator::Optimize starts from (extrinsics_calibrator.cpp:116-130). */
void cc_affine_to_qt(const float* T16, double* q_wxyz, double* t_xyz);

#ifdef __cplusplus
}
#endif
#endif

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

#ifdef __cplusplus
}
#endif
#endif

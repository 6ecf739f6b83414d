/* Host build of the device's sin / cos restatement (csrc/d2d_sincos.h) for tests/test_sincos.py. */
#include <stdint.h>
#include "d2d_sincos.h"
void d2d_sin_host_array(const double *in, double *out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = d2d_sin(in[i]);
}
void d2d_cos_host_array(const double *in, double *out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = d2d_cos(in[i]);
}
/* the one-pass form the gaze stage uses (lanes that want sines and lanes that want cosines in one walk) */
void d2d_sin1_host_array(const double *in, double *out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = d2d_sin_or_cos(in[i], 0);
}
void d2d_cos1_host_array(const double *in, double *out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = d2d_sin_or_cos(in[i], 1);
}

/* Host build of the device's tan restatement (csrc/d2d_tan.h) for tests/test_tan.py. */
#include <stdint.h>
#include "d2d_tan.h"
void d2d_tan_host_array(const double *in, double *out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = d2d_tan(in[i]);
}

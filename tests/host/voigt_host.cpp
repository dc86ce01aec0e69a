// Host build of vamp_amd/csrc/voigt_math.hpp for CPU-side unit tests of the arithmetic
// (tests/test_voigt_math_host.py).  Test infrastructure only: the product never loads this.
#include "../../vamp_amd/csrc/voigt_math.hpp"
#include <cstdint>

extern "C" void voigt_H_host(int64_t n, const double* x, const double* y, double* out) {
    double dtab[vamp::DTAB_N];
    for (int64_t i = 0; i < n; ++i) {
        for (int k = 0; k < vamp::DTAB_N; ++k) dtab[k] = vamp::core_dtab_entry(k, y[i]);
        out[i] = vamp::voigt_H(fabs(x[i]), y[i], dtab, vamp::core_pole_factor(y[i]), vamp::core_hy(y[i]));
    }
}
extern "C" void humlicek_w4_host(int64_t n, const float* x, const float* y, float* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = vamp::humlicek_w4_re(x[i], y[i]);
}
extern "C" void cos_small_host(int64_t n, const double* a, double* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = vamp::cos_small(a[i]);
}
extern "C" void exp_taylor_host(int64_t n, const double* a, double* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = vamp::exp_taylor(a[i]);
}

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
// H(x, y) through the per-line Taylor table (x in [0, 8)); one table per distinct y would be the
// product's use, here it is rebuilt per point for simplicity
extern "C" void voigt_H_table_host(int64_t n, const double* x, const double* y, double* out) {
    double dtab[vamp::DTAB_N], tab[vamp::TAB_LINE];
    double ylast = -1.0;
    for (int64_t i = 0; i < n; ++i) {
        if (y[i] != ylast) {
            for (int k = 0; k < vamp::DTAB_N; ++k) dtab[k] = vamp::core_dtab_entry(k, y[i]);
            for (int r = 0; r < vamp::TAB_NI; ++r)
                vamp::taylor_table_row(r, y[i], dtab, vamp::core_pole_factor(y[i]), vamp::core_hy(y[i]), tab + r * vamp::TAB_NT);
            ylast = y[i];
        }
        out[i] = vamp::INV_SQRT_PI * vamp::taylor_table_eval(tab, fabs(x[i]));
    }
}
// sqrt(pi) w at the interval centres (real and imaginary part)
extern "C" void core_centre_host(int64_t n, const int* idx, const double* y, double* re, double* im) {
    double dtab[vamp::DTAB_N];
    for (int64_t i = 0; i < n; ++i) {
        for (int k = 0; k < vamp::DTAB_N; ++k) dtab[k] = vamp::core_dtab_entry(k, y[i]);
        vamp::core_centre(idx[i], y[i], dtab, vamp::core_pole_factor(y[i]), vamp::core_hy(y[i]), re[i], im[i]);
    }
}
// H(x, y) through the per-line fp32 Taylor rows of fp32 contexts (x in [0, 8))
extern "C" void voigt_H_table32_host(int64_t n, const double* x, const double* y, double* out) {
    double dtab[vamp::DTAB_N];
    float tab[vamp::TAB32_LINE];
    double ylast = -1.0;
    for (int64_t i = 0; i < n; ++i) {
        if (y[i] != ylast) {
            for (int k = 0; k < vamp::DTAB_N; ++k) dtab[k] = vamp::core_dtab_entry(k, y[i]);
            for (int r = 0; r < vamp::TAB_NI; ++r)
                vamp::taylor_table_row32(r, y[i], dtab, vamp::core_pole_factor(y[i]), vamp::core_hy(y[i]), tab + r * vamp::TAB32_NT);
            ylast = y[i];
        }
        out[i] = (double)vamp::taylor_table32_eval(tab, (float)fabs(x[i]));
    }
}

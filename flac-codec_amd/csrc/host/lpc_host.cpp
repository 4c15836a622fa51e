#include "lpc_host.h"

#include <cmath>
#include <cstring>

namespace flacenc {

namespace {
// f64::total_cmp as a key (the reference's min_by / max_by use total_cmp, encode.rs:3343, 3695)
inline long long total_key(double x) {
    long long b;
    std::memcpy(&b, &x, 8);
    b ^= static_cast<long long>(static_cast<unsigned long long>(b >> 63) >> 1);
    return b;
}

// Levinson-Durbin up to `upto` orders (encode.rs:3536-3580): c = coefficients of order `upto`,
// errs[i] = error of order i + 1.  Every product and sum is rounded separately (the build uses
// -ffp-contract=off), in the reference's order.
void levinson(const double *ac, uint32_t upto, double *c, double *errs) {
    double cn[32];
    double k = ac[1] / ac[0];
    c[0] = k;
    double err = ac[0] * (1.0 - k * k);
    errs[0] = err;
    for (uint32_t i = 1; i < upto; i++) {
        double s = -0.0;   // f64 `sum()` identity
        for (uint32_t j = 0; j < i; j++) {
            const double prod = ac[i - j] * c[j];
            s = s + prod;
        }
        const double q = ac[i + 1] - s;
        const double kk = q / err;
        for (uint32_t j = 0; j < i; j++) {
            const double t = kk * c[i - 1 - j];
            cn[j] = c[j] - t;
        }
        cn[i] = kk;
        for (uint32_t j = 0; j <= i; j++) c[j] = cn[j];
        err = err * (1.0 - kk * kk);
        errs[i] = err;
    }
}
}  // namespace

void lpc_from_autocorr(const double *ac, uint32_t L, uint32_t n, uint32_t bps, HostLpc *out) {
    std::memset(out, 0, sizeof *out);
    if (n <= L) {   // InsufficientLpcSamples, encode.rs:3300
        out->status = 1;
        return;
    }
    // precision table, encode.rs:3305-3315
    const uint32_t precision = n <= 192 ? 7 : n <= 384 ? 8 : n <= 576 ? 9 : n <= 1152 ? 10
                               : n <= 2304 ? 11 : n <= 4608 ? 12 : 13;
    double c[32], errs[32];
    levinson(ac, L, c, errs);
    // compute_best_order, encode.rs:3656-3702: ln() is the host libm's; bits per residual NOT clamped
    const double error_scale = 0.5 / static_cast<double>(n);
    const double denom = 2.0 * 0.693147180559945309417232121458176568;
    int best = -1;
    double best_bits = 0.0;
    for (uint32_t i = 0; i < L; i++) {
        if (!(errs[i] > 0.0)) break;   // take_while(error > 0.0)
        const uint32_t order = i + 1;
        const double header_bits = static_cast<double>(order * (bps + precision));
        const double bpr = std::log(errs[i] * error_scale) / denom;
        const double bits = std::fma(bpr, static_cast<double>(n - order), header_bits);
        if (best < 0 || total_key(bits) < total_key(best_bits)) {   // first minimum
            best = static_cast<int>(i);
            best_bits = bits;
        }
    }
    if (best < 0) {   // NoBestLpcOrder
        out->status = 2;
        return;
    }
    const uint32_t order = static_cast<uint32_t>(best) + 1;
    levinson(ac, order, c, errs);
    // quantize, encode.rs:3334-3401
    const int32_t max_coeff = (1 << (precision - 1)) - 1, min_coeff = -(1 << (precision - 1));
    double l = std::fabs(c[0]);
    for (uint32_t i = 1; i < order; i++) {
        const double a = std::fabs(c[i]);
        if (total_key(a) >= total_key(l)) l = a;   // max_by(total_cmp): the last maximum
    }
    if (!(l > 0.0)) {   // ZeroLpCoefficients (also NaN)
        out->status = 3;
        return;
    }
    // (precision - 1) - floor(log2(l)) as i32 - 1, capped at 15 (encode.rs:3360): `as i32` saturates
    const double fl_d = std::floor(std::log2(l));
    const int64_t fl = fl_d >= 2147483647.0 ? 2147483647ll : fl_d <= -2147483648.0 ? -2147483648ll
                                                                                  : static_cast<int64_t>(fl_d);
    int32_t sh = static_cast<int32_t>(static_cast<uint32_t>(precision - 1) - static_cast<uint32_t>(fl) - 1u);
    if (sh > 15) sh = 15;
    if (sh < -16) {   // LpNegativeShiftError
        out->status = 4;
        return;
    }
    double error = 0.0;
    const double scale = static_cast<double>(1 << (sh >= 0 ? sh : -sh));
    for (uint32_t i = 0; i < order; i++) {
        const double sum = sh >= 0 ? std::fma(c[i], scale, error) : (c[i] / scale) + error;
        const double rr = std::round(sum);
        int32_t q = (rr != rr) ? 0 : rr >= 2147483647.0 ? INT32_MAX : rr <= -2147483648.0 ? INT32_MIN
                                                                                         : static_cast<int32_t>(rr);
        q = q < min_coeff ? min_coeff : q > max_coeff ? max_coeff : q;
        error = sum - static_cast<double>(q);
        out->qlp[i] = q;
    }
    out->status = 0;
    out->order = static_cast<uint8_t>(order);
    out->precision = static_cast<uint8_t>(precision);
    out->shift = static_cast<uint8_t>(sh >= 0 ? sh : 0);
}

}  // namespace flacenc

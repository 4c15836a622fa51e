// lpc_host.h -- the LPC parameter chain on the HOST, with the host's libm (product code, not the
// oracle): used to re-decide the candidates whose two best order estimates the device found closer
// than its log() can be trusted to separate, so that the output is what the reference produces
// with this host's libm (encode.rs:3536-3580 lp_coefficients, 3656-3702 compute_best_order,
// 3334-3401 quantize).
#pragma once
#include <cstdint>

namespace flacenc {

struct HostLpc {
    int32_t status;   // 0 ok; 1 Insufficient samples, 2 NoBestLpcOrder, 3 ZeroLpCoefficients, 4 LpNegativeShiftError
    uint8_t order, precision, shift, pad;
    int32_t qlp[32];
};

// ac[0..=max_order]: the autocorrelation the device computed (exact summation order); n: block
// length; bps: the candidate's effective bits per sample (after wasted-bit removal)
void lpc_from_autocorr(const double *ac, uint32_t max_order, uint32_t n, uint32_t bps, HostLpc *out);

}  // namespace flacenc

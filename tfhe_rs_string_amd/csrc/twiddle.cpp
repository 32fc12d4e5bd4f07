// twiddle.cpp -- host-side generation of the f64 tables the kernels use (compiled with g++).
//
// Device sin/cos are not bit-identical to any host libm, and host libm versions differ between
// machines, so both tables are computed here in binary128 soft-float (libgcc, no libm) by Taylor
// series and rounded once to f64: the result is the correctly rounded value on every machine.
//
//   twist[j] = exp(+i * pi * j / N) evaluated at the f64 argument (j as f64) * (PI / (2n)), n = N/2
//              -- the reference's Twisties::new (tfhe/src/core_crypto/fft_impl/fft64/math/fft/mod.rs:58-69)
//   W[k]     = exp(-2 pi i k / n), exact angle -- twiddles of the engine's own radix-4 schedule
//              (the reference delegates this to concrete-fft)
#include <cstdint>

namespace {

void sincos_q(__float128 x, __float128 *s, __float128 *c)
{
    const __float128 x2 = x * x;
    __float128 term = x, sum = x;
    for (int i = 1; i < 40; i++) {
        term = -term * x2 / (__float128)((2 * i) * (2 * i + 1));
        sum += term;
    }
    *s = sum;
    term = 1;
    sum = 1;
    for (int i = 1; i < 40; i++) {
        term = -term * x2 / (__float128)((2 * i - 1) * (2 * i));
        sum += term;
    }
    *c = sum;
}

__float128 pi_q()
{
    // pi = 0x3.243F6A8885A3 08D313198A2E 03707344A409 3822299F31D0 ...
    const __float128 p1 = 0x3.243F6A8885A3p0;
    const __float128 p2 = 0x08D313198A2Ep-96;
    const __float128 p3 = 0x03707344A409p-144;
    const __float128 p4 = 0x3822299F31D0p-192;
    return ((p4 + p3) + p2) + p1;
}

// (sin, cos) of x in [0, pi/2], rounded to f64
void sincos_round(__float128 x, double *s, double *c)
{
    const __float128 pi = pi_q();
    __float128 sq, cq;
    if (x <= pi / 4)
        sincos_q(x, &sq, &cq);
    else
        sincos_q(pi / 2 - x, &cq, &sq);
    *s = (double)sq;
    *c = (double)cq;
}

} // namespace

// tw / w: n interleaved (re, im) pairs each
extern "C" __attribute__((visibility("hidden"))) void tfhe_hip_make_tables(uint32_t n, double *tw, double *w)
{
    const double unit = 3.14159265358979323846 / (2.0 * (double)n);
    for (uint32_t i = 0; i < n; i++) {
        const double arg = (double)i * unit;
        double s, c;
        sincos_round((__float128)arg, &s, &c);
        tw[2 * i] = c;
        tw[2 * i + 1] = s;
    }
    const __float128 pi = pi_q();
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t quad = (4 * k) / n, r = k - quad * (n / 4);
        double s, c, cr, sr;
        sincos_round(pi * (__float128)(2 * r) / (__float128)n, &s, &c);
        switch (quad) {
        case 0: cr = c; sr = s; break;
        case 1: cr = -s; sr = c; break;
        case 2: cr = -c; sr = -s; break;
        default: cr = s; sr = -c; break;
        }
        w[2 * k] = cr;
        w[2 * k + 1] = -sr;
    }
}

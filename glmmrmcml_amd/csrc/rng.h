// rng.h -- the RNG contract on the device (CPU twin:
// oracle/mcml_oracle.c).
//
// The reference draws momenta from R's rnorm (mhmcmc.h:48-51,62) and seeds its
// accept stream from std::random_device (:55): neither is reproducible.  The
// build keeps the accept stream's ENGINE and DISTRIBUTION exactly
// (std::minstd_rand + libstdc++ uniform_real_distribution<double>: two engine
// draws per uniform) and replaces rnorm by a counter-based generator:
// Philox4x32-10 -> 52-bit uniform -> Wichura AS241 inverse normal, written with
// +,*,/ ,sqrt and a table-free log built from the same exact operations, so the
// CPU and the GPU produce identical bits.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace mcml {

__host__ __device__ static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                     uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__host__ __device__ static inline double rng_u52(uint32_t w0, uint32_t w1)
{
    double a = (double)(w0 >> 6);
    double b = (double)(w1 >> 6);
    return (a * 67108864.0 + b + 0.5) * (1.0 / 4503599627370496.0);
}

__host__ __device__ static inline double rng_dlog(double x)
{
    union { double d; uint64_t u; } cv;
    cv.d = x;
    int e = (int)((cv.u >> 52) & 0x7ff) - 1023;
    cv.u = (cv.u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = cv.d;
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0);
    double s2 = s * s;
    double p = 1.0 / 27.0;
    p = p * s2 + 1.0 / 25.0;
    p = p * s2 + 1.0 / 23.0;
    p = p * s2 + 1.0 / 21.0;
    p = p * s2 + 1.0 / 19.0;
    p = p * s2 + 1.0 / 17.0;
    p = p * s2 + 1.0 / 15.0;
    p = p * s2 + 1.0 / 13.0;
    p = p * s2 + 1.0 / 11.0;
    p = p * s2 + 1.0 / 9.0;
    p = p * s2 + 1.0 / 7.0;
    p = p * s2 + 1.0 / 5.0;
    p = p * s2 + 1.0 / 3.0;
    p = p * s2;
    double ed = (double)e;
    double hi = ed * 0.693147180369123816490;
    double lo = ed * 1.90821492927058770002e-10;
    double t = 2.0 * s;
    return hi + (t + (t * p + lo));
}

__host__ __device__ static inline double rng_ppnd16(double p)
{
    double q = p - 0.5, r, val;
    if (fabs(q) <= 0.425) {
        r = 0.180625 - q * q;
        double num = 2.5090809287301226727e+3;
        num = num * r + 3.3430575583588128105e+4;
        num = num * r + 6.7265770927008700853e+4;
        num = num * r + 4.5921953931549871457e+4;
        num = num * r + 1.3731693765509461125e+4;
        num = num * r + 1.9715909503065514427e+3;
        num = num * r + 1.3314166789178437745e+2;
        num = num * r + 3.3871328727963666080e0;
        double den = 5.2264952788528545610e+3;
        den = den * r + 2.8729085735721942674e+4;
        den = den * r + 3.9307895800092710610e+4;
        den = den * r + 2.1213794301586595867e+4;
        den = den * r + 5.3941960214247511077e+3;
        den = den * r + 6.8718700749205790830e+2;
        den = den * r + 4.2313330701600911252e+1;
        den = den * r + 1.0;
        return q * num / den;
    }
    r = (q < 0.0) ? p : 1.0 - p;
    r = sqrt(-rng_dlog(r));
    if (r <= 5.0) {
        r = r - 1.6;
        double num = 7.74545014278341407640e-4;
        num = num * r + 2.27238449892691845833e-2;
        num = num * r + 2.41780725177450611770e-1;
        num = num * r + 1.27045825245236838258e0;
        num = num * r + 3.64784832476320460504e0;
        num = num * r + 5.76949722146069140550e0;
        num = num * r + 4.63033784615654529590e0;
        num = num * r + 1.42343711074968357734e0;
        double den = 1.05075007164441684324e-9;
        den = den * r + 5.47593808499534494600e-4;
        den = den * r + 1.51986665636164571966e-2;
        den = den * r + 1.48103976427480074590e-1;
        den = den * r + 6.89767334985100004550e-1;
        den = den * r + 1.67638483018380384940e0;
        den = den * r + 2.05319162663775882187e0;
        den = den * r + 1.0;
        val = num / den;
    } else {
        r = r - 5.0;
        double num = 2.01033439929228813265e-7;
        num = num * r + 2.71155556874348757815e-5;
        num = num * r + 1.24266094738807843860e-3;
        num = num * r + 2.65321895265761230930e-2;
        num = num * r + 2.96560571828504891230e-1;
        num = num * r + 1.78482653991729133580e0;
        num = num * r + 5.46378491116411436990e0;
        num = num * r + 6.65790464350110377720e0;
        double den = 2.04426310338993978564e-15;
        den = den * r + 1.42151175831644588870e-7;
        den = den * r + 1.84631831751005468180e-5;
        den = den * r + 7.86869131145613259100e-4;
        den = den * r + 1.48753612908506148525e-2;
        den = den * r + 1.36929880922735805310e-1;
        den = den * r + 5.99832206555887937690e-1;
        den = den * r + 1.0;
        val = num / den;
    }
    return (q < 0.0) ? -val : val;
}

// standard normal addressed by (seed; element, global chain id, proposal, tag)
// tag: 0 initial state, 2 momentum; + 16 * MCML iteration
__host__ __device__ static inline double rng_normal(uint64_t seed, uint32_t elem, uint32_t chain,
                                                    uint32_t prop, uint32_t tag)
{
    uint32_t o[4];
    philox4x32_10(elem, chain, prop, tag, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    return rng_ppnd16(rng_u52(o[0], o[1]));
}

// uniform on (0, 1) with the same addressing (the No-U-Turn sampler's initial values, nuts.h)
__host__ __device__ static inline double rng_uniform(uint64_t seed, uint32_t elem, uint32_t chain,
                                                     uint32_t prop, uint32_t tag)
{
    uint32_t o[4];
    philox4x32_10(elem, chain, prop, tag, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    return rng_u52(o[0], o[1]);
}

// std::minstd_rand (mhmcmc.h:27)
__host__ __device__ static inline uint32_t minstd_next(uint32_t& x)
{
    x = (uint32_t)(((uint64_t)x * 48271u) % 2147483647u);
    return x;
}

// libstdc++ uniform_real_distribution<double>(0,1)(minstd_rand)  (mhmcmc.h:28,85)
__host__ __device__ static inline double minstd_canonical(uint32_t& x)
{
    const double R = 2147483646.0;
    double sum = 0.0, tmp = 1.0;
    sum += (double)(minstd_next(x) - 1u) * tmp;
    tmp *= R;
    sum += (double)(minstd_next(x) - 1u) * tmp;
    tmp *= R;
    double ret = sum / tmp;
    if (ret >= 1.0) ret = 0.99999999999999988897769753748;   // nextafter(1,0)
    return ret;
}

__host__ __device__ static inline uint32_t chain_minstd_seed(uint64_t seed, uint32_t chain, uint32_t iter)
{
    uint32_t o[4];
    philox4x32_10(0u, chain, iter, 3u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    uint32_t s = o[0] % 2147483647u;
    return s == 0u ? 1u : s;
}

}  // namespace mcml

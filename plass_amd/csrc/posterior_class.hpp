// plasship: the posterior probability behind CompareNuclResultByScore (src/assembler/nuclassembleresult.cpp:36-58) as the nucleotide
// extension kernels evaluate it.  Product code (assemble.hip); plain C++ as well, so that tests/test_host.py can compile it with g++ and
// compare it with the reference's formula under the host's libm without a GPU.
#pragma once
#include <cmath>
#if defined(__HIPCC__)
#define PLASSHIP_PC_HD __host__ __device__ __forceinline__
#else
#define PLASSHIP_PC_HD static inline
#endif
namespace plasship {
// lgamma of a positive integer: log((n - 1)!) from the exact product below 16, Stirling's series above (truncation < 2e-14)
PLASSHIP_PC_HD double lgammaInt(unsigned n) {
    if (n < 16) { double f = 1.0; for (unsigned i = 2; i < n; i++) f *= (double) i; return log(f); }
    const double x = (double) n, r = 1.0 / x, r2 = r * r;
    return (x - 0.5) * log(x) - x + 0.91893853320467274178 +
           r * (1.0 / 12.0 - r2 * (1.0 / 360.0 - r2 * (1.0 / 1260.0 - r2 * (1.0 / 1680.0 - r2 * (1.0 / 1188.0)))));
}
// Round 4: the reference evaluates p = sum_{i < alpha2} exp(log_r_i + log_c) with four lgamma and, per term, one exp and five log in
// double.  Only the CLASS of p is used (p < 0.45, p > 0.55, between), and a p within nuclPosteriorBand of a threshold is decided by
// the host's libm anyway (assemble.hip, nuclPosteriorClassDev), so the device may take any route that is accurate to well below that
// band: the terms are t_0 = exp(log_c), t_{i+1} = t_i (alpha1 + i)(beta2 + i) / ((i + 1)(i + alpha1 + beta1 + beta2)) — one division per
// term instead of six transcendentals (rescaled when they grow: the sequence rises, then falls) — and the arguments of lgamma are
// integers.  The contigs of the late nucleotide iterations of configs[4] overlap in thousands of columns with dozens of mismatches:
// such tuples miss the memo (CMP_LEN, CMP_MM), and the heap of a query with 200 hits asks for thousands of them.
PLASSHIP_PC_HD double nuclPosteriorP(unsigned alpha1, unsigned beta1, unsigned alpha2, unsigned beta2) {
    const double log_c = (lgammaInt(beta1 + beta2) + lgammaInt(alpha1 + beta1)) - (lgammaInt(alpha1 + beta1 + beta2) + lgammaInt(beta1));
    double t = 1.0, sum = 0.0, logScale = 0.0;
    const double S = (double) alpha1 + (double) beta1 + (double) beta2;
    for (unsigned idx = 0; idx < alpha2; idx++) {
        sum += t;
        const double i = (double) idx;
        t *= (((double) alpha1 + i) * ((double) beta2 + i)) / ((i + 1.0) * (i + S));
        if (t > 1e200) { t *= 1e-200; sum *= 1e-200; logScale += 460.51701859880913680; }      // 200 ln 10
    }
    return sum > 0.0 ? exp(log_c + logScale + log(sum)) : 0.0;
}
// half-width of the band around 0.45 / 0.55 inside which the class is taken from the host-evaluated table: log_c is a difference of
// numbers of the size of the overlap lengths times their logarithm, so its rounding error — here and in the host's lgamma — grows
// with them, and the band does too
PLASSHIP_PC_HD double nuclPosteriorBand(unsigned alpha1, unsigned beta1, unsigned alpha2, unsigned beta2) {
    return 1e-9 + 1e-13 * ((double) alpha1 + (double) beta1 + (double) beta2 + (double) alpha2);
}
}  // namespace plasship

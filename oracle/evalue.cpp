// ORACLE (test infrastructure): constant tables + E-value / bit-score arithmetic, restated.
//   tables:   oracle/ref_tables.h (DATA captured from the unmodified reference, see tools/make_tables.sh)
//   evaluer:  mm/alignment/EvalueComputation.h:18-45,150-160
//   ALP:      lib/mmseqs/lib/alp/sls_alignment_evaluer.hpp:154-162 (evaluePerArea, bitScore)
//             lib/mmseqs/lib/alp/sls_alignment_evaluer.cpp:989-1029 (area)
//             lib/mmseqs/lib/alp/sls_pvalues.cpp:366-545 (get_appr_tail_prob_with_cov_without_errors)
//             lib/mmseqs/lib/alp/sls_basic.hpp:195-198 (normal_probability = 0.5*erfc(-sqrt(.5)x))
// Note on floating point: the reference is built -O3 -march=native (FMA available, GCC default
// contraction); `lambda*score - logK` and `logK + bits*ln2` are single multiply-adds there, which is
// what std::fma reproduces bit-exactly (checked against REF_KAT_* below by tests/test_oracle_kat.py).
#include "oracle.hpp"
#include "ref_tables.h"
#include <cmath>

namespace oracle {

const signed char *asciiSubMat(bool nucl) { return nucl ? REF_NUC_ASCII_SUBMAT : REF_AA_ASCII_SUBMAT; }

const unsigned char *aa2num(bool nucl, int alphabetSize) {
    if (nucl) return REF_NUC_AA2NUM;
    return (alphabetSize == 21) ? REF_AA21_AA2NUM : REF_AA13_AA2NUM;
}

Evaluer::Evaluer(bool nucl, uint64_t dbResidues)
    : g(nucl ? REF_NUC_GAPLESS_GUMBEL : REF_AA_GAPLESS_GUMBEL), logK(std::log(g[1])), dbRes((double) dbResidues) {}

// gapped evaluer on the nucleotide matrix (proteinaln2nucl.cpp:54-58); only the penguin workflow's 5/2 penalties have
// captured parameters (ALP simulates them at start-up with a fixed seed, EvalueComputation.h:46-51,92-100)
Evaluer Evaluer::nuclGapped(int gapOpen, int gapExtend, uint64_t dbResidues, bool *ok) {
    Evaluer e(true, dbResidues);
    const bool have = (gapOpen == 5 && gapExtend == 2);
    if (ok) *ok = have;
    if (have) { e.g = REF_NUC_GAPPED_5_2_GUMBEL; e.logK = std::log(e.g[1]); }
    return e;
}

// evaluer.bitScore(score, logK) = (lambda*score - logK)/log(2.0)
double Evaluer::bitScore(double score) const { return std::fma(g[0], score, -logK) / std::log(2.0); }

// computeRawScoreFromBitScore = (logK + bitScore*log(2.0)) / lambda
double Evaluer::rawFromBit(double bits) const { return std::fma(bits, std::log(2.0), logK) / g[0]; }

static inline double normalProbability(double x) { return 0.5 * std::erfc(-std::sqrt(0.5) * x); }

// area(score, seqlen1 = query length, seqlen2 = db residues): called as
// get_appr_tail_prob_with_cov_without_errors(par, blast, y=score, m=seqlen2, n=seqlen1, …)
double Evaluer::area(double y, double seqLen) const {
    const double pi = 3.1415926535897932384626433832795;
    const double const_val = 1 / std::sqrt(2.0 * pi);
    const double ai = g[2], bi = g[3], alphai = g[4], betai = g[5];
    const double aj = g[6], bj = g[7], alphaj = g[8], betaj = g[9];
    const double sigma = g[10], tau = g[11], vi_thr = g[12], vj_thr = g[13], c_thr = g[14];
    const double m_ = dbRes, n_ = seqLen;

    double m_li_y = m_ - (ai * y + bi);
    double vi_y = std::fmax(vi_thr, alphai * y + betai);
    double sqrt_vi_y = std::sqrt(vi_y);
    double m_F = (sqrt_vi_y == 0.0) ? 1e100 : m_li_y / sqrt_vi_y;
    double P_m_F = normalProbability(m_F);
    double E_m_F = -const_val * std::exp(-0.5 * m_F * m_F);
    double p1 = m_li_y * P_m_F - sqrt_vi_y * E_m_F;

    double n_lj_y = n_ - (aj * y + bj);
    double vj_y = std::fmax(vj_thr, alphaj * y + betaj);
    double sqrt_vj_y = std::sqrt(vj_y);
    double n_F = (sqrt_vj_y == 0.0) ? 1e100 : n_lj_y / sqrt_vj_y;
    double P_n_F = normalProbability(n_F);
    double E_n_F = -const_val * std::exp(-0.5 * n_F * n_F);
    double p2 = n_lj_y * P_n_F - sqrt_vj_y * E_n_F;

    double c_y = std::fmax(c_thr, sigma * y + tau);
    return p1 * p2 + c_y * P_m_F * P_n_F;
}

double Evaluer::evalue(double score, double seqLen) const {
    const double epa = g[1] * std::exp(-g[0] * score);   // evaluePerArea
    return epa * area(score, seqLen);
}

}  // namespace oracle

// ref_toms917_shim.cpp -- C binding over the REFERENCE's own Wright-omega
// (modules/toms917/toms917.cpp, compiled from /root/reference where it lies; see Makefile
// target `ref`).  TEST INFRASTRUCTURE ONLY.  This file contains no algorithm: it only
// exposes the reference's `std::complex<double> wrightomega(std::complex<double>)`
// (toms917.hpp:5) and `wrightomega_ext` (toms917.hpp:6-7) with C linkage so that tests can
// pin oracle/wdf_oracle.c against the real thing.  The call shape mirrors
// Toms917DiodePairT::tomsOmega (plugin/src/dsp/diode_clipper/Toms917DiodePair.h:64-67).
#include <complex>
#include <cstdint>
#include <toms917.hpp>

extern "C" {

double ref_toms917_omega_real(double x)
{
    return std::real(wrightomega(std::complex<double>(x)));
}

void ref_toms917_omega_vec(const double* x, double* w, int64_t n)
{
    for (int64_t i = 0; i < n; ++i)
        w[i] = std::real(wrightomega(std::complex<double>(x[i])));
}

// returns omega, and the penultimate residual r (real part) so the tests can see whether
// the reference ran one or two FSC iterations on the same input
double ref_toms917_omega_ext(double x, double* e_out, double* r_out)
{
    std::complex<double> w, e, r, cond;
    wrightomega_ext(std::complex<double>(x), w, e, r, cond);
    if (e_out) *e_out = std::real(e);
    if (r_out) *r_out = std::real(r);
    return std::real(w);
}

// The diode-pair root exactly as the reference C++ evaluates it
// (Toms917DiodePair.h:28-59): float state, omega through complex<double> and back.
float ref_toms917_diode_pair(float a, float R, float Is, float Vt, float nDiodes)
{
    const float VtN = nDiodes * Vt;                 // :31
    const float twoVt = 2.0f * VtN;                 // :32
    const float oneOverVt = 1.0f / VtN;             // :33
    const float R_Is = R * Is;                      // :39
    const float R_Is_overVt = R_Is * oneOverVt;     // :40
    const float logR_Is_overVt = std::log(R_Is_overVt);   // :41
    const float lambda = (float)((0.0f < a) - (a < 0.0f));   // chowdsp::signum, :54
    const float lambda_a_over_vt = lambda * a * oneOverVt;   // :55
    auto om = [](float x) { return (float)std::real(wrightomega(std::complex<double>((double)x))); };
    return a - twoVt * lambda * (om(logR_Is_overVt + lambda_a_over_vt) - om(logR_Is_overVt - lambda_a_over_vt));   // :56
}

}

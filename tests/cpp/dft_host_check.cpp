// CPU check of the register-resident butterflies (ni-slam_amd/csrc/kcc_fft2.h dft_run: Good-Thomas prime-factor splits and
// Cooley-Tukey splits over the base radices of kcc_fft.h) against a direct double-precision DFT, both directions, for every
// radix a plan uses.  Output k must sit in register dft_pos<R>(k).  Built with g++ against tests/cpp/hipstub.
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>

#include "kcc_fft2.h"

using namespace kcc;

template <int R, bool INV> double check() {
    cf2 v[R]; std::complex<double> in[R];
    for (int i = 0; i < R; ++i) { v[i] = mk2((float)sin(1.0 + 3.7 * i), (float)cos(0.3 + 1.9 * i * i)); in[i] = { v[i].x, v[i].y }; }
    dft_run<R, INV>(v);
    double worst = 0;
    bool used[R] = {};
    for (int k = 0; k < R; ++k) {
        std::complex<double> acc = 0;
        for (int n = 0; n < R; ++n) acc += in[n] * std::polar(1.0, (INV ? 2.0 : -2.0) * M_PI * (double)((n * k) % R) / R);
        const int pos = dft_pos<R>(k);
        if (pos < 0 || pos >= R || used[pos]) return 1e9;       // dft_pos must be a permutation
        used[pos] = true;
        worst = std::max(worst, std::abs(acc - std::complex<double>(v[pos].x, v[pos].y)));
    }
    printf("R=%d inv=%d pfa=%d worst=%.3g\n", R, (int)INV, (int)is_pfa(R), worst);
    return worst;
}
template <class P> void plan_radices(double& w);
int main() {
    double w = 0;
#define T(R) w = std::max(w, check<R, false>()); w = std::max(w, check<R, true>());
    T(2) T(3) T(4) T(5) T(7) T(8) T(6) T(9) T(10) T(12) T(14) T(15) T(16) T(18) T(20) T(24) T(25) T(30)
    printf("worst %.3g\n", w);
    return w < 4e-6 ? 0 : 1;          // R <= 30 points of magnitude <= 1: a few float32 ulps of sqrt(R)
}

// toolchain_pins.cpp -- the two C++ standard-library semantics the oracle assumes for the reference's Eigen expressions, checked
// against THIS image's libstdc++ / glibc (the reference's own toolchain family: g++ -O3, CMakeLists.txt:27-33).  oracle/RECALLED.md
// rows 16 and 18.  Prints one line of numbers; tests/test_toolchain_pins.py asserts on them.
//   1. Array::pow(int) -> std::pow(float, int): the C++11 overload promotes to double and returns double (correlation_flow.cc:213);
//      the oracle evaluates (float)pow((double)x, (double)p), the HIP kernels (float)((double)x * x * x) for p = 3.
//   2. complex<float> division (correlation_flow.cc:171, T / (Kzz + lambda)): libstdc++'s operator/ against the textbook formula
//      the oracle and the HIP ridge solve use, ((ac + bd) / |z|^2, (bc - ad) / |z|^2).
//   3. fft_result.abs() (correlation_flow.cc:92): std::abs(complex<float>) against sqrtf(re^2 + im^2).
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <type_traits>

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint64_t next() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static inline float unit() { return (float)((next() >> 40) * (1.0 / 16777216.0)); }
static inline int ulps(float a, float b) {
    int32_t x, y; memcpy(&x, &a, 4); memcpy(&y, &b, 4);
    if (x < 0) x = (int32_t)0x80000000 - x;
    if (y < 0) y = (int32_t)0x80000000 - y;
    const long long d = (long long)x - (long long)y;
    return (int)(d < 0 ? -d : d);
}

int main() {
    static_assert(std::is_same<decltype(std::pow(1.0f, 3)), double>::value, "std::pow(float, int) must return double (C++11 promotion)");
    // 0. cvRound (RECALLED row 6: lrint / cvtsd2si under the default rounding mode) rounds halves to even -- what the oracle's
    //    and the host tables' lrint / lrintf do in this libc
    if (lrint(0.5) != 0 || lrint(1.5) != 2 || lrint(2.5) != 2 || lrint(-0.5) != 0 || lrint(-1.5) != -2 || lrintf(3.5f) != 4 || lrintf(4.5f) != 4) {
        printf("{\"lrint_half_even\": false}\n"); return 1;
    }
    // 1. cubes over the range of (xz + offset): correlation values from ~1e-1 to ~1e5, both signs
    long n_pow = 0, pow_vs_oracle = 0, pow_vs_cube = 0, float_cube_differs = 0;
    for (int i = 0; i < 6000000; ++i) {
        const float mag = std::exp(unit() * 14.0f - 2.3f);            // e^-2.3 .. e^11.7
        const float x = (next() & 1) ? mag : -mag;
        const float lib = (float)std::pow(x, 3);                        // what Eigen's pow(int) evaluates
        const float ora = (float)pow((double)x, 3.0);                   // oracle/kcc_oracle.c pow_int
        const float hip = (float)((double)x * (double)x * (double)x);   // kernel_value<KT_POLY3>
        const float f32 = x * x * x;                                    // (what a float-only cube would give: NOT the semantics)
        ++n_pow; pow_vs_oracle += lib != ora; pow_vs_cube += lib != hip; float_cube_differs += lib != f32;
    }
    // 2. T / (Kzz + lambda): numerators +-1, denominators from tiny to the DC bin's ~1e5, any phase
    long n_div = 0; int worst = 0; long differ = 0;
    for (int i = 0; i < 6000000; ++i) {
        const float mag = std::exp(unit() * 16.0f - 4.0f);
        const float ph = unit() * 6.2831853f;
        const std::complex<float> z(mag * std::cos(ph) + 0.1f, mag * std::sin(ph));
        const std::complex<float> t((next() & 1) ? 1.0f : -1.0f, 0.0f);
        const std::complex<float> lib = t / z;
        const float d = z.real() * z.real() + z.imag() * z.imag();
        const float re = (t.real() * z.real() + t.imag() * z.imag()) / d, im = (t.imag() * z.real() - t.real() * z.imag()) / d;
        const int u = ulps(lib.real(), re) > ulps(lib.imag(), im) ? ulps(lib.real(), re) : ulps(lib.imag(), im);
        // (compare only where the component is not a cancellation residue: both forms agree to ulps there)
        const float big = std::fabs(re) > std::fabs(im) ? std::fabs(re) : std::fabs(im);
        const float small = std::fabs(re) > std::fabs(im) ? std::fabs(im) : std::fabs(re);
        if (small > 1e-3f * big) { ++n_div; differ += u != 0; if (u > worst) worst = u; }
    }
    // 3. |F| (correlation_flow.cc:92, fft_result.abs()): std::abs(complex<float>) (hypot) against sqrtf(re^2 + im^2), what the oracle
    //    and the HIP kernel evaluate
    long n_abs = 0, abs_differ = 0; int abs_worst = 0;
    for (int i = 0; i < 6000000; ++i) {
        const float mag = std::exp(unit() * 16.0f - 4.0f), ph = unit() * 6.2831853f;
        const std::complex<float> z(mag * std::cos(ph), mag * std::sin(ph));
        const float lib = std::abs(z), mine = sqrtf(z.real() * z.real() + z.imag() * z.imag());
        const int u = ulps(lib, mine);
        ++n_abs; abs_differ += u != 0; if (u > abs_worst) abs_worst = u;
    }
    printf("{\"abs_samples\": %ld, \"abs_differ\": %ld, \"abs_worst_ulps\": %d, ", n_abs, abs_differ, abs_worst);
    printf("\"pow_samples\": %ld, \"libpow_ne_oracle\": %ld, \"libpow_ne_double_cube\": %ld, \"libpow_ne_float_cube\": %ld, "
           "\"div_samples\": %ld, \"div_differ\": %ld, \"div_worst_ulps\": %d}\n",
           n_pow, pow_vs_oracle, pow_vs_cube, float_cube_differs, n_div, differ, worst);
    return 0;
}

// kcc_fft.h -- complex helpers and the base-radix butterflies {2,3,4,5,7,8} (in registers) that the
// register-resident FFT engine of kcc_fft2.h composes into large radices.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

namespace kcc {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a * conj(b)
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
// multiply by -i (forward) or +i (inverse)
template <bool INV> __device__ __forceinline__ float2 mul_mi(float2 a) { return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x); }

__host__ __device__ constexpr int pick_radix(int rem) {
    return rem % 8 == 0 ? 8 : rem % 4 == 0 ? 4 : rem % 2 == 0 ? 2 : rem % 3 == 0 ? 3 : rem % 5 == 0 ? 5 : rem % 7 == 0 ? 7 : rem;
}
// every prime factor must be in {2,3,5,7}
__host__ __device__ constexpr bool fft_len_ok(int n) {
    while (n % 2 == 0) n /= 2;
    while (n % 3 == 0) n /= 3;
    while (n % 5 == 0) n /= 5;
    while (n % 7 == 0) n /= 7;
    return n == 1;
}

template <int R, bool INV> struct Radix;

template <bool INV> struct Radix<2, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[2]) {
        float2 a = v[0], b = v[1];
        v[0] = cadd(a, b); v[1] = csub(a, b);
    }
};
template <bool INV> struct Radix<3, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[3]) {
        const float S = 0.86602540378443864676f;
        float2 t = cadd(v[1], v[2]);
        float2 d = csub(v[1], v[2]);
        float2 m = make_float2(v[0].x - 0.5f * t.x, v[0].y - 0.5f * t.y);
        float2 r = mul_mi<INV>(make_float2(S * d.x, S * d.y));   // -i*S*d (fwd)
        v[0] = cadd(v[0], t);
        v[1] = cadd(m, r);
        v[2] = csub(m, r);
    }
};
template <bool INV> struct Radix<4, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[4]) {
        float2 a0 = cadd(v[0], v[2]), a1 = csub(v[0], v[2]);
        float2 b0 = cadd(v[1], v[3]), b1 = mul_mi<INV>(csub(v[1], v[3]));
        v[0] = cadd(a0, b0); v[2] = csub(a0, b0);
        v[1] = cadd(a1, b1); v[3] = csub(a1, b1);
    }
};
template <bool INV> struct Radix<5, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[5]) {
        const float C1 = 0.30901699437494742410f, C2 = -0.80901699437494742410f;
        const float S1 = 0.95105651629515357212f, S2 = 0.58778525229247312917f;
        float2 a1 = cadd(v[1], v[4]), a2 = cadd(v[2], v[3]);
        float2 b1 = csub(v[1], v[4]), b2 = csub(v[2], v[3]);
        float2 r1 = make_float2(v[0].x + C1 * a1.x + C2 * a2.x, v[0].y + C1 * a1.y + C2 * a2.y);
        float2 r2 = make_float2(v[0].x + C2 * a1.x + C1 * a2.x, v[0].y + C2 * a1.y + C1 * a2.y);
        float2 i1 = mul_mi<INV>(make_float2(S1 * b1.x + S2 * b2.x, S1 * b1.y + S2 * b2.y));
        float2 i2 = mul_mi<INV>(make_float2(S2 * b1.x - S1 * b2.x, S2 * b1.y - S1 * b2.y));
        v[0] = cadd(v[0], cadd(a1, a2));
        v[1] = cadd(r1, i1); v[4] = csub(r1, i1);
        v[2] = cadd(r2, i2); v[3] = csub(r2, i2);
    }
};
template <bool INV> struct Radix<7, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[7]) {
        const float C[3] = { 0.62348980185873353053f, -0.22252093395631440429f, -0.90096886790241912624f };
        const float S[3] = { 0.78183148246802980871f, 0.97492791218182360702f, 0.43388373911755812048f };
        float2 a[3], b[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) { a[q] = cadd(v[q + 1], v[6 - q]); b[q] = csub(v[q + 1], v[6 - q]); }
        float2 y[7];
        y[0] = cadd(v[0], cadd(a[0], cadd(a[1], a[2])));
#pragma unroll
        for (int k = 1; k <= 3; ++k) {
            float2 re = v[0], im = make_float2(0.f, 0.f);
#pragma unroll
            for (int q = 1; q <= 3; ++q) {
                int t = (k * q) % 7;                 // cos(2*pi*t/7), sin(2*pi*t/7)
                float c = t <= 3 ? C[t - 1] : C[6 - t];
                float s = t <= 3 ? S[t - 1] : -S[6 - t];
                re.x += c * a[q - 1].x; re.y += c * a[q - 1].y;
                im.x += s * b[q - 1].x; im.y += s * b[q - 1].y;
            }
            float2 r = mul_mi<INV>(im);
            y[k] = cadd(re, r); y[7 - k] = csub(re, r);
        }
#pragma unroll
        for (int q = 0; q < 7; ++q) v[q] = y[q];
    }
};
template <bool INV> struct Radix<8, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[8]) {
        const float H = 0.70710678118654752440f;
        float2 e[4] = { v[0], v[2], v[4], v[6] };
        float2 o[4] = { v[1], v[3], v[5], v[7] };
        Radix<4, INV>::run(e);
        Radix<4, INV>::run(o);
        // W8^1 = (1 - i)/sqrt2 (fwd), W8^2 = -i, W8^3 = (-1 - i)/sqrt2 ; conj for inverse
        float2 o1 = INV ? make_float2(H * (o[1].x - o[1].y), H * (o[1].x + o[1].y))
                        : make_float2(H * (o[1].x + o[1].y), H * (o[1].y - o[1].x));
        float2 o2 = mul_mi<INV>(o[2]);
        float2 o3 = INV ? make_float2(-H * (o[3].x + o[3].y), H * (o[3].x - o[3].y))
                        : make_float2(H * (o[3].y - o[3].x), -H * (o[3].x + o[3].y));
        v[0] = cadd(e[0], o[0]); v[4] = csub(e[0], o[0]);
        v[1] = cadd(e[1], o1);   v[5] = csub(e[1], o1);
        v[2] = cadd(e[2], o2);   v[6] = csub(e[2], o2);
        v[3] = cadd(e[3], o3);   v[7] = csub(e[3], o3);
    }
};

}  // namespace kcc

// kcc_fft.h -- complex arithmetic of the FFT engine and the base-radix butterflies {2,3,4,5,7,8} (in registers) that
// kcc_fft2.h composes into large radices.  gfx950 only (plus a plain-C++ rendering of the same operations for the host check
// tests/cpp/dft_host_check.cpp).
//
// A complex number is ONE 64-bit register pair (cf2).  On gfx950 every operation below is a single packed-FP32 VALU
// instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) on that pair, with the swaps and sign flips complex arithmetic
// needs -- multiplication by +-i, conjugation, the cross terms of a complex product -- expressed through the instruction's
// op_sel / op_sel_hi (which half of a source feeds the low / high lane) and neg_lo / neg_hi modifiers instead of through
// extra moves: a complex add or subtract is 1 instruction instead of 2, "a +- i b" is 1 instead of 2, a complex product is 2
// instead of 4.  The compiler does not find these forms by itself (SLP-packed code paid the gain back in v_mov / v_xor:
// DESIGN.md 4.2), hence the inline assembly.  A packed instruction issues in 1.57x the time of a scalar one for twice the
// work (tools/ubench/pkrate.hip).  -DKCC_PK=0 builds the same algebra from scalar operations.
#pragma once
#include <hip/hip_runtime.h>

#ifndef KCC_PK
#define KCC_PK 1
#endif

namespace kcc {

#if defined(__clang__)
typedef float cf2 __attribute__((ext_vector_type(2)));
#else
struct cf2 { float x, y; };                                  // (g++ host check)
#endif
__host__ __device__ __forceinline__ cf2 mk2(float x, float y) { cf2 v; v.x = x; v.y = y; return v; }

#if defined(__HIP_DEVICE_COMPILE__) && KCC_PK
#define KCC_PK_ASM 1
#else
#define KCC_PK_ASM 0
#endif

// ---- one-instruction primitives -----------------------------------------------------------------------------------
#if KCC_PK_ASM
#define KCC_PK2(name, text)                                                                                     \
    __device__ __forceinline__ cf2 name(cf2 a, cf2 b) { cf2 d; asm(text : "=v"(d) : "v"(a), "v"(b)); return d; }
KCC_PK2(cadd,         "v_pk_add_f32 %0, %1, %2")                                                               // a + b
KCC_PK2(csub,         "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]")                                     // a - b
KCC_PK2(add_ib,       "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")                     // a + i b = (a.x - b.y, a.y + b.x)
KCC_PK2(sub_ib,       "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")                     // a - i b = (a.x + b.y, a.y - b.x)
KCC_PK2(cadd_conj,    "v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]")                                                  // a + conj(b)
KCC_PK2(csub_conj,    "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]")                                                  // a - conj(b)
KCC_PK2(conj_add_ib,  "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,1]")        // conj(a + i b) = (a.x - b.y, -a.y - b.x)
KCC_PK2(conj_sub_ib,  "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]")                     // conj(a - i b) = (a.x + b.y, -a.y + b.x)
KCC_PK2(pmul,         "v_pk_mul_f32 %0, %1, %2")                                                               // componentwise product
#undef KCC_PK2
__device__ __forceinline__ cf2 pfma(cf2 a, cf2 b, cf2 c) { cf2 d; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
// a * w and a * conj(w): the real part of a times w, then the imaginary part's cross terms fused on top
__device__ __forceinline__ cf2 cmul(cf2 a, cf2 w) {
    cf2 t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));                                    // (a.x w.x, a.x w.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(a), "v"(w), "v"(t));     // (-a.y w.y + t.x, a.y w.x + t.y)
    return d;
}
__device__ __forceinline__ cf2 cmulc(cf2 a, cf2 w) {
    cf2 t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));                                    // (a.x w.x, a.x w.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,0,1]" : "=v"(d) : "v"(a), "v"(w), "v"(t));     // (a.y w.y + t.x, a.y w.x - t.y)
    return d;
}
// real scalar s (a compile-time constant or a uniform value: lives in a scalar register pair) times a; a * s + c
__device__ __forceinline__ cf2 scale(cf2 a, float s) { cf2 d; const cf2 sv = mk2(s, s); asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "s"(sv)); return d; }
__device__ __forceinline__ cf2 fma_s(cf2 a, float s, cf2 c) { cf2 d; const cf2 sv = mk2(s, s); asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(sv), "v"(c)); return d; }
// the same with a per-lane scalar (vector register)
__device__ __forceinline__ cf2 scale_v(cf2 a, float s) { return pmul(a, mk2(s, s)); }
// a * w for a COMPILE-TIME w: the constant pair sits in scalar registers
__device__ __forceinline__ cf2 cmul_k(cf2 a, float wx, float wy) {
    cf2 t, d; const cf2 w = mk2(wx, wy);
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(a), "s"(w), "v"(t));
    return d;
}
__device__ __forceinline__ cf2 cmulc_k(cf2 a, float wx, float wy) {
    cf2 t, d; const cf2 w = mk2(wx, wy);
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,0,1]" : "=v"(d) : "v"(a), "s"(w), "v"(t));
    return d;
}
#else
__host__ __device__ __forceinline__ cf2 cadd(cf2 a, cf2 b) { return mk2(a.x + b.x, a.y + b.y); }
__host__ __device__ __forceinline__ cf2 csub(cf2 a, cf2 b) { return mk2(a.x - b.x, a.y - b.y); }
__host__ __device__ __forceinline__ cf2 add_ib(cf2 a, cf2 b) { return mk2(a.x - b.y, a.y + b.x); }
__host__ __device__ __forceinline__ cf2 sub_ib(cf2 a, cf2 b) { return mk2(a.x + b.y, a.y - b.x); }
__host__ __device__ __forceinline__ cf2 cadd_conj(cf2 a, cf2 b) { return mk2(a.x + b.x, a.y - b.y); }
__host__ __device__ __forceinline__ cf2 csub_conj(cf2 a, cf2 b) { return mk2(a.x - b.x, a.y + b.y); }
__host__ __device__ __forceinline__ cf2 conj_add_ib(cf2 a, cf2 b) { return mk2(a.x - b.y, -a.y - b.x); }
__host__ __device__ __forceinline__ cf2 conj_sub_ib(cf2 a, cf2 b) { return mk2(a.x + b.y, -a.y + b.x); }
__host__ __device__ __forceinline__ cf2 pmul(cf2 a, cf2 b) { return mk2(a.x * b.x, a.y * b.y); }
__host__ __device__ __forceinline__ cf2 pfma(cf2 a, cf2 b, cf2 c) { return mk2(a.x * b.x + c.x, a.y * b.y + c.y); }
__host__ __device__ __forceinline__ cf2 cmul(cf2 a, cf2 b) { return mk2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__host__ __device__ __forceinline__ cf2 cmulc(cf2 a, cf2 b) { return mk2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a * conj(b)
__host__ __device__ __forceinline__ cf2 scale(cf2 a, float s) { return mk2(a.x * s, a.y * s); }
__host__ __device__ __forceinline__ cf2 fma_s(cf2 a, float s, cf2 c) { return mk2(a.x * s + c.x, a.y * s + c.y); }
__host__ __device__ __forceinline__ cf2 scale_v(cf2 a, float s) { return mk2(a.x * s, a.y * s); }
__host__ __device__ __forceinline__ cf2 cmul_k(cf2 a, float wx, float wy) { return cmul(a, mk2(wx, wy)); }
__host__ __device__ __forceinline__ cf2 cmulc_k(cf2 a, float wx, float wy) { return cmulc(a, mk2(wx, wy)); }
#endif
__host__ __device__ __forceinline__ cf2 cconj(cf2 a) { return mk2(a.x, -a.y); }
// a +- (the direction's quarter turn) * b: forward transforms multiply by -i, inverse ones by +i
template <bool INV> __host__ __device__ __forceinline__ cf2 add_rot(cf2 a, cf2 b) { return INV ? add_ib(a, b) : sub_ib(a, b); }
template <bool INV> __host__ __device__ __forceinline__ cf2 sub_rot(cf2 a, cf2 b) { return INV ? sub_ib(a, b) : add_ib(a, b); }
// multiply by -i (forward) or +i (inverse) on its own (rare: the trivial twiddles of a Cooley-Tukey split)
template <bool INV> __host__ __device__ __forceinline__ cf2 mul_mi(cf2 a) { return INV ? mk2(-a.y, a.x) : mk2(a.y, -a.x); }

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
    static __host__ __device__ __forceinline__ void run(cf2 (&v)[2]) {
        const cf2 a = v[0], b = v[1];
        v[0] = cadd(a, b); v[1] = csub(a, b);
    }
};
template <bool INV> struct Radix<3, INV> {
    static __host__ __device__ __forceinline__ void run(cf2 (&v)[3]) {
        const float S = 0.86602540378443864676f;
        const cf2 t = cadd(v[1], v[2]);
        const cf2 sd = scale(csub(v[1], v[2]), S);
        const cf2 m = fma_s(t, -0.5f, v[0]);
        v[0] = cadd(v[0], t);
        v[1] = add_rot<INV>(m, sd);                          // m -+ i S d
        v[2] = sub_rot<INV>(m, sd);
    }
};
template <bool INV> struct Radix<4, INV> {
    static __host__ __device__ __forceinline__ void run(cf2 (&v)[4]) {
        const cf2 a0 = cadd(v[0], v[2]), a1 = csub(v[0], v[2]);
        const cf2 b0 = cadd(v[1], v[3]), d = csub(v[1], v[3]);
        v[0] = cadd(a0, b0); v[2] = csub(a0, b0);
        v[1] = add_rot<INV>(a1, d); v[3] = sub_rot<INV>(a1, d);
    }
};
template <bool INV> struct Radix<5, INV> {
    static __host__ __device__ __forceinline__ void run(cf2 (&v)[5]) {
        const float C1 = 0.30901699437494742410f, C2 = -0.80901699437494742410f;
        const float S1 = 0.95105651629515357212f, S2 = 0.58778525229247312917f;
        const cf2 a1 = cadd(v[1], v[4]), a2 = cadd(v[2], v[3]);
        const cf2 b1 = csub(v[1], v[4]), b2 = csub(v[2], v[3]);
        const cf2 r1 = fma_s(a2, C2, fma_s(a1, C1, v[0]));
        const cf2 r2 = fma_s(a2, C1, fma_s(a1, C2, v[0]));
        const cf2 i1 = fma_s(b2, S2, scale(b1, S1));
        const cf2 i2 = fma_s(b2, -S1, scale(b1, S2));
        v[0] = cadd(v[0], cadd(a1, a2));
        v[1] = add_rot<INV>(r1, i1); v[4] = sub_rot<INV>(r1, i1);
        v[2] = add_rot<INV>(r2, i2); v[3] = sub_rot<INV>(r2, i2);
    }
};
template <bool INV> struct Radix<7, INV> {
    static __host__ __device__ __forceinline__ void run(cf2 (&v)[7]) {
        const float C[3] = { 0.62348980185873353053f, -0.22252093395631440429f, -0.90096886790241912624f };
        const float S[3] = { 0.78183148246802980871f, 0.97492791218182360702f, 0.43388373911755812048f };
        cf2 a[3], b[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) { a[q] = cadd(v[q + 1], v[6 - q]); b[q] = csub(v[q + 1], v[6 - q]); }
        cf2 y[7];
        y[0] = cadd(v[0], cadd(a[0], cadd(a[1], a[2])));
#pragma unroll
        for (int k = 1; k <= 3; ++k) {
            cf2 re = v[0], im = mk2(0.f, 0.f);
#pragma unroll
            for (int q = 1; q <= 3; ++q) {
                const int t = (k * q) % 7;               // cos(2*pi*t/7), sin(2*pi*t/7)
                const float c = t <= 3 ? C[t - 1] : C[6 - t];
                const float s = t <= 3 ? S[t - 1] : -S[6 - t];
                re = fma_s(a[q - 1], c, re);
                im = q == 1 ? scale(b[0], s) : fma_s(b[q - 1], s, im);
            }
            y[k] = add_rot<INV>(re, im); y[7 - k] = sub_rot<INV>(re, im);
        }
#pragma unroll
        for (int q = 0; q < 7; ++q) v[q] = y[q];
    }
};
template <bool INV> struct Radix<8, INV> {
    static __host__ __device__ __forceinline__ void run(cf2 (&v)[8]) {
        const float H = 0.70710678118654752440f;
        cf2 e[4] = { v[0], v[2], v[4], v[6] };
        cf2 o[4] = { v[1], v[3], v[5], v[7] };
        Radix<4, INV>::run(e);
        Radix<4, INV>::run(o);
        // W8^1 = (1 - i)/sqrt2 (fwd): o1 W = H (o1 - i o1);  W8^3 = (-1 - i)/sqrt2: o3 W = -H (o3 + i o3);  conjugates for the inverse
        const cf2 o1 = scale(add_rot<INV>(o[1], o[1]), H);
        const cf2 o3 = scale(sub_rot<INV>(o[3], o[3]), -H);
        v[0] = cadd(e[0], o[0]); v[4] = csub(e[0], o[0]);
        v[1] = cadd(e[1], o1);   v[5] = csub(e[1], o1);
        v[2] = add_rot<INV>(e[2], o[2]); v[6] = sub_rot<INV>(e[2], o[2]);      // W8^2 = -+ i
        v[3] = cadd(e[3], o3);   v[7] = csub(e[3], o3);
    }
};

}  // namespace kcc

// kcc_fft2.h -- register-resident FFT engine for gfx950.
//
// A line of N complex points is transformed by 2 or 3 large-radix Stockham passes.  One thread owns one
// butterfly of a pass: its R points live in VGPRs, the radix-R DFT is a fully unrolled Cooley-Tukey
// recursion over base radices {2,3,4,5,7,8} with compile-time twiddle constants, and passes exchange
// data through one padded LDS buffer per line (write, one barrier, read).  The inverse transform uses the
// radices in reverse order, so the register layout after the last pass of one direction IS the first-pass
// input layout of the other direction: fused inverse -> pointwise -> forward (and forward -> pointwise ->
// inverse) stages never leave registers.
#pragma once
#include <hip/hip_runtime.h>

#include "kcc_consts.h"
#include "kcc_fft.h"   // base radices Radix<2,3,4,5,7,8>, complex helpers

namespace kcc {

__host__ __device__ constexpr bool is_base_radix(int r) { return r == 2 || r == 3 || r == 4 || r == 5 || r == 7 || r == 8; }
__host__ __device__ constexpr int gcd_(int a, int b) { return b == 0 ? a : gcd_(b, a % b); }
// Factor split R = A * B of a composite radix.  Coprime splits come first (largest base radix A with gcd(A, R/A) = 1): they
// run as a Good-Thomas prime-factor transform with NO twiddle multiplications between the two stages (15 = 3 x 5,
// 18 = 2 x 9, 20 = 4 x 5, 24 = 8 x 3, 10 = 2 x 5 ...; a radix-24 butterfly loses 21 of its complex multiplications, a
// 480-point transform a fifth of its arithmetic).  Otherwise Cooley-Tukey with compile-time twiddles (16 = 4 x 4, 9 = 3 x 3).
#ifndef KCC_PFA
#define KCC_PFA 1
#endif
__host__ __device__ constexpr int pfa_factor(int r) {
    if (!KCC_PFA) return 0;
    const int cand[6] = { 8, 7, 5, 4, 3, 2 };
    for (int i = 0; i < 6; ++i) if (r % cand[i] == 0 && r / cand[i] > 1 && gcd_(cand[i], r / cand[i]) == 1) return cand[i];
    return 0;
}
__host__ __device__ constexpr int first_factor(int r) {
    return is_base_radix(r) ? r : pfa_factor(r) ? pfa_factor(r) : (r % 4 == 0 ? 4 : r % 2 == 0 ? 2 : r % 3 == 0 ? 3 : r % 5 == 0 ? 5 : 7);
}
__host__ __device__ constexpr bool is_pfa(int r) { return !is_base_radix(r) && pfa_factor(r) != 0; }
// register index holding output k of dft_run<R>.  Cooley-Tukey (k = k1 + A k2): row k1 = k % A, column = the position of
// k2 = k / A inside the B-point sub-transform.  Prime-factor (k = k1 mod A, k = k2 mod B): row k % A, column position of k % B.
template <int R> __host__ __device__ constexpr int dft_pos(int k) {
    if constexpr (is_base_radix(R)) {
        return k;
    } else {
        constexpr int A = first_factor(R), B = R / A;
        return B * (k % A) + dft_pos<B>(is_pfa(R) ? k % B : k / A);
    }
}

// In-register DFT of R points (R = product of base radices).  Output k ends up in v[dft_pos<R>(k)].
template <int R, bool INV>
__device__ __forceinline__ void dft_run(cf2 (&v)[R]) {
    if constexpr (is_base_radix(R)) {
        Radix<R, INV>::run(v);
    } else if constexpr (is_pfa(R)) {
        constexpr int A = first_factor(R), B = R / A;
        // Good-Thomas: input n = (B n1 + A n2) mod R, output k with k = k1 (mod A), k = k2 (mod B):
        //   W_R^(n k) = W_A^(n1 k1) W_B^(n2 k2) -- a plain A x B two-dimensional transform.  All index maps are compile-time
        //   register renames.
        cf2 t[A][B];
#pragma unroll
        for (int n2 = 0; n2 < B; ++n2) {
            cf2 c[A];
#pragma unroll
            for (int n1 = 0; n1 < A; ++n1) c[n1] = v[(B * n1 + A * n2) % R];
            dft_run<A, INV>(c);
#pragma unroll
            for (int k1 = 0; k1 < A; ++k1) t[k1][n2] = c[dft_pos<A>(k1)];
        }
#pragma unroll
        for (int k1 = 0; k1 < A; ++k1) {
            dft_run<B, INV>(t[k1]);                           // output k2 of row k1 sits at t[k1][dft_pos<B>(k2)]
#pragma unroll
            for (int i = 0; i < B; ++i) v[B * k1 + i] = t[k1][i];
        }
    } else {
        constexpr int A = first_factor(R), B = R / A;
        // n = B*n1 + n2, k = k1 + A*k2:  DFT_A over n1, twiddle W_R^(n2*k1), DFT_B over n2
#pragma unroll
        for (int n2 = 0; n2 < B; ++n2) {
            cf2 t[A];
#pragma unroll
            for (int n1 = 0; n1 < A; ++n1) t[n1] = v[B * n1 + n2];
            dft_run<A, INV>(t);
#pragma unroll
            for (int k1 = 0; k1 < A; ++k1) v[B * k1 + n2] = t[dft_pos<A>(k1)];
        }
#pragma unroll
        for (int k1 = 1; k1 < A; ++k1) {
#pragma unroll
            for (int n2 = 1; n2 < B; ++n2) {
                const int idx = (n2 * k1) % R;
                cf2& x = v[B * k1 + n2];
                if ((4 * idx) % R == 0) {                     // W = (-i)^(4 idx / R): free
                    const int e = ((4 * idx) / R) & 3;
                    const int ee = INV ? ((4 - e) & 3) : e;
                    if (ee == 1) x = mk2(x.y, -x.x);
                    else if (ee == 2) x = mk2(-x.x, -x.y);
                    else if (ee == 3) x = mk2(-x.y, x.x);
                } else {
                    x = INV ? cmulc_k(x, Wc<R>::c[idx], Wc<R>::s[idx]) : cmul_k(x, Wc<R>::c[idx], Wc<R>::s[idx]);
                }
            }
        }
#pragma unroll
        for (int k1 = 0; k1 < A; ++k1) {
            cf2 u[B];
#pragma unroll
            for (int n2 = 0; n2 < B; ++n2) u[n2] = v[B * k1 + n2];
            dft_run<B, INV>(u);
#pragma unroll
            for (int k2 = 0; k2 < B; ++k2) v[B * k1 + k2] = u[k2];      // output k2 stays at u's position dft_pos<B>(k2)
        }
    }
}

__host__ __device__ constexpr int cmax(int a, int b) { return a > b ? a : b; }

#ifndef KCC_SWZ16
#define KCC_SWZ16 1
#endif
// FFT plan: N = R1*R2 (*R3).  Forward applies R1, R2, R3; inverse applies them reversed.
template <int N_, int R1_, int R2_, int R3_ = 1>
struct Plan {
    static constexpr int N = N_, R1 = R1_, R2 = R2_, R3 = R3_;
    static constexpr int NP = R3_ > 1 ? 3 : 2;
    static constexpr bool PRIME = false;
    static_assert(R1_ * R2_ * R3_ == N_, "plan radices must multiply to N");
    // threads per line = most butterflies in any pass
    static constexpr int T = cmax(cmax(N_ / R1_, N_ / R2_), R3_ > 1 ? N_ / R3_ : 0);
    template <bool INV> static constexpr int RF() { return INV ? (NP == 3 ? R3_ : R2_) : R1_; }   // first radix
    template <bool INV> static constexpr int RM() { return R2_; }                                  // middle (NP==3)
    template <bool INV> static constexpr int RL() { return INV ? R1_ : (NP == 3 ? R3_ : R2_); }   // last radix
    // exchange buffer: index i is stored at i + padc(RF) * (i / RF) (RF = first radix of the direction): the pass-1 writes of
    // neighbouring threads (stride RF) become stride RF + padc, which is always ODD -- an even stride in cf2 puts the 16
    // lanes of a write group on 16 or fewer banks (an odd first radix with one pad element, e.g. the inverse 240 = 15 x 16
    // plan, put all of them on ONE bank pair: 16-way conflicts in every inverse kernel of the 640x480 image family)
    // SWZ plans (240 = 16 x 15): no padding at all.  The radix-16 direction stores index i at i ^ ((i >> 4) & 15) -- its
    // pass-1 writes (stride 16) and last-pass reads (16 consecutive) both touch 16 different bank pairs -- and the radix-15
    // direction needs none (an odd stride already does).  The buffer shrinks from 272 to 240 elements per line, below the
    // natural-order pitch, and the 240-point A kernels fit FIVE workgroups per CU instead of four (32 KB each was the limit).
    static constexpr bool SWZ = KCC_SWZ16 && R3_ == 1 && ((R1_ == 16 && N_ / R2_ == 16) || (R2_ == 16 && N_ / R1_ == 16)) && ((R1_ == 16 ? R2_ : R1_) % 2 == 1);
    // SWZ3 (640 = 8 x 8 x 10): likewise no padding.  Forward: exchange 1 stores i at i ^ ((i >> 5) & 3), exchange 2 at
    // i ^ (((i >> 6) & 1) << 3); inverse (first radix 10: stride 10 is conflict-free as it is): plain.  Every write and read of
    // both directions is conflict-free (tools/lds_sim3.py), and the line shrinks from 720 to 640 elements: the two-plane
    // 640-point B kernels fit five workgroups per CU instead of four.
    static constexpr bool SWZ3 = KCC_SWZ16 && N_ == 640 && R1_ == 8 && R2_ == 8 && R3_ == 10;
    static constexpr int padc(int rf) { return (SWZ || SWZ3) ? 0 : ((rf % 2 == 0) ? 1 : 2); }
    static constexpr int ext_dir(int rf) { return N_ + padc(rf) * (N_ / rf); }
    // (ext_dir is exact to within one element: the largest physical index is N - 1 + padc * ((N - 1) / rf) < ext_dir(rf))
    static constexpr int EXT = cmax(ext_dir(R1_), ext_dir(R3_ > 1 ? R3_ : R2_));
};

// instantiated lengths
template <int N> struct PlanFor;
template <> struct PlanFor<30> : Plan<30, 5, 6> {};
template <> struct PlanFor<60> : Plan<60, 6, 10> {};
template <> struct PlanFor<120> : Plan<120, 10, 12> {};
#ifndef KCC_P240
#define KCC_P240 16, 15
#endif
// (forward kernels of the 720-row planes: a 20-point first pass makes the u8 de-rotation's source bands 36 rows tall -- 3 LDS
// pieces per box row instead of 5, three workgroups per CU instead of two at 1280x720 -- and suits the polar gather of that
// geometry (-11 %); the spectrum-in kernels keep 18 x 20, PlanInv below)
#ifndef KCC_P360
#define KCC_P360 20, 18
#endif
#ifndef KCC_P480
#define KCC_P480 20, 24
#endif
#ifndef KCC_P640
#define KCC_P640 8, 8, 10
#endif
template <> struct PlanFor<240> : Plan<240, KCC_P240> {};
template <> struct PlanFor<360> : Plan<360, KCC_P360> {};
template <> struct PlanFor<80> : Plan<80, 8, 10> {};
template <> struct PlanFor<160> : Plan<160, 10, 16> {};
template <> struct PlanFor<320> : Plan<320, 16, 20> {};
template <> struct PlanFor<480> : Plan<480, KCC_P480> {};
template <> struct PlanFor<640> : Plan<640, KCC_P640> {};
#ifndef KCC_P1280
#define KCC_P1280 20, 8, 8
#endif
template <> struct PlanFor<1280> : Plan<1280, KCC_P1280> {};
// reference configs/config_geekplus.yaml (448 x 448) and configs/config_HD.yaml (1600 x 1200)
template <> struct PlanFor<224> : Plan<224, 14, 16> {};
template <> struct PlanFor<448> : Plan<448, 7, 8, 8> {};
template <> struct PlanFor<600> : Plan<600, 24, 25> {};
// 512 x 512 cameras (round 5; the reference transforms whatever it is given, correlation_flow.cc:53-63): half-rows 256 = 16 x 16
// (16 threads per line: wave-local chains), lines 512 = 8 x 8 x 8 for the two-plane B kernels and 16 x 32 for the single-plane ones
#ifndef KCC_P256
#define KCC_P256 16, 16
#endif
#ifndef KCC_P512
#define KCC_P512 8, 8, 8
#endif
template <> struct PlanFor<256> : Plan<256, KCC_P256> {};
template <> struct PlanFor<512> : Plan<512, KCC_P512> {};
// 1024 x 768: half-rows 384 = 16 x 24, lines 1024 = 8 x 8 x 16 (two-plane kernels) / 32 x 32 (single-plane ones)
template <> struct PlanFor<384> : Plan<384, 16, 24> {};
template <> struct PlanFor<1024> : Plan<1024, 8, 8, 16> {};
template <> struct PlanFor<1600> : Plan<1600, 10, 10, 16> {};

// Lines whose length has ONE large prime factor: N = 16 x PR (752 = 16 x 47, the 752 x 480 cameras of the reference's EuRoC-class
// data sets -- correlation_flow.cc:53-63 plans any size).  The smooth part runs as radix-16 butterflies in registers, the prime part
// as a direct PR-point DFT whose outputs are shared out over the line's PR threads (thread t computes output k2 = t of all 16
// sub-transforms: every input is an LDS broadcast read, every term one complex multiply-add with W_PR^(n t mod PR) from a
// PR-entry table appended to the pass table).  From outside it is a plan whose register arrays are [16] in both directions
// with stride PR (in: x[j + q PR], out: X[j + q PR], j < PR): the B kernels' loads, stores and pointwise code are unchanged.
template <int N_, int PR_>
struct PlanPrime {
    static constexpr int N = N_, PR = PR_, R1 = 16, R2 = PR_, R3 = 1, NP = 2;
    static_assert(16 * PR_ == N_, "prime plan: N = 16 x PR");
    static constexpr bool PRIME = true, SWZ = false, SWZ3 = false;
    static constexpr int T = PR_ + 1;                        // PR threads carry the line; one more makes the prime pass split evenly
    template <bool INV> static constexpr int RF() { return 16; }
    template <bool INV> static constexpr int RM() { return 1; }
    template <bool INV> static constexpr int RL() { return 16; }
    static constexpr int padc(int) { return 0; }
    static constexpr int EXT = N_ + PR_ + 1;                 // index i lives at i + i / 16
};
template <> struct PlanFor<752> : PlanPrime<752, 47> {};
template <> struct PlanFor<848> : PlanPrime<848, 53> {};          // 848 x 480 (depth-camera colour streams): 848 = 16 x 53

// Plan of the SINGLE-plane B kernels (fwd_abs_inv, the Kzz-cached *_x / solve_cached / zz_inv modes, plain fwd / inv): with one
// plane per thread a two-pass plan's large radices fit the register budget, and 640 = 20 x 32 (32 threads per line: two passes,
// three barriers fewer per transform) beats 8 x 8 x 10 there (fwd_abs_inv 0.187 -> 0.169 ms); the two-plane kernels lose with it
// (registers) and keep PlanFor.  Default: the same plan.
template <int N> struct PlanAlt : PlanFor<N> {};
#ifndef KCC_PA640
#define KCC_PA640 20, 32
#endif
template <> struct PlanAlt<640> : Plan<640, KCC_PA640> {};
#ifndef KCC_PA512
#define KCC_PA512 16, 32
#endif
template <> struct PlanAlt<512> : Plan<512, KCC_PA512> {};
template <> struct PlanAlt<1024> : Plan<1024, 32, 32> {};
// 1280 points: the two-plane kernels (PlanFor) and the single-plane ones want different plans (HD workload, kB<1280,*> ms per 128 pairs:
// solve_inv / fwd_mul_inv / fwd_abs_inv = 0.437 / 0.385 / 0.284 with 8 x 10 x 16 everywhere, 0.406 / 0.354 / 0.330 with 20 x 8 x 8)
#ifndef KCC_PA1280
#define KCC_PA1280 8, 10, 16
#endif
template <> struct PlanAlt<1280> : Plan<1280, KCC_PA1280> {};
// (measured and not adopted: 1280 = 32 x 40 / 40 x 32 for the single-plane 1280-point kernels -- fwd_abs_inv 0.30 -> 0.34 / 0.37 ms;
// 480 = 16 x 30 for the Kzz-cached 480-point kernels -- 122.7 k -> 118.3 k candidates/s)

// Plan used by the spectrum-in (inverse) A-type kernels; may differ from PlanFor (their tile width, hence their
// thread budget, differs).  Default: the same plan.
template <int N> struct PlanInv : PlanFor<N> {};
#ifdef KCC_PI240
template <> struct PlanInv<240> : Plan<240, KCC_PI240> {};
#endif
#ifndef KCC_PI360
#define KCC_PI360 18, 20
#endif
template <> struct PlanInv<360> : Plan<360, KCC_PI360> {};

// Twiddle table layout (built on the host, see kcc_api.hip build_plan_tables):
//   2 passes, direction d:  tw[q*RF + k] = W_N^(+-q*k),            q < RL, k < RF
//   3 passes, direction d:  tw[q*RF + k] = W_(RF*RM)^(+-q*k),      q < RM, k < RF          (pass 2)
//                           tw[OFF3 + q*(RF*RM) + k] = W_N^(+-q*k), q < RL, k < RF*RM      (pass 3)
//   with OFF3 = RM*RF.  Forward tables use the minus sign; inverse tables the plus sign (so the kernels
//   always multiply, never conjugate).
template <class P, bool INV> struct Dir {
    static constexpr int N = P::N;
    static constexpr int RF = P::template RF<INV>(), RM = P::template RM<INV>(), RL = P::template RL<INV>();
    static constexpr int MF = N / RF, ML = N / RL;           // butterflies (= active threads) in first / last pass
    static constexpr int MM = P::NP == 3 ? N / RM : 0;
    static constexpr int PAD = RF, PADC = P::padc(RF);
    static constexpr bool SWZ = P::SWZ && RF == 16;          // XOR-swizzled exchange (see Plan::SWZ)
    static constexpr bool S3F = P::SWZ3 && !INV;             // XOR-swizzled exchanges of the 3-pass forward direction
    static constexpr int OFF3 = RM * RF;
    // strided reads i = j + q*M map to phys(j) + q*(M + PADC*M/PAD) when PAD divides M (true for every plan here)
    static_assert(P::PRIME || (ML % PAD == 0 && (P::NP == 2 || MM % PAD == 0)), "pad must divide the pass strides");
    static constexpr int SL = ML + PADC * (ML / PAD);        // phys stride of last-pass reads
    static constexpr int SM = P::NP == 3 ? MM + PADC * (MM / PAD) : 0;
    __device__ static __forceinline__ unsigned phys(unsigned i) { return i + (unsigned)PADC * (i / (unsigned)PAD); }
};

// Run all passes of one direction on NV independent lines owned by this thread (same j, different data),
// each with its own exchange buffer ex[v] (LDS, >= P::EXT cf2).
//   in : vin[v][q]  = x[j + q*MF]          (valid for j < MF)
//   out: vout[v][q] = X[j + q*ML]          (valid for j < ML), natural order
// All threads of the workgroup must call this (it contains barriers); the caller must place a barrier
// between two chains that share an exchange buffer.
// Synchronisation between the passes of a line.  When all threads of a line sit in ONE wavefront (WAVE: the
// caller maps line = tid / T with T dividing 64), the LDS exchange needs no workgroup barrier: a wave's LDS
// operations execute in order, so only the compiler has to be kept from reordering them.  The waves of a workgroup
// then run their transforms independently (no lock-step with the slowest wave).
template <bool WAVE> __device__ __forceinline__ void line_sync() {
    if constexpr (WAVE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}
// acc + a * w (complex), two packed FMAs
__device__ __forceinline__ cf2 cmac(cf2 acc, cf2 a, cf2 w) {
#if KCC_PK_ASM
    cf2 t, d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(t) : "v"(a), "v"(w), "v"(acc));                    // (a.x w.x + acc.x, a.x w.y + acc.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(a), "v"(w), "v"(t));       // (-a.y w.y + t.x, a.y w.x + t.y)
    return d;
#else
    return mk2(acc.x + a.x * w.x - a.y * w.y, acc.y + a.x * w.y + a.y * w.x);
#endif
}
// The chain of a PlanPrime line (see PlanPrime).  N = 16 PR, n = PR n1 + n2, k = k1 + 16 k2:
//   step A (thread j = n2):  y[k1][n2] = W_N^(n2 k1) * sum_n1 x[PR n1 + n2] W_16^(n1 k1)        -- dft_run<16> + 15 twiddles
//   step B (PR + 1 threads): X[k1 + 16 k2] = sum_n2 y[k1][n2] W_PR^(n2 k2)                      -- two outputs k2 of eight k1 per thread
//   transpose:               X back into the strided register layout through the same buffer
// Tables: forward tw[n2 * 16 + k1] = W_N^-(n2 k1), inverse tw[k1 * PR + n2] = W_N^+(n2 k1) (plan_table's two-pass layout for the
// radices (16, PR)); both followed, at tw[N + m], by W_PR^(-+ m), m < PR.  Index i of the exchange buffer lives at i + i / 16.
template <class P, bool INV, int NV, bool WAVE>
__device__ __forceinline__ void fft_chain_prime(cf2 (&vin)[NV][16], cf2 (&vout)[NV][16], unsigned j, cf2* const (&ex)[NV], const cf2* __restrict__ tw) {
    constexpr int N = P::N, PR = P::PR;
    const bool act = j < (unsigned)PR;
    if (act) {
        cf2 wa[16];
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) wa[k1] = INV ? tw[k1 * PR + j] : tw[j * 16 + k1];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            dft_run<16, INV>(vin[v]);
            cf2* w = ex[v] + j * 17;
            w[0] = vin[v][dft_pos<16>(0)];
#pragma unroll
            for (int k1 = 1; k1 < 16; ++k1) w[k1] = cmul(vin[v][dft_pos<16>(k1)], wa[k1]);
        }
    }
    line_sync<WAVE>();
    // step B on PR + 1 threads: thread t owns the outputs k2 = 2 (t / 2), + 1 of HALF the sub-transforms (k1 = 8 (t % 2) + i): every
    // y it reads feeds two multiply-adds, and a wave-instruction reads only two distinct addresses (broadcast) -- half the LDS
    // instructions of one output per thread (the LDS pipeline, not the VALU, bounds this pass)
    static_assert(PR % 2 == 1 && P::T == PR + 1, "prime plan: PR + 1 threads per line");
    const unsigned kh = (j & 1u) * 8u, ka = (j >> 1) * 2u, kb = ka + 1u;
    const bool actb = j < (unsigned)(PR + 1), hasb = kb < (unsigned)PR;
    cf2 acc[NV][2][8];
    if (actb) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[v][0][i] = acc[v][1][i] = ex[v][kh + i];        // n2 = 0: W = 1
        const cf2* wp = tw + N;
        unsigned ia = ka, ib = hasb ? kb : 0u;
#pragma unroll 1
        for (int n2 = 1; n2 < PR; ++n2) {
            const cf2 wa2 = wp[ia], wb2 = wp[ib];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const cf2* y = ex[v] + n2 * 17 + kh;
#pragma unroll
                for (int i = 0; i < 8; ++i) { const cf2 yy = y[i]; acc[v][0][i] = cmac(acc[v][0][i], yy, wa2); acc[v][1][i] = cmac(acc[v][1][i], yy, wb2); }
            }
            ia += ka; ia -= ia >= (unsigned)PR ? (unsigned)PR : 0u;
            ib += kb; ib -= ib >= (unsigned)PR ? (unsigned)PR : 0u;
        }
    }
    line_sync<WAVE>();                                       // every y consumed before the buffer takes the spectrum
    if (actb) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                ex[v][17 * ka + kh + i] = acc[v][0][i];                                   // X[k1 + 16 k2] lives at 17 k2 + k1
                if (hasb) ex[v][17 * kb + kh + i] = acc[v][1][i];
            }
    }
    line_sync<WAVE>();
    if (act) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int q = 0; q < 16; ++q) { const unsigned i = j + (unsigned)(PR * q); vout[v][q] = ex[v][i + (i >> 4)]; }
    }
}
template <class P, bool INV, int NV, bool WAVE = false, int RFV = 0, int RLV = 0>
__device__ __forceinline__ void fft_chain(cf2 (&vin)[NV][RFV], cf2 (&vout)[NV][RLV],
                                          unsigned j, cf2* const (&ex)[NV], const cf2* __restrict__ tw) {
    using D = Dir<P, INV>;
    constexpr int RF = D::RF, RL = D::RL, RM = D::RM;
    static_assert(RFV == RF && RLV == RL, "register arrays must match the plan's first / last radix");
    if constexpr (P::PRIME) {
        fft_chain_prime<P, INV, NV, WAVE>(vin, vout, j, ex, tw);
        return;
    } else {
    // ---- pass 1: radix RF, Ns = 1 (no twiddles); output q of butterfly j goes to phys(j*RF + q) = j*(RF+PADC) + q
    if (j < (unsigned)D::MF) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            dft_run<RF, INV>(vin[v]);
            if constexpr (D::SWZ) {
                cf2* w = ex[v] + j * 16;
                const unsigned t = j & 15u;
#pragma unroll
                for (int q = 0; q < RF; ++q) w[t ^ (unsigned)q] = vin[v][dft_pos<RF>(q)];
            } else if constexpr (D::S3F) {
                static_assert(RF == 8 && RM == 8 && D::MM == 80 && D::ML == 64, "swizzle constants are the 8 x 8 x 10 plan's");
                cf2* w = ex[v] + j * 8;
                const unsigned t = (j >> 2) & 3u;                // (i >> 5) & 3 for i = 8 j + q
#pragma unroll
                for (int q = 0; q < RF; ++q) w[t ^ (unsigned)q] = vin[v][dft_pos<RF>(q)];
            } else {
                cf2* w = ex[v] + j * (RF + D::PADC);
#pragma unroll
                for (int q = 0; q < RF; ++q) w[q] = vin[v][dft_pos<RF>(q)];
            }
        }
    }
    if constexpr (P::NP == 3) {
        // ---- pass 2: radix RM, Ns = RF
        constexpr int MM = D::MM;
        cf2 vm[NV][RM];
        cf2 w2[RM];
        const unsigned k = j % (unsigned)RF, jb = j / (unsigned)RF;
        const bool act = j < (unsigned)MM;
        if (act) {
#pragma unroll
            for (int q = 1; q < RM; ++q) w2[q] = tw[q * RF + k];
        }
        line_sync<WAVE>();
        if (act) {
            const unsigned pj = D::phys(j);
            if constexpr (D::S3F) {
                unsigned ri[RM];
#pragma unroll
                for (int q = 0; q < RM; ++q) { const unsigned i = j + (unsigned)(q * MM); ri[q] = i ^ ((i >> 5) & 3u); }
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int q = 0; q < RM; ++q) vm[v][q] = ex[v][ri[q]];
            } else {
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int q = 0; q < RM; ++q) vm[v][q] = ex[v][pj + q * D::SM];
            }
        }
        line_sync<WAVE>();
        if (act) {
            // phys(jb*RF*RM + k + q*RF) = jb*(RF*RM + PADC*RM) + k + q*(RF+PADC)
            const unsigned wb = jb * (RF * RM + D::PADC * RM) + k;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
#pragma unroll
                for (int q = 1; q < RM; ++q) vm[v][q] = cmul(vm[v][q], w2[q]);
                dft_run<RM, INV>(vm[v]);
                if constexpr (D::S3F) {
                    const unsigned t2 = (jb & 1u) << 3;          // ((i >> 6) & 1) << 3 for i = 64 jb + k + 8 q
#pragma unroll
                    for (int q = 0; q < RM; ++q) ex[v][wb + ((unsigned)(q * RF) ^ t2)] = vm[v][dft_pos<RM>(q)];
                } else {
#pragma unroll
                    for (int q = 0; q < RM; ++q) ex[v][wb + q * (RF + D::PADC)] = vm[v][dft_pos<RM>(q)];
                }
            }
        }
    }
    // ---- last pass: radix RL, Ns = N/RL = ML, k = j
    cf2 wl[RL];
    const bool actl = j < (unsigned)D::ML;
    if (actl) {
        const cf2* t = tw + (P::NP == 3 ? D::OFF3 : 0) + j;
#pragma unroll
        for (int q = 1; q < RL; ++q) wl[q] = t[q * D::ML];
    }
    line_sync<WAVE>();
    if (actl) {
        const unsigned pj = D::phys(j);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            cf2 t[RL];
            if constexpr (D::SWZ) {
                static_assert(D::ML == 16 && P::NP == 2, "swizzle assumes 16 last-pass butterflies");
#pragma unroll
                for (int q = 0; q < RL; ++q) t[q] = ex[v][16 * q + (j ^ (unsigned)(q & 15))];
            } else if constexpr (D::S3F) {
                const unsigned j8 = j ^ 8u;                      // i ^ (((i >> 6) & 1) << 3) for i = j + 64 q
#pragma unroll
                for (int q = 0; q < RL; ++q) t[q] = ex[v][((q & 1) ? j8 : j) + q * 64];
            } else {
#pragma unroll
                for (int q = 0; q < RL; ++q) t[q] = ex[v][pj + q * D::SL];
            }
#pragma unroll
            for (int q = 1; q < RL; ++q) t[q] = cmul(t[q], wl[q]);
            dft_run<RL, INV>(t);
#pragma unroll
            for (int q = 0; q < RL; ++q) vout[v][q] = t[dft_pos<RL>(q)];
        }
    }
    }
}

}  // namespace kcc

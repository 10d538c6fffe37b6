// kcc_pointwise.h -- the small element-wise device functions both kernel families share (the tiled fast kernels of
// kcc_kernels.hip and the any-size kernels of kcc_generic.hip): OpenCV's bilinear blend and border rule, the exact u8 -> [0,1]
// conversion, the kernel functions of EstimateTrans.  Device code only.
#pragma once
#include <hip/hip_runtime.h>

#include "kcc_kernels.h"

namespace kcc {

static __device__ __noinline__ int wrap_idx(int p, int len) {         // cv::borderInterpolate(BORDER_WRAP), general (slow) form
    if (p < 0) p -= ((p - len + 1) / len) * len;
    if (p >= len) p %= len;
    return p;
}

// cv::remap bilinear weights from 1/32-pixel fractions, summed in OpenCV's order.  Contraction is
// switched off so the arithmetic is the same mul/add sequence the CPU executes (bit-exact gathers).
__device__ __forceinline__ float bilerp(float v0, float v1, float v2, float v3, int fx, int fy) {
#pragma clang fp contract(off)
    const float s = 1.f / 32.f;
    const float tx1 = (float)fx * s, tx0 = 1.f - tx1;
    const float ty1 = (float)fy * s, ty0 = 1.f - ty1;
    const float w0 = ty0 * tx0, w1 = ty0 * tx1, w2 = ty1 * tx0, w3 = ty1 * tx1;
    float acc = v0 * w0;
    acc = acc + v1 * w1;
    acc = acc + v2 * w2;
    acc = acc + v3 * w3;
    return acc;
}
// saturate_cast<int>(m*c*AB_SCALE) and saturate_cast<int>((m1*r + m2)*AB_SCALE) of cv::warpAffine, in double
__device__ __forceinline__ int affine_delta(double m, int c) {
#pragma clang fp contract(off)
    const double t = m * (double)c;
    return __double2int_rn(t * 1024.0);
}
__device__ __forceinline__ int affine_base(double m1, int r, double m2) {
#pragma clang fp contract(off)
    const double t = m1 * (double)r;
    const double u = t + m2;
    return __double2int_rn(u * 1024.0);
}

// (xz + offset)^power as Eigen's Array::pow(int) does it: double pow, rounded to float.  The general-power
// path is kept out of line (it is ~100 instructions of libm pow per element).
// oracle/RECALLED.md row 16 is an OPEN question: Eigen 3.3 may promote the integer exponent to the array's scalar first
// (promote_scalar_arg), i.e. evaluate powf(x, 3.0f) -- one ulp away from the double evaluation on 6.6e-4 of the samples with
// glibc (tests/test_toolchain_pins.py), never enough to move an index.  -DKCC_POLY_POWF=1 builds that reading (with the device
// library's powf, itself within an ulp of glibc's); the oracle has the same switch (ora_set_pow_mode).
#ifndef KCC_POLY_POWF
#define KCC_POLY_POWF 0
#endif
static __device__ __noinline__ float pow_generic(float b, int p) { return KCC_POLY_POWF ? powf(b, (float)p) : (float)pow((double)b, (double)p); }
enum { KT_POLY3 = 0, KT_POLYN = 1, KT_GAUSS = 2 };
template <int KT>
__device__ __forceinline__ float kernel_value(const KernelFn& fn, float xz, float gauss_bias, float gauss_scale) {
    if (KT == KT_POLY3) {
        if (KCC_POLY_POWF) return powf(xz + fn.offset, 3.0f);
        const double b = (double)(xz + fn.offset);
        return (float)(b * b * b);
    } else if (KT == KT_POLYN) {
        return pow_generic(xz + fn.offset, fn.power);
    } else {
        // gaussian: exp(-1/sigma^2 * (xx + zz - 2 xz)/N)   (correlation_flow.cc:189-190)
        return expf((gauss_bias - 2.f * xz) * gauss_scale);
    }
}

// (float)v / 255.0f for an 8-bit v, correctly rounded, in two FP operations: v*chi + RN(v*clo) with chi + clo = 1/255 to
// 48 bits (exhaustively equal to the IEEE division for v = 0..255: tests/test_host_tables.py).  ConvertMatToNormalizedArray
// (utils.cc:110-118) on the fly.
__device__ __forceinline__ float unit_u8(unsigned v) {
    const float x = (float)v;
    return __builtin_fmaf(x, 0x1.010102p-8f, x * -0x1.fdfdfep-33f);
}


}  // namespace kcc

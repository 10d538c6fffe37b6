// Host stand-in for <hip/hip_runtime.h>: just enough for g++ to parse the FFT engine headers (kcc_fft.h, kcc_fft2.h) so that
// their compile-time machinery and in-register butterflies can be checked on the CPU (tests/cpp/dft_host_check.cpp).
#pragma once
#define __device__
#define __host__
#define __forceinline__ inline
struct float2 { float x, y; };
static inline float2 make_float2(float x, float y) { return { x, y }; }
static inline void __syncthreads() {}
static inline void __builtin_amdgcn_fence(int, const char*) {}
static inline void __builtin_amdgcn_wave_barrier() {}

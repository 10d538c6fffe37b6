// kcc_generic.h -- launch interface of the any-size kernels (kcc_generic.hip): see that file's header.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kcc_kernels.h"

namespace kcc {

enum { GPLAN_MAX_RADICES = 16, G_ARGMAX_CHUNK = 8192 };
typedef float gcf2 __attribute__((ext_vector_type(2)));      // (= kcc::cf2 of kcc_fft.h: the same memory layout as float2)
// run-time Stockham plan of one line length: radices in execution order, tw[k] = exp(-2 pi i k / n), k < n (built in double)
struct GPlan { int n, nr; int radix[GPLAN_MAX_RADICES]; const gcf2* tw; };
GPlan gplan_make(int n, const float2* tw);
struct GFamily { PlaneGeom g; GPlan prow, pcol; };        // real plane rows x cols: lines along rows (halved axis), then along cols

enum { GF_C2C = 0, GF_R2C = 1, GF_C2R = 2 };
struct GFArgs {
    GPlan p; int mode, waves, tw_lds;
    const void* in; void* out;
    size_t in_item_stride, out_item_stride;              // elements (float on a real side, complex on a complex side)
    size_t in_line_stride, out_line_stride;
    int in_elem_stride, out_elem_stride;                 // complex sides only
    const int* in_idx; const int* out_idx;               // optional: item -> plane index (frame-store slots)
    int n_lines; float scale;
};

void g_rfft2(hipStream_t s, int n_items, const GFamily& f, const float* real, size_t real_item_stride, int pitch, const int* real_idx,
             float2* spec, size_t spec_item_stride, const int* spec_idx);
void g_irfft2(hipStream_t s, int n_items, const GFamily& f, float2* spec, size_t spec_item_stride, const int* spec_idx,
              float* real, size_t real_item_stride, int pitch);
void g_u8_load(hipStream_t s, int n, const uint8_t* src, size_t src_stride, float* real, size_t real_stride, uint8_t* keep, size_t keep_stride,
               int keep_pitch, const int* keep_slot, int H, int W);
void g_cvt_u8(hipStream_t s, int n, const uint8_t* arena_u8, size_t u8_stride, int u8_pitch, const int* slot, float* arena_img, int H, int W, int PH);
void g_abs(hipStream_t s, int n, const float2* src, size_t src_stride, const int* src_idx, float2* dst, size_t dst_stride, size_t elems);
void g_shift_fix(hipStream_t s, int n, const float* p, size_t p_stride, float* S, size_t s_stride, int H, int W);
void g_polar(hipStream_t s, int n, const float* S, size_t s_stride, const uint32_t* map, float* out, size_t out_stride, int H, int PD, int PC);
void g_rotate(hipStream_t s, int n, const uint8_t* arena_u8, size_t u8_stride, int u8_pitch, const float* arena_img, size_t img_stride, int img_pitch,
              const int* slot, const int* rot_tab, const int* rot_index, float* out, size_t out_stride, int H, int W);
void g_mul(hipStream_t s, int n, const float2* X, size_t x_stride, const int* x_idx, const float2* Z, size_t z_stride, const int* z_idx,
           float2* out, size_t item_stride, size_t plane_stride, size_t elems, unsigned* maxbuf);
void g_kernel(hipStream_t s, int n_items, float* planes, size_t plane_stride, size_t elems, KernelFn fn, const float* energy, unsigned* maxbuf);
void g_solve(hipStream_t s, int n, const float2* kk, size_t item_stride, size_t plane_stride, const unsigned* maxbuf, float lambda, float2* G, size_t g_stride,
             int cols, size_t elems);
int g_argmax_blocks(int rows, int cols);
void g_argmax(hipStream_t s, int n, const float* g, size_t g_stride, int rows, int cols, Partial* partials, int partial_stride,
              const int* win_row, const int* win_col, int radius, int mirror);

}  // namespace kcc

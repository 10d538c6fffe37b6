// kcc_kernels.h -- host-visible launch interface of the HIP kernels (internal to libnislam_kcc_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kcc {

// raw reduction output of one response surface (arg-max + moments), per item
struct SurfaceResult {
    double sum, sumsq;
    float  peak;
    int    idx;          // linear column-major index  (col * rows + row)
};

// per-block partial of the same
struct Partial {
    double sum, sumsq;
    float  peak;
    int    idx;
};

// kernel-function parameters (CFConfig subset), reference include/read_configs.h:15-25
struct KernelFn {
    int   type;          // 0 polynomial, 1 gaussian
    float offset;
    int   power;
    float sigma;
    float lambda;
};

// De-rotation table: for every candidate angle, the fixed-point terms of cv::warpAffine's WarpAffineInvoker
//   int adelta[W], bdelta[W], X0[H], Y0[H]      (2W + 2H ints per angle; X0/Y0 include round_delta)
// so that the source coordinate of dst (r, c) is ((X0[r] + adelta[c]) >> 5, (Y0[r] + bdelta[c]) >> 5) in 1/32 px.

// Geometry of one "plane family": real plane rows x cols (column-major: a line = one column of `rows`
// contiguous floats), spectrum stored k-major: [rows/2+1][cols] float2 (cols contiguous).
struct PlaneGeom {
    int rows, cols;      // rows even
    int hr;              // rows/2+1
};

bool fft_half_supported(int h);   // rows/2 instantiated?
bool fft_line_supported(int n);   // cols instantiated?

// pass-twiddle tables of one FFT plan, one per direction (layout: kcc_fft2.h "Twiddle table layout")
struct Tables {
    const float2* half_f;    // plan of length rows/2, forward
    const float2* half_i;    // plan of length rows/2, inverse
    const float2* halfI_f;   // plan of the spectrum-in A kernels (PlanInv), forward / inverse
    const float2* halfI_i;
    const float2* tw_full;   // W_{2h}^k, k < h: r2c / c2r split twiddles
    const float2* cols_f;    // plan of length cols, forward
    const float2* cols_i;    // plan of length cols, inverse
    const float2* colsA_f;   // plan of the single-plane B kernels (PlanAlt), forward / inverse
    const float2* colsA_i;
};

// radices of the instantiated plan for length n (np = 0 if n is not instantiated)
struct PlanDesc { int n, np, r[3], t, prime; };      // t: threads per line; prime: PR of a 16 x PR plan (kcc_fft2.h PlanPrime), else 0
PlanDesc plan_desc(int n);
PlanDesc plan_desc_inv(int n);
PlanDesc plan_desc_alt(int n);   // plan of the single-plane B kernels for line length n   // plan of the spectrum-in A-type kernels for half length n

// ---- u8 frame-store image -> f32 column-major plane of the same slot (ConvertMatToNormalizedArray on demand) ----
// f32 planes have column pitch PH >= H + 4 (rows H..H+3 repeat rows 0..3: vertical wrap taps are contiguous)
void launch_cvt_u8(hipStream_t s, int n, const uint8_t* arena_u8, size_t u8_stride, int u8_pitch, const int* d_slot, float* arena_img,
                   int H, int W, int PH);
void launch_img_wrap(hipStream_t s, float* img, int H, int W, int PH);     // refresh the wrap rows of one plane
// Camera::UndistortImage (camera.cc:92-93): cv::remap with the fixed-point maps; u8 -> u8
void launch_undistort_u8(hipStream_t s, int n, const uint8_t* d_in, uint8_t* d_out, const int16_t* map1, const uint16_t* map2, int H, int W);

// MapStitcher::AddImageToOccupancy (map_stitcher.cc:36-133): frame -> temporary cells, temporary cell -> map cell
struct StitchPose { double r00, r01, r10, r11, x, y, cx, cy; };   // RotationMatrix2D(theta), image pose, image centre (W/2, H/2)
void launch_stitch_scatter(hipStream_t s, const uint8_t* img, int H, int W, const StitchPose& P, int size, int cx0, int cy0,
                           int ncx, int ncy, int* tmp_data, int* tmp_weight);
void launch_stitch_touched(hipStream_t s, const int* tmp_weight, int n_cells, int csz, int* flags);
void launch_stitch_merge(hipStream_t s, int* data, int* weight, const int* tmp_data, const int* tmp_weight, int n, int existing);
// 2 x 2 box-filtered half-resolution image (pyramid level)
void launch_downsample_u8(hipStream_t s, int n, const uint8_t* in, uint8_t* out, int H, int W);
void launch_downsample_pyr(hipStream_t s, int steps, const uint8_t* a, const uint8_t* b, int na, int nf, int H, int W, uint8_t* const* out);
// 8-bit RGB/BGR (interleaved) -> gray with OpenCV's integer luma weights
void launch_rgb2gray(hipStream_t s, const uint8_t* rgb, uint8_t* gray, size_t npix, int bgr);

// ---- A-type: lines along `rows` (r2c / c2r), transposed spectrum access ----
// forward from a real plane: src plane index = src_idx ? src_idx[item] : item
void launch_A_fwd_plane(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float* src, size_t src_stride, int src_pitch,
                        const int* src_idx, float2* dst, size_t dst_stride);
// forward from a u8 row-major image batch (ConvertMatToNormalizedArray fused into the load); every tile is also copied
// into the u8 frame store (row pitch keep_pitch = W + 16; columns 0..15 repeated behind column W-1) when keep != null
void launch_A_fwd_u8(hipStream_t s, int n_items, PlaneGeom g, Tables t, const uint8_t* src, size_t src_stride, int src_pitch,
                     uint8_t* keep, size_t keep_stride, int keep_pitch, const int* keep_slot, float2* dst, size_t dst_stride);
// forward from the de-rotated image (RotateArray fused into the load): f32 frames (column-major, wrap rows) ...
void launch_A_fwd_rot(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float* arena_img, size_t img_stride, int img_pitch,
                      const int* img_slot, const int* rot_tab, const int* rot_index,
                      float2* dst, size_t dst_stride, float* dbg_plane = nullptr);
// ... and u8 frames (row-major, wrap columns; source bands staged in LDS)
void launch_A_fwd_rot8(hipStream_t s, int n_items, PlaneGeom g, Tables t, const uint8_t* arena_u8, size_t img_stride, int img_pitch,
                       const int* img_slot, const int* rot_tab, const int* rot_index,
                       float2* dst, size_t dst_stride, float* dbg_plane = nullptr);
// geometry of the forward A kernels for half length hh (thread (line, j), j < mf, owns first-pass points j + q*mf, q < rf)
// qs_opts: the segment sizes (first-pass points per thread staged at once) the polar kernel is instantiated for, largest first
struct FwdGeom { int lines, threads, rf, mf; size_t lds_bytes; int qs_opts[3]; };
FwdGeom fwd_geom(int hh);
// LDS box geometry of the u8 de-rotation (a band of band_rows dst rows x 16 columns -> a box of box_rows x pitch bytes)
struct Rot8Geom { int band_rows, bands, box_rows, pitch, lds_bytes; };
Rot8Geom rot8_geom(int hh);
// gather tables of the polar forward kernel (built on the host by build_polar_plan, kcc_tables.cpp)
struct PolarPlan {
    const uint32_t* chunks;      // staging chunks: source offset of 16 consecutive floats, per tile and segment
    const int* seg_first;        // [tiles * nseg + 1] first descriptor of every (tile, segment)
    const uint4* pts;            // [tiles][rf][lines*threads] entries of a thread's first-pass point (two samples):
                                 //   x/z: LDS float offset of tap column sx :16 | fx:5 | fy:5,  y/w: LDS float offset of column sx+1
    int qs;                      // first-pass points per thread and segment (one of FwdGeom::qs_opts)
    size_t lds_bytes;            // staging bytes of the largest segment
};
// forward from polar(S), S = shifted zero-bordered planes [W+1][H+2] (gather fused into the load); g = polar geometry
void launch_A_fwd_polar(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float* S, size_t s_stride,
                        int H, int W, const PolarPlan& pp, float2* dst, size_t dst_stride, float* dbg_plane = nullptr);
// inverse to a real plane, scaled by 1/(rows*cols)
void launch_A_inv_real(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float2* src, size_t src_stride,
                       float* dst, size_t dst_stride);
// inverse, /(rows*cols), written fftshift-ed into the zero-bordered planes S (column pitch rows+2).
// need_cols > 0: the plane is real and even (the zero-phase image) and only its columns |c| <= need_cols are consumed:
// transform the column tiles covering [0, min(W/2, need_cols)] and write each column's mirror too; 0 = the whole plane.
// fix_zero (only with need_cols > 0): RemoveZeroComponent applied on the way -- the plane launch_fix_zero would leave
void launch_A_inv_shifted(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float2* src, size_t src_stride,
                          float* S, size_t s_stride, int need_cols = 0, bool fix_zero = false);
// RemoveZeroComponent patch of the shifted planes (one workgroup per item)
void launch_fix_zero(hipStream_t s, int n_items, float* S, size_t s_stride, int H, int W);
// item order of the A / B kernels launched next by this host thread: 0 = front to back, 1 = back to front (speed only)
void set_launch_reverse(int rev);
void launch_make_shifted(hipStream_t s, const float* p, float* S, int H, int W);
// inverse -> /(rows*cols) -> kernel function -> running max|k| -> forward; in place on `buf`.
// buf holds 2 planes per item (zz then xz), maxbuf 2 uints per item (float bits, zeroed by caller),
// energy 2 floats per item (gaussian: sum|X|^2, sum|Z|^2 over the half spectrum).
void launch_A_inv_kernel_fwd(hipStream_t s, int n_items, PlaneGeom g, Tables t, float2* buf, size_t item_stride,
                             size_t plane_stride, KernelFn fn, unsigned* maxbuf, const float* energy,
                             int plane_first = 0, int n_planes = 2, bool zz_half = false);
// inverse -> /(rows*cols) -> arg-max + moments partials
void launch_A_inv_argmax(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float2* src, size_t src_stride,
                         Partial* partials, int partial_stride);
// the same with the arg-max candidates restricted to a cyclic (2*radius+1)^2 window around (win_row[item], win_col[item])
// (mirror: also around row + rows/2); the PSR moments still cover the whole surface
void launch_A_inv_argmax_win(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float2* src, size_t src_stride,
                             Partial* partials, int partial_stride, const int* win_row, const int* win_col, int radius, int mirror);
int  argmax_blocks(PlaneGeom g);
// running max of the kernel planes: kernel_fwd files one part per wave and column tile in maxbuf[item][plane][KCC_MAXPARTS]
// (float bits); the ridge solve folds them.  kfwd_parts: parts per plane (half: the Hermitian-half zz plane)
enum { KCC_MAXPARTS = 1024 };
int  kfwd_parts(PlaneGeom g, bool half);
// column counts of the symmetry shortcuts (the stage profiler prices the launches on them):
// zz_half_columns: columns [0, W/2] rounded up to whole kernel_fwd tiles -- what is kept of the Hermitian Kzz kernel plane;
// shifted_columns: columns [0, min(W/2, need)] rounded up to whole tiles -- what the even-half inverse row pass of the zero-phase image reads
int  zz_half_columns(PlaneGeom g);
int  shifted_columns(PlaneGeom g, int need);

// ---- B-type: contiguous spectrum lines along `cols` ----
void launch_B_fwd(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float2* src, size_t src_stride,
                  float2* dst_base, size_t dst_stride, const int* dst_slot);
void launch_B_inv(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float2* src, size_t src_stride,
                  float2* dst, size_t dst_stride);
// F = fwd(src) -> dst (slot), |F| -> inverse -> tmp
void launch_B_fwd_abs_inv(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float2* src, size_t src_stride,
                          float2* dstF_base, size_t dstF_stride, const int* dst_slot,
                          float2* tmp, size_t tmp_stride, int need_cols = 0);   // need_cols: as launch_A_inv_shifted
// out planes (item_stride apart, plane_stride between zz and xz): inv(|Z|^2), inv(X conj Z).
// X: x_fwd ? fwd(xsrc line) : xsrc line.  X plane index = x_idx ? x_idx[item] : item.
void launch_B_mul_inv(hipStream_t s, int n_items, PlaneGeom g, Tables t, bool x_fwd,
                      const float2* xsrc, size_t x_stride, const int* x_idx,
                      const float2* zsrc, size_t z_stride, const int* z_idx,
                      float2* out, size_t item_stride, size_t plane_stride, unsigned* maxbuf_zero,
                      float2* xstore = nullptr, size_t xstore_stride = 0, const int* xstore_slot = nullptr, bool zz_half = false);   // x_fwd: also keep X
// G = T/(Kzz/Mzz + lambda) * Kxz/Mxz with Kzz = fwd(buf plane 0), Kxz = fwd(buf plane 1); out = inv(G)
void launch_B_solve_inv(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float2* buf, size_t item_stride,
                        size_t plane_stride, const unsigned* maxbuf, float lambda, float2* out, size_t out_stride, bool zz_half = false);

// ---- per-keyframe Kzz cache (SURVEY 8d "with Kzz cached"): the zz / xz halves of the kernel stage on their own
// plane 0 := inv(|Z|^2)
void launch_B_zz_inv(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float2* zsrc, size_t z_stride, const int* z_idx,
                     float2* out, size_t item_stride, unsigned* maxbuf_zero);
// plane 1 := inv(X conj Z)
void launch_B_mul_inv_x(hipStream_t s, int n_items, PlaneGeom g, Tables t, bool x_fwd,
                        const float2* xsrc, size_t x_stride, const int* x_idx,
                        const float2* zsrc, size_t z_stride, const int* z_idx,
                        float2* out, size_t item_stride, size_t plane_stride, unsigned* maxbuf_zero,
                      float2* xstore = nullptr, size_t xstore_stride = 0, const int* xstore_slot = nullptr);   // x_fwd: also keep X
// G from the cached Kzz spectrum / max of slot z_idx[item] and Kxz = fwd(buf plane 1)
void launch_B_solve_cached(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float2* buf, size_t item_stride,
                           size_t plane_stride, const unsigned* maxbuf, const float2* kzz, size_t kzz_stride,
                           const unsigned* mzz, const int* z_idx, float lambda, float2* out, size_t out_stride);
void launch_store_mzz(hipStream_t s, int n, PlaneGeom g, const unsigned* maxbuf, const int* slots, unsigned* mzz);

// half-spectrum energies for the gaussian kernel: energy[item] = {sum|X|^2, sum|Z|^2}
void launch_energy(hipStream_t s, int n_items, PlaneGeom g, const float2* xsrc, size_t x_stride, const int* x_idx,
                   const float2* zsrc, size_t z_stride, const int* z_idx, float* energy);

// reduce partials -> SurfaceResult per item; rot_index (optional): de-rotation table index of the n_hyp
// translation items of each pair, variant(h) * PD + arg-max row
void launch_finalize(hipStream_t s, int n_items, const Partial* partials, int partial_stride, int n_partials,
                     SurfaceResult* out, int* rot_index, int n_hyp, int PD, SurfaceResult* host_out = nullptr);   // host_out: pinned mirror of out (or null)


// residual statistics of a batch call from its raw surface results: stats[4] = [sum PSR_t, sum PSR_r, sum |t|^2, count]
void launch_residual_stats(hipStream_t s, const SurfaceResult* rot, const SurfaceResult* trans, int n, int n_hyp,
                           int H, int W, int PD, int PC, double* stats);
void launch_stats_sum(hipStream_t s, const double* parts, int n_parts, double* total);

// window centres of a level from the surface results of the level above (coarse-to-fine chaining on the device)
void launch_predict_windows(hipStream_t s, const SurfaceResult* rot, const SurfaceResult* trans, int n, int PDu, int PCu, int Hu, int Wu,
                            int PD, int PC, int H, int W, int* wrr, int* wrc, int* wtr, int* wtc);

// layout conversion for export / import: reference [cols][hr] <-> internal [hr][cols]
void launch_transpose_c(hipStream_t s, const float2* src, float2* dst, int src_rows, int src_cols);

}  // namespace kcc

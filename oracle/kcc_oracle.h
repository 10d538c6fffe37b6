/*
 * kcc_oracle.h -- CPU restatement (the parity ORACLE) of NI-SLAM's KCC front end.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing on the product path (ni-slam_amd/) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it -- as the checker / reported baseline.
 *
 * PARITY UNPINNED: the reference (sair-lab/ni-slam) ships no tests, fixtures or
 * golden vectors, and cannot be built in this image (needs FFTW3f, Eigen3,
 * OpenCV 4.2 -- all absent).  This file restates the reference's algorithm
 *   /root/reference/src/correlation_flow.cc:37-243
 *   /root/reference/src/utils.cc:110-175
 *   /root/reference/include/circ_shift.h:131-154,238-244
 * in dependency-free C (own float32 mixed-radix FFT standing in for FFTW3f,
 * literal restatements of the OpenCV 4.2 warpPolar/warpAffine/remap fixed-point
 * bilinear paths, Eigen's column-major maxCoeff tie-break).  It is pinned by
 * (1) analytic known-answer tests and (2) an independent numpy/scipy
 * restatement (oracle/np_restatement.py) whose outputs are committed under
 * tests/golden/.
 *
 * Array conventions (same as the reference):
 *   real image / plane  : Eigen::ArrayXXf, column-major rows x cols  -> a[c*rows + r]
 *   spectrum            : Eigen::ArrayXXcf column-major (rows/2+1) x cols,
 *                         interleaved (re,im)                        -> s[c*(rows/2+1) + k]
 *   u8 image            : cv::Mat row-major rows x cols              -> m[r*cols + c]
 */
#ifndef KCC_ORACLE_H
#define KCC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } ora_cf32;

/* mirrors CFConfig, /root/reference/include/read_configs.h:15-25 */
typedef struct {
    int   width;
    int   height;
    float lambda;
    int   kernel;            /* 0 polynomial, 1 gaussian */
    float sigma;
    float offset;
    int   power;
    int   rotation_divisor;  /* polar rows (angle bins), 720 */
    int   rotation_channel;  /* polar cols (radius bins), 480 */
} ora_config;

typedef struct ora_ctx ora_ctx;

/* per-call debug taps (all optional integers / floats the tests compare) */
typedef struct {
    int   rot_row, rot_col;        /* arg-max of the rotation surface (720x480)            */
    int   trans_row[2], trans_col[2]; /* arg-max of the translation surface, hypothesis 0/1 */
    float psr_rot;
    float rot_peak, rot_mirror;    /* g at the arg-max and at its 180-degree mirror row (row +- PD/2): the polar
                                      source is point-symmetric, so these two are a near-tie decided by FFT noise */
    float psr_trans[2];
    float degree_used[2];          /* de-rotation angle fed to RotateArray (= -degree)     */
    float degree_final;
    int   chosen;                  /* 0 = orig, 1 = veri (+180)                             */
    int   n_hyp;
    int   rot_forced;              /* 1 if the rotation arg-max was imposed by ora_force_rotation (test hook)     */
} ora_pose_debug;

/* CorrelationFlow::CorrelationFlow  correlation_flow.cc:37-44 */
ora_ctx* ora_create(const ora_config* cfg, int image_height, int image_width);
void     ora_destroy(ora_ctx* ctx);
int      ora_rows(const ora_ctx*);   /* H  */
int      ora_cols(const ora_ctx*);   /* W  */

/* ConvertMatToNormalizedArray  utils.cc:110-118 */
void ora_normalize_u8(const uint8_t* img_rowmajor, int rows, int cols, float* out_colmajor);

/* CorrelationFlow::FFT / IFFT  correlation_flow.cc:53-77 (any even rows) */
void ora_fft (ora_ctx* ctx, const float* x, int rows, int cols, ora_cf32* xf);
void ora_ifft(ora_ctx* ctx, const ora_cf32* xf, int hrows, int cols, float* x);

/* RemoveZeroComponent :79-87, fftshift circ_shift.h:238-244, polar :228-236 */
void ora_remove_zero(const float* x, int rows, int cols, float* y);
void ora_fftshift(const float* x, int rows, int cols, float* y);
void ora_polar(ora_ctx* ctx, const float* x /*H x W*/, float* out /*PD x PC*/);

/* RotateArray utils.cc:154-161 ; WarpArray utils.cc:163-171 */
void ora_rotate(const float* x, int rows, int cols, float degree, float* out);
void ora_warp  (const float* x, int rows, int cols, float tx, float ty, float degree, float* out);

/* NormalizeDegree utils.cc:173-175 */
double ora_normalize_degree(double angle_degree);
/* polynomial kernel's power (correlation_flow.cc:213,223 `Array::pow(int)`): 0 (default) double pow rounded to float, 1 powf with
 * the exponent promoted to float (oracle/RECALLED.md row 16, an open question: the switch is for whoever can pin it) */
void ora_set_pow_mode(int mode);
int  ora_get_pow_mode(void);

/* CorrelationFlow::ComputeIntermedium :89-95 */
void ora_intermedium(ora_ctx* ctx, const float* image, ora_cf32* fft_result, ora_cf32* fft_polar);

/* CorrelationFlow::EstimateTrans :145-179.  which=0: translation (H x W, target_fft),
 * which=1: rotation (PD x PC, target_rotation_fft).  Returns PSR; trans[2] as the reference;
 * row/col receive the raw arg-max; g_out (optional) receives the response surface.
 * Returns NaN and sets *err=-1 for an invalid kernel id (reference throws invalid_argument). */
float ora_estimate_trans(ora_ctx* ctx, const ora_cf32* last_fft, const ora_cf32* cur_fft,
                         int which, double trans[2], int* row, int* col, float* g_out, int* err);

/* CorrelationFlow::GetInfo :238-243 */
float ora_get_info(const float* g, long n, float response);

/* CorrelationFlow::ComputePose :97-143.  faithful!=0 also executes the dead `rectify`
 * work of :141 (timing baseline only; never changes outputs).  Returns 0, or -1 for an
 * invalid kernel id. */
int ora_compute_pose(ora_ctx* ctx, const ora_cf32* last_fft_result, const float* image,
                     const ora_cf32* last_fft_polar, const ora_cf32* fft_polar,
                     int not_large_rotation, int faithful,
                     double pose[3], double info[3], ora_pose_debug* dbg);

/* Convenience for tests / the CPU baseline: n independent pairs (key image, current image),
 * both u8 row-major.  Per pair: ComputeIntermedium(key) [untimed part of a tracker's life, done
 * here so the call is self-contained], then the timed unit of SURVEY 8(d):
 * ComputeIntermedium(cur) + ComputePose(key, cur, not_large_rotation).
 * nthreads>1 runs pairs in parallel with OpenMP (one ctx clone per thread).
 * seconds_unit (optional) receives the wall time of the timed units only. */
int ora_track_pairs(const ora_config* cfg, int H, int W, int n,
                    const uint8_t* key_imgs, const uint8_t* cur_imgs,
                    int not_large_rotation, int faithful, int nthreads,
                    double* poses /*n x 3*/, double* infos /*n x 3*/,
                    ora_pose_debug* dbgs /*n or NULL*/, double* seconds_unit);


/* ---- camera undistortion: the step right before the path ----------------------------------------------
 * /root/reference/src/camera.cc:45-47 (getOptimalNewCameraMatrix(alpha=0) + initUndistortRectifyMap(CV_16SC2))
 * and :92-93 (cv::remap(INTER_LINEAR), u8, BORDER_CONSTANT 0).  OpenCV 4.2 semantics restated from the
 * published algorithm [recalled, could not be validated against a real OpenCV here].
 * K = {fx, cx, fy, cy}; D = {k1, k2, p1, p2, k3}. */
void ora_optimal_new_camera_matrix(const double K[4], const double D[5], int width, int height, double newK[4]);
void ora_undistort_maps(const double K[4], const double D[5], const double newK[4], int width, int height,
                        int16_t* map1 /* H*W*2: (sx, sy) */, uint16_t* map2 /* H*W: fy*32 + fx */);
void ora_remap_u8(const uint8_t* src_rowmajor, int width, int height, const int16_t* map1, const uint16_t* map2,
                  uint8_t* dst_rowmajor);

/* ---- coarse-to-fine extension (BASELINE config 3; no reference counterpart): restrict the arg-max of the next
 * ora_estimate_trans / ora_compute_pose calls to cyclic (2*radius+1)^2 windows (rotation surface: also around the
 * mirror row); radius < 0 switches it off.  2x2 box down-sampling of a u8 image. */
/* test hook for near-ties of the rotation surface: the next ora_compute_pose calls take (row, col) as the rotation
 * arg-max instead of searching (row < 0 switches it off); rot_peak then reports g at that position. */
void ora_force_rotation(ora_ctx* ctx, int row, int col);
void ora_set_window(ora_ctx* ctx, int rot_row, int rot_col, int trans_row, int trans_col, int radius);
void ora_downsample_u8(const uint8_t* src_rowmajor, int width, int height, uint8_t* dst_rowmajor);

#ifdef __cplusplus
}
#endif
#endif

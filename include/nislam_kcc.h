/*
 * nislam_kcc.h -- C ABI of libnislam_kcc_hip.so: the MI355X (gfx950) implementation of
 * NI-SLAM's Kernel-Cross-Correlator front end.
 *
 * The reference has no FFI: its de-facto boundary is the public interface of class
 * CorrelationFlow (reference include/correlation_flow.h:8-33), called from
 * MapBuilder (src/map_builder.cc:23,72-75,127-131) and LoopClosure (src/loop_closure.cc:55-59).
 * Each entry point below names the reference interface it replaces.  The header-only C++
 * adaptor ni-slam_amd/correlation_flow_hip.h re-creates that class on top of this ABI.
 *
 * Conventions
 *   - plain C types only; all functions return NIK_OK (0) or a negative nik_status; nothing
 *     throws across the ABI.  nik_last_error() returns a human-readable message.
 *   - host arrays use the reference's layouts:
 *       image     : Eigen::ArrayXXf  column-major H x W           -> a[c*H + r], values in [0,1]
 *       spectrum  : Eigen::ArrayXXcf column-major (H/2+1) x W     -> s[c*(H/2+1) + k], (re,im) floats
 *       polar sp. : Eigen::ArrayXXcf column-major (PD/2+1) x PC
 *       u8 image  : cv::Mat CV_8UC1 row-major H x W               -> m[r*stride + c]
 *   - "dev" entry points take DEVICE pointers (HBM-resident inputs) and are stream-ordered.
 *   - a nik_frame is a slot in the context's device-resident keyframe store (the analogue of
 *     reference Frame's _fft_result/_fft_polar members, include/frame.h:32-39).
 */
#ifndef NISLAM_KCC_H
#define NISLAM_KCC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    NIK_OK = 0,
    NIK_ERR_INVALID_ARG = -1,
    NIK_ERR_UNSUPPORTED_SIZE = -2,   /* FFT length not instantiated / odd dimension */
    NIK_ERR_INVALID_KERNEL = -3,     /* reference: throw std::invalid_argument("Received invalid kernel type"), correlation_flow.cc:167-168 */
    NIK_ERR_HIP = -4,                /* a HIP runtime call failed (no device, OOM, launch failure) */
    NIK_ERR_CAPACITY = -5,           /* batch larger than max_batch / frame slot out of range */
    NIK_ERR_NOT_READY = -6           /* frame slot holds no spectra yet */
} nik_status;

/* mirrors CFConfig, reference include/read_configs.h:15-25 (same field order and types) */
typedef struct {
    int   width;
    int   height;
    float lambda;
    int   kernel;             /* 0 polynomial, 1 gaussian; anything else -> NIK_ERR_INVALID_KERNEL at pose time */
    float sigma;
    float offset;
    int   power;
    int   rotation_divisor;   /* polar rows (angle bins) */
    int   rotation_channel;   /* polar cols (radius bins) */
} nik_config;

typedef struct nik_ctx nik_ctx;
typedef int32_t nik_frame;    /* slot index in [0, max_frames) */

/* raw per-pair results of ComputePose (reference correlation_flow.cc:97-143) */
typedef struct {
    double pose[3];           /* (x px, y px, theta rad)  == reference `pose` out-param            */
    double info[3];           /* (PSR_t, PSR_t, PSR_r)    == reference return value                 */
    int32_t rot_row, rot_col; /* arg-max of the rotation surface (column-major first max)           */
    int32_t trans_row[2], trans_col[2];   /* arg-max of the translation surface, hypothesis 0 / 1   */
    float   psr_rot, psr_trans[2];
    float   degree_final;     /* degrees, after the >180 fold of :134                               */
    int32_t chosen;           /* 0 = `orig`, 1 = `veri` (+180) hypothesis (:121-131)                */
    int32_t n_hyp;            /* 1 (not_large_rotation) or 2                                        */
} nik_pose_result;

/* ---- lifetime -------------------------------------------------------------------------- */

/* replaces CorrelationFlow::CorrelationFlow(CFConfig&, double& H, double& W) (correlation_flow.cc:37-44).
 * As in the reference, cfg->height/width are overridden by image_height/image_width.
 * max_batch  : largest number of frame pairs (or loop-closure candidates) per batched call.
 * max_frames : capacity of the device keyframe store.
 * device     : HIP device ordinal. */
int  nik_create(const nik_config* cfg, int image_height, int image_width,
                int max_batch, int max_frames, int device, nik_ctx** out);
void nik_destroy(nik_ctx* ctx);
const char* nik_last_error(const nik_ctx* ctx);      /* ctx may be NULL: last create() error */
/* geometry queries (H, W, PD, PC, max_batch, max_frames) */
int  nik_get_dims(const nik_ctx* ctx, int dims[6]);
/* Which plane families of the context run the any-size kernel family (kcc_generic.hip): a MASK -- bit 0 (1) the image family
 * (H x W), bit 1 (2) the polar family (PD x PC); 0 = both on the tiled kernels, 3 = both on the any-size kernels.  (Test it
 * with `!= 0` or per bit, not `== 1`.)  A family is any-size when its geometry is outside the tiled kernels' instantiated set --
 * half-rows {30,60,120,224,240,256,360,384,600}, lines {80,160,320,448,480,512,640,752,848,1024,1280,1600}, W and PC multiples of 16,
 * aspect within 2:1 -- or when $NIK_GENERIC forces it (1 both, 2 polar only, 4 image only).  Same results either way; the
 * any-size family is slower. */
int  nik_is_generic(const nik_ctx* ctx);
/* first of the context's streams (a hipStream_t, returned as void*) */
void* nik_stream(const nik_ctx* ctx);
int   nik_device(const nik_ctx* ctx);                        /* HIP device ordinal the context lives on */
/* A batched call is split over up to `n` concurrent HIP streams ("lanes", default 3 or $NIK_STREAMS, max 4);
 * returns the number now active.  Outputs do not depend on it. */
int  nik_set_streams(nik_ctx* ctx, int n);
/* A batched call of more than streams * `pairs` pairs is cut into chunks of at most `pairs` pairs, dealt to the streams in
 * turn (0: one chunk per stream; default $NIK_CHUNK or the library's tuned value).  A chunk's intermediates stay in the
 * 256 MiB Infinity Cache between the kernel that writes them and the one that reads them.  Outputs do not depend on it.
 * Returns the previous value. */
int  nik_set_chunk(nik_ctx* ctx, int pairs);
int  nik_synchronize(nik_ctx* ctx);
/* Per-keyframe cache of the key-side kernel Kzz = FFT(kernel(IFFT(|Z|^2))) and its max (both families), built the
 * first time a slot is used as a key and dropped when the slot is rewritten.  The reference recomputes Kzz in every
 * EstimateTrans call (correlation_flow.cc:160,164) although it depends on the keyframe only; outputs are identical.
 * Default off ($NIK_KZZ_CACHE=1 turns it on at creation); costs (H/2+1)*W + (PD/2+1)*PC complex per slot. */
int  nik_set_kzz_cache(nik_ctx* ctx, int enable);

/* ---- Camera undistortion: the step right before the path (MapBuilder::AddNewInput, map_builder.cc:31-33) ----
 * nik_camera_maps     host only, no GPU: the map construction of Camera::Camera (camera.cc:46-47) --
 *                     getOptimalNewCameraMatrix(K, D, size, alpha 0, size) and initUndistortRectifyMap(K, D, I,
 *                     new_K, size, CV_16SC2): K = {fx, cx, fy, cy}, D = {k1, k2, p1, p2, k3};  outputs new_K (same
 *                     order), map1[H*W*2] = integer source (x, y), map2[H*W] = fy*32 + fx (1/32 px fractions).
 * nik_set_undistort   installs the maps (host pointers, copied; NULL, NULL removes them).  While installed, EVERY u8
 *                     entry point (nik_intermedium_u8 / _batch_dev, nik_track_batch_dev, nik_tracker_push_*)
 *                     takes the RAW camera frame: Camera::UndistortImage (camera.cc:92-93, cv::remap INTER_LINEAR,
 *                     border 0) is fused into the u8 -> f32 conversion.  f32 entry points are unaffected.
 * nik_undistort_dev   Camera::UndistortImage itself for n device images (u8 row-major -> u8), e.g. for a stitcher. */
int  nik_camera_maps(const double K[4], const double D[5], int width, int height, double new_K[4],
                     int16_t* map1, uint16_t* map2);
int  nik_set_undistort(nik_ctx* ctx, const int16_t* map1, const uint16_t* map2);
int  nik_undistort_dev(nik_ctx* ctx, int n, const uint8_t* d_raw, uint8_t* d_out);

/* ---- ComputeIntermedium (correlation_flow.cc:89-95) ------------------------------------- */

/* replaces MapBuilder::ComputeFFTResult (map_builder.cc:72-75): ConvertMatToNormalizedArray
 * (utils.cc:110-118) + ComputeIntermedium.  `gray` is a host CV_8UC1 image, row stride in bytes. */
int nik_intermedium_u8(nik_ctx* ctx, const uint8_t* gray, int stride, nik_frame dst);
/* replaces CorrelationFlow::ComputeIntermedium(const ArrayXXf&, ArrayXXcf&, ArrayXXcf&): host f32 image,
 * column-major H x W.  Results stay on the device in slot `dst`; fetch them with nik_frame_export. */
int nik_intermedium_f32(nik_ctx* ctx, const float* image_colmajor, nik_frame dst);

/* ---- host frames -> device on the context's own UPLOAD stream (never a compute lane) ---------------------------------
 * The reference's caller hands over host images (main.cpp:55-65, map_builder.cc:30-33: Dataset::GetImage -> AddNewInput).  A
 * streamed caller uploads the next window of frames while the current one is registered:
 *   ticket = nik_upload_u8_async(ctx, n, frames, stride, frame_stride, d_dst)   copies enqueued, returns at once (>= 0)
 *   nik_upload_fence(ctx, ticket)        every compute lane waits ON THE DEVICE for that upload (call it before the *_dev
 *                                        entry point that reads d_dst; uploads enqueued later do not delay that work)
 *   nik_upload_wait(ctx)                 host-side: the source buffers of all uploads so far may be reused
 *   nik_upload_after_compute(ctx)        the upload stream waits ON THE DEVICE for everything enqueued on the compute lanes so
 *                                        far (call it before an upload that overwrites a device buffer earlier *_dev calls read)
 * A source in pinned memory (hipHostMalloc / hipHostRegister, e.g. a camera driver's DMA ring) is read by the copy engine
 * directly; a pageable one is staged through two pinned buffers of the context.  At most four uploads may be un-fenced.
 * nik_dev_malloc / nik_dev_free: device buffers on the context's GPU for callers that do not link the HIP runtime. */
int nik_upload_u8_async(nik_ctx* ctx, int n, const uint8_t* gray, int stride, size_t frame_stride, uint8_t* d_dst);
int nik_upload_fence(nik_ctx* ctx, int ticket);
int nik_upload_wait(nik_ctx* ctx);
int nik_upload_after_compute(nik_ctx* ctx);
int nik_dev_malloc(nik_ctx* ctx, size_t bytes, void** out);
int nik_dev_free(nik_ctx* ctx, void* p);
/* batched, device-resident inputs: n u8 images [n][H][W] (row-major, tightly packed) already in HBM. */
int nik_intermedium_batch_dev(nik_ctx* ctx, int n, const uint8_t* d_gray, const nik_frame* dst);

/* copy a slot's contents to host arrays in the reference layouts (any pointer may be NULL):
 * the image (reference Frame::_frame), fft_result, fft_polar. */
int nik_frame_export(nik_ctx* ctx, nik_frame f, float* image_colmajor,
                     float* fft_result /*2*(H/2+1)*W floats*/, float* fft_polar /*2*(PD/2+1)*PC floats*/);
/* load host arrays (reference layouts) into a slot -- lets spectra computed elsewhere (e.g. by the
 * reference itself) be used as keys: the ComputePose(const ArrayXXcf& last_fft_result, ...) arguments. */
int nik_frame_import(nik_ctx* ctx, nik_frame f, const float* image_colmajor,
                     const float* fft_result, const float* fft_polar);

/* ---- ComputePose (correlation_flow.cc:97-143) -------------------------------------------- */

/* replaces CorrelationFlow::ComputePose(last_fft_result, image, last_fft_polar, fft_polar, pose,
 * not_large_rotation) for one pair: `key` supplies last_fft_result/last_fft_polar, `cur` supplies
 * image/fft_polar.  pose/info as the reference.  res (optional) receives the raw arg-max indices. */
int nik_pose(nik_ctx* ctx, nik_frame key, nik_frame cur, int not_large_rotation,
             double pose[3], double info[3], nik_pose_result* res);
/* n independent pairs (MapBuilder::Tracking over a batch, map_builder.cc:127-131). */
int nik_pose_batch(nik_ctx* ctx, int n, const nik_frame* keys, const nik_frame* curs,
                   int not_large_rotation, nik_pose_result* res);
/* the same without waiting: res is final after nik_synchronize (or once as many further calls as the call depth -- 2 by default, nik_set_call_depth -- have been enqueued on every stream) */
int nik_pose_batch_async(nik_ctx* ctx, int n, const nik_frame* keys, const nik_frame* curs,
                         int not_large_rotation, nik_pose_result* res);
/* the results of ONE asynchronous batch: returns when every in-flight call that writes into res[0, n) has finished and its
 * results are final (calls of a stream retire in order, so whatever that stream was given earlier is finished too); calls
 * enqueued later keep running.  With it a caller keeps several batches in flight and consumes them in order (kcc_tracker.cpp's
 * look-ahead batches) instead of draining the context with nik_synchronize. */
int nik_wait_results(nik_ctx* ctx, const nik_pose_result* res, int n);
/* on: successive batched calls start on successive streams of the context (by default every call starts on the first one, so
 * small calls queue behind each other); outputs are unchanged.  Set by nik_tracker_create for its context. */
int nik_set_lane_rotation(nik_ctx* ctx, int on);

/* The benchmark unit of SURVEY.md 8(d): for each of n pairs,
 *   ComputeIntermedium(current image) + ComputePose(key, current, not_large_rotation).
 * d_gray: n u8 images in HBM; keys[i]: slot holding pair i's keyframe spectra; cur_dst[i]: slot that
 * receives the current frame's image + spectra (so it can become a key later, map_builder.cc:99-106).
 * Asynchronous on nik_stream(); results land in `res` (host) after nik_synchronize(), or call with
 * sync=1 to block.  res may be pageable host memory; with sync=0 it is WRITTEN LATER (when a later call of the same
 * stream retires this one, or at nik_synchronize): it must stay allocated until then. */
int nik_track_batch_dev(nik_ctx* ctx, int n, const uint8_t* d_gray, const nik_frame* keys,
                        const nik_frame* cur_dst, int not_large_rotation, nik_pose_result* res, int sync);

/* replaces the per-candidate loop of LoopClosure::FindLoopClosure (loop_closure.cc:40-66): runs
 * ComputePose(cand_i, query, not_large_rotation=false) for n candidates and returns every result
 * plus the index of the candidate with the largest response.sum() (first such in `cands` order),
 * or -1 when n == 0.  The frame-gap / distance filters (:43-53) stay with the caller.  n may exceed max_batch
 * (candidates are processed max_batch at a time). */
int nik_match(nik_ctx* ctx, nik_frame query, int n, const nik_frame* cands,
              int* best, nik_pose_result* res /* n entries, may be NULL */, nik_pose_result* best_res);

/* Extension (no reference counterpart): two-stage search.  Stage 1 ranks all n candidates by
 * the rotation-stage PSR (info[2]; needs only the cached polar spectra), stage 2 runs the reference's full
 * ComputePose on the k best and applies nik_match's selection rule to them.  shortlist (optional, k ints)
 * receives the candidate indices that reached stage 2. */
int nik_match_topk(nik_ctx* ctx, nik_frame query, int n, const nik_frame* cands, int k, int* best,
                   nik_pose_result* best_res, int* shortlist);

/* Extension (config 4): interleaved 8-bit RGB (bgr=0) or BGR (bgr=1) images in HBM -> 8-bit gray with OpenCV's
 * integer luma weights (R*4899 + G*9617 + B*1868 + 8192) >> 14; n images of H x W.  The reference loads gray. */
int nik_rgb_to_gray_dev(nik_ctx* ctx, int n, const uint8_t* d_rgb, int bgr, uint8_t* d_gray);
/* the same without returning to the host: ordered on the device before every call enqueued later (and after everything
 * enqueued before); results are ready when those later calls are */
int  nik_rgb_to_gray_async(nik_ctx* ctx, int n, const uint8_t* d_rgb, int bgr, uint8_t* d_gray);

/* ---- sequence driver: the tracking subset of MapBuilder ---------------- */

/* Camera intrinsics after undistortion (reference Camera::_new_K, _height, _extrinsics; src/camera.cc:20-75)
 * and KeyframeSelectionConfig (include/read_configs.h:27-32).  Undistortion itself (cv::remap) is not done
 * here: frames are expected undistorted. */
typedef struct {
    double fx, fy, cx, cy;
    double height;                 /* camera height above the ground plane */
    double extrinsics[9];          /* row-major 3x3, applied to (x, y, theta) as the reference does (camera.cc:207) */
    double max_distance, max_angle, lower_response_thr, upper_response_thr;
} nik_tracker_config;

typedef struct nik_tracker nik_tracker;

typedef struct {
    int32_t frame_id;              /* MapBuilder::_frame_id of this input                                     */
    int32_t inserted;              /* AddNewInput's return value: the frame became a keyframe                 */
    int32_t good_tracking;         /* Tracking(): PSR_t and PSR_r above lower_response_thr                    */
    int32_t key_frame_id;          /* frame id of the keyframe this frame was registered against (-1: first)  */
    nik_frame slot;                /* device slot holding the frame's spectra if it was inserted, else -1     */
    double  response[3];           /* ComputePose's return value                                              */
    double  cf_pose[3];            /* _current_cf_pose (image plane, pixels)                                  */
    double  robot_pose[3];         /* _current_pose                                                           */
    double  distance;              /* _distance: accumulated travel of the keyframes so far (SetFrameDistance) */
    int32_t optimized;             /* 1: this keyframe triggered CheckAndOptimize and the pose graph was optimised;
                                      cf_pose / robot_pose are the values after UpdateValueAfterLoop                */
    int32_t reserved_;
} nik_track_output;

/* replaces MapBuilder::MapBuilder's tracking members (map_builder.cc:18-28); the tracker borrows ctx (which must outlive it)
 * and configures it for its batches: nik_set_lane_rotation(ctx, 1), nik_set_call_depth(ctx, 4). */
int  nik_tracker_create(nik_ctx* ctx, const nik_tracker_config* cfg, nik_tracker** out);
void nik_tracker_destroy(nik_tracker* t);
/* replaces MapBuilder::AddNewInput (map_builder.cc:30-70) minus undistortion / map / loop closure, for n
 * consecutive frames already in HBM (u8, [n][H][W]).  Frames are registered speculatively against the current
 * keyframe in one batch and re-registered after every keyframe switch, so the outputs are exactly those of n
 * sequential calls.  n <= max_batch of the context.
 * Look-ahead batches: the coming keyframes are guessed from the history of their gaps, and asynchronous pose batches planned
 * along that chain ($NIK_TRK_DEPTH of them, default 2, of at most $NIK_TRK_FLIGHT pairs, default max_batch) stay in flight --
 * also across calls, over the frames of windows already handed to nik_tracker_prefetch_dev -- while the previous batch's
 * results are applied.  A result is used only if its key is the frame the rule above really made the keyframe; a wrong guess
 * wastes its GPU work, never changes an output.  The `out` array of a call is complete when the call returns. */
int  nik_tracker_push_dev(nik_tracker* t, int n, const uint8_t* d_gray, nik_track_output* out);
/* diagnostics of the look-ahead batches: [guesses held, guesses failed, batches enqueued, registrations enqueued, registrations
 * consumed, registrations of batches still in flight when a guess failed, 0, 0] (the first three: nik_tracker_speculation) */
int  nik_tracker_stats(const nik_tracker* t, long out[8]);
/* ComputeIntermedium of the NEXT window started now (it does not depend on the key frame): it runs beside the current window's
 * registrations; the nik_tracker_push_dev with the same pointer and n picks the spectra up (at most two windows under way).
 * Outputs unchanged. */
int  nik_tracker_prefetch_dev(nik_tracker* t, int n, const uint8_t* d_gray);
/* one host frame (cv::Mat CV_8UC1) */
int  nik_tracker_push_u8(nik_tracker* t, const uint8_t* gray, int stride, nik_track_output* out);
/* n host frames (H rows of `stride` bytes each, `frame_stride` bytes apart): the reference's per-frame loop (main.cpp:51-86,
 * map_builder.cc:30-33) for a streamed caller.  Windows of max_batch frames; window k+1 is uploaded on the context's upload
 * stream and window k+1's spectra are computed while window k is registered.  Outputs are exactly those of n
 * nik_tracker_push_u8 calls. */
int  nik_tracker_push_host(nik_tracker* t, int n, const uint8_t* gray, int stride, size_t frame_stride, nik_track_output* out);
/* number of keyframes inserted so far and their slots (for loop closure: nik_match over these) */
int  nik_tracker_keyframes(const nik_tracker* t, nik_frame* slots, int cap, int* n);

/* ---- keyframe map + loop-closure candidate management (src/map.cc, src/loop_closure.cc) ------------------
 * Decides WHICH keyframes a new keyframe is registered against; the registrations are one nik_match call.
 * Candidates are visited in ascending frame id (the reference's grid overload iterates unordered_sets, i.e. in
 * unspecified order; with its strict '>' the order only breaks exact ties -- here the lowest id wins). */
typedef struct {
    double  grid_scale;              /* MapConfig::grid_scale (read_configs.h:34-36), > 0                       */
    int32_t to_find_loop;            /* LoopClosureConfig (read_configs.h:38-44); informational here            */
    int32_t frame_gap_thr;           /* > 0: skip candidates closer than this many frame ids                    */
    double  distance_thr;            /* > 0: skip candidates whose accumulated travel differs by less than this */
    double  position_response_thr, angle_response_thr;   /* found = PSR_t > position_thr && PSR_r > angle_thr   */
} nik_loop_config;

typedef struct {
    int32_t  found;                  /* LoopClosureResult::found                                                */
    int32_t  cur_frame_id, loop_frame_id;   /* loop_frame_id = -1: no candidate                                 */
    nik_frame loop_slot;
    int32_t  n_candidates;           /* candidates that passed the filters (= ComputePose calls of the reference) */
    double   response[3];            /* best candidate's ComputePose return value; (-1,-1,-1) if none           */
    double   relative_pose[3];       /* its pose (image-centre based: apply ConvertCenterToPrincipal as MapBuilder::FindLoopClosure does) */
} nik_loop_result;

typedef struct nik_map nik_map;
/* replaces Map::Map + LoopClosure::LoopClosure.  ctx may be NULL for candidate queries only (no GPU needed). */
int  nik_map_create(nik_ctx* ctx, const nik_loop_config* cfg, nik_map** out);
void nik_map_destroy(nik_map* m);
/* Map::AddFrame (map.cc:17-30; the first frame's id is forced to 0) + Map::SetFrameDistance (:32-34; distance may be
 * NULL = never set, GetFrameDistance then reports -1).  `slot` is the device slot holding the keyframe's spectra. */
int  nik_map_add_frame(nik_map* m, int frame_id, nik_frame slot, const double pose[3], const double* distance);
int  nik_map_size(const nik_map* m);
/* Map::UpdatePoses (map.cc:73-79): new poses for the listed frames (the grid cells stay as inserted, as in the reference) */
int  nik_map_update_poses(nik_map* m, int n, const int32_t* frame_ids, const double* poses /*[n][3]*/);
/* the frames LoopClosure::FindLoopClosure would call ComputePose on for keyframe cur_frame_id (already added):
 * all frames (prior_pose NULL, loop_closure.cc:10-15) or those in the 3 x 3 grid cells around prior_pose
 * (:17-33), minus the frame-gap and travel-distance filters (:43-53). */
int  nik_map_candidates(const nik_map* m, int cur_frame_id, const double* prior_pose, int* frame_ids, int cap, int* n);
/* LoopClosure::FindLoopClosure (loop_closure.cc:36-73): every candidate against the current keyframe
 * (not_large_rotation = false) in one batched nik_match; the largest response.sum() wins. */
int  nik_map_find_loop(nik_map* m, int cur_frame_id, const double* prior_pose, nik_loop_result* out);

/* MapBuilder's map side for the tracker (map_builder.cc:61-65,168-178): with a map attached (before the first frame;
 * borrowed), every keyframe is added to it with its robot pose and accumulated distance, and -- if to_find_loop --
 * searched for a loop closure around that pose.  Loops found at consecutive keyframes accumulate like
 * MapBuilder::_loop_matches (relative_pose already passed through ConvertCenterToPrincipal); the first keyframe WITHOUT
 * a loop runs CheckAndOptimize (map_builder.cc:108-116): with >= 2 accumulated loops their edges are added, the pose
 * graph of all keyframes (KCC edges between consecutive keyframes + loop edges, identity information) is optimised
 * (nik_pose_graph_optimize), the map's and the tracker's poses are replaced (Map::UpdatePoses, UpdateValueAfterLoop) and
 * that frame's output carries optimized = 1 -- the caller then refreshes its occupancy map (nik_tracker_poses ->
 * nik_stitcher_recompute, MapStitcher::RecomputeOccupancy); the accumulated loops are cleared either way. */
int  nik_tracker_attach_map(nik_tracker* t, nik_map* m, int to_find_loop);
int  nik_tracker_loops(const nik_tracker* t, nik_loop_result* out, int cap, int* n);       /* every loop found so far */
int  nik_tracker_pending_loops(const nik_tracker* t);                                       /* MapBuilder::_loop_matches.size() */
/* robot poses of all keyframes (ascending frame id), as last written by the tracker or the optimiser */
int  nik_tracker_poses(const nik_tracker* t, int32_t* frame_ids, double* poses /*[cap][3]*/, int cap, int* n);

/* ---- MapStitcher (src/map_stitcher.cc, include/map_stitcher.h): occupancy map of the key frames -----------------
 * Cells are cell_size x cell_size int32 planes (data, weight), row-major [y in cell][x in cell], addressed by the
 * reference's floor-divided (cell_x, cell_y).  The reference's arithmetic is kept literally (a cell's first frame
 * stores raw sums and counts; later frames blend data*weight + sum*count and divide by the new weight).
 * insert_dev   MapStitcher::InsertFrame: d_image = the undistorted u8 frame (device, row-major H x W; copied);
 *              image_pose = the frame's pose in the image plane, centre based (ConvertRobotPoseToImagePlane +
 *              ConvertPrincipalToCenter, map_stitcher.cc:40-42 -- the caller's camera math).
 * recompute    MapStitcher::RecomputeOccupancy after a pose-graph update: new image poses for the listed frames, then
 *              every stored frame replayed in ascending frame id (the reference's order is unspecified). */
typedef struct nik_stitcher nik_stitcher;
int  nik_stitcher_create(nik_ctx* ctx, int cell_size, nik_stitcher** out);
void nik_stitcher_destroy(nik_stitcher* s);
int  nik_stitcher_insert_dev(nik_stitcher* s, int frame_id, const uint8_t* d_image, const double image_pose[3]);
int  nik_stitcher_recompute(nik_stitcher* s, int n, const int32_t* frame_ids, const double* image_poses /*[n][3]*/);
int  nik_stitcher_cells(const nik_stitcher* s, int32_t* locs /*[cap][2]: cell_x, cell_y*/, int cap, int* n);
int  nik_stitcher_read_cell(const nik_stitcher* s, int cell_x, int cell_y, int32_t* data, int32_t* weight);

/* ---- coarse-to-fine registration over an image pyramid (BASELINE config 3; NO reference counterpart) ----------
 * nik_pose_batch_window  ComputePose (small-rotation mode) with the arg-max of both correlation surfaces restricted to
 *                        the cyclic (2*radius+1)^2 window around centers[i] = {rot_row, rot_col, trans_row, trans_col}
 *                        (surface indices; rotation rows: also around the 180-degree mirror row).  PSR moments still
 *                        cover the whole surface.
 * nik_downsample_u8_dev  2x2 box filter, rounded: n device images H x W -> H/2 x W/2 (ctx's geometry is the source's).
 * nik_pyramid_*          `levels` contexts (level l: (H, W) >> l; polar (PD, PC) * {1, 2/3, 1/3, 1/6, ...}); track_dev
 *                        registers n (key, current) pairs: the coarsest level globally, every finer level inside the
 *                        window predicted from the level above.  res: [levels][n], level 0 first. */
int  nik_pose_batch_window(nik_ctx* ctx, int n, const nik_frame* keys, const nik_frame* curs, const int32_t* centers /*[n][4]*/,
                           int radius, nik_pose_result* res);
int  nik_downsample_u8_dev(nik_ctx* ctx, int n, const uint8_t* d_in, uint8_t* d_out);
typedef struct nik_pyramid nik_pyramid;
int  nik_pyramid_create(const nik_config* cfg, int H, int W, int levels, int max_batch, int device, nik_pyramid** out);
void nik_pyramid_destroy(nik_pyramid* p);
int  nik_pyramid_levels(const nik_pyramid* p, int* dims /* [levels][4]: H, W, PD, PC; may be NULL */);
int  nik_pyramid_track_dev(nik_pyramid* p, int n, const uint8_t* d_key, const uint8_t* d_cur, int radius, nik_pose_result* res);
/* The same without the final wait: successive batches pipeline (the coarse levels of batch k+1 run beside the fine
 * levels of batch k).  d_key, d_cur and res must stay valid, and res is not final, until nik_pyramid_synchronize or
 * until two further batches have been enqueued. */
int  nik_pyramid_track_dev_async(nik_pyramid* p, int n, const uint8_t* d_key, const uint8_t* d_cur, int radius, nik_pose_result* res);
int  nik_pyramid_synchronize(nik_pyramid* p);
const char* nik_pyramid_last_error(const nik_pyramid* p, int level);

/* ---- 2-D pose-graph optimisation (MapBuilder::OptimizeMap, map_builder.cc:195-271) ------------------------
 * The Ceres problem of src/optimization_2d/pose_graph_2d.cc:53-109,187-200 solved by an own Levenberg-Marquardt
 * (host, double): residual = chol_lower(information) * [R(yaw_a)^T (p_b - p_a) - p_ab; Normalize(yaw_b - yaw_a -
 * yaw_ab)], yaw updated through NormalizeAngle, the pose with id 0 constant.  Same minimum as Ceres (same cost,
 * same LM scheme and tolerances); the iterates are not Ceres' bit for bit. */
typedef struct {
    int32_t id_begin, id_end;        /* Constraint2d (include/optimization_2d/types.h:80-96)                     */
    double  x, y, yaw_radians;       /* pose of id_end in the frame of id_begin                                  */
    double  information[9];          /* row-major symmetric positive-definite 3x3 (x, y, yaw)                    */
} nik_pg_constraint;
enum { NIK_PG_CONVERGENCE = 0, NIK_PG_NO_CONVERGENCE = 1, NIK_PG_FAILURE = 2, NIK_PG_NO_CONSTRAINTS = 3 };
typedef struct {
    int32_t termination;             /* NIK_PG_*: NO_CONVERGENCE = max_iterations reached (still usable)          */
    int32_t iterations, successful_steps;
    int32_t inexact_solves;          /* device solves that hit the PCG iteration limit before its tolerance (their step is
                                      * still tried on its merits by the trust-region test); 0 on the host path.  Occupies
                                      * what was padding: size and the other offsets are unchanged                         */
    double  initial_cost, final_cost;   /* 0.5 * sum of squared residuals, as Ceres reports                       */
} nik_pg_summary;
/* poses: n_poses x (x, y, yaw), updated in place; ids: their frame ids (must contain 0); max_iterations <= 0: 300. */
int nik_pose_graph_optimize(int n_poses, const int32_t* ids, double* poses, int n_constraints,
                            const nik_pg_constraint* constraints, int max_iterations, nik_pg_summary* summary);
/* The same solver with the residuals and the normal equations (J^T J blocks, J^T r, cost) evaluated on HIP device `device`
 * (kcc_posegraph_dev.hip: one thread per constraint, a gather per free pose over its incident constraints, a wave-shuffle
 * reduction of the cost -- all in double, fixed summation orders); the damped solve stays on the host. */
int nik_pose_graph_optimize_dev(int device, int n_poses, const int32_t* ids, double* poses, int n_constraints,
                                const nik_pg_constraint* constraints, int max_iterations, nik_pg_summary* summary);
/* cost = 0.5 sum |r|^2, gradient J^T r [n_poses][3] and the J^T J diagonal blocks [n_poses][9] at `poses` (rows of the constant
 * pose and of poses no constraint touches are zero); device < 0: on the host.  Outputs may be NULL. */
int nik_pose_graph_linearize(int device, int n_poses, const int32_t* ids, const double* poses, int n_constraints,
                             const nik_pg_constraint* constraints, double* cost, double* gradient, double* jtj_diag);
/* A shard of the constraints resident on one GPU, and its cost left on that device (one double, valid after the work on
 * *stream, a hipStream_t): the operand of nik_group_pose_graph_cost. */
typedef struct nik_pg_shard nik_pg_shard;
int  nik_pg_shard_create(int device, int n_poses, const int32_t* ids, const double* poses, int n_constraints,
                         const nik_pg_constraint* constraints, nik_pg_shard** out);
void nik_pg_shard_destroy(nik_pg_shard* s);
int  nik_pg_shard_device(const nik_pg_shard* s);         /* the GPU the shard lives on (< 0: error) */
int  nik_pg_shard_cost_dev(nik_pg_shard* s, const double* poses /* NULL: unchanged */, double** d_cost, void** stream);

/* the tracker's pose graph: Map::_edges as OptimizeMap would feed them to the solver (robot units, identity information;
 * types[i]: 0 = KCC edge between consecutive keyframes, 1 = loop edge), and how often CheckAndOptimize has optimised */
int  nik_tracker_edges(const nik_tracker* t, nik_pg_constraint* out, int32_t* types, int cap, int* n);
int  nik_tracker_optimizations(const nik_tracker* t, nik_pg_summary* last /* may be NULL */);
/* nik_tracker_push_dev registers, in the batch that serves the current keyframe, also the frames behind the frames it GUESSES to
 * become the next keyframes (nik_tracker_guess_gap); out = [guesses that held, guesses that failed, batched pose calls so far] */
int  nik_tracker_speculation(const nik_tracker* t, long out[3]);
/* what push_dev expects the next keyframe gap to be after the gaps gaps[0..n) (oldest first): the gap that followed the most
 * recent earlier occurrence of the longest matching suffix (up to sixteen gaps), else the last gap; 0 for an empty history. */
int  nik_tracker_guess_gap(const int32_t* gaps, int n);

/* ---- measurement ---------------------------------------------------------------------------- */

/* Per-kernel timing with HIP events recorded on nik_stream() around every hot-path launch.
 * bytes = the launch's NOMINAL traffic (its input planes read once + its output planes written once, whole planes: the
 * SURVEY 8(d) accounting of that pass), summed over launches.
 * bytes_design = what the launch is BUILT to move: the nominal planes minus what its symmetry shortcuts leave out (the
 * Hermitian half of the Kzz kernel plane; the columns |c| <= Rmax + 1 of the zero-phase image) -- the figure a per-kernel
 * GB/s must be priced on (a kernel priced on bytes it never touches can "exceed" the HBM peak) -- plus what it files beside its
 * transform (the u8 kernel's copy of the image into the frame store).  Measured L2-fill traffic agrees with it within 2 % for
 * every kernel but two (profiles/r06_design_vs_moved.txt).
 * Enabling resets the accumulators; reading synchronises the stream. */
typedef struct {
    char    name[64];     /* kernel<length,mode> */
    double  ms;           /* total device time of the launches */
    int64_t launches;
    double  bytes;
    double  bytes_design;
} nik_stage_stat;
int nik_profile_enable(nik_ctx* ctx, int enable);
int nik_profile_read(nik_ctx* ctx, nik_stage_stat* out, int cap, int* n);

/* Small batches are bound by the latency of their ~16 dependent kernel launches: with max_pairs > 0, nik_pose / nik_pose_batch
 * calls of at most that many stored u8 frames (one stream, Kzz cache off) are captured once into a hipGraph per batch
 * size and replayed.  Results are identical.  0 = off (default; $NIK_GRAPH sets the default). */
int nik_set_graphs(nik_ctx* ctx, int max_pairs);

/* ---- coarse-to-fine chaining on the device (used by nik_pyramid; BASELINE config 3, an extension) ------------------
 * nik_pose_batch_window with the window centres predicted, on the device, from the peaks `upper` (a coarser level's
 * context on the same device) found in its latest pose call over the same n pairs.  No host round trip between levels. */
int nik_pose_batch_chained(nik_ctx* ctx, int n, const nik_frame* keys, const nik_frame* curs, nik_ctx* upper, int radius,
                           nik_pose_result* res, int sync);
/* every stream of ctx waits for the work `other` (same device) has enqueued so far */
int nik_wait_for(nik_ctx* ctx, nik_ctx* other);
/* nik_downsample_u8_dev, asynchronous on nik_stream(ctx) */
int nik_downsample_u8_async(nik_ctx* ctx, int n, const uint8_t* d_in, uint8_t* d_out);
/* the same on a caller-owned hipStream_t of ctx's device (no ordering against ctx's own streams) */
int nik_downsample_u8_stream(nik_ctx* ctx, int n, const uint8_t* d_in, uint8_t* d_out, void* stream);
/* `steps` (1..3) consecutive levels in one launch on `stream`: na frames of d_a followed by nb frames of d_b (d_b may be NULL
 * with nb = 0) of ctx's geometry; out[d] receives the na + nb frames of level d + 1 back to back -- the same integers as chained
 * nik_downsample_u8 calls.  H, W and the pointers must be aligned to 2^steps, else NIK_ERR_UNSUPPORTED_SIZE. */
int nik_downsample_pyr_u8_stream(nik_ctx* ctx, int steps, int na, const uint8_t* d_a, int nb, const uint8_t* d_b, uint8_t* const* out, void* stream);
/* `stream` waits for everything ctx has enqueued so far / every stream of ctx waits for what `stream` holds so far */
int nik_stream_wait_ctx(nik_ctx* ctx, void* stream);
int nik_ctx_wait_stream(nik_ctx* ctx, void* stream);
/* number of asynchronous calls (sync = 0) a stream of ctx keeps in flight before the next call blocks on the oldest:
 * 1..4, default 2.  Drains the context. */
int nik_set_call_depth(nik_ctx* ctx, int depth);

/* ---- residual statistics of a batch, reduced on the device --------------------------------------
 * stats = [sum PSR_t (chosen hypothesis), sum PSR_r, sum |t|^2 (px^2), count] over the pairs of the latest
 * nik_pose_batch / nik_track_batch_dev / nik_match call: the per-batch "residual sum" that a multi-GPU run all-reduces
 * (nik_group_allreduce_residual).  Off by default; when on, every batch call appends one tiny reduction kernel per stream. */
int nik_set_residual_stats(nik_ctx* ctx, int enable);
/* device pointer to the 4 doubles of the latest batch, valid in stream order on *stream (a hipStream_t owned by the
 * context; asynchronous: nothing is waited for).  Work enqueued on that stream delays nothing but the next batch's own
 * statistics. */
int nik_residual_stats_dev(nik_ctx* ctx, double** d_stats4, void** stream);
/* the same on the host (waits for the batch) */
int nik_residual_stats(nik_ctx* ctx, double stats[4]);

/* ---- multi-GPU: nik_group (kcc_group.cpp) ----------------------------------------------------
 * The path shards by independent units (frame pairs; loop-closure candidates), so a group is a set of contexts, one per
 * GPU, plus an RCCL communicator for the two small exchanges: the all-reduce of the per-batch residual statistics and
 * the all-gather of each GPU's best loop-closure candidate (reference src/loop_closure.cc:61-65).
 *   one process, every GPU of the node : nik_group_create_local  (a C++ MapBuilder linking this library)
 *   one process per GPU                : nik_group_unique_id on rank 0, broadcast by the launcher, nik_group_create_rank
 * RCCL is loaded on first use (dlopen librccl.so.1); groups of one GPU need no RCCL at all. */
typedef struct nik_group nik_group;
#define NIK_GROUP_ID_BYTES 128
const char* nik_group_last_error(const nik_group* g);          /* g may be NULL: the last create error of this thread */
int  nik_group_unique_id(uint8_t id[NIK_GROUP_ID_BYTES]);
int  nik_group_create_rank(nik_ctx* ctx, int rank, int world, const uint8_t id[NIK_GROUP_ID_BYTES], nik_group** out);   /* borrows ctx */
int  nik_group_create_local(const nik_config* cfg, int image_height, int image_width, int max_batch, int max_frames,
                            int n_devices, const int* devices /* NULL: 0..n-1 */, nik_group** out);                    /* owns its contexts */
void nik_group_destroy(nik_group* g);
int  nik_group_world(const nik_group* g);                      /* GPUs in the group */
int  nik_group_local_count(const nik_group* g);                /* members driven by this process (world, or 1) */
nik_ctx* nik_group_ctx(nik_group* g, int local_index);
int  nik_group_rank(const nik_group* g, int local_index);
/* contiguous shard [begin, end) of n units for `rank` of `world` (sizes differ by at most one; rank order = unit order) */
void nik_group_shard(int n, int world, int rank, int* begin, int* end);
/* the reference's winner rule (loop_closure.cc:61-65) over `world` gathered records of 8 doubles [score, global index (< 0:
 * none), pose x3, info x3]: strictly larger score wins, equal scores go to the lowest rank = the first candidate in global
 * order.  Returns the winning rank, -1 if no rank has a candidate.  Host-only (what nik_group_gather_best applies). */
int  nik_group_pick_best(const double* records, int world);
/* the RCCL the library bound: path of the file ncclAllReduce came from (NULL: RCCL not loaded / not found) and whether it is the
 * copy the host process had already loaded (*shared_with_host = 1, e.g. PyTorch's) or one the library loaded itself.
 * nik_group_create_rank gives ncclCommInitRank $NIK_GROUP_INIT_TIMEOUT seconds (default 90; <= 0: no limit) and fails with a
 * message instead of blocking for good. */
const char* nik_group_rccl_library(int* shared_with_host);
/* ranks the group's RCCL communicator spans (ncclCommCount); 0 = the group runs without RCCL (one member) */
int  nik_group_comm_ranks(const nik_group* g);
/* sum over the group of every member's latest-batch statistics (nik_set_residual_stats is switched on by the group),
 * reduced on the devices and all-reduced with RCCL; asynchronous when out == NULL (fetch with nik_group_residual_result) */
int  nik_group_allreduce_residual(nik_group* g, double out[4]);
int  nik_group_residual_result(nik_group* g, double out[4]);      /* once per all-reduce: a second fetch is NIK_ERR_NOT_READY */
/* cost of a pose graph whose constraints are sharded over the group (shards[i]: local member i's nik_pg_shard, on its
 * device): each GPU reduces its shard, ONE double is all-reduced with RCCL ("the final pose-graph residual sum") */
int  nik_group_pose_graph_cost(nik_group* g, nik_pg_shard* const* shards, const double* poses, double* cost);
/* every local member's best candidate (global index, -1: none) -> the group's winner by the reference's rule */
int  nik_group_gather_best(nik_group* g, const int* global_index, const nik_pose_result* local_best, int* best_index, nik_pose_result* best);
/* local groups: a batch of n pairs (host u8 images) sharded over the GPUs; keys[i] / cur_dst[i] are slots of the member
 * that owns pair i (nik_group_shard) */
int  nik_group_track_batch(nik_group* g, int n, const uint8_t* h_gray, const nik_frame* keys, const nik_frame* cur_dst,
                           int not_large_rotation, nik_pose_result* res);
/* local groups: FindLoopClosure's candidate loop over a key-frame store sharded over the GPUs (cands[r]: slots of member r) */
int  nik_group_match(nik_group* g, const uint8_t* h_query, nik_frame query_slot, const int* n_cands, const nik_frame* const* cands,
                     int* best_member, int* best_local, nik_pose_result* best);

/* ---- debug / parity taps (used by tests only) --------------------------------------------- */

/* CorrelationFlow::FFT / IFFT (correlation_flow.cc:53-77) on host arrays in the reference layouts.
 * which: 0 = image geometry (H x W), 1 = polar geometry (PD x PC). */
int nik_dbg_fft (nik_ctx* ctx, int which, const float* x_colmajor, float* xf_out);
int nik_dbg_ifft(nik_ctx* ctx, int which, const float* xf, float* x_out);
/* RotateArray (utils.cc:154-161) of slot f's image by `degree2`/2 degrees (degree2 = 2*degree, integer). */
int nik_dbg_rotate(nik_ctx* ctx, nik_frame f, int degree2, float* out_colmajor);
/* polar(fftshift(RemoveZeroComponent(x))) (correlation_flow.cc:93-94) of a host H x W plane. */
int nik_dbg_polar(nik_ctx* ctx, const float* x_colmajor, float* out_colmajor /*PD x PC*/);


/* The response surface g = IFFT(G) of one EstimateTrans call (correlation_flow.cc:171-173), which the hot path never stores.
 * which 0: rotation stage of (key, cur) -> PD x PC;  which 1: translation stage against cur's image de-rotated by
 * degree2/2 degrees -> H x W.  Column-major like every real plane of the reference. */
int nik_dbg_response(nik_ctx* ctx, int which, nik_frame key, nik_frame cur, int degree2, float* g_colmajor);

/* ---- host-side gather tables (tests only; no GPU needed) --------------------------------------
 * The tables the two gather kernels consume, built exactly as nik_create builds them, so that CPU tests can replay
 * the kernels' staging / sampling arithmetic against the oracle (tests/test_host_tables.py). */
/* polar gather plan for geometry (H, W, PD, PC).  dims = {qs, nseg, tiles, lines, threads, rf, mf, lds_bytes}.
 * chunks[n_chunks]: source offset of 16 consecutive floats;  seg_first[tiles*nseg+1];  pts[tiles*rf*lines*threads*4].
 * The arrays are malloc'ed; release with nik_host_free. */
int nik_host_polar_plan(int H, int W, int PD, int PC, int dims[8], uint32_t** chunks, int* n_chunks, int** seg_first, uint32_t** pts);
void nik_host_free(void* p);
/* cv::warpAffine fixed-point terms of RotateArray(image, degree): out[2W+2H] = adelta | bdelta | X0 | Y0 */
int nik_host_rot_terms(int H, int W, float degree, int* out);
/* LDS box geometry of the u8 de-rotation for image height H: geom = {band_rows, bands, box_rows, pitch, lds_bytes} */
/* radices (execution order, 0-terminated) of the run-time FFT plan the any-size kernels use for n points; returns their count */
int  nik_host_fft_plan(int n, int radices[16]);
int nik_host_rot8_geom(int H, int geom[5]);

#ifdef __cplusplus
}
#endif
#endif

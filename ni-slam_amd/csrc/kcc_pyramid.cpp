// kcc_pyramid.cpp -- coarse-to-fine registration over an image pyramid: BASELINE config 3 ("4-level correlation
// pyramid with radius-4 lookup").  NO REFERENCE COUNTERPART (the reference registers at one resolution); defined in
// SURVEY 8(d) as an extension and checked against this repository's own CPU restatement only:
//   level l image = 2x2 box filter (rounded) of level l-1;  level l polar size = (PD, PC) * {1, 2/3, 1/3, 1/6, ...};
//   the coarsest level runs the plain KCC pose (small-rotation mode); every finer level runs it with the arg-max of
//   both correlation surfaces restricted to the (2R+1)^2 cyclic window around the peak predicted by the level above
//   (rotation surface: also around the 180-degree mirror row -- its source is point-symmetric).
//   A peak at index idx of a surface of size n_from predicts index n_to/2 + lround((idx - n_from/2) * n_to / n_from)
//   (cyclic) on the finer surface; the prediction runs on the device (k_predict_windows), so the levels chain without
//   a host round trip and their streams overlap.
// One nik_ctx per level; host code only (the kernels are the ordinary ones plus the windowed arg-max).
#include "../../include/nislam_kcc.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <vector>

struct nik_pyramid {
    int levels = 0, max_batch = 0, H = 0, W = 0;
    std::vector<nik_ctx*> ctx;                 // [level]
    std::vector<int> h, w, pd, pc;
    std::vector<uint8_t*> d_key;               // [level >= 1] downsampled frames, 2 * max_batch per level: the n key frames, then
                                               // the n current frames, so that both go through ONE intermedium call of 2n frames
                                               // (level 0 is the caller's buffers)
    hipStream_t ds = nullptr;                  // the box-filter chain runs beside the levels' own streams
};

extern "C" {

void nik_pyramid_destroy(nik_pyramid* p) {
    if (!p) return;
    for (nik_ctx* c : p->ctx) if (c) nik_destroy(c);
    for (uint8_t* b : p->d_key) if (b) (void)hipFree(b);
    if (p->ds) (void)hipStreamDestroy(p->ds);
    delete p;
}

int nik_pyramid_create(const nik_config* cfg, int H, int W, int levels, int max_batch, int device, nik_pyramid** out) {
    if (!cfg || !out || levels < 1 || levels > 6 || max_batch < 1) return NIK_ERR_INVALID_ARG;
    if ((H % (1 << levels)) || (W % (1 << levels))) return NIK_ERR_INVALID_ARG;      // every level keeps even sizes
    nik_pyramid* p = new nik_pyramid();
    p->levels = levels; p->max_batch = max_batch; p->H = H; p->W = W;
    p->ctx.assign(levels, nullptr); p->d_key.assign(levels, nullptr);
    for (int l = 0; l < levels; ++l) {
        nik_config c = *cfg;
        // polar geometry per level: 1, 2/3, 1/3, 1/6, ... of the base (720x480 -> 480x320 -> 240x160 -> 120x80)
        const int num = l == 0 ? 3 : 2, den = l == 0 ? 3 : 3 << (l - 1);
        if ((cfg->rotation_divisor * num) % den || (cfg->rotation_channel * num) % den) { nik_pyramid_destroy(p); return NIK_ERR_UNSUPPORTED_SIZE; }
        c.rotation_divisor = cfg->rotation_divisor * num / den; c.rotation_channel = cfg->rotation_channel * num / den;
        c.height = H >> l; c.width = W >> l;
        p->h.push_back(H >> l); p->w.push_back(W >> l); p->pd.push_back(c.rotation_divisor); p->pc.push_back(c.rotation_channel);
        // (below level 0 the contexts are sized for 2 * max_batch frames per call: key and current frames of a level are
        // transformed together.  Level 0 -- whose buffers are the big ones -- never runs a 2n-frame call: its key and current
        // frames arrive in two caller buffers and go through two n-frame calls, so max_batch items with 2 * max_batch slots do.
        // One stream per level: batches this small lose more to the doubled launch count than a second stream gains)
        int rc = nik_create(&c, H >> l, W >> l, l == 0 ? max_batch : 2 * max_batch, 2 * max_batch, device, &p->ctx[l]);
        if (!rc) rc = nik_set_call_depth(p->ctx[l], 4);            // spectra + pose per batch, two batches in flight
        if (!rc) rc = nik_set_streams(p->ctx[l], 1) == 1 ? 0 : NIK_ERR_INVALID_ARG;     // (a 2n-frame call of a coarse level would otherwise split)
        // chained calls must be cut exactly like the level above (nik_pose_batch_chained): the $NIK_CHUNK tuning knob, which
        // nik_create reads, would chunk the plain call of the coarsest level and not the chained ones
        if (!rc) rc = nik_set_chunk(p->ctx[l], 0) >= 0 ? 0 : NIK_ERR_INVALID_ARG;
        if (rc) { nik_pyramid_destroy(p); return rc; }
        if (l > 0) {
            const size_t frame = (size_t)(H >> l) * (W >> l);
            if (hipMalloc(&p->d_key[l], 2 * frame * max_batch) != hipSuccess) { nik_pyramid_destroy(p); return NIK_ERR_HIP; }
        }
    }
    if (hipStreamCreateWithFlags(&p->ds, hipStreamNonBlocking) != hipSuccess) { nik_pyramid_destroy(p); return NIK_ERR_HIP; }
    *out = p;
    return NIK_OK;
}

int nik_pyramid_levels(const nik_pyramid* p, int* dims /* [levels][4]: H, W, PD, PC */) {
    if (!p) return 0;
    if (dims) for (int l = 0; l < p->levels; ++l) { dims[4 * l] = p->h[l]; dims[4 * l + 1] = p->w[l]; dims[4 * l + 2] = p->pd[l]; dims[4 * l + 3] = p->pc[l]; }
    return p->levels;
}

int nik_pyramid_track_dev_async(nik_pyramid* p, int n, const uint8_t* d_key, const uint8_t* d_cur, int radius, nik_pose_result* res) {
    if (!p || !d_key || !d_cur || !res || n < 0 || radius < 0) return NIK_ERR_INVALID_ARG;
    if (n == 0) return NIK_OK;
    if (n > p->max_batch) return NIK_ERR_CAPACITY;
    const int L = p->levels;
    int rc;
    std::vector<const uint8_t*> key(L), cur(L);
    key[0] = d_key; cur[0] = d_cur;
    // every level's frames first: level l is the 2x2 box filter of level l - 1, produced on the pyramid's own stream -- up to three
    // levels per launch (nik_downsample_pyr_u8_stream), both frame sets at once.  That stream waits, per level, for the level's
    // PREVIOUS spectra call only (recorded right behind it, see below): the box filters of batch k+1 neither queue behind the
    // poses of batch k nor return to the host.
    for (int l = 1; l < L; ++l) { key[l] = p->d_key[l]; cur[l] = p->d_key[l] + (size_t)n * p->h[l] * p->w[l]; }
    for (int l0 = 0; l0 + 1 < L; ) {
        const int steps = std::min(3, L - 1 - l0);
        uint8_t* outs[3] = { nullptr, nullptr, nullptr };
        for (int d = 0; d < steps; ++d) outs[d] = p->d_key[l0 + 1 + d];
        // level 0 arrives in two caller buffers; below it both frame sets sit in one buffer (2n frames)
        rc = l0 == 0 ? nik_downsample_pyr_u8_stream(p->ctx[0], steps, n, key[0], n, cur[0], outs, p->ds)
                     : nik_downsample_pyr_u8_stream(p->ctx[l0], steps, 2 * n, key[l0], 0, nullptr, outs, p->ds);
        if (rc == NIK_ERR_UNSUPPORTED_SIZE) {                 // unaligned caller buffers: one level at a time
            for (int l = l0 + 1; l <= l0 + steps; ++l)
                if ((rc = nik_downsample_u8_stream(p->ctx[l - 1], n, key[l - 1], p->d_key[l], p->ds)) ||
                    (rc = nik_downsample_u8_stream(p->ctx[l - 1], n, cur[l - 1], p->d_key[l] + (size_t)n * p->h[l] * p->w[l], p->ds))) return rc;
        } else if (rc) return rc;
        l0 += steps;
    }
    for (int l = 1; l < L; ++l) if ((rc = nik_ctx_wait_stream(p->ctx[l], p->ds))) return rc;
    // slots 0..n-1 hold the key frames, n..2n-1 the current frames; below level 0 both sets sit in one buffer, the current
    // frames directly behind the n key frames (d_key + n frames), so one 2n-frame call transforms them all
    std::vector<nik_frame> ks(n), cs(n), all(2 * (size_t)n);
    for (int i = 0; i < n; ++i) { ks[i] = i; cs[i] = n + i; all[i] = i; all[n + i] = n + i; }
    for (int l = L - 1; l >= 1; --l) {
        if ((rc = nik_intermedium_batch_dev(p->ctx[l], 2 * n, key[l], all.data()))) return rc;
        // (the next batch's box filter overwrites d_key[l]: it has to wait for this call, and for nothing behind it)
        if ((rc = nik_stream_wait_ctx(p->ctx[l], p->ds))) return rc;
    }
    if ((rc = nik_intermedium_batch_dev(p->ctx[0], n, key[0], ks.data())) ||
        (rc = nik_intermedium_batch_dev(p->ctx[0], n, cur[0], cs.data()))) return rc;
    // coarsest level: plain KCC; finer levels: windows predicted on the device from the level above
    for (int l = L - 1; l >= 0; --l) {
        nik_ctx* c = p->ctx[l];
        nik_pose_result* out = res + (size_t)l * n;
        if (l == L - 1) { if ((rc = nik_pose_batch_async(c, n, ks.data(), cs.data(), 1, out))) return rc; }
        else if ((rc = nik_pose_batch_chained(c, n, ks.data(), cs.data(), p->ctx[l + 1], radius, out, 0))) return rc;
    }
    return NIK_OK;
}

int nik_pyramid_synchronize(nik_pyramid* p) {
    if (!p) return NIK_ERR_INVALID_ARG;
    int rc;
    for (int l = 0; l < p->levels; ++l) if ((rc = nik_synchronize(p->ctx[l]))) return rc;
    return NIK_OK;
}

int nik_pyramid_track_dev(nik_pyramid* p, int n, const uint8_t* d_key, const uint8_t* d_cur, int radius, nik_pose_result* res) {
    int rc = nik_pyramid_track_dev_async(p, n, d_key, d_cur, radius, res);
    return rc ? rc : nik_pyramid_synchronize(p);
}

const char* nik_pyramid_last_error(const nik_pyramid* p, int level) {
    return (p && level >= 0 && level < p->levels) ? nik_last_error(p->ctx[level]) : "";
}

}  // extern "C"

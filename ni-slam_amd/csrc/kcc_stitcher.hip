// kcc_stitcher.hip -- the reference's MapStitcher (src/map_stitcher.cc:11-145, include/map_stitcher.h) on the device:
// every key frame's (undistorted, 100/255-scaled) image is scattered into an occupancy map made of cell_size^2 integer
// cells, with the reference's arithmetic kept literally -- including its quirks: a cell's FIRST frame stores raw sums
// and counts, later frames blend `data*weight + sum*count` and divide by the new weight (integer division).
// The merge is not commutative, so RecomputeOccupancy needs an order: the reference iterates an unordered_map keyed by
// pointers (unspecified); here frames are replayed in ascending frame id.
// Per key frame (not per frame) work; the scatter is one thread per pixel with integer atomics.
#include "../../include/nislam_kcc.h"
#include "kcc_kernels.h"

#include <algorithm>
#include <cmath>
#include <map>
#include <vector>

using namespace kcc;

namespace {
struct CellBuf { int* data = nullptr; int* weight = nullptr; };
struct RawFrame { uint8_t* d_img = nullptr; double pose[3]; };
}

struct nik_stitcher {
    nik_ctx* ctx = nullptr;
    int H = 0, W = 0, size = 0;
    hipStream_t stream = nullptr;
    std::map<std::pair<int, int>, CellBuf> cells;          // MapStitcher::_occupancy_data
    std::map<int, RawFrame> raw;                            // MapStitcher::_raw_images (by frame id)
    int* tmp = nullptr; size_t tmp_cells = 0;               // temporary cells of the frame being added: [2][tmp_cells][size^2]
    int* d_flags = nullptr; size_t flag_cap = 0;            // per temporary cell: touched?

    static void cell_position(int x, int size, int& cell, int& pos) {      // MapStitcher::ComputeCellPosition (:24-34)
        cell = x >= 0 ? x / size : (x - size + 1) / size;
        pos = x - cell * size;
    }

    int add(const uint8_t* d_img, const double pose[3]) {                  // MapStitcher::AddImageToOccupancy (:36-133)
        StitchPose P;
        const double c = std::cos(pose[2]), s = std::sin(pose[2]);         // RotationMatrix2D
        P.r00 = c; P.r01 = -s; P.r10 = s; P.r11 = c; P.x = pose[0]; P.y = pose[1]; P.cx = (double)W / 2; P.cy = (double)H / 2;
        // which cells may be used: the four corner pixels (:64-80)
        int minx = 0, maxx = 0, miny = 0, maxy = 0; bool first = true;
        for (int i : { 0, W - 1 })
            for (int j : { 0, H - 1 }) {
                const double wi = (double)i - P.cx, hj = (double)j - P.cy;
                const int x = (int)((P.r00 * wi + P.x) + P.r01 * hj), y = (int)((P.r10 * wi + P.y) + P.r11 * hj);
                if (first) { minx = maxx = x; miny = maxy = y; first = false; }
                minx = std::min(minx, x); maxx = std::max(maxx, x); miny = std::min(miny, y); maxy = std::max(maxy, y);
            }
        int cx0, cx1, cy0, cy1, dummy;
        cell_position(minx, size, cx0, dummy); cell_position(maxx, size, cx1, dummy);
        cell_position(miny, size, cy0, dummy); cell_position(maxy, size, cy1, dummy);
        const int ncx = cx1 - cx0 + 1, ncy = cy1 - cy0 + 1;
        const size_t need = (size_t)ncx * ncy, csz = (size_t)size * size;
        if (need > tmp_cells) {
            (void)hipFree(tmp); tmp = nullptr; tmp_cells = 0;
            if (hipMalloc(&tmp, sizeof(int) * 2 * need * csz) != hipSuccess) return NIK_ERR_HIP;
            tmp_cells = need;
        }
        int* tmp_data = tmp; int* tmp_weight = tmp + tmp_cells * csz;
        if (hipMemsetAsync(tmp, 0, sizeof(int) * 2 * tmp_cells * csz, stream) != hipSuccess) return NIK_ERR_HIP;
        launch_stitch_scatter(stream, d_img, H, W, P, size, cx0, cy0, ncx, ncy, tmp_data, tmp_weight);
        // which temporary cells were touched (locations[...], :82-90,108): a cell of the corner-derived range can stay
        // empty, so the device reports per cell whether any weight is non-zero
        if (need > flag_cap) {
            (void)hipFree(d_flags); d_flags = nullptr; flag_cap = 0;
            if (hipMalloc(&d_flags, sizeof(int) * need) != hipSuccess) return NIK_ERR_HIP;
            flag_cap = need;
        }
        if (hipMemsetAsync(d_flags, 0, sizeof(int) * need, stream) != hipSuccess) return NIK_ERR_HIP;
        launch_stitch_touched(stream, tmp_weight, (int)need, (int)csz, d_flags);
        std::vector<int> touched(need, 0);
        if (hipMemcpyAsync(touched.data(), d_flags, sizeof(int) * need, hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess) return NIK_ERR_HIP;
        for (int a = 0; a < ncx; ++a)
            for (int b = 0; b < ncy; ++b) {
                const size_t k = (size_t)a * ncy + b;
                if (!touched[k]) continue;
                const auto key = std::make_pair(cx0 + a, cy0 + b);
                auto it = cells.find(key);
                const bool existing = it != cells.end();
                if (!existing) {
                    CellBuf nb;
                    if (hipMalloc(&nb.data, sizeof(int) * csz) != hipSuccess) return NIK_ERR_HIP;
                    if (hipMalloc(&nb.weight, sizeof(int) * csz) != hipSuccess) { (void)hipFree(nb.data); return NIK_ERR_HIP; }
                    it = cells.emplace(key, nb).first;
                }
                launch_stitch_merge(stream, it->second.data, it->second.weight, tmp_data + k * csz, tmp_weight + k * csz, (int)csz, existing ? 1 : 0);
            }
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return NIK_ERR_HIP;
        return NIK_OK;
    }
    void clear_cells() {
        for (auto& kv : cells) { (void)hipFree(kv.second.data); (void)hipFree(kv.second.weight); }
        cells.clear();
    }
};

extern "C" {

int nik_stitcher_create(nik_ctx* ctx, int cell_size, nik_stitcher** out) {
    if (!ctx || !out || cell_size < 1 || cell_size > 8192) return NIK_ERR_INVALID_ARG;
    int dims[6];
    int rc = nik_get_dims(ctx, dims);
    if (rc) return rc;
    nik_stitcher* s = new nik_stitcher();
    s->ctx = ctx; s->H = dims[0]; s->W = dims[1]; s->size = cell_size; s->stream = (hipStream_t)nik_stream(ctx);
    *out = s;
    return NIK_OK;
}

void nik_stitcher_destroy(nik_stitcher* s) {
    if (!s) return;
    s->clear_cells();
    for (auto& kv : s->raw) (void)hipFree(kv.second.d_img);
    (void)hipFree(s->tmp); (void)hipFree(s->d_flags);
    delete s;
}

int nik_stitcher_insert_dev(nik_stitcher* s, int frame_id, const uint8_t* d_image, const double image_pose[3]) {
    if (!s || !d_image || !image_pose) return NIK_ERR_INVALID_ARG;
    if (s->raw.count(frame_id)) return NIK_ERR_INVALID_ARG;
    int rc = nik_synchronize(s->ctx);                          // the image may have been produced on the context's streams
    if (rc) return rc;
    RawFrame f; f.pose[0] = image_pose[0]; f.pose[1] = image_pose[1]; f.pose[2] = image_pose[2];
    const size_t n = (size_t)s->H * s->W;
    if (hipMalloc(&f.d_img, n) != hipSuccess) return NIK_ERR_HIP;
    if (hipMemcpyAsync(f.d_img, d_image, n, hipMemcpyDeviceToDevice, s->stream) != hipSuccess) { (void)hipFree(f.d_img); return NIK_ERR_HIP; }
    s->raw[frame_id] = f;                                      // _raw_images[frame] (:20); the 100/255 scaling happens in the scatter
    return s->add(f.d_img, f.pose);
}

int nik_stitcher_recompute(nik_stitcher* s, int n, const int32_t* frame_ids, const double* image_poses) {
    if (!s || n < 0 || (n > 0 && (!frame_ids || !image_poses))) return NIK_ERR_INVALID_ARG;
    for (int i = 0; i < n; ++i) {                              // Map::UpdatePoses: frames not in the map are ignored (map.cc:73-79)
        auto it = s->raw.find(frame_ids[i]);
        if (it == s->raw.end()) continue;
        for (int k = 0; k < 3; ++k) it->second.pose[k] = image_poses[3 * (size_t)i + k];
    }
    s->clear_cells();                                          // RecomputeOccupancy (:135-141), replayed in ascending frame id
    for (auto& kv : s->raw) { const int rc = s->add(kv.second.d_img, kv.second.pose); if (rc) return rc; }
    return NIK_OK;
}

int nik_stitcher_cells(const nik_stitcher* s, int32_t* locs, int cap, int* n) {
    if (!s || !n) return NIK_ERR_INVALID_ARG;
    *n = (int)s->cells.size();
    int i = 0;
    for (const auto& kv : s->cells) { if (i < cap && locs) { locs[2 * i] = kv.first.first; locs[2 * i + 1] = kv.first.second; } ++i; }
    return NIK_OK;
}

int nik_stitcher_read_cell(const nik_stitcher* s, int cell_x, int cell_y, int32_t* data, int32_t* weight) {
    if (!s) return NIK_ERR_INVALID_ARG;
    const auto it = s->cells.find(std::make_pair(cell_x, cell_y));
    if (it == s->cells.end()) return NIK_ERR_NOT_READY;
    const size_t b = sizeof(int) * (size_t)s->size * s->size;
    if (data && hipMemcpy(data, it->second.data, b, hipMemcpyDeviceToHost) != hipSuccess) return NIK_ERR_HIP;
    if (weight && hipMemcpy(weight, it->second.weight, b, hipMemcpyDeviceToHost) != hipSuccess) return NIK_ERR_HIP;
    return NIK_OK;
}

}  // extern "C"

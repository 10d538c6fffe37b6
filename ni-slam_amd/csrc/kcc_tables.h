// kcc_tables.h -- host-side gather tables (kcc_tables.cpp); internal to libnislam_kcc_hip.so.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace kcc {

// natural polar map [PC][PD]: (sx*(H+2)+sy) | fx<<22 | fy<<27
int build_polar_map(int H, int W, int PD, int PC, std::vector<uint32_t>& tab, std::string& err);

// host image of kcc_kernels.h PolarPlan
struct PolarPlanHost {
    std::vector<uint32_t> chunks;    // source offset of 16 consecutive floats
    std::vector<int> seg_first;      // [tiles*nseg + 1]
    std::vector<uint32_t> pts;       // [tiles][rf][lines*threads][4]
    int qs = 0, nseg = 0, tiles = 0, lines = 0, threads = 0, rf = 0, mf = 0;
    size_t lds_bytes = 0;
};
// lines/threads/rf/mf/fft_lds_bytes/qs_opts: kcc_kernels.h fwd_geom(PD/2)
// aligned = G > 0: a span is staged as back-to-back 16-pixel chunks from its start rounded down to G pixels (16: whole 64-byte
// pieces of the plane) -- kcc_tables.cpp plan_segment; 0: 16-pixel chunks every 15 pixels, started up to x % 8 pixels early
#ifndef KCC_POLAR_ALIGNED_DEFAULT
#define KCC_POLAR_ALIGNED_DEFAULT 0
#endif
int build_polar_plan(int H, int W, int PD, int PC, int lines, int threads, int rf, int mf, size_t fft_lds_bytes, const int qs_opts[3],
                     PolarPlanHost& out, std::string& err, int aligned = KCC_POLAR_ALIGNED_DEFAULT);

// [adelta W | bdelta W | X0 H | Y0 H] of cv::warpAffine for RotateArray(image, degree_arg)
void rotation_terms(int H, int W, float degree_arg, int* out);

}  // namespace kcc

// kcc_tables.cpp -- host-side tables of the two gathers (no device code, no HIP calls: testable without a GPU).
//
//   polar map   cv::warpPolar's float maps quantised as cv::remap does (reference correlation_flow.cc:228-236)  [recalled]
//   polar plan  the same map re-expressed for the LDS-staged gather kernel (kA_fwd<., SRC_POLAR_*>): per tile and
//               angular segment the source-pixel runs to stage, per thread the LDS positions of its samples' taps
//   rotation terms  fixed-point coordinate terms of cv::warpAffine for RotateArray (utils.cc:154-161)  [recalled]
#include "kcc_tables.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace kcc {

static inline int cv_round_f(float v) { return (int)lrintf(v); }

// Entry [rho * PD + phi] = (sx*(H+2)+sy) | fx<<22 | fy<<27: the top-left tap of sample (angle phi, radius rho) in the
// shifted zero-bordered plane S[W+1][H+2] (column pitch H+2) and its 1/32-pixel fractions.
int build_polar_map(int H, int W, int PD, int PC, std::vector<uint32_t>& tab, std::string& err) {
    tab.assign((size_t)PD * PC, 0);
    const float cx = (float)W / 2, cy = (float)H / 2;
    const double maxRadius = (double)std::min(H / 2, W / 2);
    const double Kangle = (2.0 * 3.1415926535897932384626433832795) / PD;
    const double Kmag = maxRadius / PC;
    std::vector<float> rhos(PC);
    for (int rho = 0; rho < PC; ++rho) rhos[rho] = (float)(rho * Kmag);
    for (int phi = 0; phi < PD; ++phi) {
        const double KKy = Kangle * phi;
        const double cp = cos(KKy), sp = sin(KKy);
        for (int rho = 0; rho < PC; ++rho) {
            const float mx = (float)(rhos[rho] * cp + cx);
            const float my = (float)(rhos[rho] * sp + cy);
            const int qx = cv_round_f(mx * 32), qy = cv_round_f(my * 32);
            const int sx = qx >> 5, sy = qy >> 5;
            // all four taps must fall inside the zero-bordered plane (taps beyond the image read 0, exactly cv::remap's
            // BORDER_CONSTANT path for a source that never leaves the image by more than 1 px)
            if (sx < 0 || sy < 0 || sx + 1 > W || sy + 1 > H) { err = "polar map leaves the image by more than one pixel"; return -1; }
            const uint32_t off = (uint32_t)sx * (uint32_t)(H + 2) + (uint32_t)sy;
            if (off >= (1u << 22)) { err = "image too large for the packed polar table"; return -1; }
            tab[(size_t)rho * PD + phi] = off | ((uint32_t)(qx & 31) << 22) | ((uint32_t)(qy & 31) << 27);
        }
    }
    return 0;
}

namespace {
struct Span { uint32_t a, b; int first_chunk; };          // source offsets [a, b] (inclusive), index of its first chunk in the segment

// one (tile, segment): stage list + entries; returns the number of chunks
int plan_segment(const std::vector<uint32_t>& map, int PD, int SP, int tile, int seg, int qs, int lines, int threads, int rf, int mf,
                 std::vector<uint32_t>& chunks, uint32_t* pts /* [rf][lines*threads][4] of this tile */, int aligned) {
    const int NT = lines * threads;
    std::vector<uint32_t> need;
    need.reserve((size_t)lines * mf * qs * 8);
    for (int line = 0; line < lines; ++line)
        for (int j = 0; j < mf; ++j)
            for (int qq = 0; qq < qs; ++qq) {
                const int m = j + (seg * qs + qq) * mf;
                for (int h = 0; h < 2; ++h) {
                    const uint32_t off = map[(size_t)(tile * lines + line) * PD + 2 * m + h] & 0x3FFFFFu;
                    need.push_back(off); need.push_back(off + 1); need.push_back(off + SP); need.push_back(off + SP + 1);
                }
            }
    std::sort(need.begin(), need.end());
    need.erase(std::unique(need.begin(), need.end()), need.end());
    // runs of needed pixels (gaps of <= 3 pixels are fetched rather than opening a new chunk)
    std::vector<Span> spans;
    int nch = 0;
    for (size_t i = 0; i < need.size();) {
        size_t k = i;
        while (k + 1 < need.size() && need[k + 1] - need[k] <= 4) ++k;
        Span s{ need[i], need[k], nch };
        // LDS banks: pixel y of a chunk sits at bank 16*(chunk & 1) + (y - chunk start), so where the annulus edge runs
        // nearly straight across neighbouring source columns (top and bottom of the ring) every column would put its taps
        // into the same banks.  Starting the chunks of source column x up to x % 8 pixels early decorrelates them: the
        // simulated bank-conflict cycles of the gather drop from 6.2x to 3.3x the conflict-free count for +19 % chunks.
        if (aligned) {
            // ALIGNED form (round 6, $NIK_POLAR_ALIGNED = G in the tuning library; off by default): the span is staged as back-to-back
            // 16-pixel chunks from its start rounded down to G pixels.  G = 16 makes every chunk exactly half a 128-byte line (the
            // item stride is a multiple of 16 floats); neighbouring source columns stay decorrelated over the LDS banks by the
            // plane's own pitch (482 = 2 mod 16).  Measured (profiles/r06_polar_aligned.txt): in the ABLATION build, whose staging
            // loop the ABL() branches serialise, G = 16 is 8 % faster (0.257 -> 0.237 ms per 256 items) although it stages 21 % more
            // chunks in four segments instead of two; G = 1 (contiguous, unaligned, 16 % FEWER chunks) is 10 % slower -- the cost of
            // the LDS-DMA is lines touched, not bytes.  On the RELEASE code generation, where all DMAs of a segment are in flight
            // together, the three forms are equal within 0.7 % (0.4468 / 0.4458 / 0.4500 ms per 512 items for G = 16 / 8 / 0):
            // the staging is hidden there and the old form stays.
            s.a = (s.a / (uint32_t)aligned) * (uint32_t)aligned;
            for (uint32_t st = s.a;; st += 16) {
                chunks.push_back(st);
                ++nch;
                if (st + 15 >= s.b) break;
            }
            spans.push_back(s);
            i = k + 1;
            continue;
        }
        {
            const uint32_t x = s.a / (uint32_t)SP, y = s.a % (uint32_t)SP;
            s.a -= std::min<uint32_t>(x % 8u, y);
        }
        // chunks of 16 pixels starting every 15 (the last one may run past the span: the kernel always fetches 16):
        // consecutive chunks share one pixel, so every vertical tap pair (y, y+1) lies inside one chunk
        for (uint32_t st = s.a;; st += 15) {
            chunks.push_back(st);
            ++nch;
            if (st + 15 >= s.b) break;
        }
        spans.push_back(s);
        i = k + 1;
    }
    auto lds_of_pair = [&](uint32_t off) -> uint32_t {     // LDS float index of source pixel `off`, with off+1 right behind it
        size_t lo = 0, hi = spans.size();
        while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (spans[mid].a <= off) lo = mid; else hi = mid; }
        const Span& s = spans[lo];
        const uint32_t rel = off - s.a, k = rel / 15;
        if (aligned) return (uint32_t)s.first_chunk * 16u + rel;
        return (uint32_t)(s.first_chunk + (int)k) * 16u + (rel - 15 * k);
    };
    for (int line = 0; line < lines; ++line)
        for (int j = 0; j < mf; ++j)
            for (int qq = 0; qq < qs; ++qq) {
                const int q = seg * qs + qq, m = j + q * mf, tid = line * threads + j;
                uint32_t* e = pts + ((size_t)q * NT + tid) * 4;
                for (int h = 0; h < 2; ++h) {
                    const uint32_t t = map[(size_t)(tile * lines + line) * PD + 2 * m + h];
                    const uint32_t off = t & 0x3FFFFFu, fx = (t >> 22) & 31, fy = t >> 27;
                    e[2 * h + 0] = lds_of_pair(off) | (fx << 16) | (fy << 21);
                    e[2 * h + 1] = lds_of_pair(off + SP);
                }
            }
    return nch;
}
}  // namespace

int build_polar_plan(int H, int W, int PD, int PC, int lines, int threads, int rf, int mf, size_t fft_lds_bytes, const int qs_opts[3],
                     PolarPlanHost& out, std::string& err, int aligned) {
    std::vector<uint32_t> map;
    if (build_polar_map(H, W, PD, PC, map, err)) return -1;
    if (lines <= 0 || PC % lines || rf * mf != PD / 2) { err = "polar tile geometry does not match the plane"; return -1; }
    const int tiles = PC / lines, NT = lines * threads, SP = H + 2;
    // the largest segment whose staged pixels fit under the FFT buffers they alias (no extra LDS, same occupancy); the
    // one-point segments may take more than that
    const size_t fit_limit = std::max<size_t>(fft_lds_bytes, 48 * 1024), any_limit = 150 * 1024;
    for (int oi = 0; oi < 3; ++oi) {
        const int qs = qs_opts[oi];
        if (qs <= 0 || rf % qs || (oi && qs == qs_opts[oi - 1])) continue;
        const int nseg = rf / qs;
        out = PolarPlanHost();
        out.qs = qs; out.nseg = nseg; out.tiles = tiles; out.lines = lines; out.threads = threads; out.rf = rf; out.mf = mf;
        out.pts.assign((size_t)tiles * rf * NT * 4, 0);
        out.seg_first.assign((size_t)tiles * nseg + 1, 0);
        int worst = 0;
        for (int t = 0; t < tiles; ++t)
            for (int s = 0; s < nseg; ++s) {
                out.seg_first[(size_t)t * nseg + s] = (int)out.chunks.size();
                const int nch = plan_segment(map, PD, SP, t, s, qs, lines, threads, rf, mf, out.chunks, out.pts.data() + (size_t)t * rf * NT * 4, aligned);
                worst = std::max(worst, nch);
            }
        out.seg_first[(size_t)tiles * nseg] = (int)out.chunks.size();
        out.lds_bytes = (size_t)worst * 64;
        if (out.lds_bytes <= (qs > 1 ? fit_limit : any_limit)) return 0;
    }
    err = "polar gather: one segment's source pixels do not fit in LDS for this geometry";
    return -1;
}

// Fixed-point terms of cv::warpAffine (WarpAffineInvoker, INTER_LINEAR) for RotateArray(image, degree_arg)
// (utils.cc:154-161): the inverse of getRotationMatrix2D(center, angle, 1) in double, then
//   adelta[c] = rint(M0*c*1024), bdelta[c] = rint(M3*c*1024), X0[r] = rint((M1*r+M2)*1024)+16, Y0[r] likewise.
// Layout: [adelta W | bdelta W | X0 H | Y0 H].
void rotation_terms(int H, int W, float degree_arg, int* out) {
    const float cx = (float)(W / 2.), cy = (float)(H / 2.);
    double angle = (double)degree_arg;
    angle *= 3.1415926535897932384626433832795 / 180;
    const double alpha = cos(angle) * 1.0, beta = sin(angle) * 1.0;
    double M[6] = { alpha, beta, (1 - alpha) * cx - beta * cy, -beta, alpha, beta * cx + (1 - alpha) * cy };
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
    const double b1 = -M[0] * M[2] - M[1] * M[5];
    const double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
    const int round_delta = 1024 / 32 / 2;
    for (int c = 0; c < W; ++c) { out[c] = (int)lrint(M[0] * c * 1024); out[W + c] = (int)lrint(M[3] * c * 1024); }
    for (int r = 0; r < H; ++r) {
        out[2 * W + r] = (int)lrint((M[1] * r + M[2]) * 1024) + round_delta;
        out[2 * W + H + r] = (int)lrint((M[4] * r + M[5]) * 1024) + round_delta;
    }
}

}  // namespace kcc

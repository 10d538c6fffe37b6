// adaptor_test.cpp -- exercises the C++ drop-in adaptor (ni-slam_amd/correlation_flow_hip.h) the way
// MapBuilder does (reference src/map_builder.cc:72-75,127-131), without Eigen: a minimal column-major
// array stands in for Eigen::ArrayXXf / ArrayXXcf / Vector3d.
// Build: g++ -std=c++17 adaptor_test.cpp -I../../ni-slam_amd -L../../ni-slam_amd -lnislam_kcc_hip
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "correlation_flow_hip.h"
#include "kcc_oracle.h"          // the CPU oracle (oracle/libkcc_oracle.so): the checker of this test

template <class T> struct ColMajor {
    std::vector<T> v; long r = 0, c = 0;
    ColMajor() {}
    ColMajor(long rows, long cols) { resize(rows, cols); }
    void resize(long rows, long cols) { r = rows; c = cols; v.assign((size_t)rows * cols, T()); }
    T* data() { return v.data(); }
    const T* data() const { return v.data(); }
    long rows() const { return r; } long cols() const { return c; }
    T& operator()(long i, long j) { return v[(size_t)j * r + i]; }
    const T& operator()(long i, long j) const { return v[(size_t)j * r + i]; }
};
struct Vec3 { double d[3] = {0, 0, 0}; double& operator[](int i) { return d[i]; } double operator[](int i) const { return d[i]; } };
struct CFConfig { int width, height; float lambda; int kernel; float sigma, offset; int power; int rotation_divisor, rotation_channel; };
using CF = nislam_kcc::CorrelationFlowT<ColMajor<float>, ColMajor<std::complex<float>>, Vec3>;

static ColMajor<float> texture(int H, int W, int dy, int dx) {
    // deterministic periodic texture with a sharp autocorrelation: white noise (LCG) under a cyclic 3x3 box filter, cyclically
    // shifted by (-dy,-dx) (camera window moved by (dy,dx)).  (A sum of a few low-frequency cosines has a correlation peak so
    // broad that float32 rounding moves its arg-max by a pixel between implementations -- not a parity question.)
    static std::vector<float> noise; static int nh = 0, nw = 0;
    if (nh != H || nw != W) {
        noise.assign((size_t)H * W, 0.f); nh = H; nw = W;
        unsigned s = 12345u;
        for (float& v : noise) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) / 16777216.f; }
    }
    auto at = [&](int y, int x) { return noise[(size_t)(((y % H) + H) % H) * W + (((x % W) + W) % W)]; };
    ColMajor<float> a(H, W);
    for (int x = 0; x < W; ++x) for (int y = 0; y < H; ++y) {
        const int yy = y + dy, xx = x + dx;
        float v = 0;
        for (int j = -1; j <= 1; ++j) for (int i = -1; i <= 1; ++i) v += at(yy + j, xx + i);
        a(y, x) = v / 9.f;
    }
    return a;
}

int main(int argc, char** argv) {
    const int H = argc > 2 ? atoi(argv[1]) : 60, W = argc > 2 ? atoi(argv[2]) : 80;
    CFConfig cfg{W, H, 0.1f, 0, 0.2f, 0.1f, 3, H == 480 ? 720 : 120, H == 480 ? 480 : 80};
    double dh = H, dw = W;
    int fails = 0;
    {
        CF flow(cfg, dh, dw);
        const int dy = 4, dx = -7;
        ColMajor<float> key = texture(H, W, 0, 0), cur = texture(H, W, dy, dx);
        ColMajor<std::complex<float>> kf, kp, cf, cp;
        flow.ComputeIntermedium(key, kf, kp);            // MapBuilder::ComputeFFTResult
        flow.ComputeIntermedium(cur, cf, cp);
        if (kf.rows() != H / 2 + 1 || kf.cols() != W || kp.rows() != cfg.rotation_divisor / 2 + 1) { printf("FAIL shapes\n"); ++fails; }
        Vec3 pose; Vec3 info = flow.ComputePose(kf, cur, kp, cp, pose, true);       // MapBuilder::Tracking
        printf("pose=(%g,%g,%g) info=(%g,%g,%g)\n", pose[0], pose[1], pose[2], info[0], info[1], info[2]);
        if (pose[0] != dx || pose[1] != dy || std::fabs(std::remainder(pose[2], 6.283185307179586)) > 1e-9) { printf("FAIL pose\n"); ++fails; }
        if (!(info[0] > 5 && info[2] > 5)) { printf("FAIL psr\n"); ++fails; }
        Vec3 pose2; flow.ComputePose(kf, cur, kp, cp, pose2, false);                 // LoopClosure call pattern
        if (pose2[0] != dx || pose2[1] != dy) { printf("FAIL pose (large-rotation mode)\n"); ++fails; }
    }
    {
        // The MapBuilder call pattern against the ORACLE, pair by pair, with COPIES of the arrays handed back (MapBuilder keeps
        // `_last_fft_result = fft_result` copies, map_builder.cc:99-106): poses equal the oracle's exactly (integer arg-max),
        // PSR to its float32 tolerance -- and the adaptor's frame side table recognises the copies: not a single import.
        CF flow(cfg, dh, dw);
        ora_config oc{W, H, 0.1f, 0, 0.2f, 0.1f, 3, cfg.rotation_divisor, cfg.rotation_channel};
        ora_ctx* ora = ora_create(&oc, H, W);
        const int PD = cfg.rotation_divisor, PC = cfg.rotation_channel;
        std::vector<ora_cf32> okf((size_t)(H / 2 + 1) * W), okp((size_t)(PD / 2 + 1) * PC), ocf(okf.size()), ocp(okp.size());
        const int moves[4][2] = { { 3, 5 }, { -6, 2 }, { 0, -9 }, { 7, 7 } };
        ColMajor<float> key = texture(H, W, 0, 0);
        ColMajor<std::complex<float>> kf, kp;
        flow.ComputeIntermedium(key, kf, kp);
        ora_intermedium(ora, key.data(), okf.data(), okp.data());
        ColMajor<std::complex<float>> last_f = kf, last_p = kp;              // MapBuilder's copies
        for (int m = 0; m < 4; ++m) {
            ColMajor<float> cur = texture(H, W, moves[m][0], moves[m][1]);
            ColMajor<std::complex<float>> cf, cp;
            flow.ComputeIntermedium(cur, cf, cp);
            ColMajor<float> cur_copy = cur; ColMajor<std::complex<float>> cp_copy = cp;
            for (int mode = 0; mode < 2; ++mode) {
                Vec3 pose; Vec3 info = flow.ComputePose(last_f, cur_copy, last_p, cp_copy, pose, mode == 0);
                ora_intermedium(ora, cur.data(), ocf.data(), ocp.data());
                double op[3], oi[3];
                ora_compute_pose(ora, okf.data(), cur.data(), okp.data(), ocp.data(), mode == 0, 0, op, oi, nullptr);
                const bool same = pose[0] == op[0] && pose[1] == op[1] && std::fabs(std::remainder(pose[2] - op[2], 6.283185307179586)) < 1e-6 &&
                                  std::fabs(info[0] - oi[0]) <= 5e-3 * std::fabs(oi[0]) && std::fabs(info[2] - oi[2]) <= 5e-3 * std::fabs(oi[2]);
                if (!same) { printf("FAIL vs oracle move %d mode %d: (%g,%g,%g | %g,%g) oracle (%g,%g,%g | %g,%g)\n", m, mode, pose[0], pose[1], pose[2], info[0], info[2], op[0], op[1], op[2], oi[0], oi[2]); ++fails; }
            }
            if (m == 1) { last_f = cf; last_p = cp; okf = ocf; okp = ocp; }    // a key-frame switch (UpdateIntermedium)
        }
        const CF::Stats st = flow.stats();
        printf("side table: %ld hits, %ld imports over %ld poses\n", st.table_hits, st.imports, st.poses);
        if (st.imports != 0 || st.table_hits != 2 * st.poses) { printf("FAIL side table: MapBuilder's pattern must not import\n"); ++fails; }
        // arrays the table does not know (a spectrum computed elsewhere: here an edited copy) ARE imported, and honoured
        ColMajor<float> cur = texture(H, W, 2, 1);
        ColMajor<std::complex<float>> cf, cp;
        flow.ComputeIntermedium(cur, cf, cp);
        ColMajor<std::complex<float>> edited = last_f;
        for (long i = 0; i < edited.rows() * edited.cols(); ++i) edited.data()[i] *= 2.0f;      // scaling the key spectrum leaves the pose unchanged
        Vec3 p1, p2;
        flow.ComputePose(last_f, cur, last_p, cp, p1, true);
        const long before = flow.stats().imports;
        flow.ComputePose(edited, cur, last_p, cp, p2, true);
        if (flow.stats().imports != before + 1) { printf("FAIL side table: an unknown key spectrum was not imported\n"); ++fails; }
        if (p1[0] != p2[0] || p1[1] != p2[1]) { printf("FAIL pose after import\n"); ++fails; }
        // an in-place edit of ONE element anywhere in an exported array (here one the old 65-sample fingerprint never looked at)
        // must be noticed: the array is imported instead of the stale device copy being used (by-value semantics)
        {
            ColMajor<std::complex<float>> one = cp;
            const long n = one.rows() * one.cols(), at = n / 2 + 37;
            one.data()[at] += std::complex<float>(1e-3f, 0.f);
            const long b2 = flow.stats().imports;
            Vec3 p3;
            flow.ComputePose(last_f, cur, last_p, one, p3, true);
            if (flow.stats().imports != b2 + 1) { printf("FAIL side table: a single edited element went unnoticed\n"); ++fails; }
            ColMajor<float> img1 = cur;
            img1.data()[(long)H * W / 3 + 5] += 0.25f;
            const long b3 = flow.stats().imports;
            flow.ComputePose(last_f, img1, last_p, cp, p3, true);
            if (flow.stats().imports <= b3) { printf("FAIL side table: an edited image went unnoticed\n"); ++fails; }
        }
        ora_destroy(ora);
    }
    {
        CFConfig bad = cfg; bad.kernel = 9;
        CF flow(bad, dh, dw);
        ColMajor<float> img = texture(H, W, 0, 0);
        ColMajor<std::complex<float>> f, p;
        flow.ComputeIntermedium(img, f, p);
        bool threw = false;
        try { Vec3 pose; flow.ComputePose(f, img, p, p, pose, true); } catch (const std::invalid_argument& e) { threw = std::string(e.what()) == "Received invalid kernel type"; }
        if (!threw) { printf("FAIL invalid kernel did not throw std::invalid_argument\n"); ++fails; }
    }
    printf(fails ? "ADAPTOR TEST FAILED\n" : "ADAPTOR TEST OK\n");
    return fails ? 1 : 0;
}

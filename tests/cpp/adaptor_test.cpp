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
    // deterministic band-limited texture, cyclically shifted by (-dy,-dx) (camera window moved by (dy,dx))
    ColMajor<float> a(H, W);
    unsigned s = 12345u; float ph[24][3];
    for (auto& p : ph) for (float& q : p) { s = s * 1664525u + 1013904223u; q = (float)(s >> 8) / 16777216.f; }
    for (int x = 0; x < W; ++x) for (int y = 0; y < H; ++y) {
        const int yy = ((y + dy) % H + H) % H, xx = ((x + dx) % W + W) % W;
        float v = 0;
        for (int k = 0; k < 24; ++k) {
            const int fy = 1 + (int)(ph[k][0] * 9), fx = 1 + (int)(ph[k][1] * 9);
            v += std::cos(6.2831853f * (fy * yy / (float)H + fx * xx / (float)W + ph[k][2]));
        }
        a(y, x) = 0.5f + v / 48.f;
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

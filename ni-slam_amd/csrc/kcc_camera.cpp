// kcc_camera.cpp -- host side of the camera model that sits right before the KCC path: the map construction of
// Camera::Camera (reference src/camera.cc:45-47)
//     _new_K = getOptimalNewCameraMatrix(_K, _D, image_size, 0, image_size);
//     initUndistortRectifyMap(_K, _D, cv::Mat(), _new_K, image_size, CV_16SC2, _map1, _map2);
// restated without OpenCV (4.2 semantics, double arithmetic, 1/32 px fixed-point maps).  The maps feed
// nik_set_undistort(); the per-frame cv::remap (Camera::UndistortImage, camera.cc:92-93) runs on the GPU.
// Runs once per camera: plain scalar code.
#include "../../include/nislam_kcc.h"

#include <cfloat>
#include <cmath>
#include <cstdint>

namespace {

constexpr int kInterBits = 5, kInterTab = 1 << kInterBits;    // cv::INTER_BITS, INTER_TAB_SIZE

struct Intrinsics { double fx, cx, fy, cy; };
struct Distortion { double k1, k2, p1, p2, k3; };

// cvUndistortPoints with the C API's default termination (5 iterations), R = P = identity: pixel -> normalised
void undistort_normalised(double u, double v, const Intrinsics& K, const Distortion& d, double& xn, double& yn) {
    const double xs = (u - K.cx) * (1.0 / K.fx), ys = (v - K.cy) * (1.0 / K.fy);
    double x = xs, y = ys;
    for (int it = 0; it < 5; ++it) {
        const double r2 = x * x + y * y;
        const double icdist = 1.0 / (1 + ((d.k3 * r2 + d.k2) * r2 + d.k1) * r2);
        if (icdist < 0) { x = xs; y = ys; break; }
        const double dx = 2 * d.p1 * x * y + d.p2 * (r2 + 2 * x * x);
        const double dy = d.p1 * (r2 + 2 * y * y) + 2 * d.p2 * x * y;
        x = (xs - dx) * icdist;
        y = (ys - dy) * icdist;
    }
    xn = x; yn = y;
}

// getOptimalNewCameraMatrix(alpha = 0, same size, centerPrincipalPoint = false): the rectangle inscribed in the
// undistorted 9 x 9 sample grid (single-precision points and rectangle, as in icvGetRectangles) fills the viewport
Intrinsics optimal_new_matrix(const Intrinsics& K, const Distortion& d, int width, int height) {
    const int N = 9;
    float left = -FLT_MAX, right = FLT_MAX, top = -FLT_MAX, bottom = FLT_MAX;
    for (int gy = 0; gy < N; ++gy)
        for (int gx = 0; gx < N; ++gx) {
            const float u = (float)gx * width / (N - 1), v = (float)gy * height / (N - 1);
            double xn, yn;
            undistort_normalised(u, v, K, d, xn, yn);
            const float xf = (float)xn, yf = (float)yn;
            if (gx == 0) left = std::fmax(left, xf);
            if (gx == N - 1) right = std::fmin(right, xf);
            if (gy == 0) top = std::fmax(top, yf);
            if (gy == N - 1) bottom = std::fmin(bottom, yf);
        }
    const float rw = right - left, rh = bottom - top;
    Intrinsics n;
    n.fx = (width - 1) / rw;  n.cx = -n.fx * left;       // (int / float: a single-precision quotient, then widened)
    n.fy = (height - 1) / rh; n.cy = -n.fy * top;
    return n;
}

}  // namespace

extern "C" int nik_camera_maps(const double K[4], const double D[5], int width, int height, double newK[4],
                               int16_t* map1, uint16_t* map2) {
    if (!K || !D || !newK || !map1 || !map2 || width <= 0 || height <= 0) return NIK_ERR_INVALID_ARG;
    if (K[0] == 0 || K[2] == 0) return NIK_ERR_INVALID_ARG;
    const Intrinsics k{ K[0], K[1], K[2], K[3] };
    const Distortion d{ D[0], D[1], D[2], D[3], D[4] };
    const Intrinsics nk = optimal_new_matrix(k, d, width, height);
    if (!std::isfinite(nk.fx) || !std::isfinite(nk.fy) || nk.fx == 0 || nk.fy == 0) return NIK_ERR_INVALID_ARG;
    newK[0] = nk.fx; newK[1] = nk.cx; newK[2] = nk.fy; newK[3] = nk.cy;
    // initUndistortRectifyMap(R = I): destination pixel (col, row) -> normalised ray through new_K^-1 -> distort -> K.
    // The reference accumulates the normalised x along a row (x += 1/fx' per column); kept, it decides roundings.
    const double step_x = 1.0 / nk.fx, x_first = -nk.cx / nk.fx, step_y = 1.0 / nk.fy, y_first = -nk.cy / nk.fy;
    for (int row = 0; row < height; ++row) {
        double x = x_first;
        const double y = row * step_y + y_first;
        for (int col = 0; col < width; ++col, x += step_x) {
            const double x2 = x * x, y2 = y * y, r2 = x2 + y2, xy2 = 2 * x * y;
            const double radial = 1 + ((d.k3 * r2 + d.k2) * r2 + d.k1) * r2;
            const double xd = x * radial + d.p1 * xy2 + d.p2 * (r2 + 2 * x2);
            const double yd = y * radial + d.p1 * (r2 + 2 * y2) + d.p2 * xy2;
            const int iu = (int)std::nearbyint((k.fx * xd + k.cx) * kInterTab);    // cvRound: round-half-even
            const int iv = (int)std::nearbyint((k.fy * yd + k.cy) * kInterTab);
            const size_t i = (size_t)row * width + col;
            map1[2 * i + 0] = (int16_t)(iu >> kInterBits);
            map1[2 * i + 1] = (int16_t)(iv >> kInterBits);
            map2[i] = (uint16_t)((iv & (kInterTab - 1)) * kInterTab + (iu & (kInterTab - 1)));
        }
    }
    return NIK_OK;
}

// correlation_flow_hip.h -- header-only C++ adaptor that re-creates the reference's
// `class CorrelationFlow` (reference include/correlation_flow.h:8-33) on top of the C ABI of
// libnislam_kcc_hip.so (include/nislam_kcc.h), so that MapBuilder (src/map_builder.cc:23,72-75,
// 127-131) and LoopClosure (src/loop_closure.cc:55-59) compile and behave unchanged.
//
//   #include "correlation_flow_hip.h"          // instead of "correlation_flow.h"
//   -> CorrelationFlow / CorrelationFlowPtr are defined here when <Eigen/Core> is available.
//
// The class is a template over the array types so it can be compiled and tested without Eigen
// (tests/cpp/adaptor_test.cpp instantiates it with a 30-line column-major array).  Requirements on
// the array types: value_type-like  data(), rows(), cols(), resize(rows, cols); Vec3: operator[].
//
// Semantics kept from the reference:
//   * ctor overrides cfg.height/width with the camera's image size (correlation_flow.cc:40-41);
//   * ComputeIntermedium(image, fft_result&, fft_polar&) fills caller-owned arrays (:89-95);
//   * ComputePose(...) returns `info` and fills `pose` (:97-143); an invalid cfg.kernel throws
//     std::invalid_argument("Received invalid kernel type") at ComputePose time (:167-168);
//   * single-threaded, synchronous, non-reentrant (one context; shared via shared_ptr like :33).
// Differences: no std::cout of pose/info (:139-140) and no dead `rectify` warp (:141).
#pragma once

#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>

#include "../include/nislam_kcc.h"

#if defined(__has_include)
#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#define NISLAM_KCC_HAVE_EIGEN 1
#endif
#endif

namespace nislam_kcc {

// mirrors CFConfig (reference include/read_configs.h:15-25); when the reference's read_configs.h is
// in the include path, pass its CFConfig directly -- any struct with these nine members works.
template <class CFConfigT>
inline nik_config to_nik_config(const CFConfigT& c) {
    nik_config n;
    n.width = c.width; n.height = c.height; n.lambda = c.lambda; n.kernel = c.kernel; n.sigma = c.sigma;
    n.offset = c.offset; n.power = c.power; n.rotation_divisor = c.rotation_divisor; n.rotation_channel = c.rotation_channel;
    return n;
}

template <class ArrayXXf, class ArrayXXcf, class Vector3d>
class CorrelationFlowT {
public:
    template <class CFConfigT>
    CorrelationFlowT(CFConfigT& cf_config, double& image_height, double& image_width, int device = 0)
        : H_((int)image_height), W_((int)image_width) {
        nik_config n = to_nik_config(cf_config);
        PD_ = n.rotation_divisor; PC_ = n.rotation_channel;
        // slots: 0 = key (last_fft_*), 1 = current; batch of 1 pair (the reference's call pattern)
        const int rc = nik_create(&n, H_, W_, /*max_batch=*/1, /*max_frames=*/2, device, &ctx_);
        if (rc != NIK_OK) throw std::runtime_error(std::string("nik_create: ") + nik_last_error(nullptr));
    }
    ~CorrelationFlowT() { nik_destroy(ctx_); }
    CorrelationFlowT(const CorrelationFlowT&) = delete;
    CorrelationFlowT& operator=(const CorrelationFlowT&) = delete;

    // void ComputeIntermedium(const ArrayXXf&, ArrayXXcf&, ArrayXXcf&)      correlation_flow.cc:89-95
    void ComputeIntermedium(const ArrayXXf& image, ArrayXXcf& fft_result, ArrayXXcf& fft_polar) {
        check(nik_intermedium_f32(ctx_, image.data(), 1));
        fft_result.resize(H_ / 2 + 1, W_);
        fft_polar.resize(PD_ / 2 + 1, PC_);
        check(nik_frame_export(ctx_, 1, nullptr, reinterpret_cast<float*>(fft_result.data()),
                               reinterpret_cast<float*>(fft_polar.data())));
    }

    // Vector3d ComputePose(last_fft_result, image, last_fft_polar, fft_polar, pose&, not_large_rotation)   :97-143
    Vector3d ComputePose(const ArrayXXcf& last_fft_result, const ArrayXXf& image, const ArrayXXcf& last_fft_polar,
                         const ArrayXXcf& fft_polar, Vector3d& pose, bool not_large_rotation) {
        check(nik_frame_import(ctx_, 0, image.data() /*flag only: key image is never read*/,
                               reinterpret_cast<const float*>(last_fft_result.data()),
                               reinterpret_cast<const float*>(last_fft_polar.data())));
        // the current frame contributes its image and polar spectrum; its fft_result is not an input of ComputePose
        check(nik_frame_import(ctx_, 1, image.data(), reinterpret_cast<const float*>(last_fft_result.data()),
                               reinterpret_cast<const float*>(fft_polar.data())));
        double p[3], i[3];
        const int rc = nik_pose(ctx_, 0, 1, not_large_rotation ? 1 : 0, p, i, nullptr);
        if (rc == NIK_ERR_INVALID_KERNEL) throw std::invalid_argument("Received invalid kernel type");
        check(rc);
        Vector3d info;
        for (int k = 0; k < 3; ++k) { pose[k] = p[k]; info[k] = i[k]; }
        return info;
    }

    nik_ctx* context() const { return ctx_; }      // for batched / device-resident use beyond the reference API

private:
    void check(int rc) const {
        if (rc != NIK_OK) throw std::runtime_error(std::string("nislam_kcc: ") + nik_last_error(ctx_));
    }
    nik_ctx* ctx_ = nullptr;
    int H_, W_, PD_ = 0, PC_ = 0;
};

}  // namespace nislam_kcc

#ifdef NISLAM_KCC_HAVE_EIGEN
// drop-in names of the reference header (include/correlation_flow.h:8,33)
using CorrelationFlow = nislam_kcc::CorrelationFlowT<Eigen::ArrayXXf, Eigen::ArrayXXcf, Eigen::Vector3d>;
typedef std::shared_ptr<CorrelationFlow> CorrelationFlowPtr;
#endif

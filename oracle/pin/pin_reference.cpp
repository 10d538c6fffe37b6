// pin_reference.cpp -- runs the REAL reference (its own, unmodified src/correlation_flow.cc + src/utils.cc, compiled in place
// from -DNISLAM_REFERENCE) and the REAL OpenCV 4.x / FFTW3f / Eigen 3 on seeded inputs and writes the vectors that pin
// oracle/kcc_oracle.c: tests/golden/ref_pairs.json, ref_recalled.json and ref_*.bin.  See README.md.  This program cannot be
// built in the image the repository was developed in (none of those libraries exist there); it is the recipe for whoever has
// them.  usage: pin_reference <dir with manifest.txt and *.u8 from make_inputs.py> <output dir (tests/golden)>
//
// The private non-inline members of CorrelationFlow (FFT, IFFT, EstimateTrans, target_fft) are reached by compiling THIS translation unit with
// `private` spelled `public`; the reference's own sources are compiled unmodified.
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include <opencv2/calib3d.hpp>
#include <opencv2/imgproc.hpp>

#define private public
#include "correlation_flow.h"
#undef private
#include "read_configs.h"
#include "utils.h"

using Eigen::ArrayXXcf;
using Eigen::ArrayXXf;

// 31-bit LCG the Python side reproduces (tests/test_ref_golden.py lcg_floats / lcg_bytes): 24-bit fractions, exact in float
struct Lcg {
    uint32_t s;
    explicit Lcg(uint32_t seed) : s(seed) {}
    uint32_t next() { s = s * 1103515245u + 12345u; return s; }
    float unit() { return (float)((next() >> 8) & 0xFFFFFFu) / 16777216.0f; }
    uint8_t byte() { return (uint8_t)((next() >> 16) & 0xFFu); }
};

static void write_bin(const std::string& path, const void* p, size_t bytes) {
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(p), (std::streamsize)bytes);
}
// column-major float array -> file (Eigen's own storage order; the Python side reads order='F')
static void dump(const std::string& path, const ArrayXXf& a) { write_bin(path, a.data(), sizeof(float) * (size_t)a.size()); }
static void dump(const std::string& path, const ArrayXXcf& a) { write_bin(path, a.data(), sizeof(float) * 2 * (size_t)a.size()); }

static ArrayXXf lcg_array(int rows, int cols, uint32_t seed) {      // filled in ROW-major order (r outer, c inner)
    Lcg g(seed);
    ArrayXXf a(rows, cols);
    for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) a(r, c) = g.unit();
    return a;
}

static CFConfig make_cfg(int H, int W, int PD, int PC) {
    CFConfig c;                                    // values of configs/config_ntu.yaml (what oracle.default_config mirrors)
    c.width = W; c.height = H; c.lambda = 0.1f; c.kernel = 0; c.sigma = 0.2f; c.offset = 0.1f; c.power = 3;
    c.rotation_divisor = PD; c.rotation_channel = PC;
    return c;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s <work dir> <golden dir>\n", argv[0]); return 2; }
    const std::string work = argv[1], out = argv[2];

    // ---------------------------------------------------------------------------------------------------------------
    // 1. whole-path vectors: ComputeIntermedium + ComputePose of the reference on the seeded tests/synth.py pairs
    // ---------------------------------------------------------------------------------------------------------------
    std::ifstream man(work + "/manifest.txt");
    if (!man) { fprintf(stderr, "no manifest.txt in %s (run oracle/pin/make_inputs.py)\n", work.c_str()); return 1; }
    std::ofstream js(out + "/ref_pairs.json");
    js.precision(17);
    js << "{\n \"note\": \"outputs of the REAL reference (sair-lab/ni-slam, its own correlation_flow.cc/utils.cc with FFTW3f, Eigen3, OpenCV "
       << CV_VERSION << ") on the seeded pairs of oracle/pin/make_inputs.py\",\n \"cases\": [\n";
    std::string name; int H, W, PD, PC, n, small_rot; bool first_case = true;
    while (man >> name >> H >> W >> PD >> PC >> n >> small_rot) {
        std::vector<uint8_t> keys((size_t)n * H * W), curs((size_t)n * H * W);
        std::ifstream(work + "/" + name + "_keys.u8", std::ios::binary).read(reinterpret_cast<char*>(keys.data()), (std::streamsize)keys.size());
        std::ifstream(work + "/" + name + "_curs.u8", std::ios::binary).read(reinterpret_cast<char*>(curs.data()), (std::streamsize)curs.size());
        CFConfig cfg = make_cfg(H, W, PD, PC);
        double dh = H, dw = W;
        CorrelationFlow cf(cfg, dh, dw);
        js << (first_case ? "" : ",\n") << "  {\"name\": \"" << name << "\", \"H\": " << H << ", \"W\": " << W << ", \"PD\": " << PD << ", \"PC\": " << PC
           << ", \"not_large_rotation\": " << small_rot << ", \"pairs\": [\n";
        first_case = false;
        for (int i = 0; i < n; ++i) {
            cv::Mat km(H, W, CV_8UC1, keys.data() + (size_t)i * H * W), cm(H, W, CV_8UC1, curs.data() + (size_t)i * H * W);
            ArrayXXf ka, ca;
            ConvertMatToNormalizedArray(km, ka);
            ConvertMatToNormalizedArray(cm, ca);
            ArrayXXcf kf, kp, xf, xp;
            cf.ComputeIntermedium(ka, kf, kp);
            cf.ComputeIntermedium(ca, xf, xp);
            Eigen::Vector3d pose;
            const Eigen::Vector3d info = cf.ComputePose(kf, ca, kp, xp, pose, small_rot != 0);
            // probes of the spectra (full planes only for the first pair of the small case: ref_small_*.bin)
            js << (i ? ",\n" : "") << "   {\"pose\": [" << pose[0] << ", " << pose[1] << ", " << pose[2] << "], \"info\": [" << info[0] << ", " << info[1] << ", "
               << info[2] << "], \"F_probe\": [" << kf(0, 0).real() << ", " << kf(1, 2).real() << ", " << kf(1, 2).imag() << ", " << kf(H / 2, W - 1).real()
               << "], \"P_probe\": [" << kp(0, 0).real() << ", " << kp(1, 2).real() << ", " << kp(1, 2).imag() << ", " << kp(PD / 2, PC - 1).real()
               << "], \"F_abs_sum\": " << (double)kf.abs().sum() << ", \"P_abs_sum\": " << (double)kp.abs().sum() << "}";
            if (i == 0 && name == "small") { dump(out + "/ref_small_F.bin", kf); dump(out + "/ref_small_P.bin", kp); }
        }
        js << "\n  ]}";
    }
    js << "\n ]\n}\n";
    js.close();

    // ---------------------------------------------------------------------------------------------------------------
    // 2. the experiments of oracle/RECALLED.md on the real libraries (60 x 80 plane, polar 120 x 80: whole arrays are small)
    // ---------------------------------------------------------------------------------------------------------------
    const int h = 60, w = 80, pd = 120, pc = 80;
    std::ofstream rj(out + "/ref_recalled.json");
    rj.precision(17);
    rj << "{\n \"opencv\": \"" << CV_VERSION << "\", \"H\": " << h << ", \"W\": " << w << ", \"PD\": " << pd << ", \"PC\": " << pc << ",\n";
    const ArrayXXf x = lcg_array(h, w, 12345u);
    {
        // #1-#3  cv::warpPolar exactly as CorrelationFlow::polar calls it (correlation_flow.cc:228-236)
        CFConfig cfg = make_cfg(h, w, pd, pc); double dh = h, dw = w;
        CorrelationFlow cf(cfg, dh, dw);
        // (CorrelationFlow::polar is an `inline` member defined inside the reference's .cc -- there is no out-of-line symbol to link
        // against -- so RECALLED.md #1 is pinned by calling cv::warpPolar here with the arguments documented there: dsize =
        // (rotation_channel, rotation_divisor), centre (cols/2, rows/2) as floats, maxRadius = min(rows/2, cols/2), linear mode)
        {
            cv::Mat polar_img, img = ConvertArrayToMat(x);
            cv::Point2f center((float)img.cols / 2, (float)img.rows / 2);
            const double radius = (double)std::min(img.rows / 2, img.cols / 2);
            cv::warpPolar(img, polar_img, cv::Size(pc, pd), center, radius, cv::INTER_LINEAR + cv::WARP_FILL_OUTLIERS);
            dump(out + "/ref_polar.bin", ConvertMatToArray(polar_img));         // [pd x pc] column-major
        }
        // #12-#14 the reference's FFT / IFFT (FFTW3f r2c / c2r with its (cols, rows) argument order)
        const ArrayXXcf xf = cf.FFT(x);
        dump(out + "/ref_fft.bin", xf);                                          // [(h/2+1) x w] column-major, interleaved
        dump(out + "/ref_ifft.bin", cf.IFFT(xf));
        // #13 c2r of a spectrum with non-zero imaginary parts in its DC and Nyquist rows
        ArrayXXcf bad = xf;
        for (int c = 0; c < w; ++c) { bad(0, c) += std::complex<float>(0.f, 3.5f); bad(h / 2, c) += std::complex<float>(0.f, -2.25f); }
        dump(out + "/ref_ifft_nonhermitian.bin", cf.IFFT(bad));
        // EstimateTrans on two LCG planes (response arg-max + PSR): pins the kernel / ridge / arg-max chain in one number set
        const ArrayXXf z = lcg_array(h, w, 777u);
        Eigen::Vector2d tr;
        const float psr = cf.EstimateTrans(cf.FFT(z), xf, cf.target_fft, h, w, tr);
        rj << " \"estimate_trans\": {\"trans\": [" << tr[0] << ", " << tr[1] << "], \"psr\": " << psr << "},\n";
    }
    {
        // #4-#6  RotateArray (utils.cc:154-161): getRotationMatrix2D + warpAffine(INTER_LINEAR, BORDER_WRAP)
        const float degs[5] = { 0.5f, 37.f, 180.f, -12.5f, 90.f };
        rj << " \"rotate_degrees\": [0.5, 37, 180, -12.5, 90],\n";
        for (int k = 0; k < 5; ++k) dump(out + "/ref_rotate_" + std::to_string(k) + ".bin", RotateArray(x, degs[k]));
    }
    {
        // #9-#10 getOptimalNewCameraMatrix + initUndistortRectifyMap(CV_16SC2) as Camera::Camera (camera.cc:45-47); #8 remap on u8
        const double K[4] = { 52.0, 39.6, 51.5, 30.2 }, D[5] = { -0.28, 0.09, 0.001, -0.0007, 0.0 };      // tests/golden "small_barrel"
        cv::Mat Km = (cv::Mat_<double>(3, 3) << K[0], 0.0, K[1], 0.0, K[2], K[3], 0.0, 0.0, 1.0);
        cv::Mat Dm = (cv::Mat_<double>(5, 1) << D[0], D[1], D[2], D[3], D[4]);
        const cv::Size size(w, h);
        cv::Mat newK = cv::getOptimalNewCameraMatrix(Km, Dm, size, 0, size), map1, map2;
        cv::initUndistortRectifyMap(Km, Dm, cv::Mat(), newK, size, CV_16SC2, map1, map2);
        rj << " \"camera\": {\"K\": [52.0, 39.6, 51.5, 30.2], \"D\": [-0.28, 0.09, 0.001, -0.0007, 0.0], \"new_K\": [" << newK.at<double>(0, 0) << ", "
           << newK.at<double>(0, 2) << ", " << newK.at<double>(1, 1) << ", " << newK.at<double>(1, 2) << "]},\n";
        write_bin(out + "/ref_map1.bin", map1.data, (size_t)h * w * 4);         // int16 pairs, row-major
        write_bin(out + "/ref_map2.bin", map2.data, (size_t)h * w * 2);         // uint16, row-major
        Lcg g(999u);
        cv::Mat img(h, w, CV_8UC1), und;
        for (int r = 0; r < h; ++r) for (int c = 0; c < w; ++c) img.at<uint8_t>(r, c) = g.byte();
        cv::remap(img, und, map1, map2, cv::INTER_LINEAR);
        write_bin(out + "/ref_remap_u8.bin", und.data, (size_t)h * w);
        // #11 cvtColor(RGB2GRAY) on u8
        cv::Mat rgb(h, w, CV_8UC3), gray;
        Lcg g2(4242u);
        for (int r = 0; r < h; ++r) for (int c = 0; c < w; ++c) rgb.at<cv::Vec3b>(r, c) = cv::Vec3b(g2.byte(), g2.byte(), g2.byte());
        cv::cvtColor(rgb, gray, cv::COLOR_RGB2GRAY);
        write_bin(out + "/ref_rgb2gray.bin", gray.data, (size_t)h * w);
    }
    {
        // #15 maxCoeff on a column-major array with two equal maxima; #16 Array::pow(int)
        ArrayXXf t = x;
        t(7, 3) = 5.f; t(2, 9) = 5.f; t(40, 3) = 5.f;
        Eigen::Index r, c;
        t.maxCoeff(&r, &c);
        rj << " \"maxcoeff_tie\": {\"set\": [[7, 3], [2, 9], [40, 3]], \"row\": " << (long)r << ", \"col\": " << (long)c << "},\n";
        const ArrayXXf p3 = (x + 0.1f).pow(3);
        dump(out + "/ref_pow3.bin", p3);
    }
    rj << " \"lcg\": \"s = s*1103515245 + 12345 (uint32); unit = ((s >> 8) & 0xFFFFFF) / 2^24; byte = (s >> 16) & 0xFF; planes filled row by row; x: seed 12345, z: seed 777\"\n}\n";
    printf("wrote %s/ref_pairs.json, ref_recalled.json, ref_*.bin\n", out.c_str());
    return 0;
}

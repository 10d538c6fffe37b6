"""ctypes binding of the CPU oracle (oracle/kcc_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from the product path (ni-slam_amd/).

Array convention: a reference ``Eigen::ArrayXXf`` (column-major rows x cols) is a C-order
numpy array of shape ``(cols, rows)``; a half spectrum ``ArrayXXcf`` ((rows/2+1) x cols) is
a complex64 array of shape ``(cols, rows//2+1)``.  u8 images are ``(rows, cols)`` like cv::Mat.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libkcc_oracle.so")


class OraConfig(C.Structure):
    """mirrors CFConfig (/root/reference/include/read_configs.h:15-25)"""
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("lambda_", C.c_float), ("kernel", C.c_int),
                ("sigma", C.c_float), ("offset", C.c_float), ("power", C.c_int),
                ("rotation_divisor", C.c_int), ("rotation_channel", C.c_int)]


class OraPoseDebug(C.Structure):
    _fields_ = [("rot_row", C.c_int), ("rot_col", C.c_int),
                ("trans_row", C.c_int * 2), ("trans_col", C.c_int * 2),
                ("psr_rot", C.c_float), ("rot_peak", C.c_float), ("rot_mirror", C.c_float),
                ("psr_trans", C.c_float * 2),
                ("degree_used", C.c_float * 2), ("degree_final", C.c_float),
                ("chosen", C.c_int), ("n_hyp", C.c_int), ("rot_forced", C.c_int)]

    def as_dict(self):
        return dict(rot_row=self.rot_row, rot_col=self.rot_col,
                    trans_row=list(self.trans_row), trans_col=list(self.trans_col),
                    psr_rot=float(self.psr_rot), rot_peak=float(self.rot_peak),
                    rot_mirror=float(self.rot_mirror), psr_trans=[float(v) for v in self.psr_trans],
                    degree_used=[float(v) for v in self.degree_used], degree_final=float(self.degree_final),
                    chosen=self.chosen, n_hyp=self.n_hyp)


def build(force=False):
    """Compile the oracle with its Makefile (gcc); no-op when up to date."""
    src = os.path.join(_HERE, "kcc_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        P = C.c_void_p
        L.ora_create.restype = P
        L.ora_create.argtypes = [C.POINTER(OraConfig), C.c_int, C.c_int]
        L.ora_destroy.argtypes = [P]
        L.ora_normalize_u8.argtypes = [P, C.c_int, C.c_int, P]
        L.ora_fft.argtypes = [P, P, C.c_int, C.c_int, P]
        L.ora_ifft.argtypes = [P, P, C.c_int, C.c_int, P]
        L.ora_remove_zero.argtypes = [P, C.c_int, C.c_int, P]
        L.ora_fftshift.argtypes = [P, C.c_int, C.c_int, P]
        L.ora_polar.argtypes = [P, P, P]
        L.ora_rotate.argtypes = [P, C.c_int, C.c_int, C.c_float, P]
        L.ora_normalize_degree.restype = C.c_double
        L.ora_normalize_degree.argtypes = [C.c_double]
        L.ora_set_pow_mode.argtypes = [C.c_int]
        L.ora_get_pow_mode.restype = C.c_int
        L.ora_intermedium.argtypes = [P, P, P, P]
        L.ora_estimate_trans.restype = C.c_float
        L.ora_estimate_trans.argtypes = [P, P, P, C.c_int, P, P, P, P, P]
        L.ora_get_info.restype = C.c_float
        L.ora_get_info.argtypes = [P, C.c_long, C.c_float]
        L.ora_compute_pose.argtypes = [P, P, P, P, P, C.c_int, C.c_int, P, P, P]
        L.ora_track_pairs.argtypes = [C.POINTER(OraConfig), C.c_int, C.c_int, C.c_int, P, P,
                                      C.c_int, C.c_int, C.c_int, P, P, P, P]
        L.ora_optimal_new_camera_matrix.argtypes = [P, P, C.c_int, C.c_int, P]
        L.ora_undistort_maps.argtypes = [P, P, P, C.c_int, C.c_int, P, P]
        L.ora_remap_u8.argtypes = [P, C.c_int, C.c_int, P, P, P]
        L.ora_set_window.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ora_force_rotation.argtypes = [P, C.c_int, C.c_int]
        L.ora_downsample_u8.argtypes = [P, C.c_int, C.c_int, P]
        _lib = L
    return _lib


def default_config(kernel=0, rotation_divisor=720, rotation_channel=480, power=3):
    """values of /root/reference/configs/config_ntu.yaml:6-17"""
    return OraConfig(width=640, height=480, lambda_=0.1, kernel=kernel, sigma=0.2, offset=0.1,
                     power=power, rotation_divisor=rotation_divisor, rotation_channel=rotation_channel)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """CorrelationFlow restated on the CPU (include/correlation_flow.h:8-31)."""

    def __init__(self, cfg, H, W):
        self.cfg, self.H, self.W = cfg, H, W
        self.PD, self.PC = cfg.rotation_divisor, cfg.rotation_channel
        self._ctx = lib().ora_create(C.byref(cfg), H, W)
        if not self._ctx:
            raise ValueError("ora_create failed (odd height / bad config)")

    def __del__(self):
        if getattr(self, "_ctx", None):
            try:
                lib().ora_destroy(self._ctx)
            except TypeError:                  # interpreter shutdown: the module globals are gone already
                pass
            self._ctx = None

    def force_rotation(self, row=-1, col=-1):
        """test hook: impose the rotation arg-max of the following compute_pose calls (row < 0: off)"""
        lib().ora_force_rotation(self._ctx, int(row), int(col))

    def set_window(self, rot_row, rot_col, trans_row, trans_col, radius):
        """coarse-to-fine extension: restrict both arg-max searches of the following calls (radius < 0: off)"""
        lib().ora_set_window(self._ctx, int(rot_row), int(rot_col), int(trans_row), int(trans_col), int(radius))

    @staticmethod
    def normalize_u8(img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        H, W = img.shape
        out = np.empty((W, H), np.float32)
        lib().ora_normalize_u8(_p(img), H, W, _p(out))
        return out

    def fft(self, x):
        x = np.ascontiguousarray(x, np.float32)
        cols, rows = x.shape
        out = np.empty((cols, rows // 2 + 1), np.complex64)
        lib().ora_fft(self._ctx, _p(x), rows, cols, _p(out))
        return out

    def ifft(self, xf):
        xf = np.ascontiguousarray(xf, np.complex64)
        cols, hr = xf.shape
        out = np.empty((cols, (hr - 1) * 2), np.float32)
        lib().ora_ifft(self._ctx, _p(xf), hr, cols, _p(out))
        return out

    @staticmethod
    def remove_zero(x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty_like(x)
        lib().ora_remove_zero(_p(x), x.shape[1], x.shape[0], _p(out))
        return out

    @staticmethod
    def fftshift(x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty_like(x)
        lib().ora_fftshift(_p(x), x.shape[1], x.shape[0], _p(out))
        return out

    def polar(self, x):
        x = np.ascontiguousarray(x, np.float32)
        assert x.shape == (self.W, self.H)
        out = np.empty((self.PC, self.PD), np.float32)
        lib().ora_polar(self._ctx, _p(x), _p(out))
        return out

    @staticmethod
    def rotate(x, degree):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty_like(x)
        lib().ora_rotate(_p(x), x.shape[1], x.shape[0], C.c_float(degree), _p(out))
        return out

    def intermedium(self, image):
        image = np.ascontiguousarray(image, np.float32)
        assert image.shape == (self.W, self.H)
        f = np.empty((self.W, self.H // 2 + 1), np.complex64)
        fp = np.empty((self.PC, self.PD // 2 + 1), np.complex64)
        lib().ora_intermedium(self._ctx, _p(image), _p(f), _p(fp))
        return f, fp

    def estimate_trans(self, last_fft, cur_fft, which, want_g=False):
        last_fft = np.ascontiguousarray(last_fft, np.complex64)
        cur_fft = np.ascontiguousarray(cur_fft, np.complex64)
        h, w = (self.PD, self.PC) if which else (self.H, self.W)
        trans = np.zeros(2, np.float64)
        row, col, err = C.c_int(0), C.c_int(0), C.c_int(0)
        g = np.empty((w, h), np.float32) if want_g else None
        psr = lib().ora_estimate_trans(self._ctx, _p(last_fft), _p(cur_fft), which, _p(trans),
                                       C.addressof(row), C.addressof(col),
                                       _p(g) if want_g else None, C.addressof(err))
        if err.value:
            raise ValueError("Received invalid kernel type")
        return float(psr), trans, row.value, col.value, g

    @staticmethod
    def get_info(g, response):
        g = np.ascontiguousarray(g, np.float32)
        return float(lib().ora_get_info(_p(g), g.size, C.c_float(response)))

    def compute_pose(self, last_fft, image, last_polar, polar, not_large_rotation=True, faithful=False):
        last_fft = np.ascontiguousarray(last_fft, np.complex64)
        last_polar = np.ascontiguousarray(last_polar, np.complex64)
        polar = np.ascontiguousarray(polar, np.complex64)
        image = np.ascontiguousarray(image, np.float32)
        pose, info, dbg = np.zeros(3), np.zeros(3), OraPoseDebug()
        rc = lib().ora_compute_pose(self._ctx, _p(last_fft), _p(image), _p(last_polar), _p(polar),
                                    int(bool(not_large_rotation)), int(bool(faithful)), _p(pose), _p(info),
                                    C.addressof(dbg))
        if rc:
            raise ValueError("Received invalid kernel type")
        return pose, info, dbg.as_dict()


def track_pairs(cfg, key_imgs, cur_imgs, not_large_rotation=True, faithful=False, nthreads=1):
    """n independent (key, current) u8 pairs -> poses[n,3], infos[n,3], debug dicts, seconds of timed units."""
    key_imgs = np.ascontiguousarray(key_imgs, np.uint8)
    cur_imgs = np.ascontiguousarray(cur_imgs, np.uint8)
    n, H, W = cur_imgs.shape
    poses, infos = np.zeros((n, 3)), np.zeros((n, 3))
    dbgs = (OraPoseDebug * n)()
    secs = C.c_double(0)
    rc = lib().ora_track_pairs(C.byref(cfg), H, W, n, _p(key_imgs), _p(cur_imgs),
                               int(bool(not_large_rotation)), int(bool(faithful)), int(nthreads),
                               _p(poses), _p(infos), C.cast(dbgs, C.c_void_p), C.addressof(secs))
    if rc:
        raise ValueError("oracle track_pairs failed rc=%d" % rc)
    return poses, infos, [d.as_dict() for d in dbgs], secs.value


# ---- camera undistortion (camera.cc:45-47, 92-93) -------------------------------------------------------
def optimal_new_camera_matrix(K, D, W, H):
    """getOptimalNewCameraMatrix(K, D, (W,H), alpha=0, (W,H)); K = (fx, cx, fy, cy), D = (k1, k2, p1, p2, k3)."""
    K = np.ascontiguousarray(K, np.float64); D = np.ascontiguousarray(D, np.float64)
    out = np.empty(4, np.float64)
    lib().ora_optimal_new_camera_matrix(_p(K), _p(D), W, H, _p(out))
    return out


def undistort_maps(K, D, newK, W, H):
    """initUndistortRectifyMap(K, D, I, newK, (W,H), CV_16SC2) -> map1 int16 (H,W,2), map2 uint16 (H,W)."""
    K = np.ascontiguousarray(K, np.float64); D = np.ascontiguousarray(D, np.float64)
    newK = np.ascontiguousarray(newK, np.float64)
    m1 = np.empty((H, W, 2), np.int16); m2 = np.empty((H, W), np.uint16)
    lib().ora_undistort_maps(_p(K), _p(D), _p(newK), W, H, _p(m1), _p(m2))
    return m1, m2


def remap_u8(img, map1, map2):
    """cv::remap(img, map1, map2, INTER_LINEAR) for a u8 row-major image (Camera::UndistortImage)."""
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    out = np.empty((H, W), np.uint8)
    lib().ora_remap_u8(_p(img), W, H, _p(np.ascontiguousarray(map1, np.int16)), _p(np.ascontiguousarray(map2, np.uint16)), _p(out))
    return out


def downsample_u8(img):
    """2 x 2 box filter, rounded (pyramid level of the coarse-to-fine extension)"""
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    out = np.empty((H // 2, W // 2), np.uint8)
    lib().ora_downsample_u8(_p(img), W, H, _p(out))
    return out

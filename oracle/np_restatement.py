"""Independent numpy/scipy restatement of NI-SLAM's KCC front end.

TEST INFRASTRUCTURE ONLY.  It exists to pin oracle/kcc_oracle.c from a second, independently
written implementation (scipy's pocketfft in float32 instead of the oracle's own FFT; vectorised
index arithmetic instead of per-pixel loops) and to generate the golden fixtures under
tests/golden/ (tests/golden/make_golden.py).  PARITY UNPINNED against the real reference: the
reference has no tests/fixtures and cannot be built here (FFTW3f/Eigen3/OpenCV absent).

Follows /root/reference/src/correlation_flow.cc:37-243, src/utils.cc:110-175,
include/circ_shift.h:238-244.  Conventions as in oracle/kcc_oracle.py: an Eigen column-major
rows x cols array is a C-order numpy array of shape (cols, rows).
"""
import math

import numpy as np
import scipy.fft as sfft

INTER_BITS, INTER_TAB = 5, 32
AB_BITS, AB_SCALE = 10, 1024
CV_PI = 3.1415926535897932384626433832795


def fft(x):
    """CorrelationFlow::FFT :53-63 -- r2c over (cols, rows) with the row axis halved, unnormalised."""
    return sfft.rfft2(np.asarray(x, np.float32)).astype(np.complex64)


def ifft(xf):
    """CorrelationFlow::IFFT :65-77 -- c2r then / size."""
    xf = np.asarray(xf, np.complex64)
    cols, hr = xf.shape
    rows = (hr - 1) * 2
    # scipy's irfft2 is normalised by 1/(rows*cols) like the reference's final division
    return sfft.irfft2(xf, s=(cols, rows)).astype(np.float32)


def remove_zero(x):
    """RemoveZeroComponent :79-87 (both statements read the original x)."""
    y = x.copy()
    y[:, 0] = (x[:, 1] + x[:, -1]) / np.float32(2)      # y.block(0,0,1,cols): row 0, all cols
    y[0, :] = (x[1, :] + x[-1, :]) / np.float32(2)      # y.block(0,0,rows,1): col 0, all rows
    return y


def fftshift(x):
    """circ_shift.h:238-244: out(r,c) = in((r-rows/2)%rows, (c-cols/2)%cols)."""
    cols, rows = x.shape
    return np.roll(x, (cols // 2, rows // 2), axis=(0, 1))


def _bilinear(src, sx, sy, fx, fy, border):
    """cv::remap INTER_LINEAR on float data with 1/32-pixel fixed-point coordinates.
    src: (cols, rows); sx/sy integer source coords; fx/fy in [0,32). border: 'constant0'|'wrap'."""
    cols, rows = src.shape
    s = np.float32(1.0 / INTER_TAB)
    tx1 = fx.astype(np.float32) * s
    tx0 = np.float32(1) - tx1
    ty1 = fy.astype(np.float32) * s
    ty0 = np.float32(1) - ty1
    w0, w1, w2, w3 = ty0 * tx0, ty0 * tx1, ty1 * tx0, ty1 * tx1

    def tap(xx, yy):
        if border == "wrap":
            return src[np.mod(xx, cols), np.mod(yy, rows)]
        ok = (xx >= 0) & (xx < cols) & (yy >= 0) & (yy < rows)
        v = src[np.clip(xx, 0, cols - 1), np.clip(yy, 0, rows - 1)]
        return np.where(ok, v, np.float32(0))

    v0, v1, v2, v3 = tap(sx, sy), tap(sx + 1, sy), tap(sx, sy + 1), tap(sx + 1, sy + 1)
    return ((v0 * w0 + v1 * w1) + v2 * w2) + v3 * w3


def polar(x, PD, PC):
    """CorrelationFlow::polar :228-236 == cv::warpPolar(linear, FILL_OUTLIERS) == remap(BORDER_CONSTANT 0)."""
    W, H = x.shape
    cx, cy = np.float32(W) / np.float32(2), np.float32(H) / np.float32(2)
    max_radius = float(min(H // 2, W // 2))
    kangle = (2.0 * CV_PI) / PD
    kmag = max_radius / PC
    rhos = (np.arange(PC, dtype=np.float64) * kmag).astype(np.float32).astype(np.float64)
    ang = kangle * np.arange(PD, dtype=np.float64)
    mapx = (rhos[None, :] * np.cos(ang)[:, None] + float(cx)).astype(np.float32)   # (PD, PC)
    mapy = (rhos[None, :] * np.sin(ang)[:, None] + float(cy)).astype(np.float32)
    qx = np.rint(mapx * np.float32(INTER_TAB)).astype(np.int64)                    # cvRound: half-even
    qy = np.rint(mapy * np.float32(INTER_TAB)).astype(np.int64)
    out = _bilinear(x, qx >> INTER_BITS, qy >> INTER_BITS, qx & (INTER_TAB - 1), qy & (INTER_TAB - 1), "constant0")
    return np.ascontiguousarray(out.T.astype(np.float32))                          # (PC, PD) == col-major PD x PC


def rotate(x, degree):
    """RotateArray utils.cc:154-161 == getRotationMatrix2D + warpAffine(INTER_LINEAR, BORDER_WRAP)."""
    cols, rows = x.shape
    cx, cy = float(np.float32(cols / 2.0)), float(np.float32(rows / 2.0))
    a = float(np.float32(degree)) * (CV_PI / 180)
    alpha, beta = math.cos(a), math.sin(a)
    M = [alpha, beta, (1 - alpha) * cx - beta * cy, -beta, alpha, beta * cx + (1 - alpha) * cy]
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    M[0] = A11
    M[1] *= -D
    M[3] *= -D
    M[4] = A22
    b1 = -M[0] * M[2] - M[1] * M[5]
    b2 = -M[3] * M[2] - M[4] * M[5]
    M[2], M[5] = b1, b2
    c = np.arange(cols, dtype=np.float64)
    r = np.arange(rows, dtype=np.float64)
    adelta = np.rint(M[0] * c * AB_SCALE).astype(np.int64)
    bdelta = np.rint(M[3] * c * AB_SCALE).astype(np.int64)
    rd = AB_SCALE // INTER_TAB // 2
    X0 = np.rint((M[1] * r + M[2]) * AB_SCALE).astype(np.int64) + rd
    Y0 = np.rint((M[4] * r + M[5]) * AB_SCALE).astype(np.int64) + rd
    X = (X0[None, :] + adelta[:, None]) >> (AB_BITS - INTER_BITS)                  # (cols, rows)
    Y = (Y0[None, :] + bdelta[:, None]) >> (AB_BITS - INTER_BITS)
    out = _bilinear(x, X >> INTER_BITS, Y >> INTER_BITS, X & (INTER_TAB - 1), Y & (INTER_TAB - 1), "wrap")
    return np.ascontiguousarray(out.astype(np.float32))


def normalize_degree(a):
    return a - 360 * math.floor((a + 180) / 360)


class CorrelationFlowNp:
    def __init__(self, H, W, PD=720, PC=480, lam=0.1, kernel=0, sigma=0.2, offset=0.1, power=3):
        self.H, self.W, self.PD, self.PC = H, W, PD, PC
        self.lam, self.kernel, self.sigma, self.offset, self.power = (np.float32(lam), kernel, np.float32(sigma),
                                                                      np.float32(offset), power)
        self.target_fft = self._target(H, W)
        self.target_rotation_fft = self._target(PD, PC)

    @staticmethod
    def _target(rows, cols):
        t = np.zeros((cols, rows), np.float32)
        t[cols // 2, rows // 2] = 1
        return fft(t)

    def intermedium(self, image):
        f = fft(image)
        power = ifft(np.abs(f).astype(np.complex64))
        high = remove_zero(power)
        return f, fft(polar(fftshift(high), self.PD, self.PC))

    def _kernel(self, xf, zf, rows, cols):
        xz = ifft(xf * np.conj(zf))
        if self.kernel == 0:
            k = np.power((xz + self.offset).astype(np.float64), float(self.power)).astype(np.float32)
        elif self.kernel == 1:
            N = np.float32(rows * cols)
            xx = np.float32(np.sum(np.abs(xf * xf), dtype=np.float64)) / N     # half spectrum only
            zz = np.float32(np.sum(np.abs(zf * zf), dtype=np.float64)) / N
            xxzz = (xx + zz - np.float32(2) * xz) / N
            k = np.exp((np.float32(-1) / (self.sigma * self.sigma)) * xxzz).astype(np.float32)
        else:
            raise ValueError("Received invalid kernel type")
        k = k / np.max(np.abs(k))
        return fft(k)

    def estimate_trans(self, last_fft, cur_fft, which):
        rows, cols = (self.PD, self.PC) if which else (self.H, self.W)
        T = self.target_rotation_fft if which else self.target_fft
        Kzz = self._kernel(last_fft, last_fft, rows, cols)
        Kxz = self._kernel(cur_fft, last_fft, rows, cols)
        G = (T / (Kzz + self.lam) * Kxz).astype(np.complex64)
        g = ifft(G)
        flat = int(np.argmax(g.reshape(-1)))          # first max in column-major (memory) order
        col, row = divmod(flat, rows)
        response = g.reshape(-1)[flat]
        n = g.size
        m = (np.sum(g, dtype=np.float64) - response) / (n - 1)
        std = math.sqrt(np.mean((g.astype(np.float64) - m) ** 2))
        psr = (response - m) / (std + 1e-7)
        return float(psr), (-(row - rows // 2), -(col - cols // 2)), row, col, g

    def compute_pose(self, last_fft, image, last_polar, polar_, not_large_rotation=True):
        psr_r, rots, rr, rc, _ = self.estimate_trans(last_polar, polar_, 1)
        degree = np.float32(rots[0] * (2.0 / self.PD) * 180)
        degree = np.float32(normalize_degree(float(degree)))
        dbg = dict(rot_row=rr, rot_col=rc, psr_rot=psr_r)
        if not_large_rotation:
            degree = np.float32(degree - 180) if abs(degree) > 90 else degree
            psr_t, trans, tr, tc, _ = self.estimate_trans(last_fft, fft(rotate(image, -degree)), 0)
            dbg.update(trans_row=[tr, 0], trans_col=[tc, 0], psr_trans=[psr_t, 0.0], chosen=0, n_hyp=1)
        else:
            p0, t0, r0, c0, _ = self.estimate_trans(last_fft, fft(rotate(image, -degree)), 0)
            p1, t1, r1, c1, _ = self.estimate_trans(last_fft, fft(rotate(image, np.float32(-degree + 180))), 0)
            if p0 > p1:
                psr_t, trans, ch = p0, t0, 0
            else:
                psr_t, trans, ch = p1, t1, 1
                degree = np.float32(degree + 180)
            dbg.update(trans_row=[r0, r1], trans_col=[c0, c1], psr_trans=[p0, p1], chosen=ch, n_hyp=2)
        if degree > 180:
            degree = np.float32(degree - 360)
        theta = np.float32(float(np.float32(degree) / np.float32(180)) * math.pi)
        dbg["degree_final"] = float(degree)
        pose = np.array([trans[1], trans[0], float(theta)], np.float64)
        info = np.array([psr_t, psr_t, psr_r], np.float64)
        return pose, info, dbg


# ---- camera undistortion (camera.cc:45-47, 92-93), vectorised restatement ------------------------------
def _undistort_points(u, v, K, D, iters=5):
    fx, cx, fy, cy = K
    k1, k2, p1, p2, k3 = D
    x = (u - cx) * (1.0 / fx); y = (v - cy) * (1.0 / fy)
    x0, y0 = x.copy(), y.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        icdist = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x = (x0 - dx) * icdist; y = (y0 - dy) * icdist
    return x, y


def optimal_new_camera_matrix(K, D, W, H):
    N = 9
    gx = (np.arange(N, dtype=np.float32) * np.float32(W) / np.float32(N - 1)).astype(np.float32)
    gy = (np.arange(N, dtype=np.float32) * np.float32(H) / np.float32(N - 1)).astype(np.float32)
    px, py = np.meshgrid(gx, gy)                      # [y, x]
    ux, uy = _undistort_points(px.astype(np.float64), py.astype(np.float64), K, D)
    ux = ux.astype(np.float32); uy = uy.astype(np.float32)
    iX0 = ux[:, 0].max(); iX1 = ux[:, N - 1].min(); iY0 = uy[0, :].max(); iY1 = uy[N - 1, :].min()
    # (int) / (float) is a float division in the reference's C++ (cv::Rect_<float>), widened to double afterwards
    fx0 = float(np.float32(W - 1) / np.float32(iX1 - iX0)); fy0 = float(np.float32(H - 1) / np.float32(iY1 - iY0))
    return np.array([fx0, -fx0 * float(iX0), fy0, -fy0 * float(iY0)])


def undistort_maps(K, D, newK, W, H):
    fx, u0, fy, v0 = K
    k1, k2, p1, p2, k3 = D
    ir0, ir2 = 1.0 / newK[0], -newK[1] / newK[0]
    ir4, ir5 = 1.0 / newK[2], -newK[3] / newK[2]
    # running sums along a row (the reference accumulates _x += ir[0] per column)
    xs = np.empty(W); acc = ir2
    for j in range(W):
        xs[j] = acc; acc += ir0
    ys = np.arange(H) * ir4 + ir5
    x = np.broadcast_to(xs[None, :], (H, W)); y = np.broadcast_to(ys[:, None], (H, W))
    x2, y2 = x * x, y * y
    r2 = x2 + y2; _2xy = 2 * x * y
    kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)
    yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy
    iu = np.rint(( fx * xd + u0) * INTER_TAB).astype(np.int64)
    iv = np.rint(( fy * yd + v0) * INTER_TAB).astype(np.int64)
    m1 = np.stack([iu >> INTER_BITS, iv >> INTER_BITS], axis=-1).astype(np.int16)
    m2 = ((iv & (INTER_TAB - 1)) * INTER_TAB + (iu & (INTER_TAB - 1))).astype(np.uint16)
    return m1, m2


def remap_u8(img, map1, map2):
    img = np.asarray(img, np.uint8); H, W = img.shape
    sx = map1[..., 0].astype(np.int64); sy = map1[..., 1].astype(np.int64)
    fx = (map2 & 31).astype(np.int64); fy = (map2 >> 5).astype(np.int64) & 31
    pad = np.zeros((H + 2, W + 2), np.int64); pad[1:-1, 1:-1] = img       # border value 0 one pixel around
    def tap(xx, yy):
        ok = (xx >= -1) & (xx <= W) & (yy >= -1) & (yy <= H)
        return np.where(ok, pad[np.clip(yy, -1, H) + 1, np.clip(xx, -1, W) + 1], 0)
    acc = ((32 - fx) * (32 - fy) * 32 * tap(sx, sy) + fx * (32 - fy) * 32 * tap(sx + 1, sy)
           + (32 - fx) * fy * 32 * tap(sx, sy + 1) + fx * fy * 32 * tap(sx + 1, sy + 1))
    return ((acc + (1 << 14)) >> 15).astype(np.uint8)

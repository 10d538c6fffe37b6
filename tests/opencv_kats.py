"""Known answers for the OpenCV calls on the path that follow from OpenCV's DOCUMENTED geometry alone -- no OpenCV build, no
recollection of its rounding rules (oracle/RECALLED.md rows 1-6 say what is recalled; these KATs shrink what that covers).

* cv::warpAffine(getRotationMatrix2D((W/2, H/2), angle, 1), INTER_LINEAR, BORDER_WRAP) for angle in {0, 90, 180, 270} degrees:
  dst(x, y) = src(M^-1 (x, y)) with M = [[a, b, (1-a) cx - b cy], [-b, a, b cx + (1-a) cy]], a = cos, b = sin (the documented
  matrix; positive angle = counter-clockwise, origin top-left).  For right angles every source coordinate is an integer, so the
  result is a pure PERMUTATION of the source pixels whatever the fixed-point format, the rounding offset or the tap order are.
* cv::warpPolar(linear, dsize = (PC cols, PD rows), centre (W/2, H/2), maxRadius, INTER_LINEAR | WARP_FILL_OUTLIERS) on the four
  axes (angle rows 0, PD/4, PD/2, 3 PD/4): dst(rho_j, phi_i) = src(cx + r_j cos(phi_i), cy + r_j sin(phi_i)), r_j = j * maxRadius / PC.
  When maxRadius / PC is a multiple of 1/32 px (0.5 at 640x480 / 480, 0.375 at 60x80 / 80) the sample positions are exact
  1/32-pixel positions on one pixel row / column: the value is the two-tap linear blend along the axis, taps outside the image
  count 0 (BORDER_CONSTANT).  With INTEGER-valued pixels every product and sum is exact in float32, so the answer does not depend
  on the order of the blend's operations either.
Arrays are the reference's ArrayXXf seen from numpy: shape (W, H), x[c, r]."""
import numpy as np


def rotate_right_angle(x, quarters):
    """RotateArray(x, 90 * quarters degrees) (utils.cc:154-161) as a permutation"""
    W, H = x.shape
    cx, cy = W // 2, H // 2
    c, r = np.meshgrid(np.arange(W), np.arange(H), indexing="ij")          # destination (x_d = c, y_d = r)
    q = quarters % 4
    if q == 0:
        xs, ys = c, r
    elif q == 1:                                                           # x_d = y_s + cx - cy, y_d = -x_s + cx + cy
        xs, ys = cx + cy - r, c - cx + cy
    elif q == 2:
        xs, ys = 2 * cx - c, 2 * cy - r
    else:                                                                  # x_d = -y_s + cx + cy, y_d = x_s - cx + cy
        xs, ys = r + cx - cy, cx + cy - c
    return x[xs % W, ys % H]


def polar_axes(x, PD, PC):
    """{angle row i: the PC samples of warpPolar's row i} for i in (0, PD/4, PD/2, 3PD/4); x integer-valued float32"""
    W, H = x.shape
    cx, cy = W // 2, H // 2
    rmax = min(H // 2, W // 2)
    step32 = rmax * 32 / PC
    assert PD % 4 == 0 and step32 == int(step32), "the KAT needs radii on the 1/32-pixel grid"
    step32 = int(step32)

    def tap(col, row):
        return np.float32(x[col, row]) if (0 <= col < W and 0 <= row < H) else np.float32(0)
    out = {}
    for i, (dx, dy) in {0: (1, 0), PD // 4: (0, 1), PD // 2: (-1, 0), 3 * PD // 4: (0, -1)}.items():
        vals = np.zeros(PC, np.float32)
        for j in range(PC):
            ix, iy = 32 * cx + dx * step32 * j, 32 * cy + dy * step32 * j      # 1/32-pixel coordinates
            sx, fx, sy, fy = ix >> 5, ix & 31, iy >> 5, iy & 31
            t = np.float32((fx if dx else fy) / 32.0)
            a = tap(sx, sy)
            b = tap(sx + 1, sy) if dx else tap(sx, sy + 1)
            vals[j] = np.float32(a * (np.float32(1) - t)) + np.float32(b * t)   # exact for integer-valued pixels
        out[i] = vals
    return out


def remove_zero_fftshift(p):
    """fftshift(RemoveZeroComponent(p)) in numpy (correlation_flow.cc:79-87, circ_shift.h:238-244): the reference's own
    arithmetic, restated independently of the oracle"""
    W, H = p.shape
    y = p.copy()
    y[:, 0] = (p[:, 1] + p[:, H - 1]) / np.float32(2)          # row 0 of every column
    y[0, :] = (p[1, :] + p[W - 1, :]) / np.float32(2)          # column 0 (reads the ORIGINAL p: y(0,0) = (p(0,1) + p(0,W-1)) / 2)
    return np.roll(y, (W // 2, H // 2), axis=(0, 1))

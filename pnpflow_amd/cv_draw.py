"""Restatement of OpenCV's `cv2.line` for thick 8-connected lines, which is what the reference's paintbrush masks are drawn with
(`MaskGenerator._generate_mask`, pnpflow/utils.py:904-924: `cv2.line(img, (x1, y1), (x2, y2), (255, 255, 255), thickness)` on a
one-channel uint8 image, thickness 8 ... int(0.08 (W + H))).

OpenCV is a dependency of the reference (`opencv-python`, unpinned in its requirements.txt) that is not installed here, so its
published algorithm is restated (opencv/modules/imgproc/src/drawing.cpp, unchanged in this part across the 4.x series):

    line()            -> ThickLine(img, p0, p1, color, thickness, LINE_8, flags = 3, shift = 0)
    ThickLine()          endpoints to 16.16 fixed point; the stroke body is the quadrilateral p0 +- dp, p1 -+ dp with
                         dp = (round(dy r), round(dx r)), r = (thickness * 2^15 + odd * 2^15) / |p1 - p0| (cvRound = round-half-even),
                         filled by FillConvexPoly(..., LINE_8, shift = 16); both ends get Circle(center, (thickness * 2^15 + 2^15) >> 16, fill)
    FillConvexPoly()     draws every edge with Line2 (fixed-point DDA), then scan-converts between a left and a right edge walker:
                         x += dx per row, dx = ((xe - xs) * 2 + (ty - y)) / (2 (ty - y)) (C division: toward zero),
                         row span [(x_left + 2^15) >> 16, (x_right + 2^15) >> 16]
    Line2()              clipLine on the 16.16 rectangle, then one pixel per step along the major axis
    Circle()             midpoint circle, filled with horizontal runs

PARITY UNPINNED: without cv2 in the image the restatement cannot be checked against the library itself; tests hold the properties
OpenCV documents / is known for (an even thickness t covers t + 1 rows, end caps, symmetry) and the product / oracle restatements
(written separately: incremental edge walkers here, closed-form rows in oracle/pnpflow_oracle.py) against each other.
"""
from __future__ import annotations

import numpy as np

XY_SHIFT = 16
XY_ONE = 1 << XY_SHIFT


def _tdiv(a: int, b: int) -> int:
    """C integer division (truncation toward zero)."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def _hline(img, y, x1, x2):
    if x2 >= x1:
        img[y, x1:x2 + 1] = 255


def clip_line(width: int, height: int, p1, p2):
    """cv::clipLine(Size2l, Point2l&, Point2l&) -> (inside, p1, p2)"""
    x1, y1 = p1; x2, y2 = p2
    right, bottom = width - 1, height - 1
    if width <= 0 or height <= 0:
        return False, p1, p2
    c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8
    c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8
    if (c1 & c2) == 0 and (c1 | c2) != 0:
        if c1 & 12:
            a = 0 if c1 < 8 else bottom
            x1 += int(float(a - y1) * (x2 - x1) / (y2 - y1))
            y1 = a
            c1 = (x1 < 0) + (x1 > right) * 2
        if c2 & 12:
            a = 0 if c2 < 8 else bottom
            x2 += int(float(a - y2) * (x2 - x1) / (y2 - y1))
            y2 = a
            c2 = (x2 < 0) + (x2 > right) * 2
        if (c1 & c2) == 0 and (c1 | c2) != 0:
            if c1:
                a = 0 if c1 == 1 else right
                y1 += int(float(a - x1) * (y2 - y1) / (x2 - x1))
                x1 = a
                c1 = 0
            if c2:
                a = 0 if c2 == 1 else right
                y2 += int(float(a - x2) * (y2 - y1) / (x2 - x1))
                x2 = a
                c2 = 0
    return (c1 | c2) == 0, (x1, y1), (x2, y2)


def line2(img, pt1, pt2):
    """Line2: 8-connected line between two 16.16 fixed-point points (one-channel image)."""
    H, W = img.shape
    ok, pt1, pt2 = clip_line(W << XY_SHIFT, H << XY_SHIFT, pt1, pt2)
    if not ok:
        return
    x1, y1 = pt1; x2, y2 = pt2
    dx, dy = x2 - x1, y2 - y1
    j = -1 if dx < 0 else 0
    ax = (dx ^ j) - j
    i = -1 if dy < 0 else 0
    ay = (dy ^ i) - i

    def put(x, y):
        if 0 <= x < W and 0 <= y < H:
            img[y, x] = 255

    if ax > ay:
        dy = (dy ^ j) - j
        if j:                                   # the XOR swaps: the line is walked left to right
            x1, x2, y1, y2 = x2, x1, y2, y1
        x_step, y_step = XY_ONE, _tdiv(dy << XY_SHIFT, ax | 1)
        ecount = (x2 - x1) >> XY_SHIFT
    else:
        dx = (dx ^ i) - i
        if i:
            x1, x2, y1, y2 = x2, x1, y2, y1
        x_step, y_step = _tdiv(dx << XY_SHIFT, ay | 1), XY_ONE
        ecount = (y2 - y1) >> XY_SHIFT
    x1 += XY_ONE >> 1
    y1 += XY_ONE >> 1
    put((x2 + (XY_ONE >> 1)) >> XY_SHIFT, (y2 + (XY_ONE >> 1)) >> XY_SHIFT)
    if ax > ay:
        x1 >>= XY_SHIFT
        while ecount >= 0:
            put(x1, y1 >> XY_SHIFT)
            x1 += 1; y1 += y_step; ecount -= 1
    else:
        y1 >>= XY_SHIFT
        while ecount >= 0:
            put(x1 >> XY_SHIFT, y1)
            x1 += x_step; y1 += 1; ecount -= 1


def fill_convex_poly(img, v, shift=XY_SHIFT):
    """FillConvexPoly(img, v, npts, color, LINE_8, shift) for vertices already in `shift` fixed point."""
    H, W = img.shape
    npts = len(v)
    delta = (1 << shift) >> 1
    delta1 = delta2 = XY_ONE >> 1                       # line_type < LINE_AA
    p0 = (v[-1][0] << (XY_SHIFT - shift), v[-1][1] << (XY_SHIFT - shift))
    xmin = xmax = v[0][0]; ymin = ymax = v[0][1]
    imin = 0
    for idx, (px, py) in enumerate(v):
        if py < ymin:
            ymin, imin = py, idx
        ymax = max(ymax, py); xmax = max(xmax, px); xmin = min(xmin, px)
        p = (px << (XY_SHIFT - shift), py << (XY_SHIFT - shift))
        line2(img, p0, p)                               # shift != 0 -> Line2 (the outline)
        p0 = p
    xmin = (xmin + delta) >> shift; xmax = (xmax + delta) >> shift
    ymin = (ymin + delta) >> shift; ymax = (ymax + delta) >> shift
    if npts < 3 or xmax < 0 or ymax < 0 or xmin >= W or ymin >= H:
        return
    ymax = min(ymax, H - 1)
    edges = npts
    e_idx = [imin, imin]; e_di = [1, npts - 1]; e_ye = [ymin, ymin]
    e_x = [-XY_ONE, -XY_ONE]; e_dx = [0, 0]
    y = ymin
    while True:
        for i in range(2):                              # line_type < LINE_AA: always
            if y >= e_ye[i]:
                idx0, di = e_idx[i], e_di[i]
                idx = idx0 + di
                if idx >= npts:
                    idx -= npts
                while True:
                    edges -= 1
                    if edges < 0:                       # `for (; edges-- > 0; )`: the test fails, edges ends below zero
                        break
                    ty = (v[idx][1] + delta) >> shift
                    if ty > y:
                        xs, xe = v[idx0][0], v[idx][0]
                        if shift != XY_SHIFT:
                            xs <<= XY_SHIFT - shift; xe <<= XY_SHIFT - shift
                        e_ye[i] = ty
                        e_dx[i] = _tdiv((xe - xs) * 2 + (ty - y), 2 * (ty - y))
                        e_x[i] = xs
                        e_idx[i] = idx
                        break
                    idx0 = idx
                    idx += di
                    if idx >= npts:
                        idx -= npts
        if edges < 0:
            break
        if y >= 0:
            left, right = (1, 0) if e_x[0] > e_x[1] else (0, 1)
            xx1 = (e_x[left] + delta1) >> XY_SHIFT
            xx2 = (e_x[right] + delta2) >> XY_SHIFT
            if xx2 >= 0 and xx1 < W:
                _hline(img, y, max(xx1, 0), min(xx2, W - 1))
        e_x[0] += e_dx[0]; e_x[1] += e_dx[1]
        y += 1
        if y > ymax:
            break


def circle_filled(img, cx, cy, radius):
    """Circle(img, center, radius, color, fill = 1)"""
    H, W = img.shape
    err, dx, dy, plus, minus = 0, radius, 0, 1, (radius << 1) - 1
    while dx >= dy:
        y11, y12, y21, y22 = cy - dy, cy + dy, cy - dx, cy + dx
        x11, x12, x21, x22 = cx - dx, cx + dx, cx - dy, cx + dy
        if x11 < W and x12 >= 0 and y21 < H and y22 >= 0:
            a, b = max(x11, 0), min(x12, W - 1)
            if 0 <= y11 < H:
                _hline(img, y11, a, b)
            if 0 <= y12 < H:
                _hline(img, y12, a, b)
            if x21 < W and x22 >= 0:
                a, b = max(x21, 0), min(x22, W - 1)
                if 0 <= y21 < H:
                    _hline(img, y21, a, b)
                if 0 <= y22 < H:
                    _hline(img, y22, a, b)
        dy += 1
        err += plus
        plus += 2
        mask = 0 if err <= 0 else -1
        err -= minus & mask
        dx += mask
        minus -= mask & 2


def thick_line(img, p0, p1, thickness):
    """cv2.line(img, p0, p1, 255, thickness) for thickness > 1 on a (H, W) uint8 array (LINE_8, shift 0), in place."""
    x0, y0 = int(p0[0]) << XY_SHIFT, int(p0[1]) << XY_SHIFT
    x1, y1 = int(p1[0]) << XY_SHIFT, int(p1[1]) << XY_SHIFT
    inv = 1.0 / XY_ONE
    dx, dy = (x0 - x1) * inv, (y1 - y0) * inv
    r = dx * dx + dy * dy
    odd = thickness & 1
    th = thickness << (XY_SHIFT - 1)
    if abs(r) > 2.220446049250313e-16:
        r = (th + odd * XY_ONE * 0.5) / np.sqrt(r)
        dpx, dpy = int(round(dy * r)), int(round(dx * r))            # cvRound: round half to even, as Python's round
        pts = [(x0 + dpx, y0 + dpy), (x0 - dpx, y0 - dpy), (x1 - dpx, y1 - dpy), (x1 + dpx, y1 + dpy)]
        fill_convex_poly(img, pts, XY_SHIFT)
    for (px, py) in ((x0, y0), (x1, y1)):
        circle_filled(img, (px + (XY_ONE >> 1)) >> XY_SHIFT, (py + (XY_ONE >> 1)) >> XY_SHIFT, (th + (XY_ONE >> 1)) >> XY_SHIFT)
    return img

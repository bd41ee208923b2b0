"""Sequential restatement of Feature_detector::detect's bookkeeping (reference
src/Feature_detection.cpp:110-153, Frame::Set_Mask src/Frame.cpp:286-298) for tests: plain loops over
the per-cell corners, its own disc painting (distance test instead of the product's span filler is
NOT equivalent to cv::circle, so the midpoint spans are restated here too, independently)."""
import numpy as np


def circle_spans(radius):
    """Rows of cv::circle(…, -1) (OpenCV 2.4 drawing.cpp Circle(), filled): dy -> half-width."""
    spans = {}
    err, dx, dy, plus, minus = 0, radius, 0, 1, (radius << 1) - 1
    while dx >= dy:
        for yy, hw in ((dy, dx), (dx, dy)):
            spans[yy] = max(spans.get(yy, -1), hw)
        dy += 1
        err += plus
        plus += 2
        mask = -1 if err > 0 else 0
        err -= minus & mask
        dx += mask
        minus -= mask & 2
    return spans


def paint(mask, cx, cy, radius):
    h, w = mask.shape
    for dy, hw in circle_spans(radius).items():
        for y in (cy - dy, cy + dy):
            if 0 <= y < h:
                x1, x2 = max(cx - hw, 0), min(cx + hw, w - 1)
                if x1 <= x2:
                    mask[y, x1:x2 + 1] = 0


def detect(cells, width, height, cell_size, max_fts, existing_px, existing_has_point, min_dist):
    """Returns the list of (x, y, level) the reference would add, given the per-cell corners."""
    score, cx, cy, cl = cells
    n = len(existing_px)
    if n >= max_fts:
        return []
    order = sorted(range(len(score)), key=lambda k: -float(score[k]))      # stable
    mask = np.full((height, width), 255, np.uint8)
    if n > 0:
        for k in range(n):
            if existing_has_point[k]:
                paint(mask, int(round(float(existing_px[k][0]))), int(round(float(existing_px[k][1]))), min_dist)
    out = []
    for k in order:
        if float(score[k]) > 20:
            x, y = int(cx[k]), int(cy[k])
            if mask[y, x] == 255:
                out.append((x, y, int(cl[k])))
                paint(mask, x, y, cell_size)
                n += 1
        if n >= max_fts:
            break
    return out

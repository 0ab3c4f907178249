#!/usr/bin/env python3
"""Fixture for drawBoxesOnRGB / drawBoxesOnGray (reference MTM/__init__.py:299-391), derived from the DEFINITIONS of the
OpenCV calls the reference makes - cv2 itself cannot be installed here - by code that shares nothing with the product:

* cv2.rectangle(img, (x, y), (x + w, y + h), color, thickness=1): the closed polyline through the four corners, both end
  points included (imgproc/drawing.cpp: rectangle -> PolyLine -> Line, 8-connected Bresenham; an axis-aligned segment is
  the run of pixels between its end points), clipped to the image.  Only thickness 1 is pinned: thicker lines are filled
  polygons with round caps (ThickLine) whose rasterisation is not restated here.
* cv2.cvtColor(img, COLOR_GRAY2RGB): the gray value in all three channels; COLOR_RGB2GRAY for 8-bit pixels:
  (9798 R + 19235 G + 3735 B + 2^14) >> 15 (imgproc/color_rgb: the 15-bit fixed-point weights of 0.299 / 0.587 / 0.114),
  16-bit pixels: (4899 R + 9617 G + 1868 B + 2^13) >> 14.
Writes tests/golden/draw_fixture.json.  Run from the repo root."""
import json
import os

H, W = 24, 32


def gray_image():
    return [[(7 * r + 13 * c + (r * c) % 5) % 256 for c in range(W)] for r in range(H)]


def rgb_image():
    g = gray_image()
    return [[[g[r][c], (g[r][c] * 3 + r) % 256, (255 - g[r][c] + c) % 256] for c in range(W)] for r in range(H)]


def outline(x, y, w, h):
    px = set()
    for xx in range(x, x + w + 1):
        px.add((y, xx))
        px.add((y + h, xx))
    for yy in range(y, y + h + 1):
        px.add((yy, x))
        px.add((yy, x + w))
    return sorted((r, c) for r, c in px if 0 <= r < H and 0 <= c < W)


def rgb2gray8(p):
    return (9798 * p[0] + 19235 * p[1] + 3735 * p[2] + (1 << 14)) >> 15


hits = [["a", [3, 2, 10, 6], 0.9], ["b", [20, 15, 11, 8], 0.8], ["c", [0, 0, 31, 23], 0.7]]     # "b" runs off the canvas: clipped
boxes = [h[1] for h in hits]
g, rgb = gray_image(), rgb_image()
on_rgb_from_gray = [[[v, v, v] for v in row] for row in g]
on_rgb = [[list(p) for p in row] for row in rgb]
on_gray_from_rgb = [[rgb2gray8(p) for p in row] for row in rgb]
on_gray = [list(row) for row in g]
for (x, y, w, h) in boxes:
    for r, c in outline(x, y, w, h):
        on_rgb_from_gray[r][c] = [255, 255, 0]          # the reference's default boxColor
        on_rgb[r][c] = [10, 20, 30]
        on_gray_from_rgb[r][c] = 255
        on_gray[r][c] = 99
samples16 = [[65535, 0, 0], [0, 65535, 0], [0, 0, 65535], [65535, 65535, 65535], [1234, 40000, 777]]
fixture = {
    "image_gray": g, "image_rgb": rgb, "hits": hits,
    "drawBoxesOnRGB(gray, thickness=1)": on_rgb_from_gray,
    "drawBoxesOnRGB(rgb, thickness=1, boxColor=(10,20,30))": on_rgb,
    "drawBoxesOnGray(rgb, thickness=1)": on_gray_from_rgb,
    "drawBoxesOnGray(gray, thickness=1, boxColor=99)": on_gray,
    "rgb2gray_uint16": {"pixels": samples16, "gray": [(4899 * p[0] + 9617 * p[1] + 1868 * p[2] + (1 << 13)) >> 14 for p in samples16]},
}
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "draw_fixture.json")
with open(out, "w") as f:
    json.dump(fixture, f, separators=(",", ":"))
print("wrote", out, os.path.getsize(out), "bytes")

"""Builds tests/golden/screenshots.npz + screenshots.json from the reference's README screenshots.

Runs in the build container only (it reads /root/reference/docs~/N.jpg, README.md:23-40).  The seven
images are editor captures: the preview plane (the reference's OUTPUT for the parameters shown) on the
left, the inspector with every parameter on the right.  This script holds no reference text; what it
writes is data: pixel crops of the preview plane and the numbers read off each inspector.

What it does, with no knowledge of the oracle (registration against a model is the TEST's nuisance fit,
done identically for the negative controls):
  1. find the preview square by the camera-background colour that frames it (102, 97, 91): the longest
     run of rows / columns of the game view that are not that colour -> a 1097 x 1097 square in every image;
  2. drop the first row / column (blended with the frame by the JPEG) -> 1096 x 1096 = 1000 cells at
     0.9124 cells per pixel;
  3. 2 x 2 box average -> 548 x 548 (1.8248 cells per pixel), rounded to uint8;
  4. keep the channels that carry a signal: grey captures (0, 1, 3, 4) -> the channel mean; the
     blue-tinted flow-map captures (2, 5, 6) -> R (the height underlay) and B (the flow-map overlay).

Usage:  python tests/golden/make_screenshot_fixtures.py
"""
import json
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
DOCS = "/root/reference/docs~"
FRAME = np.array([102.0, 97.0, 91.0], np.float32)   # camera background around the preview plane
GAME_VIEW_COLS = 1138                               # the inspector panel starts right of this column
SIDE = 1096                                         # pixels kept per side (even; 1097 found, 1 dropped)
HALF = SIDE // 2

# Read off the inspector of each capture (FBM Source / Kernel Filter / Erosion Filter / Flow Map Filter).
# "clicked" = the component whose Enabled box carries the focus highlight = the last one applied;
# "readme" = the caption /root/reference/README.md:23-40 gives the picture.
INSPECTOR = {
    "0": dict(readme="Cellular basis", noiseType="Cellular", resolution=1000, hurst=1.0, octaves=13, xpos=0, zpos=0,
              noiseSize=1757, filter="Gauss 5", filterIterations=17, erosionIterations=5, flowIterations=5, clicked=None),
    "1": dict(readme="Gauss 5 (17 iterations)", noiseType="Cellular", resolution=1000, hurst=1.0, octaves=13, xpos=0, zpos=0,
              noiseSize=1757, filter="Gauss 5", filterIterations=17, erosionIterations=5, flowIterations=5,
              clicked="Kernel Filter"),
    "2": dict(readme="Flow Map", noiseType="Cellular", resolution=1000, hurst=1.0, octaves=13, xpos=0, zpos=0,
              noiseSize=1757, filter="Gauss 5", filterIterations=17, erosionIterations=5, flowIterations=5,
              clicked="Flow Map Filter"),
    "3": dict(readme="Simplex basis", noiseType="Simplex", resolution=1000, hurst=0.422, octaves=13, xpos=0, zpos=424,
              noiseSize=1757, filter="Gauss 5", filterIterations=17, erosionIterations=5, flowIterations=5, clicked=None),
    "4": dict(readme="Gauss 5 (17 iterations)", noiseType="Simplex", resolution=1000, hurst=0.422, octaves=13, xpos=0,
              zpos=424, noiseSize=1757, filter="Gauss 5", filterIterations=17, erosionIterations=5, flowIterations=5,
              clicked="Kernel Filter"),
    "5": dict(readme="Flow Map", noiseType="Simplex", resolution=1000, hurst=0.422, octaves=13, xpos=0, zpos=424,
              noiseSize=1757, filter="Gauss 5", filterIterations=17, erosionIterations=5, flowIterations=5,
              clicked="Flow Map Filter"),
    "6": dict(readme="Value Erosion", noiseType="Simplex", resolution=1000, hurst=0.422, octaves=13, xpos=0, zpos=424,
              noiseSize=1757, filter="Gauss 5", filterIterations=18, erosionIterations=5, flowIterations=5,
              clicked="Flow Map Filter"),
}
CHANNELS = {"0": "L", "1": "L", "3": "L", "4": "L", "2": "RB", "5": "RB", "6": "RB"}


def longest_run(ix):
    best, start, prev = (0, -1), ix[0], ix[0]
    for v in list(ix[1:]) + [None]:
        if v is None or v != prev + 1:
            if prev - start > best[1] - best[0]:
                best = (int(start), int(prev))
            start = v
        prev = v
    return best


def preview_square(rgb):
    """(x0, y0, side) of the preview plane inside the game view."""
    is_frame = np.abs(rgb - FRAME).max(axis=2) < 10
    col_frac = is_frame[40:-20, :GAME_VIEW_COLS].mean(axis=0)
    row_frac = is_frame[:, 5:GAME_VIEW_COLS - 8].mean(axis=1)
    c0, c1 = longest_run(np.where(col_frac < 0.5)[0])
    r0, r1 = longest_run(np.where(row_frac < 0.5)[0])
    return c0, r0, c1 - c0 + 1, r1 - r0 + 1


def main():
    planes, meta = {}, {}
    for name in sorted(INSPECTOR):
        rgb = np.asarray(Image.open(os.path.join(DOCS, name + ".jpg")).convert("RGB")).astype(np.float32)
        x0, y0, w, h = preview_square(rgb)
        assert w == 1097 and h == 1097, (name, w, h)
        crop = rgb[y0 + 1:y0 + 1 + SIDE, x0 + 1:x0 + 1 + SIDE]
        half = crop.reshape(HALF, 2, HALF, 2, 3).mean(axis=(1, 3))
        if CHANNELS[name] == "L":
            planes["L" + name] = np.rint(half.mean(axis=2)).astype(np.uint8)
        else:
            planes["R" + name] = np.rint(half[..., 0]).astype(np.uint8)
            planes["B" + name] = np.rint(half[..., 2]).astype(np.uint8)
        meta[name] = dict(INSPECTOR[name], channels=CHANNELS[name], crop_xy=[x0 + 1, y0 + 1], crop_side=SIDE,
                          pixels=HALF, cells_per_pixel=INSPECTOR[name]["resolution"] / HALF)
        print(name, "preview square at", (x0, y0), "side", w, "->", [k for k in planes if k.endswith(name)])
    np.savez_compressed(os.path.join(HERE, "screenshots.npz"), **planes)
    with open(os.path.join(HERE, "screenshots.json"), "w") as f:
        json.dump(dict(source="/root/reference/docs~/N.jpg via README.md:23-40", images=meta), f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote screenshots.npz (%d bytes)" % os.path.getsize(os.path.join(HERE, "screenshots.npz")))


if __name__ == "__main__":
    main()

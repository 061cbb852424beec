"""Writes tests/golden/image_descriptors.npz: 256-bit binary descriptors of the two images that ship with the reference
(/root/reference/1.png, 2.png — the desk scene its main.py demo matches), so that the parity tests also see REAL
descriptor statistics (correlated bits, near-duplicates on repeated texture, many tied distances) and not only uniform
random bytes.

cv2 (ORB) is not installable here, so the descriptors are made by this script's own small ORB-like extractor — Harris
corners with non-maximum suppression, intensity-centroid orientation, a fixed seeded pattern of 256 rotated pixel-pair
comparisons on the smoothed image — written in numpy / scipy.  They are NOT OpenCV's ORB descriptors and pin nothing about
cv2; they are inputs.  The fixture holds data only (descriptor bytes and keypoint pixels); the reference's images are
read, not copied.  Run once in the build container (the GPU box has no /root/reference):  python tests/golden/make_image_descriptors.py
"""
import os

import numpy as np
from PIL import Image
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
N_FEATURES = 600          # the reference asks ORB for 200 (slam.py:23); more rows exercise more of the tie rules
PATCH = 15                # pattern radius


def gray(path):
    return np.asarray(Image.open(path).convert("L"), np.float64)


def harris_keypoints(img, n):
    sm = ndimage.gaussian_filter(img, 1.0)
    gy, gx = np.gradient(sm)
    a, b, c = (ndimage.gaussian_filter(v, 2.0) for v in (gx * gx, gx * gy, gy * gy))
    resp = a * c - b * b - 0.04 * (a + c) ** 2
    peak = (resp == ndimage.maximum_filter(resp, size=7)) & (resp > 0)
    m = PATCH + 8
    peak[:m] = peak[-m:] = False
    peak[:, :m] = peak[:, -m:] = False
    ys, xs = np.nonzero(peak)
    order = np.argsort(-resp[ys, xs], kind="stable")[:n]
    return ys[order], xs[order]


def describe(img, ys, xs, pattern):
    sm = ndimage.gaussian_filter(img, 2.0)
    r = np.arange(-PATCH, PATCH + 1)
    yy, xx = np.meshgrid(r, r, indexing="ij")
    disk = (yy * yy + xx * xx) <= PATCH * PATCH
    out = np.zeros((len(ys), 32), np.uint8)
    for i, (y, x) in enumerate(zip(ys, xs)):
        patch = img[y - PATCH:y + PATCH + 1, x - PATCH:x + PATCH + 1] * disk
        ang = np.arctan2((patch * yy).sum(), (patch * xx).sum())          # intensity centroid (ORB's orientation)
        ca, sa = np.cos(ang), np.sin(ang)
        p = pattern.astype(np.float64)
        ax = np.rint(ca * p[:, 0] - sa * p[:, 1]).astype(int) + x
        ay = np.rint(sa * p[:, 0] + ca * p[:, 1]).astype(int) + y
        bx = np.rint(ca * p[:, 2] - sa * p[:, 3]).astype(int) + x
        by = np.rint(sa * p[:, 2] + ca * p[:, 3]).astype(int) + y
        out[i] = np.packbits((sm[ay, ax] < sm[by, bx]).astype(np.uint8))
    return out


def main():
    rng = np.random.default_rng(228)                                      # the repo's own seed (main.py:65)
    pattern = np.clip(np.rint(rng.normal(0, PATCH / 2.5, (256, 4))), -PATCH, PATCH).astype(np.int32)
    arrays = {"pattern": pattern}
    for k, name in ((1, "1.png"), (2, "2.png")):
        img = gray(os.path.join(REF, name))
        ys, xs = harris_keypoints(img, N_FEATURES)
        arrays[f"desc{k}"] = describe(img, ys, xs, pattern)
        arrays[f"kp{k}"] = np.stack([xs, ys], 1).astype(np.int32)
        print(name, img.shape, "->", arrays[f"desc{k}"].shape)
    np.savez_compressed(os.path.join(HERE, "image_descriptors.npz"), **arrays)
    d1, d2 = arrays["desc1"], arrays["desc2"]
    dist = np.bitwise_count(d2[:, None, :] ^ d1[None, :, :]).sum(-1)
    print("nearest-neighbour distance 2 -> 1: min", dist.min(), "median", int(np.median(dist.min(1))),
          "| rows with a tie for the nearest:", int((np.sort(dist, 1)[:, 0] == np.sort(dist, 1)[:, 1]).sum()))


if __name__ == "__main__":
    main()

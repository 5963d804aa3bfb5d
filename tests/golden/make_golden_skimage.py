"""Known answers of the scikit-image primitives the reference calls on this path, produced by the REAL library.

Run in the build container with the Anaconda interpreter that ships scikit-image (the system Python 3.10 has no skimage):

    /opt/conda/bin/python3.9 tests/golden/make_golden_skimage.py      # scikit-image 0.18.3, numpy 1.26

Call sites in the reference: skimage.draw.polygon occupancy_map.py:54,86,234,294,334,373; skimage.draw.line box_delivery_env.py:1219;
skimage.measure.approximate_polygon box_delivery_env.py:1240; skimage.measure.block_reduce occupancy_map.py:106;
skimage.morphology.disk / binary_dilation box_delivery_env.py:1168-1172.  (The reference does not pin scikit-image, requirements.txt.)

Output: tests/golden/skimage_golden.npz -- inputs and outputs only; tests/test_skimage_golden.py checks the oracle's restatements against it.
"""
import os

import numpy as np
import skimage
from skimage.draw import line, polygon
from skimage.measure import approximate_polygon, block_reduce
from skimage.morphology import binary_dilation, disk

rng = np.random.RandomState(20260401)
out = {"skimage_version": np.array(skimage.__version__)}

# ---- draw.polygon: the docstring triangle, random convex polygons with float vertices (the env rasterises convex hulls in pixel
#      coordinates), some reaching outside the image (clipping by shape), some with vertices exactly on pixel centres / rows -------------
polys = [(np.array([1.0, 2.0, 8.0]), np.array([1.0, 7.0, 4.0]), (10, 10))]
for k in range(60):
    n = rng.randint(3, 21)
    ang = np.sort(rng.uniform(0, 2 * np.pi, n))
    rad = rng.uniform(2.0, 30.0)
    cr, cc_ = rng.uniform(-5, 70), rng.uniform(-5, 70)
    r = cr + rad * np.sin(ang) * rng.uniform(0.6, 1.0, n)
    c = cc_ + rad * np.cos(ang) * rng.uniform(0.6, 1.0, n)
    # convex hull order is not needed by the scanline rule, but the env only draws convex polygons: keep the star-shaped outline convex-ish
    if k % 5 == 0:
        r = np.round(r); c = np.round(c)          # integer vertices: the vertex / edge rules of point_in_polygon
    if k % 7 == 0:
        r[0] = np.floor(r[0]) + 0.5               # a vertex on a half-pixel row
    polys.append((r, c, (64, 64)))
pr, pc, pshape, pmask = [], [], [], []
for r, c, shape in polys:
    img = np.zeros(shape, np.uint8)
    rr, cc = polygon(r, c, shape)
    img[rr, cc] = 1
    pr.append(r); pc.append(c); pshape.append(shape); pmask.append(np.packbits(img, axis=None))
out["poly_n"] = np.array([len(r) for r in pr])
out["poly_r"] = np.concatenate(pr); out["poly_c"] = np.concatenate(pc)
out["poly_shape"] = np.array(pshape)
out["poly_mask_len"] = np.array([len(m) for m in pmask]); out["poly_mask"] = np.concatenate(pmask)

# ---- draw.line: every direction and slope class, integer end points -------------------------------------------------------------------
ends = [(1, 1, 8, 8), (0, 0, 0, 0), (5, 9, 5, 2), (9, 5, 2, 5)]
for k in range(200):
    ends.append(tuple(int(v) for v in rng.randint(0, 60, 4)))
lr, lc, ln = [], [], []
for r0, c0, r1, c1 in ends:
    rr, cc = line(r0, c0, r1, c1)
    lr.append(rr); lc.append(cc); ln.append(len(rr))
out["line_ends"] = np.array(ends); out["line_n"] = np.array(ln)
out["line_r"] = np.concatenate(lr); out["line_c"] = np.concatenate(lc)

# ---- measure.approximate_polygon(coords, tolerance=1): 8-connected pixel paths like the spfa paths it is applied to ---------------------
paths, ap = [], []
paths.append(np.array([[0, 0], [1, 1], [2, 2], [3, 2], [4, 2], [5, 3], [6, 4], [6, 5], [6, 6]]))
for k in range(80):
    n = rng.randint(2, 120)
    steps = np.array([[0, 1], [1, 1], [1, 0], [1, -1], [0, -1], [-1, -1], [-1, 0], [-1, 1]])
    d = rng.randint(0, 8)
    p = [np.array([rng.randint(20, 80), rng.randint(20, 80)])]
    for i in range(n - 1):
        if rng.rand() < 0.15:
            d = (d + rng.choice([-1, 1])) % 8
        p.append(p[-1] + steps[d])
    paths.append(np.array(p))
for p in paths:
    ap.append(approximate_polygon(p, tolerance=1))
out["path_n"] = np.array([len(p) for p in paths]); out["path_xy"] = np.concatenate(paths)
out["approx_n"] = np.array([len(a) for a in ap]); out["approx_xy"] = np.concatenate(ap)

# ---- measure.block_reduce(img, (5, 5), np.mean) (global observation) and binary_dilation(img, disk(r)) (configuration space) -----------
img = (rng.rand(60, 200) < 0.3).astype(np.float64)
out["br_in"] = np.packbits(img.astype(np.uint8), axis=None); out["br_shape"] = np.array(img.shape)
out["br_out"] = block_reduce(img, (5, 5), np.mean)
dimg = (rng.rand(80, 90) < 0.02)
out["dil_in"] = np.packbits(dimg.astype(np.uint8), axis=None); out["dil_shape"] = np.array(dimg.shape)
for rad in (3, 5, 7):
    out["disk_%d" % rad] = disk(rad)
    out["dil_out_%d" % rad] = np.packbits(binary_dilation(dimg, disk(rad)).astype(np.uint8), axis=None)

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "skimage_golden.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes; scikit-image", skimage.__version__)

"""The oracle's restatements of the scikit-image primitives on this path against known answers of the REAL library (scikit-image 0.18.3, produced by
tests/golden/make_golden_skimage.py with the Anaconda interpreter of the build container).  Call sites in the reference:
skimage.draw.polygon occupancy_map.py:54, draw.line box_delivery_env.py:1219, measure.approximate_polygon box_delivery_env.py:1240,
measure.block_reduce occupancy_map.py:106, morphology.disk / binary_dilation box_delivery_env.py:1168-1172."""
import os

import numpy as np

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "skimage_golden.npz"))


def _unpack(bits, shape):
    n = int(shape[0]) * int(shape[1])
    return np.unpackbits(bits)[:n].reshape(int(shape[0]), int(shape[1]))


def test_draw_polygon_equals_skimage():
    from oracle.oracle import draw_polygon
    o, mo = 0, 0
    npix = 0
    for k, n in enumerate(G["poly_n"]):
        r, c = G["poly_r"][o:o + n], G["poly_c"][o:o + n]
        shape = G["poly_shape"][k]
        want = _unpack(G["poly_mask"][mo:mo + G["poly_mask_len"][k]], shape)
        got = np.zeros(tuple(shape), np.uint8)
        rr, cc = draw_polygon(r, c, tuple(int(v) for v in shape))
        got[rr, cc] = 1
        assert np.array_equal(got, want), ("polygon", k, r.tolist(), c.tolist())
        npix += int(want.sum())
        o += n; mo += G["poly_mask_len"][k]
    assert npix > 10000      # the set is not trivial
    # the published docstring example of skimage.draw.polygon is case 0 (triangle (1,1) (2,7) (8,4) on a 10 x 10 image)
    assert G["poly_n"][0] == 3 and tuple(G["poly_shape"][0]) == (10, 10)


def test_draw_line_equals_skimage():
    from oracle.oracle_bd import sk_line
    o = 0
    for k, (r0, c0, r1, c1) in enumerate(G["line_ends"]):
        n = G["line_n"][k]
        rr, cc = sk_line(int(r0), int(c0), int(r1), int(c1))
        assert np.array_equal(rr, G["line_r"][o:o + n]) and np.array_equal(cc, G["line_c"][o:o + n]), ("line", k)
        o += n


def test_approximate_polygon_equals_skimage():
    from oracle.oracle_bd import approx_polygon
    o, ao = 0, 0
    for k, n in enumerate(G["path_n"]):
        p = G["path_xy"][o:o + n]
        want = G["approx_xy"][ao:ao + G["approx_n"][k]]
        got = approx_polygon(p, 1.0)
        assert np.array_equal(got, want), ("approximate_polygon", k)
        o += n; ao += G["approx_n"][k]


def test_binary_dilation_with_disk_equals_skimage():
    from oracle.oracle_bd import dilate_disk
    img = _unpack(G["dil_in"], G["dil_shape"]).astype(np.float32)
    for rad in (3, 5, 7):
        want = _unpack(G["dil_out_%d" % rad], G["dil_shape"])
        got = (dilate_disk(img, rad) > 0).astype(np.uint8)
        assert np.array_equal(got, want), rad
        d = G["disk_%d" % rad]
        yy, xx = np.mgrid[-rad:rad + 1, -rad:rad + 1]
        assert np.array_equal(d, (xx * xx + yy * yy <= rad * rad).astype(d.dtype))


def test_block_mean_equals_skimage_block_reduce():
    img = _unpack(G["br_in"], G["br_shape"]).astype(np.float64)
    h, w = img.shape
    got = img.reshape(h // 5, 5, w // 5, 5).mean(axis=(1, 3))       # what the global observation does (tests/golden/make_golden_obs_pipeline.py)
    assert np.allclose(got, G["br_out"], rtol=0, atol=1e-15)

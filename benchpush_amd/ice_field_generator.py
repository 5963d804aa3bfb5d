"""Ice-field experiment generator with the reference's pipeline (generate_rand_exp, ship_ice_nav_mujoco/ship_ice_utils.py:660-887):
pack the channel with circles of random radii, turn every circle into a random convex polygon, clip to the channel, then add or delete
floes until the polygon concentration is within TOL of the request; pick the start pose; write the pickle the environments read
(ship_ice_env.py:76-80).  Host-side tooling, not part of the step path.

Third-party pieces that are absent here are restated:
  * ``packcircles.pack`` -> :func:`pack_circles`, the front-chain algorithm of Wang et al. 2006 ("Visualization of large hierarchical data by
    circle packing") in its common sibling-packing form.  **Parity unpinned**: the package is not in the image, so the layouts are valid
    dense packings but are not claimed to be the ones packcircles would produce for the same radii.
  * ``skimage.draw.polygon`` -> :func:`polygon_pixels` (crossing-number rule of skimage's point_in_polygon, vectorised over the pixel box).
Random draws use ``np.random`` / ``random`` in the reference's order (np.random: radii, shuffle, window offsets, choices, start pose;
random: the polygon sampler).
"""
import math
import pickle
import random as _random

import numpy as np

from .scenario import generate_polygon, poly_area

OBSTACLE = {"min_r": 0.45, "max_r": 0.70, "min_y": 3.0, "circular": False, "exp_dist": False}
TOL = 0.01      # ship_ice_utils.py:55
SCALE = 8       # ship_ice_utils.py:56 (cells per metre of the bookkeeping raster)


# ---------------------------------------------------------------------------------------------------------------------------------------
def pack_circles(radii):
    """Front-chain packing: circles are placed one after another, each tangent to two circles of the chain that surrounds the cluster,
    starting at the pair closest to the origin.  Yields (x, y, r) in input order.  Deterministic."""
    radii = [float(r) for r in radii]
    n = len(radii)
    x, y = [0.0] * n, [0.0] * n
    nxt, prv = list(range(n)), list(range(n))

    def place(b, a, c):
        dx, dy = x[b] - x[a], y[b] - y[a]
        d2 = dx * dx + dy * dy
        if d2:
            a2 = (radii[a] + radii[c]) ** 2
            b2 = (radii[b] + radii[c]) ** 2
            if a2 > b2:
                px = (d2 + b2 - a2) / (2 * d2)
                py = math.sqrt(max(0.0, b2 / d2 - px * px))
                x[c], y[c] = x[b] - px * dx - py * dy, y[b] - px * dy + py * dx
            else:
                px = (d2 + a2 - b2) / (2 * d2)
                py = math.sqrt(max(0.0, a2 / d2 - px * px))
                x[c], y[c] = x[a] + px * dx - py * dy, y[a] + px * dy + py * dx
        else:
            x[c], y[c] = x[a] + radii[c], y[a]

    def intersects(a, b):
        dr = radii[a] + radii[b] - 1e-6
        dx, dy = x[b] - x[a], y[b] - y[a]
        return dr > 0 and dr * dr > dx * dx + dy * dy

    def score(a):
        b = nxt[a]
        ab = radii[a] + radii[b]
        dx = (x[a] * radii[b] + x[b] * radii[a]) / ab
        dy = (y[a] * radii[b] + y[b] * radii[a]) / ab
        return dx * dx + dy * dy

    if n >= 1:
        x[0], y[0] = 0.0, 0.0
    if n >= 2:
        x[0], x[1], y[1] = -radii[1], radii[0], 0.0
    if n >= 3:
        place(1, 0, 2)
        a, b, c = 0, 1, 2
        nxt[a], prv[c] = b, b
        nxt[b], prv[a] = c, c
        nxt[c], prv[b] = a, a
        i = 3
        while i < n:
            c = i
            place(a, b, c)
            j, k, sj, sk = nxt[b], prv[a], radii[b], radii[a]
            hit = False
            while True:
                if sj <= sk:
                    if intersects(j, c):
                        b = j
                        nxt[a], prv[b] = b, a
                        hit = True
                        break
                    sj += radii[j]
                    j = nxt[j]
                else:
                    if intersects(k, c):
                        a = k
                        nxt[a], prv[b] = b, a
                        hit = True
                        break
                    sk += radii[k]
                    k = prv[k]
                if j == nxt[k]:
                    break
            if hit:
                continue            # try the same circle against the shortened chain
            prv[c], nxt[c] = a, b
            nxt[a], prv[b] = c, c
            b = c
            aa = score(a)
            c = nxt[c]
            while c != b:
                ca = score(c)
                if ca < aa:
                    a, aa = c, ca
                c = nxt[c]
            b = nxt[a]
            i += 1
    for i in range(n):
        yield (x[i], y[i], radii[i])


def polygon_pixels(r, c, shape):
    """skimage.draw.polygon(r, c, shape): (rr, cc) of the raster cells whose centres lie in / on the polygon (third-party, restated)."""
    r, c = np.asarray(r, np.float64), np.asarray(c, np.float64)
    minr, maxr = int(max(0, r.min())), int(math.ceil(r.max()))
    minc, maxc = int(max(0, c.min())), int(math.ceil(c.max()))
    maxr, maxc = min(shape[0] - 1, maxr), min(shape[1] - 1, maxc)
    if maxr < minr or maxc < minc:
        return np.zeros(0, np.intp), np.zeros(0, np.intp)
    yy, xx = np.mgrid[minr:maxr + 1, minc:maxc + 1]
    x, y = xx.astype(np.float64), yy.astype(np.float64)
    eps = 1e-12
    l_cross = np.zeros(x.shape, np.int64)
    r_cross = np.zeros(x.shape, np.int64)
    vertex = np.zeros(x.shape, bool)
    x1, y1 = c[-1] - x, r[-1] - y
    with np.errstate(divide="ignore", invalid="ignore"):
        for i in range(len(c)):
            x0, y0 = c[i] - x, r[i] - y
            vertex |= (np.abs(x0) < eps) & (np.abs(y0) < eps)
            up = (y0 > 0) != (y1 > 0)
            dn = (y0 < 0) != (y1 < 0)
            q = (x0 * y1 - x1 * y0) / (y1 - y0)
            r_cross += up & (q > 0)
            l_cross += dn & (q < 0)
            x1, y1 = x0, y0
    inside = vertex | ((r_cross & 1) != (l_cross & 1)) | ((r_cross & 1) == 1)
    return yy[inside], xx[inside]


# ---------------------------------------------------------------------------------------------------------------------------------------
def _floe(vertices, centre, radius, map_shape, scale):
    im_shape = (int(map_shape[0] * scale), int(map_shape[1] * scale))
    rr, cc = polygon_pixels(vertices[:, 1] * scale, vertices[:, 0] * scale, im_shape)
    return {"vertices": vertices, "centre": centre, "radius": radius, "pixels": (rr, cc), "area": poly_area(vertices)}


def compute_poly_ob_concentration(polys, map_shape, obstacle=OBSTACLE, scale=SCALE):
    im = np.zeros((int(map_shape[0] * scale), int(map_shape[1] * scale)))
    area = 0
    for p in polys:
        area += p["area"]
        rr, cc = p["pixels"]
        im[rr, cc] = 1
    return area / (map_shape[1] * (map_shape[0] - obstacle["min_y"])), im


def increase_concentration(obstacles, desired, map_shape, obstacle=OBSTACLE, scale=SCALE, tol=TOL, rng=None):
    """ship_ice_utils.py:672-729: drop new floes into empty square windows of the raster until the concentration is reached."""
    actual, im = compute_poly_ob_concentration(obstacles, map_shape, obstacle, scale)
    max_r = obstacle["max_r"]
    while actual < desired - tol and max_r - obstacle["min_r"] > 0.05:
        r = np.random.uniform(obstacle["min_r"], max_r)
        slice_shape = int(max(1 / scale, r * 2) * scale)
        rand_offset_x = np.random.choice(np.arange(slice_shape))
        rand_offset_y = np.random.choice(np.arange(slice_shape))
        centres = []
        for i in range(rand_offset_y, im.shape[0] - slice_shape + 1, slice_shape):
            for j in range(rand_offset_x, im.shape[1] - slice_shape + 1, slice_shape):
                if obstacle["min_y"] * scale <= i and i + slice_shape <= map_shape[0] * scale:
                    if im[i: i + slice_shape, j: j + slice_shape].sum() == 0:
                        centres.append([(j + slice_shape / 2) / scale, (i + slice_shape / 2) / scale])
        if len(centres) == 0:
            max_r = r
            continue
        for ind in np.random.choice(len(centres), size=int(len(centres) * 0.5), replace=False):
            x, y = centres[ind]
            r = slice_shape / scale / 2
            vertices = generate_polygon(diameter=r * 2, origin=(x, y), rng=rng)
            obstacles.append(_floe(vertices, (x, y), r, map_shape, scale))
            actual, im = compute_poly_ob_concentration(obstacles, map_shape, obstacle, scale)
    return obstacles


def decrease_concentration(obstacles, desired, map_shape, obstacle=OBSTACLE, scale=SCALE, tol=TOL):
    """ship_ice_utils.py:732-745: delete random floes (and reshuffle) until the concentration is reached."""
    obstacles = list(obstacles)
    actual, _ = compute_poly_ob_concentration(obstacles, map_shape, obstacle, scale)
    while actual > desired + tol:
        ind = int(np.random.choice(np.arange(len(obstacles))))
        del obstacles[ind]
        order = np.arange(len(obstacles))
        np.random.shuffle(order)
        obstacles = [obstacles[k] for k in order]
        actual, _ = compute_poly_ob_concentration(obstacles, map_shape, obstacle, scale)
    return obstacles


def find_best_start_x(obstacles, map_shape, slice_shape=(10, 3), scale=SCALE):
    """ship_ice_utils.py:748-776: centre of the 3 m wide strip (first 10 m) with the least ice, nearest to the channel centre on ties."""
    im = np.zeros((int(map_shape[0] * scale), int(map_shape[1] * scale)))
    for ob in obstacles:
        rr, cc = polygon_pixels(ob["vertices"][:, 1] * scale, ob["vertices"][:, 0] * scale, im.shape)
        im[rr, cc] = 1
    c = []
    for i in range((im.shape[1] - slice_shape[1] * scale) // scale):
        if i == 0:
            c.append(np.inf)
        sub = im[: slice_shape[0] * scale, i * scale: (i + slice_shape[1]) * scale]
        c.append(sub.sum() / np.multiply(*sub.shape))
    min_idx = np.where(np.asarray(c) == np.min(c))[0]
    best = min_idx[np.argmin(np.abs((min_idx + (min_idx + slice_shape[1])) // 2 - map_shape[1] / 2))].item()
    return (best + (best + slice_shape[1])) // 2


def generate_rand_exp(conc, map_shape=(40, 12), ship_state=None, goal=(0, 9.0), max_trials=100, filename=None, obstacle=OBSTACLE,
                      scale=SCALE, tol=TOL, seed=None):
    """ship_ice_utils.py:779-887.  ``map_shape`` = (length, width) in metres; ``ship_state`` = dict(range_x | None, range_y, range_theta).
    Returns the experiment dict ``{'meta_data': ..., 'exp': {conc: {trial: {'goal', 'ship_state', 'obstacles'}}}}`` in the layout the
    environments unpickle (the reference's generator keys trials directly under 'exp'; the shipped files carry the concentration level
    that ship_ice_env.py:79 indexes), and writes it to ``filename`` if given."""
    if ship_state is None:
        ship_state = {"range_x": [1.0, map_shape[1] - 1.0], "range_y": [1.0, 1.0], "range_theta": [np.pi / 2, np.pi / 2]}
    rng = None
    if seed is not None:
        np.random.seed(seed)
        rng = _random.Random(seed)
    exp = {i: {"goal": None, "ship_state": None, "obstacles": None} for i in range(max_trials)}
    avg_r = obstacle["min_r"] * 1.5 if obstacle["exp_dist"] else (obstacle["min_r"] + obstacle["max_r"]) / 2
    num_circ = (np.pi * (((map_shape[0] ** 2 + map_shape[1] ** 2) ** 0.5) / 2) ** 2) / (np.pi * avg_r ** 2)
    for i in range(max_trials):
        if obstacle["exp_dist"]:
            radii = np.maximum(obstacle["min_r"], np.minimum(obstacle["max_r"], np.random.exponential(scale=avg_r, size=int(num_circ))))
        else:
            radii = np.random.uniform(obstacle["min_r"], obstacle["max_r"], size=int(num_circ))
        circles = np.asarray(list(pack_circles(radii)))
        circles[:, 1] += -circles[:, 1].min()
        circles[:, 0] += map_shape[1]
        circles = circles[np.logical_and(circles[:, 0] >= 0, circles[:, 0] <= map_shape[1])]
        circles = circles[np.logical_and(circles[:, 1] >= 0, circles[:, 1] <= map_shape[0])]
        circles = circles[np.logical_and(circles[:, 1] >= obstacle.get("min_y", 0), circles[:, 1] <= map_shape[0])]
        np.random.shuffle(circles)
        obstacles = []
        for (x, y, radius) in circles:
            vertices = generate_polygon(diameter=radius * 2, origin=(x, y), rng=rng)
            vertices[:, 0][vertices[:, 0] < 0] = 0
            vertices[:, 0][vertices[:, 0] >= map_shape[1]] = map_shape[1]
            min_y = obstacle.get("min_y", False) or 0
            vertices[:, 1][vertices[:, 1] < min_y] = min_y
            vertices[:, 1][vertices[:, 1] > map_shape[0]] = map_shape[0]
            obstacles.append(_floe(vertices, (x, y), radius, map_shape, scale))
        poly_conc, _ = compute_poly_ob_concentration(obstacles, map_shape, obstacle, scale)
        if abs(conc - poly_conc) > tol:
            if conc > poly_conc:
                obstacles = increase_concentration(obstacles, conc, map_shape, obstacle, scale, tol, rng)
            else:
                obstacles = decrease_concentration(obstacles, conc, map_shape, obstacle, scale, tol)
        exp[i]["obstacles"] = obstacles
        exp[i]["goal"] = goal
        if ship_state["range_x"] is None:
            x = find_best_start_x(obstacles, map_shape, scale=scale)
        else:
            x = np.random.uniform(low=ship_state["range_x"][0], high=ship_state["range_x"][1])
        y = np.random.uniform(low=ship_state["range_y"][0], high=ship_state["range_y"][1])
        theta = np.random.uniform(low=ship_state["range_theta"][0], high=ship_state["range_theta"][1])
        exp[i]["ship_state"] = (x, y, theta)
    out = {"meta_data": {"concentration": conc, "map_shape": map_shape, "obstacle_config": dict(obstacle), "ship_state_config": ship_state,
                         "goal": goal, "scale": scale},
           "exp": {conc: exp}}
    if filename:
        with open(filename, "wb") as f:
            pickle.dump(out, f)
    return out

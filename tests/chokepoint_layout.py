"""A box-delivery layout whose configuration space has a one-cell-wide DIAGONAL corridor (test data, shared by the CPU and the GPU test).

Two convex obstacle quads fill the strip -3 m <= x <= 0 m of the 10 x 5 m room from wall to wall, except for a 45-degree channel of 1.30 m between
them; the configuration space dilates obstacles by the robot radius (14 px at 22.4 px/m), which leaves a single diagonal line of free cells: the
pocket on the left (where the robot starts) and the room on the right (boxes, receptacle) are 8-connected but not 4-connected.  A shortest-path
search from the robot therefore walks ~65 cells whose distances grow by sqrt(2) per cell while the pocket has long been exhausted: distance buckets
of width 1 are skipped (86, 89, 93, ...), which is what the bucketed searches of k_bd_robot_map / k_bd_finish must survive (ADVICE r2 item 1;
reference: box_delivery_env.py:1131-1138, spfa through `configuration_space`)."""
import numpy as np

from benchpush_amd import box_delivery_scenario as S

GAP, X0, X1, YA = 1.30, -3.0, 0.0, -1.5


def make_chokepoint_trial(cfg):
    rs = np.random.RandomState(0)
    start = (-4.0, YA - 0.3, np.pi / 2)
    boundary, start = S.generate_boundary(cfg, rs, start)
    yb = YA + (X1 - X0)
    o = np.array([-1.0, 1.0]) / np.sqrt(2.0) * GAP / 2
    m0, m1 = np.array([X0, YA]), np.array([X1, yb])
    upper = [tuple(m0 + o), tuple(m1 + o), (X1, 3.0), (X0, 3.0)]       # above / left of the channel, up to (beyond) the top wall
    lower = [tuple(m1 - o), tuple(m0 - o), (X0, -3.0), (X1, -3.0)]     # below / right of it, down to the bottom wall
    ins = [dict(type="column", position=(X0, 2.0), vertices=upper, length=1, width=1),
           dict(type="column", position=(X1, -2.0), vertices=lower, length=1, width=1)]
    k = max(i for i, b in enumerate(boundary) if b["type"] == "wall") + 1
    boundary = boundary[:k] + ins + boundary[k:]
    boxes = np.array([[2.0, -1.0, 0.3], [3.0, 0.5, 1.0], [1.5, 1.0, 2.0]], np.float64)   # in the room on the right: their paths to the receptacle stay there
    return dict(start=np.array(start, np.float64), boxes=boxes, boundary=boundary, statics=S.static_shapes(boundary))

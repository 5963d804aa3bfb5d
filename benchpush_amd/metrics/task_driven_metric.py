"""TaskDrivenMetric for area-clearing (reference: benchpush/common/metrics/task_driven_metric.py:8-155).

efficiency = MST cost over {robot start, cleared boxes, each box's nearest goal point} / robot path length (plus the success rate),
effort = (m0 l0 + sum_i min-goal-distance_i * area_i) / (m0 l0 + total_work).  The reference builds shapely Polygons and a networkx
graph; here the polygon area / centroid are the shoelace formulas shapely evaluates and the minimum spanning tree is Prim's (the MST
weight does not depend on the algorithm).
"""
import numpy as np

from .base_metric import BaseMetric


def _area_centroid(poly):
    p = np.asarray(poly, np.float64)
    x, y = p[:, 0], p[:, 1]
    xn, yn = np.roll(x, -1), np.roll(y, -1)
    cr = x * yn - xn * y
    a = cr.sum() / 2
    cx, cy = ((x + xn) * cr).sum() / (6 * a), ((y + yn) * cr).sum() / (6 * a)
    return abs(a), (cx, cy)


def _mst_weight(n, edges):
    if n == 0:
        return 0.0
    adj = {i: [] for i in range(n)}
    for a, b, w in edges:
        adj[a].append((b, w)); adj[b].append((a, w))
    total = 0.0
    seen_all = set()
    for root in range(n):          # forest over the components, like networkx.minimum_spanning_tree
        if root in seen_all:
            continue
        seen = {root}
        best = {}
        for b, w in adj[root]:
            best[b] = min(w, best.get(b, np.inf))
        while best:
            v = min(best, key=best.get)
            total += best.pop(v)
            seen.add(v)
            for b, w in adj[v]:
                if b not in seen:
                    best[b] = min(w, best.get(b, np.inf))
        seen_all |= seen
    return total


class TaskDrivenMetric(BaseMetric):
    def __init__(self, alg_name, robot_mass, box_mass=None) -> None:
        super().__init__(alg_name=alg_name)
        self.eps_reward = 0
        self.total_mass_dist = 0
        self.robot_mass = robot_mass
        self.total_robot_dist = 0
        self.box_mass = box_mass
        if not hasattr(self, "success_rates"):
            self.success_rates = []

    def compute_mst_cost_for_successful_boxes(self):
        if not any(self.box_completed_statuses):
            return 0
        done = [c for c, s in zip(self._centroids, self.box_completed_statuses) if s]
        k = len(done)
        robot = 2 * k
        edges = []
        for i in range(k):
            for j in range(i + 1, k):
                edges.append((i, j, float(np.linalg.norm(np.array(done[i]) - np.array(done[j])))))
            edges.append((robot, i, float(np.linalg.norm(np.array(self.initial_robot_state[:2]) - np.array(done[i])))))
            edges.append((i, i + k, min(float(np.linalg.norm(np.array(done[i]) - np.array(g))) for g in self.goal_positions)))
        return _mst_weight(2 * k + 1, edges)

    def compute_efficiency_score(self, mst_cost):
        success_rate = sum(self.box_completed_statuses) / len(self.box_completed_statuses)
        return success_rate, mst_cost / self.total_robot_dist

    def compute_effort_score(self):
        min_mass_dist = 0
        for (area, c), s in zip(zip(self._areas, self._centroids), self.box_completed_statuses):
            if not s:
                continue
            d = min(float(np.linalg.norm(np.array([g[0], g[1]]) - np.array(c))) for g in self.goal_positions)
            min_mass_dist += d * (self.box_mass if self.box_mass is not None else area)
        own = self.robot_mass * self.total_robot_dist
        return (own + min_mass_dist) / (own + self.total_mass_dist)

    def update(self, info, reward, eps_complete=False):
        self.eps_reward += reward
        self.total_mass_dist = info["total_work"]
        self.box_completed_statuses = info["box_completed_statuses"]
        cur = info["state"]
        self.total_robot_dist += np.linalg.norm(np.array(self.robot_state[:2]) - np.array(cur[:2]))
        self.robot_state = cur
        if eps_complete:
            self.rewards.append(self.eps_reward)
            mst_cost = self.compute_mst_cost_for_successful_boxes()
            success_rate, efficiency = self.compute_efficiency_score(mst_cost)
            self.success_rates.append(success_rate)
            self.efficiency_scores.append(efficiency)
            self.effort_scores.append(self.compute_effort_score())

    def reset(self, info):
        self.eps_reward = 0
        self.total_mass_dist = 0
        self.total_robot_dist = 0
        self.trial_success = False
        self.robot_state = info["state"]
        self.initial_robot_state = info["state"]
        boxes = info["obs"]
        ac = [_area_centroid(b) for b in boxes]
        self._areas = [a for a, _ in ac]
        self._centroids = [c for _, c in ac]
        self.goal_positions = [[g[0], g[1]] if not hasattr(g, "x") else [g.x, g.y] for g in info["goal_positions"]]
        self.box_completed_statuses = [False] * len(boxes)

"""BaseMetric: per-algorithm lists of episode scores + box plots.

API-identical to the reference's benchpush/common/metrics/base_metric.py:7-193 (attributes ``rewards``,
``efficiency_scores``, ``effort_scores``, ``success_rates``, ``alg_name``; abstract ``compute_efficiency_score``,
``compute_effort_score``, ``update(info, reward, eps_complete)``, ``reset(info)``; static plot helpers taking the tuples
returned by ``BasePolicy.evaluate``).  matplotlib is imported lazily so that headless rollouts do not pay for it.
"""
import os
from abc import ABC, abstractmethod


def _plt():
    import matplotlib
    matplotlib.use("Agg", force=False)
    from matplotlib import pyplot as plt
    return plt


class BaseMetric(ABC):
    def __init__(self, alg_name) -> None:
        self.rewards = []
        self.efficiency_scores = []
        self.effort_scores = []
        self.success_rates = []
        self.alg_name = alg_name

    def plot_scores(self, save_fig_dir):
        """One box plot per score list of this algorithm (reference: base_metric.py:21-64)."""
        plt = _plt()
        os.makedirs(save_fig_dir, exist_ok=True)
        fig, ax = plt.subplots()
        for data, title, ylabel, suffix in (
                (self.efficiency_scores, "Efficiency Plot", "Efficiency Scores", "_efficiency.png"),
                (self.effort_scores, "Effort Plot", "Effort Scores", "_effort.png"),
                (self.rewards, "Rewards Plot", "Rewards", "_rewards.png"),
                (self.success_rates, "Success Rates Plot", "Success Rates", "_success_rates.png")):
            ax.clear()
            ax.boxplot([data], showmeans=True)
            ax.set_title(title)
            ax.set_xlabel("Trials")
            ax.set_ylabel(ylabel)
            fig.savefig(os.path.join(save_fig_dir, self.alg_name + suffix))
        plt.close("all")

    @staticmethod
    def plot_algs_score(scores, score_name, alg_names, save_fig_dir, filename, legend=True):
        plt = _plt()
        os.makedirs(save_fig_dir, exist_ok=True)
        fig, ax = plt.subplots()
        colors = [(0.43, 0.64, 0.68), (0.84, 0.39, 0.26), (0.65, 0.65, 0.65), (0.3, 0.3, 0.3), (0.1, 0.1, 0.1)]
        boxes, positions = [], []
        for i, score in enumerate(scores):
            pos = 1.5 * i + 1
            bp = ax.boxplot([score], positions=[pos], showmeans=False, widths=0.8, patch_artist=True,
                            boxprops=dict(facecolor=colors[i % len(colors)]), medianprops=dict(color="black"))
            boxes.append(bp["boxes"][0])
            positions.append(pos)
        ax.set_xticks(positions)
        ax.set_xticklabels(alg_names)
        if legend:
            ax.legend(boxes, alg_names, loc="lower right")
        ax.set_xlim(0, 1.5 * len(scores) + 0.5)
        fig.savefig(os.path.join(save_fig_dir, filename + ".png"))
        plt.close(fig)

    @staticmethod
    def plot_algs_scores(benchmark_results, save_fig_dir: str, plot_success=False) -> None:
        """benchmark_results: list of tuples returned by policy.evaluate() (base_metric.py:107-137)."""
        eff, effort, rew, names, succ = [], [], [], [], []
        for res in benchmark_results:
            if plot_success:
                s, e, f, r, n = res
                succ.append(s)
            else:
                e, f, r, n = res
            eff.append(e)
            effort.append(f)
            rew.append(r)
            names.append(n)
        BaseMetric.plot_algs_score(eff, "Efficiency Score", names, save_fig_dir, "efficiency_benchmark")
        BaseMetric.plot_algs_score(effort, "Effort Score", names, save_fig_dir, "effort_benchmark")
        BaseMetric.plot_algs_score(rew, "Rewards", names, save_fig_dir, "reward_benchmark")
        if plot_success:
            BaseMetric.plot_algs_score(succ, "Task Success Score", names, save_fig_dir, "success_benchmark")

    @abstractmethod
    def compute_efficiency_score(self):
        raise NotImplementedError

    @abstractmethod
    def compute_effort_score(self):
        raise NotImplementedError

    @abstractmethod
    def update(self, info, reward, eps_complete=False):
        raise NotImplementedError

    @abstractmethod
    def reset(self, info):
        raise NotImplementedError

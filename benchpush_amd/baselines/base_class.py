"""BasePolicy: the policy interface of the reference (benchpush/baselines/base_class.py:5-38), unchanged."""
from abc import ABC, abstractmethod
from typing import List, Tuple


class BasePolicy(ABC):
    def __init__(self) -> None:
        ...

    def train(self):
        """Train the policy."""
        raise NotImplementedError

    @abstractmethod
    def evaluate(self, num_eps: int, model_eps: str = "latest") -> Tuple[List[float], List[float], List[float], str]:
        """Evaluate for `num_eps` episodes -> (efficiency scores, effort scores, rewards, algorithm name)."""
        raise NotImplementedError

    @abstractmethod
    def act(self, observation, **kwargs):
        """Compute an action given the observation."""
        raise NotImplementedError

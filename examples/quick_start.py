"""README quick start: python examples/quick_start.py (on an MI355X box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, benchpush_amd
from benchpush_amd.envs.ship_ice import BatchedShipIceEnv
env = benchpush_amd.make("ship-ice-v0", cfg={"concentration": 0.3})
obs, info = env.reset(); obs, reward, terminated, truncated, info = env.step(0.2)
print(obs.shape, reward, terminated, truncated, sorted(info.keys()))
venv = BatchedShipIceEnv(64, cfg={"concentration": 0.3})
obs, info = venv.reset()
obs, rew, term, trunc, info = venv.step(torch.zeros(64, dtype=torch.float64, device="cuda"))
venv.reset(term)
cost = venv.cost_maps(scale=5, m=76, n=12)
print(obs.shape, cost.shape, float(cost[:, :, 1:-1].max()))

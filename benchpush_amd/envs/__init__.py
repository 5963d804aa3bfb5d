"""Environment registration, mirroring benchpush/environments/__init__.py:3-7 (id + max_episode_steps)."""
from ..gym_shim import register

register(
    id="ship-ice-v0",
    entry_point="benchpush_amd.envs.ship_ice:ShipIceEnv",
    max_episode_steps=300,
)

register(
    id="maze-NAMO-v0",
    entry_point="benchpush_amd.envs.maze_namo:MazeNAMO",
    max_episode_steps=400,
)

register(
    id="box-delivery-v0",
    entry_point="benchpush_amd.envs.box_delivery:BoxDeliveryEnv",
    max_episode_steps=30000,
)

register(
    id="area-clearing-v0",
    entry_point="benchpush_amd.envs.area_clearing:AreaClearingEnv",
    max_episode_steps=30000,
)

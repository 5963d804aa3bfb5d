from .base_metric import BaseMetric  # noqa: F401
from .ship_ice_metric import BatchedShipIceMetric, ShipIceMetric  # noqa: F401
from .maze_namo_metric import MazeNamoMetric  # noqa: F401

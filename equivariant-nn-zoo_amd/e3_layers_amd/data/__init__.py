from .data import Data, Batch
from .compute_edge import computeEdgeVector, computeEdgeIndex

__all__ = ["Data", "Batch", "computeEdgeVector", "computeEdgeIndex"]

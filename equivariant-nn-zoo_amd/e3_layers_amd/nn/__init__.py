from .sequential import Module, SequentialGraphNetwork
from .core import Linear, FullyConnectedNet, FullyConnectedTensorProduct, Gate, NormActivation, UVUTensorProduct
from .pointwise import PointwiseLinear, LayerNormalization, TensorProductExpansion, Concat
from .embedding import (
    symmetricCutoff, _poly_cutoff, PolynomialCutoff, BesselBasis, SphericalEncoding, RadialBasisEncoding,
    Broadcast, OneHotEncoding, RelativePositionEncoding,
)
from .message_passing import FactorizedConvolution, MessagePassing
from .scaling import PerTypeScaleShift
from .output import GradientOutput, Pooling

__all__ = [
    "Module", "SequentialGraphNetwork", "Linear", "FullyConnectedNet", "FullyConnectedTensorProduct", "Gate", "NormActivation",
    "UVUTensorProduct", "PointwiseLinear", "LayerNormalization", "TensorProductExpansion", "Concat",
    "symmetricCutoff", "_poly_cutoff", "PolynomialCutoff", "BesselBasis", "SphericalEncoding", "RadialBasisEncoding",
    "Broadcast", "OneHotEncoding", "RelativePositionEncoding", "FactorizedConvolution", "MessagePassing",
    "PerTypeScaleShift", "GradientOutput", "Pooling",
]

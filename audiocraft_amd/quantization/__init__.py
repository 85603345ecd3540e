from .vq import ResidualVectorQuantizer, BaseQuantizer, QuantizedResult  # noqa: F401

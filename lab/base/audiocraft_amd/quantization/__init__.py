from .vq import ResidualVectorQuantizer, BaseQuantizer  # noqa: F401

from .projection import ProjectionMatrixBuilder

__all__ = ["ProjectionMatrixBuilder"]

"""MI355X-native implementation of SUAVE's ELBO training step and generative decode."""

from .types import ColumnSpec, Schema  # noqa: F401

__all__ = ["ColumnSpec", "Schema"]

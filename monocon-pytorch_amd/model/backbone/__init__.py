"""DLA-34 backbone and its IDA / DLA up-sampling neck (parameter containers + sub-module forward API)."""
from .dla import DLA
from .dla_neck import DLAUp, IDAUp

__all__ = ("DLA", "DLAUp", "IDAUp")

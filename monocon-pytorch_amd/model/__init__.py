"""Public model surface (`from model import DLA, DLAUp, IDAUp, MonoConDenseHeads, AttnBatchNorm2d, MonoConDetector`),
same names as the reference package; every class routes its arithmetic to libmonocon_hip."""
from .backbone import DLA, DLAUp, IDAUp
from .dense_heads import MonoConDenseHeads
from .detector import MonoConDetector
from .norm import AttnBatchNorm2d

__all__ = ("DLA", "DLAUp", "IDAUp", "AttnBatchNorm2d", "MonoConDenseHeads", "MonoConDetector")

"""The detector that ties backbone, neck and dense heads to the fused HIP plans."""
from .monocon_detector import MonoConDetector

__all__ = ("MonoConDetector",)

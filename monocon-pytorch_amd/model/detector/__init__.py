from .monocon_detector import MonoConDetector

__all__ = ['MonoConDetector']

from .dla import DLA
from .dla_neck import IDAUp, DLAUp

__all__ = ['DLA', 'IDAUp', 'DLAUp']

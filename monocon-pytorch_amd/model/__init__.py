from .backbone import *
from .norm import *
from .dense_heads import *
from .detector import *

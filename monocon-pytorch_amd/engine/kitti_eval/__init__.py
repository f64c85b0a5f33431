from .eval import kitti_eval, do_eval

__all__ = ["kitti_eval", "do_eval"]

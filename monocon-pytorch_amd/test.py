"""`python test.py --config_file cfg.yaml --checkpoint_file x.pth [--evaluate]` (reference test.py:12-70)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from engine.monocon_engine import MonoconEngine        # noqa: E402
from utils.engine_utils import load_cfg, tprint         # noqa: E402


def main():
    # (behind the __main__ guard: the loaders' workers come from a fork server and import this module again)
    ap = argparse.ArgumentParser('MonoCon Tester for KITTI 3D Object Detection Dataset')
    ap.add_argument('--config_file', type=str, required=True)
    ap.add_argument('--checkpoint_file', type=str, required=True)
    ap.add_argument('--gpu_id', type=int, default=0)
    ap.add_argument('--evaluate', action='store_true')
    args = ap.parse_args()

    cfg = load_cfg(args.config_file)
    cfg.GPU_ID = args.gpu_id
    engine = MonoconEngine(cfg, auto_resume=False, is_test=True)
    engine.load_checkpoint(args.checkpoint_file, verbose=True)
    if args.evaluate:
        tprint("Mode: Evaluation")
        print(engine.evaluate())


if __name__ == '__main__':
    main()

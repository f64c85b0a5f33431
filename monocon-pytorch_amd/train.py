"""`python train.py [--config_file cfg.yaml] [KEY VALUE ...]` -- same flow as the reference's
train.py:19-45 (default config -> seed -> MonoconEngine(cfg).train()).  Under
`python -m torch.distributed.run --nproc-per-node N train.py` it trains data-parallel."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from engine.monocon_engine import MonoconEngine                                   # noqa: E402
from utils.engine_utils import generate_random_seed, get_default_cfg, load_cfg, set_random_seed, tprint   # noqa: E402


def main():
    # (a function behind the __main__ guard: the loaders' workers come from a fork server and import this module again)
    ap = argparse.ArgumentParser()
    ap.add_argument('--config_file', type=str, default=None)
    ap.add_argument('opts', nargs='*', help="KEY VALUE overrides, e.g. DATA.ROOT synthetic DATA.BATCH_SIZE 8")
    args = ap.parse_args()

    cfg = load_cfg(args.config_file) if args.config_file else get_default_cfg()
    if args.opts:
        import yaml
        cfg.set_new_allowed(True)
        cfg.merge_from_list([yaml.safe_load(v) if i % 2 else v for i, v in enumerate(args.opts)])
    from hipmonocon import dist as hdist                                              # noqa: E402
    hdist.init_from_env()
    # one seed for all ranks: the replicas must start from the same weights (rank 0's draw is broadcast)
    seed = hdist.broadcast_seed(generate_random_seed(cfg.get('SEED', -1)))
    set_random_seed(seed)
    cfg.SEED = seed
    tprint("Using Random Seed %d" % seed)
    engine = MonoconEngine(cfg)
    engine.train()


if __name__ == '__main__':
    main()

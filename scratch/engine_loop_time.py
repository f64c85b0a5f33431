"""ms per step of MonoconEngine.train_one_epoch on the synthetic dataset (B=32, 384x1280), default loop vs
MONOCON_HIP_SYNC_LOOP=1 (the reference's loop as written).  usage: python scratch/engine_loop_time.py [workers] [steps]"""
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "monocon-pytorch_amd"), REPO]
import torch


class PooledDataset(torch.utils.data.Dataset):
    """the synthetic dataset costs ~0.4 s per sample (it draws the frame and the labels): a pool of 16 samples drawn once,
    handed out as copies, so that the loop is what is timed and not the generator"""

    def __init__(self, base, length, pool=16):
        self.base, self.length = base, length
        self.pool = [base[i] for i in range(pool)]
        self.collate_fn = base.collate_fn

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        d = self.pool[idx % len(self.pool)]
        return {'img': d['img'].clone(), 'label': {k: v.clone() for k, v in d['label'].items()}, 'calib': d['calib'],
                'img_metas': dict(d['img_metas'], sample_idx=idx)}

    def evaluate(self, *a, **k):
        return self.base.evaluate(*a, **k)


class Timed:
    """time the consumer spends blocked in the loader's next()"""

    def __init__(self, loader):
        self.loader, self.wait = loader, 0.0

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        it = iter(self.loader)
        while True:
            t0 = time.perf_counter()
            try:
                b = next(it)
            except StopIteration:
                return
            self.wait += time.perf_counter() - t0
            yield b


def run(sync, workers, steps):
    from engine.monocon_engine import MonoconEngine
    from torch.utils.data import DataLoader
    from utils.engine_utils import get_default_cfg
    os.environ["MONOCON_HIP_SYNC_LOOP"] = "1" if sync else "0"
    cfg = get_default_cfg()
    cfg.set_new_allowed(True)
    cfg.OUTPUT_DIR = tempfile.mkdtemp()
    cfg.SEED = 3
    cfg.DATA.ROOT = 'synthetic'
    cfg.DATA.SYNTHETIC_LENGTH = 32
    cfg.DATA.BATCH_SIZE = 32
    cfg.DATA.NUM_WORKERS = workers
    cfg.MODEL.BACKBONE.IMAGENET_PRETRAINED = False
    cfg.SOLVER.OPTIM.NUM_EPOCHS = 2
    cfg.PERIOD.EVAL_PERIOD = 100
    cfg.SOLVER.SCHEDULER.ENABLE = False      # (its length was fixed from the 1-batch loader the engine built)
    cfg.PERIOD.LOG_PERIOD = 50
    eng = MonoconEngine(cfg)
    ds = PooledDataset(eng.train_dataset, 32 * steps)
    if os.environ.get("RING", "1") == "1" and workers > 0:
        from hipmonocon.feed import RingLoader
        eng.train_loader = RingLoader(ds, 32, workers, shuffle=True, collate_fn=ds.collate_fn)
    else:
        eng.train_loader = DataLoader(ds, batch_size=32, num_workers=workers, shuffle=True, collate_fn=ds.collate_fn,
                                      pin_memory=True, persistent_workers=workers > 0, prefetch_factor=4 if workers else None)
    eng.train_loader = Timed(eng.train_loader)
    eng.model.set_precision(os.environ.get("PREC", "f16x2"))
    eng.train_one_epoch()          # warm-up epoch: plan build, autotune, worker start
    torch.cuda.synchronize()
    eng.train_loader.wait = 0.0
    t0 = time.perf_counter()
    eng.train_one_epoch()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("sync_loop=%d workers=%d: %.2f ms/step over %d steps (%.1f img/s), blocked in the loader %.2f ms/step, last loss %.4f"
          % (sync, workers, dt / steps * 1e3, steps, 32 * steps / dt, eng.train_loader.wait / steps * 1e3, eng.entire_losses[-1]),
          flush=True)
    del eng
    torch.cuda.empty_cache()


if __name__ == "__main__":
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    for sync in (0, 1, 0):
        run(sync, workers, steps)

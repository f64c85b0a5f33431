"""where a train loop's time goes with each feed: resident batch / DataLoader(pin_memory) + DevicePrefetcher / RingLoader +
DevicePrefetcher, B=32 384x1280 float32 frames out of a pool of pre-drawn samples.
usage: python scratch/feed_time.py [workers] [steps]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "monocon-pytorch_amd"), REPO, os.path.join(REPO, "scratch")]
import torch
from torch.utils.data import DataLoader

from engine_loop_time import PooledDataset


def main():
    from dataset.synthetic_dataset import SyntheticMonoConDataset
    from hipmonocon.feed import DeferredScalars, DevicePrefetcher, RingLoader
    from model import MonoConDetector
    from solver import AdamW
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    B = 32
    ds = PooledDataset(SyntheticMonoConDataset(length=16, seed=1), B * steps)
    m = MonoConDetector(34, pretrained_backbone=False).cuda().train().set_precision("f16x2")
    opt = AdamW(m.parameters(), lr=1e-4, weight_decay=1e-5, betas=(0.95, 0.99), max_grad_norm=35.0)

    def loop(feed, tag):
        losses = DeferredScalars()
        t_wait = t_enq = 0.0
        evs = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        it = iter(feed)
        n = 0
        while True:
            ta = time.perf_counter()
            try:
                batch = next(it)
            except StopIteration:
                break
            tb = time.perf_counter()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            opt.zero_grad()
            _, loss = m(batch)
            total = sum(loss.values())
            total.backward()
            losses.push(total)
            opt.step()
            e1.record()
            evs.append((e0, e1))
            losses.ready(1)
            tc = time.perf_counter()
            t_wait += tb - ta
            t_enq += tc - tb
            n += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        busy = sum(a.elapsed_time(b) for a, b in evs[2:]) / max(len(evs) - 2, 1)
        gap = sum(evs[i][1].elapsed_time(evs[i + 1][0]) for i in range(2, len(evs) - 1)) / max(len(evs) - 3, 1)
        print("%-34s %7.2f ms/step (%6.1f img/s): waiting for the batch %6.2f, enqueueing the step %6.2f ms; on the device: step %6.2f ms, gap to the next %5.2f ms"
              % (tag, dt / n * 1e3, B * n / dt, t_wait / n * 1e3, t_enq / n * 1e3, busy, gap), flush=True)

    resident = ds.collate_fn([ds[i] for i in range(B)])
    resident = {"img": resident["img"].cuda(), "label": {k: v.cuda() for k, v in resident["label"].items()},
                "img_metas": resident["img_metas"], "calib": resident["calib"]}
    loop([resident] * 6, "warm-up")
    loop([resident] * steps, "resident batch")
    host = ds.collate_fn([ds[i] for i in range(B)])
    pinned = [dict(host, img=host["img"].clone().pin_memory(), label={k: v.clone().pin_memory() for k, v in host["label"].items()})
              for _ in range(3)]
    loop(DevicePrefetcher([pinned[i % 3] for i in range(steps)], "cuda:0", m), "pinned host batches + prefetcher (no workers)")
    loop(DevicePrefetcher([pinned[i % 3] for i in range(steps)], "cuda:0", m), "pinned host batches + prefetcher (no workers)")
    ring = RingLoader(ds, B, workers, shuffle=True, collate_fn=ds.collate_fn)
    for rep in range(3):
        loop(DevicePrefetcher(ring, "cuda:0", m), "RingLoader + prefetcher (%d w), epoch %d" % (workers, rep))
    ring.close()
    del ring


if __name__ == "__main__":
    main()

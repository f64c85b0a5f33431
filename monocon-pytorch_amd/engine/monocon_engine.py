"""MonoconEngine: the reference's train / evaluate loop (engine/monocon_engine.py:23-194) on top of
the MI355X hot path.  ``train_one_epoch`` is the reference's step sequence -- zero_grad, H2D,
``self.model(data_dict)``, ``sum(losses).backward()``, clip, ``optimizer.step()``,
``scheduler.step()``, periodic logging -- with the clip folded into the fused HIP AdamW and, under
``torch.distributed.run``, a sharded sampler + the in-backward RCCL gradient all-reduce.

Data-parallel policy (N ranks): ``DATA.BATCH_SIZE`` is the PER-RANK batch (global batch = N x BATCH_SIZE),
the sampler shards the training set so an epoch has 1/N as many steps and the one-cycle schedule's
``total_steps`` shrinks with it; ``SOLVER.OPTIM.LR`` is used exactly as configured -- nothing is scaled
automatically, scale it in the config if the larger global batch calls for it.  All ranks start from rank 0's
weights (``hipmonocon.dist.sync_module_state``); rank 0 alone evaluates and writes checkpoints.
"""
import os
from typing import Dict, List

import torch
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from engine.base_engine import BaseEngine
from model import MonoConDetector
from solver import AdamW, CyclicScheduler
from utils.decorators import decorator_timer
from hipmonocon.feed import WORKER_CONTEXT, DeferredScalars, DevicePrefetcher, RingLoader, prepare_worker_context
from utils.engine_utils import move_data_device, progress_to_string_bar, reduce_loss_dict, tprint


class _SeedWorker:
    """worker_init_fn (module level: the workers of a fork server get it pickled).  torch seeds numpy per WORKER from a base seed
    that is the same on every rank (replicas start from rank 0's seed): offset it per rank, or all ranks would draw the same
    augmentation decisions for their shards"""

    def __init__(self, rank):
        self.rank = rank

    def __call__(self, worker_id):
        import numpy as np
        np.random.seed((torch.initial_seed() + 1000003 * (self.rank + 1)) % (2 ** 32))


class MonoconEngine(BaseEngine):
    def __init__(self, cfg, **kwargs):
        super().__init__(cfg, **kwargs)

    def build_model(self):
        detector = MonoConDetector(num_dla_layers=self.cfg.MODEL.BACKBONE.NUM_LAYERS,
                                   pretrained_backbone=self.cfg.MODEL.BACKBONE.IMAGENET_PRETRAINED)
        return detector.to(self.current_device)

    def build_solver(self):
        assert self.model is not None and self.train_loader is not None
        clip = self.cfg.SOLVER.CLIP_GRAD
        if clip.ENABLE and float(clip.NORM_TYPE) != 2.0:
            raise NotImplementedError("the fused optimizer clips the global L2 norm (reference default)")
        optimizer = AdamW(self.model.parameters(), lr=self.cfg.SOLVER.OPTIM.LR,
                          weight_decay=self.cfg.SOLVER.OPTIM.WEIGHT_DECAY, betas=(0.95, 0.99),
                          max_grad_norm=float(clip.MAX_NORM) if clip.ENABLE else None)
        scheduler = None
        if self.cfg.SOLVER.SCHEDULER.ENABLE:
            total_steps = len(self.train_loader) * self.cfg.SOLVER.OPTIM.NUM_EPOCHS
            scheduler = CyclicScheduler(optimizer, total_steps=total_steps, target_lr_ratio=(10, 1E-04),
                                        target_momentum_ratio=(0.85 / 0.95, 1.0), period_up=0.4)
        return optimizer, scheduler

    def build_loader(self, is_train: bool = True):
        if str(self.cfg.DATA.ROOT).startswith('synthetic'):
            from dataset.synthetic_dataset import SyntheticMonoConDataset
            n = int(self.cfg.DATA.get('SYNTHETIC_LENGTH', 64))
            hw = self.cfg.DATA.get('SYNTHETIC_HW', (384, 1280))
            dataset = SyntheticMonoConDataset(length=n if is_train else max(n // 4, 1), height=int(hw[0]), width=int(hw[1]),
                                              max_objs=self.cfg.MODEL.HEAD.MAX_OBJS, seed=1 if is_train else 2)
        else:
            # the KITTI file dataset (dataset/monocon_dataset.py: PIL decode; the 'train' split runs the reference's random
            # augmentations, transforms/augmentations.py, every other split the deterministic list)
            from dataset.monocon_dataset import MonoConDataset
            # DATA.DEVICE_AUGMENT (default on beside a device): the samples carry the decoded uint8 frame + 24 parameters and the
            # detector forms the float32 frame on the device (mc_preprocess_augmented, bit-identical to the host transforms);
            # a worker then spends its time on the PNG and the labels, not on ~80 ms of float32 colour arithmetic per frame
            dataset = MonoConDataset(base_root=self.cfg.DATA.ROOT,
                                     split=self.cfg.DATA.TRAIN_SPLIT if is_train else self.cfg.DATA.TEST_SPLIT,
                                     max_objs=self.cfg.MODEL.HEAD.MAX_OBJS,
                                     filter_configs={k.lower(): v for k, v in dict(self.cfg.DATA.FILTER).items()},
                                     device_image=torch.cuda.is_available() and bool(self.cfg.DATA.get('DEVICE_AUGMENT', True)))
        sampler = None
        if self.world > 1 and is_train:
            sampler = DistributedSampler(dataset, num_replicas=self.world, rank=self.rank, shuffle=True, drop_last=True)
        seed_worker = _SeedWorker(self.rank)
        if is_train and self.cfg.DATA.NUM_WORKERS == 0 and self.world > 1:
            seed_worker(0)
        shuffle, drop_last = (is_train and sampler is None), (self.world > 1 and is_train)
        init_fn = seed_worker if (is_train and self.cfg.DATA.NUM_WORKERS > 0) else None
        loader = None
        if self.cfg.DATA.NUM_WORKERS > 0 and torch.cuda.is_available() and bool(self.cfg.DATA.get('RING_LOADER', True)):
            # worker processes + a HIP device: the frames go through a shared, page-locked ring of batch slots instead of the
            # workers' queues (hipmonocon/feed.py: a DataLoader delivers a 32-frame float32 batch every 75-105 ms, the ring
            # every ~10).  Same batches in the same order; DATA.RING_LOADER: False is the reference's DataLoader.
            try:
                loader = RingLoader(dataset, self.cfg.DATA.BATCH_SIZE, self.cfg.DATA.NUM_WORKERS, shuffle=shuffle, sampler=sampler,
                                    drop_last=drop_last, collate_fn=dataset.collate_fn, worker_init_fn=init_fn)
            except (RuntimeError, OSError) as e:        # e.g. /dev/shm too small for the ring
                self._say("RingLoader unavailable (%s): using torch's DataLoader." % (e,))
        if loader is None:
            # pin_memory: a loader thread page-locks every batch, so that DevicePrefetcher's uploads are asynchronous -- and
            # beside page-locked memory the workers must not be forks of this process (hipmonocon/feed.py, RingLoader)
            beside_device = torch.cuda.is_available() and self.cfg.DATA.NUM_WORKERS > 0
            if beside_device:
                prepare_worker_context()
            loader = DataLoader(dataset, batch_size=self.cfg.DATA.BATCH_SIZE, num_workers=self.cfg.DATA.NUM_WORKERS,
                                shuffle=shuffle, sampler=sampler, collate_fn=dataset.collate_fn, drop_last=drop_last,
                                pin_memory=torch.cuda.is_available(), worker_init_fn=init_fn,
                                multiprocessing_context=WORKER_CONTEXT if beside_device else None,
                                persistent_workers=beside_device)
        return dataset, loader

    @decorator_timer
    def train_one_epoch(self) -> float:
        epoch_losses = []
        if hasattr(getattr(self.train_loader, 'sampler', None), 'set_epoch'):      # DistributedSampler (RingLoader shows it guarded)
            self.train_loader.sampler.set_epoch(self.epochs)
        # the batches arrive on the device one step ahead (copy stream, labels checked on the host) and the loss of a step is
        # read back while the NEXT one runs: nothing between two log lines drains the stream (hipmonocon/feed.py)
        losses = DeferredScalars()

        def collect(keep):
            got = losses.ready(keep)
            epoch_losses.extend(got)
            self.entire_losses.extend(got)

        # MONOCON_HIP_SYNC_LOOP=1: the reference's loop as written (upload on the compute stream, loss read inside the step)
        sync_loop = os.environ.get("MONOCON_HIP_SYNC_LOOP", "0") == "1"
        feed = self.train_loader if sync_loop else DevicePrefetcher(self.train_loader, self.current_device, self.model)
        for batch_idx, data_dict in enumerate(feed):
            self.optimizer.zero_grad()
            data_dict = move_data_device(data_dict, self.current_device)      # (a no-op for what the prefetcher uploaded)
            _, loss_dict = self.model(data_dict)
            total_loss = reduce_loss_dict(loss_dict)
            total_loss.backward()                       # HIP backward + (N > 1) gradient all-reduce
            losses.push(total_loss)
            self.optimizer.step()                       # fused clip (SOLVER.CLIP_GRAD) + AdamW
            if self.scheduler is not None:
                self.scheduler.step()
            logging = self.global_iters % self.log_period == 0 and self.is_main
            collect(0 if (logging or sync_loop) else 1)
            if logging:
                bar = progress_to_string_bar(batch_idx + 1, len(self.train_loader), bins=20)
                recent = sum(self.entire_losses[-100:]) / len(self.entire_losses[-100:])
                print("| Progress %s | LR %.6f | Loss %8.4f (%8.4f) |" % (bar, self.current_lr, self.entire_losses[-1], recent))
                self._update_dict_to_writer(loss_dict, tag='loss')
            self._iter_update()
        collect(0)
        self._epoch_update()
        return sum(epoch_losses) / max(len(epoch_losses), 1)

    @torch.no_grad()
    def evaluate(self) -> Dict[str, float]:
        was_training = self.model.training
        if was_training:
            self.model.eval()
            self._say("Model is converted to eval mode.")
        container = {'img_bbox': [], 'img_bbox2d': []}
        synthetic = str(self.cfg.DATA.ROOT).startswith('synthetic')
        for test_data in DevicePrefetcher(self.test_loader, self.current_device):
            test_data = move_data_device(test_data, self.current_device)
            if synthetic:            # KITTI text export needs dataset metadata the synthetic set does not have
                res = self.model.batch_eval(test_data, get_vis_format=True)
                container['img_bbox'].extend(r['img_bbox'] for r in res)
                container['img_bbox2d'].extend(r['img_bbox2d'] for r in res)
            else:
                res = self.model.batch_eval(test_data)
                for field in ('img_bbox', 'img_bbox2d'):
                    container[field].extend(res[field])
        eval_dict = self.test_dataset.evaluate(container, eval_classes=['Pedestrian', 'Cyclist', 'Car'], verbose=self.is_main)
        if was_training:
            self.model.train()
            self._say("Model is converted to train mode.")
        return eval_dict

    @torch.no_grad()
    def visualize(self, output_dir: str, draw_items: List[str] = ('bev', '2d', '3d')):
        raise NotImplementedError("drawing (cv2) is outside the MI355X hot path; use the reference's utils/visualizer.py "
                                  "on the output of model.batch_eval(data, get_vis_format=True)")

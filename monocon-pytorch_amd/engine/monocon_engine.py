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
from utils.engine_utils import move_data_device, progress_to_string_bar, reduce_loss_dict, tprint


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
            dataset = MonoConDataset(base_root=self.cfg.DATA.ROOT,
                                     split=self.cfg.DATA.TRAIN_SPLIT if is_train else self.cfg.DATA.TEST_SPLIT,
                                     max_objs=self.cfg.MODEL.HEAD.MAX_OBJS,
                                     filter_configs={k.lower(): v for k, v in dict(self.cfg.DATA.FILTER).items()})
        sampler = None
        if self.world > 1 and is_train:
            sampler = DistributedSampler(dataset, num_replicas=self.world, rank=self.rank, shuffle=True, drop_last=True)
        rank = self.rank

        def seed_worker(worker_id):
            # torch seeds numpy per WORKER from a base seed that is the same on every rank (replicas start from rank 0's
            # seed): offset it per rank, or all ranks would draw the same augmentation decisions for their shards
            import numpy as np
            np.random.seed((torch.initial_seed() + 1000003 * (rank + 1)) % (2 ** 32))
        if is_train and self.cfg.DATA.NUM_WORKERS == 0 and self.world > 1:
            seed_worker(0)
        loader = DataLoader(dataset, batch_size=self.cfg.DATA.BATCH_SIZE, num_workers=self.cfg.DATA.NUM_WORKERS,
                            shuffle=(is_train and sampler is None), sampler=sampler, collate_fn=dataset.collate_fn,
                            drop_last=(self.world > 1 and is_train),
                            worker_init_fn=seed_worker if (is_train and self.cfg.DATA.NUM_WORKERS > 0) else None)
        return dataset, loader

    @decorator_timer
    def train_one_epoch(self) -> float:
        epoch_losses = []
        if isinstance(getattr(self.train_loader, 'sampler', None), DistributedSampler):
            self.train_loader.sampler.set_epoch(self.epochs)
        for batch_idx, data_dict in enumerate(self.train_loader):
            self.optimizer.zero_grad()
            data_dict = move_data_device(data_dict, self.current_device)
            _, loss_dict = self.model(data_dict)
            total_loss = reduce_loss_dict(loss_dict)
            total_loss.backward()                       # HIP backward + (N > 1) gradient all-reduce
            step_loss = total_loss.detach().item()
            epoch_losses.append(step_loss)
            self.entire_losses.append(step_loss)
            self.optimizer.step()                       # fused clip (SOLVER.CLIP_GRAD) + AdamW
            if self.scheduler is not None:
                self.scheduler.step()
            if self.global_iters % self.log_period == 0 and self.is_main:
                bar = progress_to_string_bar(batch_idx + 1, len(self.train_loader), bins=20)
                recent = sum(self.entire_losses[-100:]) / len(self.entire_losses[-100:])
                print("| Progress %s | LR %.6f | Loss %8.4f (%8.4f) |" % (bar, self.current_lr, step_loss, recent))
                self._update_dict_to_writer(loss_dict, tag='loss')
            self._iter_update()
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
        for test_data in self.test_loader:
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

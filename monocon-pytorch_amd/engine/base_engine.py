"""Engine base: epoch loop, checkpoints, logging.  Same public surface as reference
engine/base_engine.py:18-278 (``BaseEngine(cfg, auto_resume=True, is_test=False)``, ``train``,
``save_checkpoint``, ``load_checkpoint``, ``current_lr``, ``current_device``), made rank-aware for
one-process-per-GPU data parallelism (rank 0 prints / writes), TensorBoard optional.
Checkpoints keep the reference's layout: ``{'engine_attrs': {...}, 'state_dict': {'model',
'optimizer', 'scheduler'}}`` in ``checkpoints/epoch_NNN[_postfix].pth``; loading uses
``weights_only=False`` because the reference pickles engine attributes (base_engine.py:171-174).
"""
import glob
import os
from datetime import datetime, timedelta
from typing import Dict, Union

import numpy as np
import torch

from config.cfgnode import CfgNode
from hipmonocon import dist as hdist
from utils.decorators import decorator_timer
from utils.engine_utils import OpaqueReferenceObject, count_trainable_params, export_cfg, load_cfg, load_checkpoint_file, tprint

try:
    from torch.utils.tensorboard import SummaryWriter
except Exception:                       # tensorboard is not installed in the MI355X image
    SummaryWriter = None


class _NullWriter:
    def add_scalar(self, *a, **k):
        pass


class BaseEngine:
    def __init__(self, cfg: Union[str, CfgNode], auto_resume: bool = True, is_test: bool = False):
        if isinstance(cfg, str):
            cfg = load_cfg(cfg_file=cfg)
        elif not isinstance(cfg, CfgNode):
            raise Exception("Argument 'cfg' must be either a string or a CfgNode.")
        self.cfg = cfg
        self.world, self.rank, self.local_rank = hdist.init_from_env()
        self.version, self.description = cfg.VERSION, cfg.DESCRIPTION
        self.epochs = 1
        self.target_epochs = cfg.SOLVER.OPTIM.NUM_EPOCHS
        assert self.epochs <= self.target_epochs
        self.global_iters = 1
        self.log_period, self.val_period = cfg.PERIOD.LOG_PERIOD, cfg.PERIOD.EVAL_PERIOD

        self.train_dataset, self.train_loader = self.build_loader(is_train=True) if not is_test else (None, None)
        self.test_dataset, self.test_loader = self.build_loader(is_train=False)
        self.model = self.build_model()
        # data parallelism averages gradients only: every replica starts from rank 0's weights and buffers
        hdist.sync_module_state(self.model)
        self.optimizer, self.scheduler = self.build_solver() if not is_test else (None, None)

        self.root = cfg.OUTPUT_DIR
        self.writer_dir = os.path.join(self.root, 'tf_logs')
        self.weight_dir = os.path.join(self.root, 'checkpoints')
        self.writer = _NullWriter()
        if not is_test:
            resumed = False
            if os.path.isdir(self.weight_dir) and auto_resume:
                pth = sorted(glob.glob(os.path.join(self.weight_dir, '*.pth')))
                if pth:
                    self.load_checkpoint(pth[-1])
                    resumed = True
                    self._say("Existing checkpoint '%s' is found and loaded automatically." % pth[-1])
            if not resumed and self.is_main:
                for d in (self.writer_dir, self.weight_dir):
                    os.makedirs(d, exist_ok=True)
            if self.is_main and SummaryWriter is not None:
                self.writer = SummaryWriter(self.writer_dir)
        self.epoch_times, self.entire_losses = [], []

    # ------------------------------------------------------------------ hooks
    def build_model(self):
        raise NotImplementedError

    def build_solver(self):
        raise NotImplementedError

    def build_loader(self, is_train: bool):
        raise NotImplementedError

    @property
    def is_main(self) -> bool:
        return self.rank == 0

    def _say(self, msg, indent=False):
        if self.is_main:
            tprint(msg, indent=indent)

    # ------------------------------------------------------------------ train loop
    def train(self, resume_from: str = None) -> None:
        assert torch.cuda.is_available(), "CUDA is not available."
        assert self.epochs < self.target_epochs or self.target_epochs == 1
        if self.is_main:
            self._print_engine_info()
            export_cfg(self.cfg, os.path.join(self.root, 'config.yaml'))
        if resume_from is not None:
            self.load_checkpoint(resume_from)
            self._say("Training resumes from '%s'. (Start Epoch: %d)" % (resume_from, self.epochs))
        self._say("Training will be proceeded from epoch %d to epoch %d." % (self.epochs, self.target_epochs))
        for epoch in range(self.epochs, self.target_epochs + 1):
            if self.is_main:
                print((" Epoch %3d / %3d " % (self.epochs, self.target_epochs)).center(90, "="))
            avg_loss, elapsed = self.train_one_epoch()
            self.epoch_times.append(elapsed)
            if self.is_main:
                ti = self._get_time_info()
                print("\n- Average Loss: %.3f\n- Epoch Time: %s\n- Remain Time: %s\n- Estimated End-Time: %s"
                      % (avg_loss, ti['epoch_time'], ti['remain_time'], ti['end_time']))
            if self.val_period > 0 and epoch % self.val_period == 0:
                self.model.eval()
                self._say("Evaluating on Epoch %d..." % epoch, indent=True)
                # rank 0 evaluates (result export / the AP evaluator write files); the others wait for its verdict.  A
                # failure on rank 0 (missing evaluator dependency, IO error) must not leave them parked in a collective
                # until the watchdog kills the job: the success flag is all-reduced, and every rank raises together
                eval_error = None
                if self.is_main:
                    try:
                        eval_dict = self.evaluate()
                        self._update_dict_to_writer(eval_dict, tag='eval')
                    except Exception as e:      # noqa: BLE001
                        eval_error = e
                if hdist.is_distributed():
                    ok = hdist.all_ranks_ok(eval_error is None, torch.device("cuda", torch.cuda.current_device()))
                    if not ok and eval_error is None:
                        raise RuntimeError("evaluation failed on rank 0 (see its log)")
                if eval_error is not None:
                    raise eval_error
                self.model.train()
                self.save_checkpoint(post_fix=None)
        self.save_checkpoint(post_fix='final')

    @decorator_timer
    def train_one_epoch(self):
        raise NotImplementedError

    @torch.no_grad()
    def evaluate(self):
        raise NotImplementedError

    # ------------------------------------------------------------------ checkpoints
    _ATTR_EXCEPT = ('cfg', 'writer', 'train_loader', 'test_loader', 'train_dataset', 'test_dataset', 'model', 'optimizer',
                    'scheduler')

    def save_checkpoint(self, post_fix: str = None, save_after_update: bool = True, verbose: bool = True) -> None:
        if not self.is_main:
            return
        ep = self.epochs - (1 if save_after_update else 0)
        name = 'epoch_%03d.pth' % ep if post_fix is None else 'epoch_%03d_%s.pth' % (ep, post_fix)
        path = os.path.join(self.weight_dir, name)
        os.makedirs(self.weight_dir, exist_ok=True)
        attrs = {k: v for k, v in self.__dict__.items() if k not in self._ATTR_EXCEPT and not callable(v)}
        torch.save({'engine_attrs': attrs,
                    'state_dict': {'model': self.model.state_dict() if self.model is not None else None,
                                   'optimizer': self.optimizer.state_dict() if self.optimizer is not None else None,
                                   'scheduler': self.scheduler.state_dict() if self.scheduler is not None else None}}, path)
        if verbose:
            tprint("Checkpoint is saved to '%s'." % path)

    def load_checkpoint(self, ckpt_file: str, verbose: bool = False) -> None:
        d = load_checkpoint_file(ckpt_file)       # (tolerant of the reference's pickled dataset / transform objects)
        # Only plain bookkeeping attributes are adopted.  The reference's save (engine/base_engine.py:171, a missing comma) lets
        # its whole pickled `test_dataset` into this dict: objects of the live run (`_ATTR_EXCEPT`: loaders, datasets, model,
        # optimizer ...) are never replaced from a file, and stand-ins for classes this tree does not have are dropped.
        for k, v in d['engine_attrs'].items():
            if k in ('world', 'rank', 'local_rank') or k in BaseEngine._ATTR_EXCEPT or isinstance(v, OpaqueReferenceObject):
                continue
            setattr(self, k, v)
        sd = d['state_dict']
        if sd['model'] is not None and self.model is not None:
            self.model.load_state_dict(sd['model'])
            hdist.sync_module_state(self.model)
        if sd['optimizer'] is not None and self.optimizer is not None:
            self.optimizer.load_state_dict(sd['optimizer'])
        if sd['scheduler'] is not None and self.scheduler is not None:
            self.scheduler.load_state_dict(sd['scheduler'])
        if verbose:
            self._say("Checkpoint is loaded from '%s'." % ckpt_file)

    # ------------------------------------------------------------------ bookkeeping
    def _epoch_update(self):
        self.epochs += 1

    def _iter_update(self):
        self.global_iters += 1

    def _update_dict_to_writer(self, data: Dict[str, Union[torch.Tensor, float]], tag: str):
        for k, v in data.items():
            self.writer.add_scalar('%s/%s' % (tag, k), scalar_value=v if isinstance(v, float) else v.detach().item(),
                                   global_step=self.global_iters)

    def _get_time_info(self) -> Dict[str, str]:
        avg = float(np.mean(self.epoch_times))
        remain = (self.target_epochs - (self.epochs - 1)) * avg
        return {'epoch_time': str(timedelta(seconds=self.epoch_times[-1]))[:-7] or '0:00:00',
                'remain_time': str(timedelta(seconds=remain))[:-7] or '0:00:00',
                'end_time': str(datetime.now() + timedelta(seconds=remain))[:-7]}

    def _print_engine_info(self):
        print("\n==================== Engine Info ====================")
        print("- Root: %s\n- Version: %s\n- Description: %s" % (self.root, self.version, self.description))
        print("\n- Seed: %s\n- Device: GPU %d (%s) x %d rank(s)" % (self.cfg.SEED, self.local_rank,
                                                                  torch.cuda.get_device_name(self.local_rank), self.world))
        print("\n- Model: %s (# Params: %d)" % (self.model.__class__.__name__, count_trainable_params(self.model)))
        print("- Optimizer: %s\n- Scheduler: %s\n" % (self.optimizer.__class__.__name__, self.scheduler.__class__.__name__))
        print("- Epoch Progress: %d/%d\n- # Train Samples: %d\n- # Test Samples: %d"
              % (self.epochs, self.target_epochs, len(self.train_dataset), len(self.test_dataset)))
        print("=====================================================\n")

    @property
    def current_lr(self) -> float:
        return self.optimizer.param_groups[0]['lr']

    @property
    def current_device(self) -> torch.device:
        return torch.device('cuda:%d' % (self.local_rank if self.world > 1 else self.cfg.GPU_ID))

"""Training driver for the MI355X hot path: the counterpart of the reference's ``src/02_train.py``
(Trainer: setup_model / training_loop / train_epoch / validation_epoch) built on the fused
``TrainStep`` instead of ``nn.DataParallel`` + autograd + ``torch.optim``.

What is kept from the reference so that its experiment directories stay interchangeable:
  * experiment parameters are the reference's ``exp_data`` dictionary (``training.learning_rate``,
    ``learning_rate_factor``, ``patience``, ``momentum``, ``optimizer``, ``nesterov``, ``scheduler``,
    ``num_epochs``, ``save_frequency``; ``model.model_name``; ``dataset.image_size``),
    ``lib/model_setup.py:100-158``;
  * epoch structure ``02_train.py:157-175``: validation epoch (first fifth of the loader,
    ``:247-249``) -> training epoch -> scheduler step (``plateau`` steps on the validation loss
    with ``mode="max"`` exactly as ``model_setup.py:141-148`` configures it; ``step`` = StepLR);
  * ``training_logs.json`` layout of ``lib/utils.py:127-156,172-205``;
  * checkpoints ``<exp>/models/checkpoint_epoch_{n|final}.pth`` holding ``epoch``,
    ``model_state_dict`` (keys carry the ``module.`` prefix of the DataParallel wrapper,
    ``02_train.py:166`` / ``model_setup.py:200-205``), ``optimizer_state_dict`` in
    ``torch.optim`` layout (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` or
    ``momentum_buffer``) and ``scheduler_state_dict``.
Data loading, augmentation and TensorBoard stay the caller's: the
loaders are any iterables of ``(imgs, target, target_weight, metadata)`` like the reference's.
"""
from __future__ import annotations

import json
import os
import time
from typing import Iterable, Optional

import numpy as np
import torch

from .hrnet import PoseHighResolutionNet, norm_device
from .inference import forward_pass
from .loss import PersonMSELoss, perceptual_affine
from .pose_parsing import accuracy
from .train_step import TrainStep

_PREFIX = "module."


def _timestamp() -> str:
    return time.strftime("%Y-%m-%d_%H-%M-%S")


def create_train_logs(exp_path: str) -> dict:
    logs = {"last_modified": _timestamp(), "iterations": 0,
            "loss": {"training": [], "validation": []}, "accuracy": {"training": [], "validation": []}}
    with open(os.path.join(exp_path, "training_logs.json"), "w") as f:
        json.dump(logs, f)
    return logs


def load_train_logs(exp_path: str) -> dict:
    with open(os.path.join(exp_path, "training_logs.json")) as f:
        return json.load(f)


def update_train_logs(exp_path: str, logs: dict, iterations: int, train_loss, valid_loss, train_acc, valid_acc) -> None:
    logs["last_modified"] = _timestamp()
    logs["iterations"] = int(iterations)
    logs["loss"]["training"].append(float(train_loss))
    logs["loss"]["validation"].append(float(valid_loss))
    logs["accuracy"]["training"].append(float(train_acc))
    logs["accuracy"]["validation"].append(float(valid_acc))
    with open(os.path.join(exp_path, "training_logs.json"), "w") as f:
        json.dump(logs, f)


class Trainer:
    def __init__(self, exp_path: str, exp_data: dict, train_loader: Iterable, valid_loader: Iterable, batch_size: int,
                 arch: str = "w32", compute_dtype: str = "mixed", checkpoint: Optional[str] = None,
                 resume_training: bool = False, process_group=None, device="cuda", acc_every: int = 1):
        self.exp_path, self.exp_data = exp_path, exp_data
        self.train_loader, self.valid_loader = train_loader, valid_loader
        self.batch_size = int(batch_size)
        self.arch, self.compute_dtype = arch, compute_dtype
        self.checkpoint, self.resume_training = checkpoint, resume_training
        self.pg, self.device = process_group, norm_device(device)
        self.rank, self.world = 0, 1
        if process_group is not None:
            import torch.distributed as dist
            self.rank, self.world = dist.get_rank(process_group), dist.get_world_size(process_group)
        self.params = None   # optional argparse-like object with .use_perceptual_loss (02_train.py:208-212)
        self.acc_every = max(1, int(acc_every))   # PCK needs the heatmaps on the host (metrics.accuracy): every n-th batch
        tr = exp_data["training"]
        self.num_epochs = int(tr["num_epochs"])
        self.save_frequency = int(tr["save_frequency"])
        self.scheduler_type = tr.get("scheduler", "plateau")
        self.train_loss = self.valid_loss = 1e18
        self.train_acc = self.valid_acc = 0.0
        self.iterations, self.cur_epoch = 0, 0
        h, w = exp_data["dataset"]["image_size"]
        self.image_size = (int(h), int(w))
        if self.rank == 0:
            os.makedirs(os.path.join(exp_path, "models"), exist_ok=True)

    # ------------------------------------------------------------------ model / optimiser / scheduler
    def setup_model(self):
        tr = self.exp_data["training"]
        self.model = PoseHighResolutionNet(self.arch, self.compute_dtype, is_train=False).to(self.device)
        self.model_name = self.exp_data.get("model", {}).get("model_name", "HRNet")
        opt = "adam" if tr["optimizer"] == "adam" else "sgd"
        self.ts = TrainStep(self.model, self.batch_size, *self.image_size, optimizer=opt, lr=float(tr["learning_rate"]),
                            momentum=float(tr.get("momentum", 0.9)), nesterov=bool(tr.get("nesterov", False)),
                            weight_decay=0.0 if opt == "adam" else 0.0005, process_group=self.pg, device=self.device)
        self.loss_function = PersonMSELoss()
        # the learning-rate schedule runs on a one-parameter shadow optimiser: same torch scheduler
        # classes and state_dict as the reference, the value is pushed into the fused step
        self._shadow = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=float(tr["learning_rate"]))
        if self.scheduler_type == "plateau":
            self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(
                self._shadow, patience=int(tr["patience"]), factor=float(tr["learning_rate_factor"]), min_lr=1e-8, mode="max")
        elif self.scheduler_type == "step":
            self.scheduler = torch.optim.lr_scheduler.StepLR(self._shadow, gamma=float(tr["learning_rate_factor"]),
                                                             step_size=int(tr["patience"]))
        else:
            self.scheduler = None
        if self.checkpoint is not None:
            self.load_checkpoint(self.checkpoint, only_model=not self.resume_training)
        self._broadcast_state()

    # ------------------------------------------------------------------ one process per GPU
    def _broadcast_state(self):
        """Replicas must start from rank 0's weights / BatchNorm buffers / optimiser state (DataParallel
        re-broadcasts module state every step, 02_train.py:109; here once is enough because every rank then
        applies the same all-reduced gradients)."""
        if self.world == 1:
            return
        import torch.distributed as dist
        st, ts = self.ts.store, self.ts
        for t in (st.master, st.bufs, st.nbt, ts.m, ts.v, ts.step_count, ts.hyper):
            if t is not None:
                dist.broadcast(t, src=dist.get_global_rank(self.pg, 0), group=self.pg)
        self.ts.invalidate_weights()

    def _mean_over_ranks(self, *vals):
        """Loaders are sharded, so per-rank epoch statistics differ: average them so that every rank logs
        and -- more importantly -- feeds ReduceLROnPlateau the same number (replicas would otherwise drift
        apart in learning rate)."""
        if self.world == 1:
            return vals
        import torch.distributed as dist
        t = torch.tensor(vals, dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
        return tuple((t / self.world).tolist())

    def _common_steps(self, loader) -> Optional[int]:
        """Steps per epoch every rank can take: the MINIMUM of the per-rank loader lengths.  Each step holds one
        all-reduce per gradient bucket, so a rank that ran out of batches early would leave the others waiting in a
        collective for ever (uneven shards: no drop_last, a last partial shard).  None with one process."""
        if self.world == 1:
            return None
        import torch.distributed as dist
        try:
            n = len(loader)
        except TypeError:
            n = -1
        t = torch.tensor([n, -n], dtype=torch.int64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.pg)
        lo = -int(t[1].item())
        if lo < 0:
            raise RuntimeError("Trainer with a process group: every rank's loader needs __len__ (a sharded sampler) so that the "
                               "ranks can agree on the number of steps per epoch")
        return lo

    def _barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier(group=self.pg)

    @property
    def lr(self) -> float:
        return float(self._shadow.param_groups[0]["lr"])

    # ------------------------------------------------------------------ checkpoints (reference layout)
    def _optimizer_state_dict(self) -> dict:
        st, ts = self.ts.store, self.ts
        names = [k for k, _ in st.reg.params]
        step = int(ts.step_count.item())
        step = step if step >= 0 else -step - 1   # (negative: the last step was skipped, see TrainStep.check_forward_range)
        state = {}
        for i, (k, shape) in enumerate(st.reg.params):
            a = st.param_off[k]
            n = int(np.prod(shape)) if shape else 1
            if ts.kind == "adam":
                state[i] = {"step": torch.tensor(float(step)), "exp_avg": ts.m[a:a + n].view(shape).cpu().clone(),
                            "exp_avg_sq": ts.v[a:a + n].view(shape).cpu().clone()}
            else:
                state[i] = {"momentum_buffer": ts.m[a:a + n].view(shape).cpu().clone()}
        hyper = ts.hyper.tolist()
        group = {"lr": self.lr, "params": list(range(len(names)))}
        if ts.kind == "adam":
            group.update(betas=(hyper[1], hyper[2]), eps=hyper[3], weight_decay=hyper[4], amsgrad=False)
        else:
            group.update(momentum=hyper[5], dampening=0, weight_decay=hyper[4], nesterov=bool(hyper[6]))
        return {"state": state, "param_groups": [group], "stl_step": step}

    def _load_optimizer_state_dict(self, sd: dict) -> None:
        st, ts = self.ts.store, self.ts
        step, have_momentum = 0, False
        for i, (k, shape) in enumerate(st.reg.params):
            s = sd["state"].get(i)
            if s is None:
                continue
            a = st.param_off[k]
            n = int(np.prod(shape)) if shape else 1
            if ts.kind == "adam":
                ts.m[a:a + n].copy_(s["exp_avg"].reshape(-1))
                ts.v[a:a + n].copy_(s["exp_avg_sq"].reshape(-1))
                step = int(float(s["step"]))
            else:
                ts.m[a:a + n].copy_(s["momentum_buffer"].reshape(-1))
                have_momentum = True
        if have_momentum:
            # torch.optim.SGD keeps no step counter; the fused kernel treats step <= 1 as "no momentum
            # buffer yet" (b = g).  With a buffer loaded the next step must be a continuation, like torch's.
            step = max(step, int(sd.get("stl_step", 1)), 1)
        ts.step_count.fill_(step)
        lr = float(sd["param_groups"][0]["lr"])
        self._shadow.param_groups[0]["lr"] = lr
        ts.set_lr(lr)

    def save_checkpoint(self, epoch: int, finished: bool = False) -> str:
        """rank 0 writes (every rank holds identical weights); the others wait at the barrier."""
        name = "checkpoint_epoch_final.pth" if finished else f"checkpoint_epoch_{epoch}.pth"
        path = os.path.join(self.exp_path, "models", name)
        if self.rank != 0:
            self._barrier()
            return path
        msd = {_PREFIX + k: v.detach().cpu() for k, v in self.model.state_dict().items()}
        torch.save({"epoch": epoch, "model_state_dict": msd, "optimizer_state_dict": self._optimizer_state_dict(),
                    "scheduler_state_dict": self.scheduler.state_dict() if self.scheduler is not None else {}}, path)
        self._barrier()
        return path

    def load_checkpoint(self, path: str, only_model: bool = False) -> None:
        ck = torch.load(path, map_location="cpu", weights_only=False)
        sd = ck["model_state_dict"] if "model_state_dict" in ck else ck
        sd = {(k[len(_PREFIX):] if k.startswith(_PREFIX) else k): v for k, v in sd.items()}
        self.model.load_state_dict(sd)   # in place: the parameters are views of the flat master buffer
        self.ts.invalidate_weights()
        if only_model:
            return
        self._load_optimizer_state_dict(ck["optimizer_state_dict"])
        if self.scheduler is not None and ck.get("scheduler_state_dict"):
            self.scheduler.load_state_dict(ck["scheduler_state_dict"])
        self.cur_epoch = int(ck["epoch"])

    # ------------------------------------------------------------------ epochs
    def training_loop(self):
        if self.rank != 0:
            self.training_logs = None
        elif self.checkpoint is None or not self.resume_training:
            self.training_logs = create_train_logs(self.exp_path)
        else:
            self.training_logs = load_train_logs(self.exp_path)
        for epoch in range(self.cur_epoch, self.num_epochs):
            self.validation_epoch(epoch)
            self.train_epoch(epoch)
            self.train_loss, self.valid_loss, self.train_acc, self.valid_acc = self._mean_over_ranks(
                self.train_loss, self.valid_loss, self.train_acc, self.valid_acc)
            if self.scheduler_type == "plateau":
                self.scheduler.step(self.valid_loss)
            elif self.scheduler_type == "step":
                self.scheduler.step()
            self.ts.set_lr(self.lr)
            if self.rank == 0:
                update_train_logs(self.exp_path, self.training_logs, self.iterations, self.train_loss, self.valid_loss,
                                  self.train_acc, self.valid_acc)
            if epoch % self.save_frequency == 0:
                self.save_checkpoint(epoch)
        self.save_checkpoint(self.num_epochs, finished=True)

    def train_epoch(self, epoch: int):
        self.model.train()
        losses, accs = [], []
        steps = self._common_steps(self.train_loader)
        for i, (imgs, target, target_weight, meta) in enumerate(self.train_loader):
            if steps is not None and i >= steps:
                break
            self.ts.load_batch(imgs.float().to(self.device, non_blocking=True), target.float().to(self.device, non_blocking=True),
                               target_weight.float().to(self.device, non_blocking=True))
            # 02_train.py:208-212: loss = apply_perceptual_loss(exp_data, params, loss, metadata["perceptual_loss"]).
            # It is an affine map of the MSE by per-batch host scalars, so it becomes (scale, offset) of the fused step.
            perc = meta.get("perceptual_loss") if isinstance(meta, dict) else None
            self.ts.set_loss_affine(*perceptual_affine(self.exp_data, self.params, perc))
            loss = self.ts.step()
            self.iterations += 1
            losses.append(loss.clone())            # stays on the device: no host sync per iteration
            if i % self.acc_every == 0:
                _, avg_acc, _, _ = accuracy(self.ts.eng.out.detach(), target)
                accs.append(avg_acc)
                self.ts.check_forward_range()      # the heat maps just came to the host: one more 4-byte read
        self.ts.check_forward_range()
        self.train_loss = float(torch.stack(losses).mean().item()) if losses else 1e18
        self.train_acc = float(np.mean(accs)) if accs else 0.0

    @torch.no_grad()
    def validation_epoch(self, epoch: int):
        self.model.eval()
        losses, accs = [], []
        try:
            limit = len(self.valid_loader) // 5
        except TypeError:
            limit = None
        for i, (imgs, target, target_weight, _meta) in enumerate(self.valid_loader):
            if limit is not None and i >= limit:
                break
            imgs = imgs.float().to(self.device)
            target = target.float().to(self.device)
            output = forward_pass(model=self.model, img=imgs, model_name=self.model_name, device=self.device, flip=False)
            losses.append(self.loss_function(output, target, target_weight.float().to(self.device)).clone())
            _, avg_acc, _, _ = accuracy(output, target)
            accs.append(avg_acc)
        self.valid_loss = float(torch.stack(losses).mean().item()) if losses else 1e18
        self.valid_acc = float(np.mean(accs)) if accs else 0.0

"""Fused train step for the HRNet hot path: weights->kernel layout, forward, masked-MSE loss,
backward, (RCCL gradient all-reduce), optimiser -- enqueued as one kernel stream and captured in
HIP graphs so that a step costs a handful of host calls.

Mirrors the reference's hot loop ``src/02_train.py:203-218`` (fwd -> PersonMSELoss -> backward ->
optimizer.step) and replaces its ``nn.DataParallel`` (``02_train.py:109``) with one process per GPU
and a bucketed all-reduce of the flat gradient buffer.  BatchNorm statistics stay
per replica, like DataParallel's.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch

from . import capi
from .dp import FlatAllReduce
from .hrnet import PoseHighResolutionNet

ADAM, SGD = "adam", "sgd"


class TrainStep:
    def __init__(self, model: PoseHighResolutionNet, batch: int, height: int, width: int, optimizer: str = ADAM,
                 lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 momentum: float = 0.9, nesterov: bool = False, process_group=None, bucket_mb: float = 32.0,
                 use_graph: bool = True, device=None, bf16_buckets: Optional[bool] = None):
        self.model = model
        from .hrnet import norm_device
        dev = norm_device(device or "cuda")
        if not model._packed(dev):
            model.to(dev)
            model._pack(dev)
        import weakref
        model._pinned_by = weakref.ref(self)   # forward() must never re-pack under us (it would orphan store/optimiser state)
        model.train()
        self.dev = dev
        self.store = model._store
        self.eng = model.engine(batch, height, width, True)
        self.kind = optimizer
        self.pg = process_group
        # buckets = the engine's own gradient buckets (contiguous slices, final at known points of backward)
        if bf16_buckets is None:
            bf16_buckets = os.environ.get("STLPOSE_BF16_BUCKETS", "0") != "0"
        self.dp = (FlatAllReduce(self.store.grads, process_group, bucket_mb, bounds=[(b["lo"], b["hi"]) for b in self.eng.buckets],
                                 bf16_buckets=bf16_buckets)
                   if process_group is not None else None)
        if self.dp is not None:
            self.eng.graph_mode = False   # the per-bucket all-reduce picks buckets up at the program's EVENTS (stl_program_wait_op)
        self._comm: Optional[torch.cuda.Stream] = None
        # per-bucket issue needs a collective that is a KERNEL on the communicator's stream (RCCL).  Decide by the backend serving
        # CUDA tensors: a default-initialised group reports e.g. "cpu:gloo,cuda:nccl" or "undefined" from get_backend()
        self._dp_rccl = False
        if process_group is not None:
            import torch.distributed as dist
            try:
                name = process_group._get_backend(torch.device("cuda")).name()
            except Exception:
                name = str(dist.get_backend(process_group))
            self._dp_rccl = "nccl" in str(name).lower()
            if not self._dp_rccl and os.environ.get("RANK", "0") == "0":
                import warnings
                warnings.warn(f"stlpose_amd.TrainStep: process group backend {name!r} is not RCCL -- gradient buckets are all-reduced "
                              "behind the whole backward pass (no overlap)")
        self._force_dp = os.environ.get("STLPOSE_DP_FORCE", "0") == "1"   # exercise the DP path with one rank (tests)
        self.world = self.dp.world if self.dp is not None else 1
        n = self.store.nparam
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev) if optimizer == ADAM else None
        gscale = self.dp.grad_scale if self.dp is not None else 1.0   # 1/world, or 1 when the bf16 buckets carry the mean
        self.hyper = torch.tensor([lr, betas[0], betas[1], eps, weight_decay, momentum, float(nesterov), gscale],
                                  dtype=torch.float32, device=dev)
        self.step_count = torch.zeros(1, dtype=torch.int32, device=dev)
        J = self.eng.out.shape[1]
        self.target = torch.zeros_like(self.eng.out)
        self.tweight = torch.ones(batch, J, dtype=torch.float32, device=dev)
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self._nblk = 512
        self._partial = torch.zeros(self._nblk, dtype=torch.float64, device=dev)
        # HIP-graph capture of a plan that forks onto >2 streams crashes inside hipStreamEndCapture
        # on ROCm 7.2 ("graphs"); such plans are replayed eagerly on their streams.
        if self.eng.nstreams > 2:
            use_graph = False
        self.use_graph = use_graph
        self._g_fb: Optional[torch.cuda.CUDAGraph] = None
        self._g_opt: Optional[torch.cuda.CUDAGraph] = None
        self._loss_scale, self._loss_offset = 1.0, 0.0
        # graph mode + process group: bucket events recorded at capture time are not re-recorded by a
        # replay, so the overlapped per-bucket all-reduce has nothing to wait on -- run eagerly instead
        if self.use_graph and process_group is not None:
            self.use_graph = False
        # Single process, native replay: optimiser slices and the next step's weight layouts run as ops of the backward
        # program (engine.attach_optimizer).  With a process group the collective sits between a bucket's reduction and
        # its optimiser, so the host issues those (_allreduce / _optim).
        self._fused_optim = (not self.use_graph and self.dp is None and os.environ.get("STLPOSE_FUSED_OPTIM", "0") != "0")   # measured neutral on one MI355X (14.93 vs 14.95 ms per step: the tail of backward it overlaps with is bandwidth-bound too): opt-in
        if self._fused_optim:
            self.eng.attach_optimizer(0 if optimizer == ADAM else 1, self.store.master.data_ptr(), self.store.grads.data_ptr(),
                                      self.m.data_ptr(), self.v.data_ptr() if self.v is not None else 0,
                                      self.hyper.data_ptr(), self.step_count.data_ptr())

    # ------------------------------------------------------------------ pieces
    def set_lr(self, lr: float):
        self.hyper[0] = lr

    def set_loss_affine(self, scale: float = 1.0, offset: float = 0.0):
        """loss <- scale * MSE + offset for the NEXT step(s): the scalar re-weighting of the reference's
        ``apply_perceptual_loss`` (lib/loss.py:97-150; 'add': scale = 1 + mean(p); lambda form: scale =
        lambda_D, offset = lambda_P * mean(p)).  The scale multiplies dL/dout inside the loss kernel."""
        scale, offset = float(scale), float(offset)
        if self.use_graph and (scale, offset) != (self._loss_scale, self._loss_offset):
            # the scale is a kernel argument baked into the capture and changes per batch with the perceptual loss
            # (scale = 1 + mean(p)): re-capturing every step would cost a device sync each time -- leave graph mode
            self.use_graph = False
            self._g_fb = self._g_opt = None
        self._loss_scale, self._loss_offset = scale, offset

    def load_batch(self, img: torch.Tensor, target: torch.Tensor, target_weight: torch.Tensor):
        self.eng.img.copy_(img, non_blocking=True)
        self.target.copy_(target, non_blocking=True)
        self.tweight.copy_(target_weight.reshape(self.tweight.shape), non_blocking=True)

    def check_forward_range(self):
        """Raise FloatingPointError if a step since the last call overflowed the 16-bit forward tensors (4-byte device read:
        call it where the host synchronises anyway -- Trainer does at every accuracy / epoch-loss read)."""
        self.eng.check_forward_range()

    def invalidate_weights(self):
        """Call after changing the model's parameters outside of step() (load_state_dict, manual edits)."""
        self._prepped = False

    def _fwd_bwd(self, update_running: bool = True, fused_optim: bool = False):
        st = torch.cuda.current_stream().cuda_stream
        e = self.eng
        # kernel-layout weights were refreshed bucket by bucket at the end of the previous step
        e.weights_ready = getattr(self, "_prepped", False)
        self._prepped = False
        e.forward(st, update_running=update_running)
        B, J = e.out.shape[:2]
        HW = e.out[0, 0].numel()
        capi.call("stl_mse_loss", e.out.data_ptr(), self.target.data_ptr(), self.tweight.data_ptr(), e.dout.data_ptr(),
                  self._partial.data_ptr(), self._nblk, None, B, J, HW, self._loss_scale, st)   # (the scalar: behind backward, below)
        if fused_optim:
            capi.call("stl_optim_begin_step", self.step_count.data_ptr(), self.eng.overflow.data_ptr(), st)
        # per-bucket issue while backward is being enqueued: RCCL only ("nccl": the collective is a kernel enqueued on the
        # communicator's stream, the host does not wait); gloo's all-reduce of a device tensor makes the host wait for the stream
        dp_on = self.dp is not None and (self.world > 1 or self._force_dp) and not self.use_graph and self._dp_rccl
        e.backward(st, fused_optim=fused_optim, on_bucket=self._issue_bucket if dp_on else None)
        self._buckets_issued = dp_on
        # the loss scalar (a one-block sum of the per-block partials): enqueued BEHIND the backward program -- between the loss
        # kernel and the head's gradient it was 16 us of a single resident block at a point where nothing else can run
        capi.call("stl_sum_partials", self._partial.data_ptr(), self._nblk, 0.5 / float(B * J * HW), self.loss.data_ptr(), 0, st)
        if self._loss_scale != 1.0 or self._loss_offset != 0.0:   # the kernel scales only dL/dout
            self.loss.mul_(self._loss_scale).add_(self._loss_offset)
        if fused_optim:
            self._prepped = True   # every bucket's kernel-layout weights were refreshed inside the program

    def _optim(self):
        st = torch.cuda.current_stream().cuda_stream
        s = self.store
        if self.kind == ADAM:
            capi.call("stl_adam_step", s.master.data_ptr(), s.grads.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                      s.nparam, self.hyper.data_ptr(), self.step_count.data_ptr(), self.eng.overflow.data_ptr(), st)
        else:
            capi.call("stl_sgd_step", s.master.data_ptr(), s.grads.data_ptr(), self.m.data_ptr(), s.nparam,
                      self.hyper.data_ptr(), self.step_count.data_ptr(), self.eng.overflow.data_ptr(), st)

    def _issue_bucket(self, i: int):
        """Enqueue gradient bucket i's all-reduce on the communication stream, behind the bucket's event.  Called by
        Engine.backward between two ranges of the backward program, i.e. while the host is still ENQUEUING backward: hardware
        queues are in-order and RCCL's stream shares one with a compute stream (GPU_MAX_HW_QUEUES = 4), so a collective issued
        after the whole program would sit behind everything that queue still has to run (round 4, measured with a stand-in
        kernel on the communication stream: bucket 0 final at 6.3 ms, its kernel started at 13.9 ms -- no overlap at all)."""
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=self.dev)
        self.eng.bucket_wait(i, self._comm.cuda_stream)
        with torch.cuda.stream(self._comm):
            self.dp.launch(upto=i + 1, force=self._force_dp)

    def _allreduce(self):
        """Bucketed all-reduce overlapped with backward.  The collectives were enqueued bucket by bucket while backward was
        being issued (_issue_bucket); here the main stream waits for them before the optimiser.  (Graph mode: the buckets'
        events do not exist, the collectives are issued here, behind the whole backward.)"""
        if self.dp is None or (self.world == 1 and not self._force_dp):
            return
        if not getattr(self, "_buckets_issued", False):
            for i in range(len(self.eng.buckets)):
                self._issue_bucket(i)
        self.dp.wait()

    def _capture(self):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):   # warm-up outside capture (lazy kernel attributes, allocator); the BatchNorm
            self._fwd_bwd(update_running=False)   # running statistics and num_batches_tracked must not see this extra pass
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._g_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_fb):
            self._fwd_bwd()
        self._g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_opt):
            self._optim()

    # ------------------------------------------------------------------ one step
    def step(self) -> torch.Tensor:
        """One train step on the currently loaded batch; returns the (device) loss scalar."""
        if self.use_graph:
            if self._g_fb is None:
                self._capture()
            self._g_fb.replay()
            self._allreduce()
            self._g_opt.replay()
        elif self._fused_optim and not self.use_graph:
            self._fwd_bwd(fused_optim=True)
        else:
            self._fwd_bwd()
            self._allreduce()
            self._optim()
        return self.loss

"""Mirror of src/model/Base.py:Sequential — the model base class (flags, output bias, loss/eval scaffold)
re-hosted on torch.nn.Module with a flat parameter arena for the fused optimizer."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
from torch import nn

from .. import ops


class Sequential(nn.Module):
    """Base.py:90-207.  Reads the same FLAGS fields as the reference (Base.py:92-104)."""

    def __init__(self, num_items, FLAGS):
        super().__init__()
        self.num_items = num_items
        self.num_units = FLAGS.num_units
        self.num_heads = FLAGS.num_heads
        self.hidden_dropout_rate = float(getattr(FLAGS, "hidden_dropout_rate", 0.0) or 0.0)
        self.attention_probs_dropout_rate = float(getattr(FLAGS, "attention_probs_dropout_rate", 0.0) or 0.0)
        self.seqslen = FLAGS.seqslen
        self.learning_rate = FLAGS.learning_rate
        self.l2_reg = float(getattr(FLAGS, "l2_reg", 0.0) or 0.0)
        self.num_train_steps = getattr(FLAGS, "num_train_steps", None)
        self.num_warmup_steps = getattr(FLAGS, "num_warmup_steps", None)
        cd = getattr(FLAGS, "compute_dtype", "bf16")
        self.act_dtype = {"bf16": torch.bfloat16, "f32": torch.float32, "fp32": torch.float32}[cd]
        self._arena: Optional[torch.Tensor] = None
        self.pad = (0, 0)       # (dh_pad, dh_true) of a channel-padded model (EasyDGL at a head dim the kernels do not tile)

    # ---- flat parameter arena -------------------------------------------------------------------------
    def finalize(self, device) -> "Sequential":
        """Move to `device` and re-home every parameter as a view of ONE flat f32 arena (plus a flat
        gradient arena, Adam moments and — in bf16 mode — a bf16 shadow refreshed by the optimizer kernel).
        Embedding tables come first so that the l2 segments (coding.py:48-55) are contiguous ranges."""
        self.to(device)
        names = [n for n, _ in self.named_parameters()]
        order = sorted(names, key=lambda n: (0 if n in self.l2_param_names() else 1, names.index(n)))
        params = dict(self.named_parameters())
        total = sum(params[n].numel() for n in order)
        # keep every view 16-byte aligned for the vector loads of the kernels
        offs, o = {}, 0
        for n in order:
            offs[n] = o
            o += (params[n].numel() + 7) // 8 * 8
        total = o
        arena = torch.zeros(total, device=device, dtype=torch.float32)
        # the gradient arena carries 8 trailing floats that are no parameter's gradient: scalars that must travel with the ONE
        # data-parallel all-reduce of a step (TrainEngine: this rank's share of the cross-entropy + TPP loss)
        self._grad_comm = torch.zeros(total + 8, device=device, dtype=torch.float32)
        grad = self._grad_comm[:total]
        for n in order:
            p = params[n]
            view = arena[offs[n]:offs[n] + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = grad[offs[n]:offs[n] + p.numel()].view(p.shape)
        self._arena, self._grad_arena, self._offsets, self._order = arena, grad, offs, order
        self._plist = [(params[n], params[n].grad) for n in order]   # (parameter, its view of the gradient arena)
        self._adam_m = torch.zeros_like(arena)
        self._adam_v = torch.zeros_like(arena)
        self._adam_state = torch.zeros(2, device=device, dtype=torch.int64)
        self._rng_state = torch.tensor([getattr(self, "seed", 9876), 0], device=device, dtype=torch.int64)
        segs: List[int] = []
        for n in self.l2_param_names():
            segs += [offs[n], offs[n] + params[n].numel()]
        self._l2_seg = torch.tensor(segs, device=device, dtype=torch.int64) if segs else None
        self._shadow = None
        if self.act_dtype != torch.float32:
            self._shadow = torch.empty(total, device=device, dtype=self.act_dtype)
            self.sync_shadow()
        # operator modules (module/coding.py, module/temporal.py) read parameters through the owning model's compute copy
        for mod in self.modules():
            if mod is self:
                continue
            if "compute" in mod.__dict__:
                mod.compute = self.compute
            if "act_dtype" in mod.__dict__:
                mod.act_dtype = self.act_dtype
        return self

    def l2_param_names(self) -> List[str]:
        return []

    def sync_shadow(self) -> None:
        """Refresh the low-precision compute copies after the f32 masters were changed outside adam_step."""
        if self._shadow is not None:
            ops.cast_to(self._arena, self.act_dtype, out=self._shadow)

    def compute(self, p: torch.Tensor) -> torch.Tensor:
        """Tensor the kernels read for parameter `p`: the master (f32 mode) or its bf16 shadow view."""
        if self._arena is None:
            raise RuntimeError("call model.finalize(device) before running the model")
        if self._shadow is None:
            return p
        off = (p.data_ptr() - self._arena.data_ptr()) // 4
        return self._shadow[off:off + p.numel()].view(p.shape)

    def zero_grad_arena(self) -> None:
        self._grad_arena.zero_()

    def detach_grads(self) -> None:
        """Before an autograd backward of a training step: with `.grad` unset autograd hands every parameter the gradient tensor
        an op produced instead of launching one `+=` kernel per parameter into the arena views."""
        for p, _ in self._plist:
            p.grad = None

    def collect_grads(self) -> None:
        """After that backward: copy the gradients into the flat arena with ONE multi-tensor launch and restore the views."""
        src, dst = [], []
        for p, view in self._plist:
            g = p.grad
            if g is None:
                view.zero_()
            else:
                src.append(g)
                dst.append(view)
            p.grad = view
        if dst:
            torch._foreach_copy_(dst, src)

    def graphed_train_step(self, features: Dict[str, torch.Tensor], labels: torch.Tensor, warmup: int = 2):
        """One `train_step` captured into a HIP graph on static copies of the batch: returns step(features, labels) -> loss
        (device scalar) that copies the new batch in and replays ~170 kernel launches as one submission.  Everything a step
        needs lives on the device (dropout step counter, Adam step count), so replays advance like eager steps.  The `warmup`
        eager steps that precede the capture are real optimizer steps."""
        static_f = {k: v.clone() for k, v in features.items()}
        static_l = labels.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        warm_loss = None
        with torch.cuda.stream(side):
            for _ in range(warmup):
                warm_loss = self.train_step(static_f, static_l).clone()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            loss = self.train_step(static_f, static_l)

        def step(f, l):
            for k, buf in static_f.items():
                buf.copy_(f[k])
            static_l.copy_(l)
            graph.replay()
            return loss
        step.graph = graph
        step.warmup_loss = warm_loss   # loss of the last eager warm-up step (a real optimizer step on the capture batch)
        return step

    def mask_padded_grads(self) -> None:
        """Channel-padded models zero the few gradients that are not zero on padded entries by themselves (EasyDGL)."""

    def settle_state(self) -> None:
        """The static engine leaves the step counters (dropout step, Adam step / learning rate) of the NEXT step in place behind
        its optimizer kernel — the single-thread update then costs nothing in front of the next step's first kernels
        (engine.TrainEngine._advance_state).  Anything else that reads or advances them (the autograd path's steps, a checkpoint)
        calls this first: the counters go back to "steps taken so far"."""
        if getattr(self, "_state_ahead", False):
            self._rng_state[1] -= 1
            self._adam_state[0] -= 1
            self._state_ahead = False

    def optimizer_step(self) -> None:
        """tf.train.AdamOptimizer(lr).minimize (Base.py:142-144) fused over the arena.  The l2 gradient is
        produced by autograd (ops.L2Fn), so no l2 is folded in here."""
        self.settle_state()
        self.mask_padded_grads()     # (channel-padded models; idempotent: every optimizer entry point keeps the padded entries at zero)
        ops.adam_step(self._arena, self._grad_arena, self._adam_m, self._adam_v, self.learning_rate, self._adam_state,
                      0.0, None, self._shadow)

    # ---- Base.py:106-113 ---------------------------------------------------------------------------------
    def make_output_bias(self) -> nn.Parameter:
        """output_bias(inf_pad=True): variable [num_items-1] zeros; the used bias is concat([-1000], var)."""
        return nn.Parameter(torch.zeros(self.num_items - 1))

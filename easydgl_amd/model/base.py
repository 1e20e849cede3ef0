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
        self._l2_parts_owner = None      # (the weights were rewritten: no engine's sums of squares describe them any more)
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
        self._table_grad_zero = None
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
        self._state_pinned = True      # (the captured launches keep the step counters' addresses: engine.TrainEngine does not swap them)
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

    # ---- channel padding (TGAT / TiSASRec / CTSMA; EasyDGL has its own forms of the same in model/easydgl.py) -----------------
    # A head dim the attention kernels do not tile (they take 16 / 32 / 64 / 128; the reference's own default is --num_units 50
    # --num_heads 1, main.py:35-37) runs at the next supported head dim with ZERO-PADDED channels: every parameter is stored at the
    # padded width, head h's true channels sit at [h * dh_pad, h * dh_pad + dh_true), the padded rows / columns are zero and stay
    # zero (the projections' padded columns, the dense layers' padded rows and columns and the tables' padded columns are 0, the
    # LayerNorms take their moments over the real channels and return nothing into the padded ones); the score scale 1 / sqrt(dh)
    # and coding.py's sqrt(num_units) are those of the TRUE width.  What is not zero on a padded entry by itself (rounding residue
    # of a softmax backward's row sums against TGAT's constant time code, the intensity MLP's padded hidden units) is zeroed by
    # mask_padded_grads in front of every optimizer launch.  tf_values() / tf_gradients() / load_tf_variables() speak the
    # reference's shapes.  A model lists its variables in _pad_specs(): (TF name, parameter, per-axis index map true -> padded or
    # None, initialiser of the TRUE-shape variable).
    def _setup_channel_pad(self, who: str, max_pad_dim: int = 128) -> None:
        from ..module import temporal as T
        self.width_true = self.num_units
        self.pad, self.qk_scale = (0, 0), 0.0
        H = self.num_heads
        if H <= 0 or self.num_units % H:
            raise ValueError(f"{who}: num_units={self.num_units} must be a multiple of num_heads={H}")
        dht = self.num_units // H
        if dht in T.SUPPORTED_HEAD_DIMS or dht > max_pad_dim:
            return
        dhp = next(d for d in T.SUPPORTED_HEAD_DIMS if d >= dht)
        self.pad = (dhp, dht)
        self.num_units = H * dhp
        self.qk_scale = float(dht) ** -0.5

    def _cmap(self, blocks: int = 1) -> torch.Tensor:
        """True channel -> padded channel, for `blocks` concatenated [C]-wide column blocks (K | V, item | position ...)."""
        dhp, dht = self.pad
        c = torch.arange(self.width_true)
        one = (c // dht) * dhp + (c % dht)
        return torch.cat([one + k * self.num_units for k in range(blocks)])

    def _pad_specs(self):
        raise NotImplementedError

    @staticmethod
    def _spec_index(t, maps):
        idx = [torch.arange(t.shape[a]) if m is None else m for a, m in enumerate(maps)]
        return torch.meshgrid(*[i.to(t.device) for i in idx], indexing="ij")

    @staticmethod
    def _spec_shape(p, maps):
        return tuple(p.shape[a] if m is None else len(m) for a, m in enumerate(maps))

    @torch.no_grad()
    def _init_padded(self, gen) -> None:
        """The reference's initialisers on the TRUE shapes (glorot limits of the true fans), scattered into zeroed padded storage."""
        import numpy as np
        from ..module.coding import glorot_uniform_
        specs = self._pad_specs()
        for _, p, _, _ in specs:
            p.data.zero_()
        for _, p, maps, kind in specs:
            shp = self._spec_shape(p, maps)
            if kind == "glorot":
                v = glorot_uniform_(torch.empty(shp), gen)
            elif kind == "ones":
                v = torch.ones(shp)
            elif kind == "linspace9":      # coding.py:104-108 TimeFunctionCoding.basis_freq
                v = torch.from_numpy(np.linspace(0, 9, shp[0]).astype(np.float32))
            else:
                v = torch.zeros(shp)
            p.data[self._spec_index(p.data, maps)] = v.to(p.device, p.dtype)

    def mask_padded_grads(self) -> None:
        """Zero every gradient entry of a padded row / column (see above); a model without padding: nothing.  ONE index_fill_ on
        the flat gradient arena (the padded entries of every parameter as one int64 index, built once) — a launch per (parameter,
        axis) was dozens of tiny kernels in front of every optimizer step of a 3-block model, captured into the graph step too."""
        if not self.pad[0]:
            return
        plan = getattr(self, "_pad_mask_plan", None)
        if plan is None:
            real = {}      # parameter -> per axis: set of real indices (None: all)
            for _, p, maps, _ in self._pad_specs():
                cur = real.setdefault(id(p), [p] + [set() if m is not None else None for m in maps])
                for a, m in enumerate(maps):
                    if m is not None:
                        cur[1 + a].update(m.tolist())
            per_param, flat = [], []
            base = self._arena.data_ptr()
            for p, *axes in real.values():
                mask = torch.zeros(p.shape, dtype=torch.bool)
                for a, idxs in enumerate(axes):
                    if idxs is not None:
                        padded = sorted(set(range(p.shape[a])) - idxs)
                        if padded:
                            idx = torch.tensor(padded, dtype=torch.int64)
                            mask.index_fill_(a, idx, True)
                            per_param.append((p, a, idx.to(p.device)))
                pos = torch.nonzero(mask.reshape(-1)).reshape(-1)
                if pos.numel():
                    flat.append(pos + (p.data_ptr() - base) // 4)
            flat_idx = torch.cat(flat).to(self._arena.device) if flat else None
            plan = (flat_idx, per_param)
            self._pad_mask_plan = plan
        flat_idx, per_param = plan
        if flat_idx is None:
            return
        g0, n = self._grad_arena.data_ptr(), self._grad_arena.numel()
        if all(p.grad is not None and g0 <= p.grad.data_ptr() < g0 + 4 * n for p, _, _ in per_param):
            self._grad_arena.index_fill_(0, flat_idx, 0.0)      # (every gradient is its view of the arena: the engine and collect_grads)
            return
        for p, a, idx in per_param:
            if p.grad is not None:
                p.grad.index_fill_(a, idx, 0.0)

    @torch.no_grad()
    def padded_leak(self) -> float:
        """Largest |value| on a padded entry of any parameter (0.0 for a healthy model; tests)."""
        worst = 0.0
        clones = {}
        for _, p, maps, _ in self._pad_specs():
            q = clones.setdefault(id(p), p.detach().clone())
            q[self._spec_index(q, maps)] = 0
        for q in clones.values():
            worst = max(worst, float(q.abs().max()))
        return worst

    @torch.no_grad()
    def _padded_values(self, grad: bool):
        out = {}
        for name, p, maps, _ in self._pad_specs():
            t = p.grad if grad else p
            out[name] = t.detach()[self._spec_index(t, maps)].clone()
        return out

    @torch.no_grad()
    def _load_padded(self, values) -> None:
        import numpy as np
        specs = self._pad_specs()
        for _, p, _, _ in specs:
            p.data.zero_()
        for name, p, maps, _ in specs:
            v = torch.as_tensor(np.asarray(values[name]), dtype=torch.float32)
            if tuple(v.shape) != self._spec_shape(p, maps):
                raise ValueError(f"{name}: expected shape {self._spec_shape(p, maps)}, got {tuple(v.shape)}")
            p.data[self._spec_index(p.data, maps)] = v.to(p.device, p.dtype)
        self.sync_shadow()

    def settle_state(self) -> None:
        """The static engine leaves the step counters (dropout step, Adam step / learning rate) of the NEXT step in place behind
        its optimizer kernel — the single-thread update then costs nothing in front of the next step's first kernels
        (engine.TrainEngine._advance_state).  Anything else that reads or advances them (the autograd path's steps, a checkpoint)
        calls this first: the counters go back to "steps taken so far"."""
        self._l2_parts_owner = None      # (whoever settles the counters is about to read or rewrite the state itself)
        self._table_grad_zero = None     # (... or the gradient arena: engine.TrainEngine zero-fills the tied table's gradient again)
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

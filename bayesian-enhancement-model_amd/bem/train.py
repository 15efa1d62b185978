"""Optimizer side of the Stage-II training step (basicsr/models/image_enhancer_model.py:200-211):
``clip_grad_norm_(net.parameters(), max_norm)`` + ``torch.optim.AdamW.step()`` as two kernel launches over ONE flat buffer.

All trainable parameters of a group live in one contiguous f32 buffer (``p.data`` are views of it), their gradients in a second
one (``p.grad`` are views; the backward kernels of bem.autograd accumulate straight into them), Adam's moments in two more:
  zero_grad  = one memset,
  clip       = one sum-of-squares reduction (f64) whose result stays on the device,
  step       = one elementwise kernel that reads the clip coefficient from the device -- no host synchronisation in the step.
Hyper-parameters, update rule and state layout follow torch.optim.AdamW (decoupled weight decay, bias correction), so a
``state_dict()`` from here loads into torch.optim.AdamW and back.
"""
from __future__ import annotations

from typing import Iterable

import torch

from . import ops


def _align4(n: int) -> int:
    return (n + 3) // 4 * 4


class StepState:
    """Per-iteration scalars of a graph-captured training step, resident in HBM.

    A HIP graph replays its kernels with the arguments they were captured with, so everything that changes from one iteration to the
    next -- the Philox epoch of the weight draws, the learning rate and Adam's bias corrections, each Bayesian leaf's EMA decay --
    is read by the kernels from this buffer instead (``stream_add`` / ``hyper`` / ``decay_dev`` of the C ABI).  ``slot(fn, k)`` hands
    out k floats whose values ``fn()`` supplies before every run; ``upload`` evaluates them and sends all words with ONE launch
    (bem_store_words); ``advance`` runs the host-side bookkeeping of the step (counters the eager path bumps inline) after it."""

    WORDS = 512

    def __init__(self, device):
        self.dev = torch.zeros(self.WORDS, device=device, dtype=torch.float32)
        self.host = torch.zeros(self.WORDS, dtype=torch.float32)
        self.epoch_dev = self.dev[:2].view(torch.int64)            # words 0..1: [epoch] << 20 (SampleCtx.next_stream's epoch field)
        self._host_epoch = self.host[:2].view(torch.int64)
        self.n = 2
        self._fns, self._advance = [], []

    def slot(self, fn, k: int = 1):
        if self.n + k > self.WORDS:
            raise RuntimeError("StepState: out of slots")
        view = self.dev[self.n:self.n + k]
        self._fns.append((self.n, k, fn))
        self.n += k
        return view

    def on_advance(self, fn):
        self._advance.append(fn)

    def upload(self, epoch: int):
        self._host_epoch[0] = (int(epoch) & 0xFFFFFF) << 20
        for off, k, fn in self._fns:
            v = fn()
            self.host[off:off + k] = torch.as_tensor(v, dtype=torch.float32).reshape(k)
        ops.store_words(self.dev, self.host, self.n)

    def advance(self):
        for fn in self._advance:
            fn()


STEP_STATE = [None]        # set while a step is being captured: leaves and the optimizer take their per-iteration scalars from it


class BemAdamW(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, **ignored):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._flat = []          # per group: dict(p, g, m, v, n, params)
        self._steps = 0
        self._lag = {}           # parameter -> steps it is behind the shared counter (skipped iterations)
        self._max_norm = 0.0
        self._sumsq = None
        self._norm = None
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.requires_grad]
            if not ps:
                self._flat.append(None)
                continue
            dev = ps[0].device
            if dev.type != "cuda" or any(p.device != dev or p.dtype != torch.float32 for p in ps):
                raise ValueError("BemAdamW: float32 parameters on one HIP device")
            total = sum(_align4(p.numel()) for p in ps)
            fp = torch.zeros(total, device=dev, dtype=torch.float32)
            fg = torch.zeros_like(fp)
            fm, fv = torch.zeros_like(fp), torch.zeros_like(fp)
            off = 0
            with torch.no_grad():
                for p in ps:
                    n = p.numel()
                    fp[off:off + n].copy_(p.detach().reshape(-1))
                    p.data = fp[off:off + n].view(p.shape)
                    p.grad = fg[off:off + n].view(p.shape)
                    self.state[p] = {"step": torch.tensor(0.0), "exp_avg": fm[off:off + n].view(p.shape), "exp_avg_sq": fv[off:off + n].view(p.shape)}
                    off += _align4(n)
            self._flat.append(dict(p=fp, g=fg, m=fm, v=fv, params=ps))
            self._sumsq = torch.zeros(1, device=dev, dtype=torch.float64)
            self._norm = torch.zeros(1, device=dev, dtype=torch.float32)
        ops.bump_weight_epoch()

    # -- gradient buffers -------------------------------------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = False):
        """One memset per group; the .grad views stay attached (set_to_none is ignored: the kernels accumulate into them)."""
        for f in self._flat:
            if f is None:
                continue
            f["g"].zero_()
            off = 0
            for p in f["params"]:
                n = p.numel()
                if p.grad is None or p.grad.data_ptr() != f["g"].data_ptr() + 4 * off:
                    p.grad = f["g"][off:off + n].view(p.shape)
                off += _align4(n)

    def all_reduce_grads(self):
        """Data-parallel training: average the flat gradient buffers over the ranks (one RCCL all-reduce per parameter group -- the
        whole-model bucket that DistributedDataParallel converges to; reference wrap: basicsr/models/base_model.py:97-100)."""
        import torch.distributed as dist
        world = dist.get_world_size()
        if world == 1:
            return
        for f in self._flat:
            if f is None:
                continue
            if dist.get_backend() == "nccl":
                dist.all_reduce(f["g"], op=dist.ReduceOp.AVG)
            else:                                                   # gloo (tests): no AVG
                dist.all_reduce(f["g"])
                f["g"].mul_(1.0 / world)

    def clip_grad_norm_(self, max_norm: float):
        """torch.nn.utils.clip_grad_norm_ over every group's gradients (L2, error_if_nonfinite=False).  The scaling happens inside
        the next step(); returns a 1-element device tensor that holds the total norm once that step has run."""
        flats = [f for f in self._flat if f is not None]
        if len(flats) != 1:
            raise NotImplementedError("BemAdamW.clip_grad_norm_: one non-empty parameter group")
        ops.grad_sumsq(flats[0]["g"], self._sumsq)
        self._max_norm = float(max_norm)
        return self._norm

    def _slices(self, p):
        """(flat parameter, gradient, first moment, second moment) views of one parameter, and its group's hyper-parameters."""
        for group, f in zip(self.param_groups, self._flat):
            if f is None:
                continue
            off = 0
            for q in f["params"]:
                n = q.numel()
                if q is p:
                    return tuple(f[k][off:off + n] for k in ("p", "g", "m", "v")), group
                off += _align4(n)
        raise KeyError("BemAdamW: parameter is not in any group")

    def graph_safe(self, skip=()):
        """True when step(skip) is the single fused launch per group (no parameter that lags behind the shared count and needs its own
        bias corrections): the form a captured step can replay."""
        skip = list(skip)
        return not any(lag and not any(p is q for q in skip) for p, lag in self._lag.items())

    def _host_advance(self, skip):
        self._steps += 1
        for f in self._flat:
            if f is not None:
                for p in f["params"]:
                    self.state[p]["step"] += 1
        for p in skip:
            self.state[p]["step"] -= 1
            self._lag[p] = self._lag.get(p, 0) + 1
        ops.bump_weight_epoch()

    @torch.no_grad()
    def step_captured(self, state: StepState, skip=()):
        """step(skip) recorded for replay: learning rate and bias corrections come from ``state`` slots (filled from the host's group['lr']
        and step count before each run), the counters advance in ``state.advance()`` after each run instead of here."""
        skip = list(skip)
        if not self.graph_safe(skip):
            raise RuntimeError("BemAdamW.step_captured: a lagging parameter needs its own step count (run step())")
        keep = [(p, p.detach().clone(), self.state[p]["exp_avg"].clone(), self.state[p]["exp_avg_sq"].clone()) for p in skip]
        for group, f in zip(self.param_groups, self._flat):
            if f is None:
                continue
            b1, b2 = group["betas"]

            def hyper(group=group, b1=b1, b2=b2):
                t = self._steps + 1
                return [float(group["lr"]), float(1.0 - b1 ** t), float((1.0 - b2 ** t) ** 0.5)]
            ops.adamw_step_(f["p"], f["g"], f["m"], f["v"], group["lr"], group["betas"], group["eps"], group["weight_decay"], 0,
                            max_norm=self._max_norm, sumsq=self._sumsq if self._max_norm > 0 else None, norm_out=self._norm,
                            hyper=state.slot(hyper, 3))
        for p, val, m, v in keep:
            p.copy_(val); self.state[p]["exp_avg"].copy_(m); self.state[p]["exp_avg_sq"].copy_(v)
        self._max_norm = 0.0
        state.on_advance(lambda: self._host_advance(skip))

    @torch.no_grad()
    def step(self, closure=None, skip=()):
        """``skip``: parameters that took no part in this iteration's graph.  torch.optim.AdamW leaves a parameter whose ``.grad`` is None
        untouched (no decay, no moment update, its own step count stands still); the fused kernel runs over the whole flat buffer with
        ONE step count, so skipped slices are put back after it, and a parameter that has fallen behind the shared count (``_lag``) and is
        updated again is redone on its own slice with its own count (bias correction as torch computes it)."""
        if closure is not None:
            raise NotImplementedError("BemAdamW.step: closures are not supported")
        skip = list(skip)
        behind = [p for p in self._lag if self._lag[p] and not any(p is q for q in skip)]
        keep = [(p, p.detach().clone(), self.state[p]["exp_avg"].clone(), self.state[p]["exp_avg_sq"].clone()) for p in skip + behind]
        self._steps += 1
        for group, f in zip(self.param_groups, self._flat):
            if f is None:
                continue
            ops.adamw_step_(f["p"], f["g"], f["m"], f["v"], group["lr"], group["betas"], group["eps"], group["weight_decay"], self._steps,
                            max_norm=self._max_norm, sumsq=self._sumsq if self._max_norm > 0 else None, norm_out=self._norm)
            for p in f["params"]:
                self.state[p]["step"] += 1
        for p, val, m, v in keep:
            p.copy_(val); self.state[p]["exp_avg"].copy_(m); self.state[p]["exp_avg_sq"].copy_(v)
        for p in skip:
            self.state[p]["step"] -= 1
            self._lag[p] = self._lag.get(p, 0) + 1
        for p in behind:
            (fp, fg, fm, fv), group = self._slices(p)
            ops.adamw_step_(fp, fg, fm, fv, group["lr"], group["betas"], group["eps"], group["weight_decay"], self._steps - self._lag[p],
                            max_norm=self._max_norm, sumsq=self._sumsq if self._max_norm > 0 else None, norm_out=self._norm)
        self._max_norm = 0.0
        ops.bump_weight_epoch()        # parameters changed behind torch's version counters: derived-weight caches are stale

    # -- checkpoints ------------------------------------------------------------------------------------------------
    def state_dict(self):
        """torch.optim.Optimizer layout ({'state': {idx: {step, exp_avg, exp_avg_sq}}, 'param_groups': [...]}) with standalone copies
        of the moments (the live ones are views of the flat buffers): loads into torch.optim.AdamW and into load_state_dict below."""
        sd = super().state_dict()
        sd["state"] = {k: {n: (t.detach().clone() if torch.is_tensor(t) else t) for n, t in st.items()} for k, st in sd["state"].items()}
        return sd

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        """Hyper-parameters via torch's loader, then the moments are copied INTO the flat buffers and the per-parameter state is pointed
        back at their views (torch's loader replaces the state tensors, which would detach them from the fused step)."""
        super().load_state_dict(state_dict)
        own = []
        for f in self._flat:
            if f is None:
                continue
            off = 0
            for p in f["params"]:
                n = p.numel()
                st = self.state.get(p, {})
                m, v = f["m"][off:off + n].view(p.shape), f["v"][off:off + n].view(p.shape)
                if "exp_avg" in st:
                    m.copy_(st["exp_avg"]); v.copy_(st["exp_avg_sq"])
                    k = int(float(st["step"]))
                else:                                        # torch.optim.AdamW keeps no state for a parameter that never had a gradient
                    m.zero_(); v.zero_()
                    k = 0
                own.append((p, k))
                self.state[p] = {"step": torch.tensor(float(k)), "exp_avg": m, "exp_avg_sq": v}
                off += _align4(n)
        # the fused step keeps ONE counter: the furthest parameter's; the others carry their distance to it (a parameter the step skipped,
        # e.g. the Stage-I mask token after the first scheduler period -- condition_generator_model.py:185-186)
        self._steps = max((k for _, k in own), default=0)
        self._lag = {p: self._steps - k for p, k in own if k != self._steps}
        ops.bump_weight_epoch()

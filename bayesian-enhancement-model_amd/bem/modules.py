"""nn.Module mirrors of the reference's hot-path modules, computing through the HIP C ABI.

Same class names, constructor arguments, parameter names/shapes (state-dict contract, SURVEY.md
section 8b) and ``forward(x)`` meaning as the reference modules cited in each docstring -- but every
forward is a sequence of hand-written gfx950 kernels (bem.ops).  There is no CPU implementation:
calling a module on CPU tensors raises ``BemNativeError``.  Inference (forward) only in this round;
the modules run under ``torch.no_grad`` semantics (outputs carry no autograd graph).
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import autograd as ag
from . import ops
from .native import BemNativeError


# ------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------
class _Cache:
    """Derived device tensors (packed / permuted weights) keyed on the source tensors' identity+version."""

    def __init__(self):
        self._d = {}

    def get(self, key, srcs, fn):
        sig = (ops.WEIGHT_EPOCH[0],) + tuple((t.data_ptr(), ops.tensor_version(t), t.device) for t in srcs)
        hit = self._d.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1]
        with torch.no_grad():
            val = fn()
        self._d[key] = (sig, val)
        return val


SCAN_RM = os.environ.get("BEM_SCAN_RM", "1") != "0"          # row-major SS2D scan (no transposes of xc / y1) where the plane size allows


def grad_mode(m: nn.Module) -> bool:
    """True when a forward has to record the training graph: module in train() mode with autograd enabled (the reference's
    training step, image_enhancer_model.py:165-216).  Everything else runs the inference kernels only."""
    return m.training and torch.is_grad_enabled()


def _need_cuda(x):
    if not x.is_cuda:
        raise BemNativeError("BEM modules run on the GPU through libbem_hip.so only (no CPU fallback); "
                             "move the model and input to a HIP device")


class SampleCtx:
    """Per-forward Bayesian sampling context.

    nsets: number of independent weight samples (= batch size: one per batch element) or 1 (shared);
    eps:   optional {'<module path>.weight'|'.bias': tensor (nsets, *shape)} injected N(0,1) draws
           (parity runs); when None the draws come from the in-kernel Philox stream (seed, counter).
    """

    _epoch = 0      # process-wide forward counter: successive forwards never reuse a Philox stream

    def __init__(self, nsets: int, eps: Optional[Dict[str, torch.Tensor]] = None, seed: int = 0, rank: int = 0, epoch: Optional[int] = None,
                 epoch_dev: Optional[torch.Tensor] = None):
        self.nsets, self.eps, self.seed, self.rank = nsets, eps, seed, rank
        self.counter = 0
        # epoch_dev: one int64 on the device holding ``epoch << 20``, added to the stream ids by the sampling kernels; the ids handed out
        # here then carry epoch 0.  A captured step (HIP graph) replays with whatever epoch the host wrote there last.
        self.epoch_dev = epoch_dev
        self.bank = None                     # EvalSampleBank whose one launch made this forward's draws (set by Network.forward)
        self.counter0 = 0                    # value of ``counter`` when the forward began (the leaves record relative stream numbers)
        if epoch_dev is not None:
            epoch = 0
        if epoch is None:
            SampleCtx._epoch += 1
            epoch = SampleCtx._epoch
        self.epoch = int(epoch)              # a caller-chosen epoch (the training iteration) makes the draws a function of (seed, rank, iteration)

    def next_stream(self):
        """A fresh Philox stream id per sampled tensor: [rank : 16 bits | forward epoch : 24 bits | tensor counter : 20 bits].
        Ranks of a multi-GPU run that share ``seed`` therefore draw different weight sets; bit 62 is reserved for the
        condition-noise draws (BEMPipeline.candidates)."""
        self.counter += 1
        if self.counter >= (1 << 20) or self.epoch >= (1 << 24):
            raise RuntimeError("SampleCtx: Philox stream id space exhausted (2^20 tensors per forward, 2^24 forwards per process)")
        return (self.rank << 44) | (self.epoch << 20) | self.counter


_SAMPLE_CTX: List[Optional[SampleCtx]] = [None]
_TRAIN_STEP: List[Optional["ag.BayesStep"]] = [None]      # set by Network.forward for the span of one training forward


class sampling:
    """``with sampling(SampleCtx(...)):`` scopes the Bayesian draws of the enclosed forwards."""

    def __init__(self, ctx: Optional[SampleCtx]):
        self.ctx = ctx

    def __enter__(self):
        self.prev = _SAMPLE_CTX[0]
        _SAMPLE_CTX[0] = self.ctx
        return self.ctx

    def __exit__(self, *a):
        _SAMPLE_CTX[0] = self.prev


# ------------------------------------------------------------------------------------------------
# leaves
# ------------------------------------------------------------------------------------------------
class LayerNorm2d(nn.LayerNorm):
    """basicsr/vmamba/models/vmamba.py:58-63.  Parameter holder: the normalisation itself is always
    fused into the prologue of the 1x1 GEMM that consumes it."""

    def forward(self, x):
        raise BemNativeError("LayerNorm2d is fused into its consumer GEMM; it has no standalone forward here")


class Linear2d(nn.Linear):
    """basicsr/vmamba/models/vmamba.py:42-55: 1x1 conv with a 2-D weight (4-D accepted on load)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._cache = _Cache()
        self.module_path = ""

    def _load_from_state_dict(self, state_dict, prefix, *args):
        if prefix + "weight" in state_dict:
            state_dict[prefix + "weight"] = state_dict[prefix + "weight"].view(self.weight.shape)
        return super()._load_from_state_dict(state_dict, prefix, *args)

    def gemm_weights(self, B):
        Wp = self._cache.get("p", [self.weight], lambda: ops.pack_pw_weight(self.weight.detach().contiguous()))
        return Wp, (self.bias.detach() if self.bias is not None else None)

    def forward(self, x, **kw):
        _need_cuda(x)
        if grad_mode(self):
            return _pw_train(self, x, kw)
        Wp, b = self.gemm_weights(x.shape[0])
        return ops.pw_gemm(x, Wp, self.out_features, bias=b, **kw)


def _pw_train(m, x, kw):
    """Training-mode forward of a plain 1x1 layer (optionally over the concatenation of two inputs)."""
    extra = set(kw) - {"x2", "in_mode"}
    if extra or (("x2" in kw) != (kw.get("in_mode", 0) == 2)):
        raise BemNativeError(f"1x1 layer: the training path covers plain and concat-input forms only (got {sorted(kw)})")
    return ag.PwFn.apply(x, kw.get("x2"), m.weight, m.bias, m)


class PwConv2d(nn.Conv2d):
    """nn.Conv2d(kernel_size=1) holder running on the MFMA pointwise GEMM."""

    def __init__(self, cin, cout, bias=True):
        super().__init__(cin, cout, 1, 1, 0, bias=bias)
        self._cache = _Cache()
        self.module_path = ""

    def gemm_weights(self, B):
        Wp = self._cache.get("p", [self.weight], lambda: ops.pack_pw_weight(
            self.weight.detach().reshape(self.out_channels, self.in_channels).contiguous()))
        return Wp, (self.bias.detach() if self.bias is not None else None)

    def forward(self, x, **kw):
        _need_cuda(x)
        if grad_mode(self):
            return _pw_train(self, x, kw)
        Wp, b = self.gemm_weights(x.shape[0])
        return ops.pw_gemm(x, Wp, self.out_channels, bias=b, **kw)


class DwConv2d(nn.Conv2d):
    """Depthwise 3x3, padding 1 (holder; epilogue chosen by the caller)."""

    def __init__(self, ch, bias=True):
        super().__init__(ch, ch, 3, 1, 1, groups=ch, bias=bias)
        self.module_path = ""

    def dw_weights(self, B):
        return self.weight.detach(), (self.bias.detach() if self.bias is not None else None)


class Conv2dK(nn.Conv2d):
    """Dense conv (3x3 s1 p1 or 4x4 s2 p1) on the direct-conv kernel."""

    def forward(self, x, relu=False, res1=None, res2=None, cin_slice=None):
        _need_cuda(x)
        if grad_mode(self) and self.weight.requires_grad:
            if relu or res1 is not None or res2 is not None:
                raise BemNativeError("Conv2dK: the training path covers the plain convolution (+ bias) only")
            return ag.Conv2dFn.apply(x, self.weight, self.bias, self, cin_slice)
        return ops.conv2d(x, self.weight.detach(), None if self.bias is None else self.bias.detach(),
                          stride=self.stride[0], pad=self.padding[0], relu=relu, res1=res1, res2=res2, cin_slice=cin_slice, dilation=self.dilation[0])


class ConvT2x2(nn.ConvTranspose2d):
    """nn.ConvTranspose2d(C, C/2, kernel 2, stride 2) = a 1x1 GEMM to 4*Cout rows + 2x2 scatter."""

    def __init__(self, cin, cout):
        super().__init__(cin, cout, kernel_size=2, stride=2, padding=0, output_padding=0)
        self._cache = _Cache()

    def forward(self, x):
        _need_cuda(x)
        if grad_mode(self):
            return ag.ConvT2x2Fn.apply(x, self.weight, self.bias, self)
        return self._forward_nograd(x)

    def _forward_nograd(self, x):
        co = self.out_channels

        def prep():
            w = self.weight.detach()                                  # (Cin, Cout, 2, 2)
            w4 = w.permute(2, 3, 1, 0).reshape(4 * co, self.in_channels).contiguous()   # row = (dy*2+dx)*Co + co
            return ops.pack_pw_weight(w4), self.bias.detach().repeat(4).contiguous()
        Wp, b4 = self._cache.get("p", [self.weight, self.bias], prep)
        return ops.pw_gemm(x, Wp, 4 * co, bias=b4, convT_Win=x.shape[3])


# ------------------------------------------------------------------------------------------------
# Bayesian leaves (basicsr/bayesian/conv.py:10-128, linear.py:8-104)
# ------------------------------------------------------------------------------------------------
class _BayesBase(nn.Module):
    def _init_common(self, sigma_init, decay, bias):
        self.deterministic = False
        self.decay, self.sigma_init, self.step = decay, sigma_init, 0
        self.bias = bias
        self.module_path = ""
        self._cache = _Cache()

    def _rho_init(self):
        return math.log(math.expm1(abs(self.sigma_init)) + 1e-20)

    def _register_priors(self):
        """prior_* buffers (non-persistent, as conv.py:57-70): copies of the freshly initialised mu / rho; the training forward moves
        them towards the current parameters (threshold EMA).  Loading a checkpoint does not touch them -- the reference's behaviour."""
        for kind in ("weight", "bias") if self.bias else ("weight",):
            self.register_buffer(f"prior_mu_{kind}", getattr(self, f"mu_{kind}").detach().clone(), persistent=False)
            self.register_buffer(f"prior_rho_{kind}", getattr(self, f"rho_{kind}").detach().clone(), persistent=False)
        self._ws = self._bs = self._eps_w = self._eps_b = self._sample_owner = None

    def _train_sample(self):
        """Once per training forward (conv.py:84-104): move the prior, draw eps, form w = mu + softplus(rho) eps.  The sampled tensors
        are what the forward / backward kernels see as this leaf's weight and bias (bem.autograd.wb)."""
        step = _TRAIN_STEP[0]
        if step is None:
            raise BemNativeError("Bayesian leaves: a training-mode forward runs under Network.forward, which scopes the weight sample")
        if self._sample_owner is step:
            return
        ctx = _SAMPLE_CTX[0]
        from .train import STEP_STATE
        st = STEP_STATE[0]
        decay = lambda: min(self.decay, (1 + self.step) / (10 + self.step))
        d, d_dev = decay(), (st.slot(decay) if st is not None else None)       # a captured step reads the iteration's decay from HBM
        for kind in ("weight", "bias") if self.bias else ("weight",):
            mu, rho = getattr(self, f"mu_{kind}"), getattr(self, f"rho_{kind}")
            ops.bnn_prior_ema_(getattr(self, f"prior_mu_{kind}"), getattr(self, f"prior_rho_{kind}"), mu.detach(), rho.detach(), d, d_dev)
            if ctx.eps is not None:
                e = ctx.eps[f"{self.module_path}.{kind}"].reshape((1,) + tuple(mu.shape)).contiguous()
            else:
                e = ops.randn((1,) + tuple(mu.shape), mu.device, ctx.seed, ctx.next_stream(), ctx.epoch_dev)
                self.__dict__.setdefault("_draw_order", {})[kind] = ctx.counter - getattr(step, "counter0", 0)   # for BayesBank: same streams
            w = ops.bnn_sample(mu.detach(), rho.detach(), 1, e)[0]
            if kind == "weight":
                self._ws, self._eps_w = w, e
            else:
                self._bs, self._eps_b = w, e
        if st is None:
            self.step += 1
        else:
            st.on_advance(self._count_step)           # recorded, not run: the counter moves when the captured step is replayed
        self._sample_owner = step
        step.leaves.append(self)

    def _count_step(self):
        self.step += 1

    def fold_sample_grads(self):
        """dmu += dw, drho += dw eps sigmoid(rho) for the sampled weight and bias of the finished backward pass, then drop the sample."""
        for w, e, kind in ((self._ws, self._eps_w, "weight"), (self._bs, self._eps_b, "bias")):
            if w is not None and w.grad is not None:
                mu, rho = getattr(self, f"mu_{kind}"), getattr(self, f"rho_{kind}")
                ops.bnn_reparam_bwd_(w.grad, e, rho.detach(), ag.grad_of(mu), ag.grad_of(rho))
        self._ws = self._bs = self._eps_w = self._eps_b = self._sample_owner = None

    def kl_terms(self):
        out = [(self.mu_weight, self.rho_weight, self.prior_mu_weight, self.prior_rho_weight)]
        if self.bias:
            out.append((self.mu_bias, self.rho_bias, self.prior_mu_bias, self.prior_rho_bias))
        return out

    def _sigma(self):
        rho = self.rho_weight
        return self._cache.get("sigma", [rho], lambda: ops.bnn_sample(torch.zeros_like(rho), rho.detach(), 1, torch.ones_like(rho))[0])

    def _sampled(self, B, packed_mk=None):
        """(weights (nsets,*shape), bias (nsets,C)|None, nsets) for this forward; with ``packed_mk = (M, K)`` the sampled
        weights come back already packed for the x6 GEMM (stochastic mode only)."""
        ctx = _SAMPLE_CTX[0]
        if self.deterministic:
            return self.mu_weight.detach()[None], (self.mu_bias.detach()[None] if self.bias else None), 1
        if self.training:
            self._train_sample()             # raises outside Network's training forward (e.g. train() mode under no_grad: call eval())
            return self._ws[None], (self._bs[None] if self.bias else None), 1
        if ctx is None:      # leaf used outside a Network forward: one-off context (fresh epoch)
            ctx = SampleCtx(B, None, seed=torch.initial_seed() & 0xFFFFFFFF)
        ns = ctx.nsets
        bank = ctx.bank
        if bank is not None:                         # this forward's draws were all made by one launch (EvalSampleBank)
            return bank.take(self, packed_mk, ns)
        ew = eb = None
        if ctx.eps is not None:
            ew = ctx.eps[self.module_path + ".weight"].contiguous()
            if self.bias:
                eb = ctx.eps[self.module_path + ".bias"].contiguous()
        rec = {"mk": packed_mk if (packed_mk is not None and ops.USE_X6) else None, "b": None}
        if packed_mk is not None and ops.USE_X6:
            # GEMM weights go straight into operand order (same Philox stream ids, same values as sample-then-pack)
            # sigma = log1p(exp(rho)) once per weight version (one launch of the sampler with mu = 0, eps = 1), not once per sample
            w = ops.bnn_sample_packed(self.mu_weight.detach(), self._sigma(), ns, packed_mk[0], packed_mk[1], ew, ctx.seed, ctx.next_stream(), sigma_given=True, stream_add=ctx.epoch_dev)
        else:
            w = ops.bnn_sample(self.mu_weight.detach(), self.rho_weight.detach(), ns, ew, ctx.seed, ctx.next_stream(), ctx.epoch_dev)
        rec["w"] = ctx.counter - ctx.counter0
        b = None
        if self.bias:
            b = ops.bnn_sample(self.mu_bias.detach(), self.rho_bias.detach(), ns, eb, ctx.seed, ctx.next_stream(), ctx.epoch_dev)
            rec["b"] = ctx.counter - ctx.counter0
        self.__dict__["_eval_draw"] = rec            # which streams of the forward this leaf drew from, and in which form (EvalSampleBank)
        return w, b, ns


class BayesBank:
    """All Bayesian tensors of a net as segments of flat arenas, so that the per-tensor steps of a training iteration -- prior EMA + eps
    draw + weight sample, KL, KL backward, reparameterisation backward (conv.py:84-112, tools.py:76-84) -- are ONE launch each over the
    whole net (bem_bnn_bank_*) instead of one per tensor (60 leaves / 90 tensors in the shipped Stage-I net: ~630 launches per step).

    The prior buffers of the leaves become views of the bank's prior arenas; the sampled weights, their eps and their gradient buffers
    are persistent views of the w / eps / gw arenas (the sampling launch zeroes gw).  Parameters and their .grad buffers are addressed by
    pointer (they usually live in BemAdamW's flat buffers); the tables are rebuilt when any of those pointers changes.  Used for
    Philox draws only -- injected eps (parity runs) take the per-leaf path."""

    def __init__(self, net):
        self.leaves = [m for m in net.modules() if isinstance(m, _BayesBase)]
        self.sig = None

    def ready(self):
        return self.sig is not None

    def try_build(self):
        """After a per-leaf Philox forward: every tensor has recorded which stream of the forward it drew from (1 .. S in execution
        order); the bank gives each segment that same stream, so its draws are the per-leaf path's draws."""
        order = sorted(m.__dict__.get("_draw_order", {}).get(kind, -1) for m, kind, _, _ in self._tensors())
        if order == list(range(1, len(order) + 1)) and not torch.cuda.is_current_stream_capturing():
            self._build()

    def _tensors(self):
        for m in self.leaves:
            for kind in ("weight", "bias") if m.bias else ("weight",):
                yield m, kind, getattr(m, f"mu_{kind}"), getattr(m, f"rho_{kind}")

    def usable(self, ctx):
        if ctx is None or ctx.eps is not None or ctx.nsets != 1 or not self.leaves:
            return False
        m0 = self.leaves[0]
        return all(m.training and not m.deterministic and m.decay == m0.decay and m.step == m0.step for m in self.leaves)

    def _signature(self):
        sig = []
        for m, kind, mu, rho in self._tensors():
            sig += [mu.data_ptr(), rho.data_ptr(), ag.grad_of(mu).data_ptr(), ag.grad_of(rho).data_ptr(),
                    getattr(m, f"prior_mu_{kind}").data_ptr(), getattr(m, f"prior_rho_{kind}").data_ptr()]
        return tuple(sig)

    def _build(self):
        if torch.cuda.is_current_stream_capturing():
            raise BemNativeError("BayesBank: its tables cannot be (re)built while a HIP graph is being captured")
        dev = self.leaves[0].mu_weight.device
        rows, blks, off = [], [], 0
        items = list(self._tensors())
        for s_, (m, kind, mu, rho) in enumerate(items):
            n = mu.numel()
            inv_n = int.from_bytes(torch.tensor(1.0 / n, dtype=torch.float32).numpy().tobytes(), "little")
            rows.append([mu.data_ptr(), rho.data_ptr(), ag.grad_of(mu).data_ptr(), ag.grad_of(rho).data_ptr(), off, n, m._draw_order[kind], inv_n])
            blks += [[s_, b] for b in range(0, n, 1024)]
            off += (n + 3) // 4 * 4
        self.total, self.nblk = off, len(blks)
        pm, pr = torch.empty(off, device=dev), torch.empty(off, device=dev)
        self.w, self.eps, self.gw = torch.zeros(off, device=dev), torch.zeros(off, device=dev), torch.zeros(off, device=dev)
        self.views = []
        for (m, kind, mu, rho), r in zip(items, rows):
            o, n = r[4], r[5]
            with torch.no_grad():
                pm[o:o + n].copy_(getattr(m, f"prior_mu_{kind}").reshape(-1))
                pr[o:o + n].copy_(getattr(m, f"prior_rho_{kind}").reshape(-1))
            setattr(m, f"prior_mu_{kind}", pm[o:o + n].view(mu.shape))
            setattr(m, f"prior_rho_{kind}", pr[o:o + n].view(mu.shape))
            wv = self.w[o:o + n].view(mu.shape)
            wv.grad = self.gw[o:o + n].view(mu.shape)
            if kind == "weight":
                m.__dict__["_bank_wv"] = wv
            self.views.append((m, kind, wv, self.eps[o:o + n].view((1,) + tuple(mu.shape))))
        self.pm, self.pr = pm, pr
        self.segs = torch.tensor(rows, dtype=torch.int64).to(dev)
        self.blks = torch.tensor(blks, dtype=torch.int32).to(dev)
        # the x6 operand forms of every 1x1 weight sample -- forward (M, K) and transposed (K, M, for the input-gradient GEMM) -- by ONE more
        # launch (bem_pack_pw_weight_x6_jobs) instead of two small packing launches per layer and iteration
        jobs, jblk, poff, self.packs = [], [], 0, []
        if ops.USE_X6:
            for (m, kind, mu, rho), r in zip(items, rows):
                if kind != "weight" or not (getattr(m, "_is_pw", False) or isinstance(m, Linear2dReparameterization)):
                    continue
                M, K = mu.shape[0], mu.shape[1]
                src = self.w.data_ptr() + 4 * r[4]
                ent = []
                for (Mj, Kj, rs, cs) in ((M, K, K, 1), (K, M, 1, K)):
                    it = ((Mj + 31) // 32) * ((Kj + 15) // 16) * 64
                    jobs.append([src, poff, Mj | (Kj << 32), rs, cs, it, 0, 0])
                    jblk += [[len(jobs) - 1, b] for b in range((it + 255) // 256)]
                    ent.append((poff, ops.packed_elems(Mj, Kj, True), (Mj, Kj)))
                    poff += ops.packed_elems(Mj, Kj, True)
                self.packs.append((m, ent))
        if jobs:
            self.parena = torch.empty(poff, device=dev, dtype=torch.float32)
            self.jobs = torch.tensor(jobs, dtype=torch.int64).to(dev)
            self.jblks, self.njblk = torch.tensor(jblk, dtype=torch.int32).to(dev), len(jblk)
            for i, (m, ent) in enumerate(self.packs):
                vs = []
                for off_, pe, mk in ent:
                    v = self.parena[off_:off_ + pe].view(1, pe)
                    v._bem_mk = mk
                    vs.append(v)
                self.packs[i] = (m, vs[0], vs[1])
        self.sig = self._signature()

    def sample(self, ctx, step, state=None):
        """One launch: every leaf's prior EMA, eps and weight sample for this forward.  The leaves are handed their views and marked as
        sampled for ``step``; their own _train_sample() then has nothing left to do."""
        if self.sig != self._signature():
            self._build()                                # a parameter, gradient or prior buffer moved (optimizer created, net.to(...))
        m0 = self.leaves[0]
        decay = lambda: min(m0.decay, (1 + m0.step) / (10 + m0.step))
        d_dev = state.slot(decay) if state is not None else None
        base = (ctx.rank << 44) | (ctx.epoch << 20) | ctx.counter
        ops.bnn_bank_sample(self, decay(), d_dev, ctx.seed, base, ctx.epoch_dev)
        ctx.counter += len(self.views)
        if self.packs:
            ops.pack_pw_weight_jobs(self.jobs, self.jblks, self.njblk, self.parena)
            for m, fwd, tr in self.packs:
                m.__dict__["_bank_wp"] = (step, fwd)                                        # gemm_weights of this training forward
                wv = m.__dict__["_bank_wv"]
                ag._derived(m).d["T"] = ((ops.WEIGHT_EPOCH[0], (wv.data_ptr(), ops.tensor_version(wv))), tr)   # autograd._pack(m, "T", [w], ...) hits
        for m, kind, wv, ev in self.views:
            if kind == "weight":
                m._ws, m._eps_w, m._bs, m._eps_b = wv, ev, None, None
                m._sample_owner = step
                step.leaves.append(m)
                if state is None:
                    m.step += 1
                else:
                    state.on_advance(m._count_step)
            else:
                m._bs, m._eps_b = wv, ev
        step.bank = self


class EvalSampleBank:
    """The N weight sets of every Bayesian tensor of a net, for one stochastic (eval) forward, drawn by ONE launch (bem_bnn_ebank_sample_f32)
    into a flat arena; the leaves then take views of it.  Leaf by leaf the Stage-I net of the Monte-Carlo loop issues 90 sampling launches
    per forward between its layers' kernels, on planes of H/16 x W/16 pixels where every dependent launch costs >= 5 us.

    Built after a leaf-by-leaf Philox forward, which records for every leaf the form it asked for (x6-packed GEMM operand or natural
    order) and the streams it drew from: the bank gives each tensor that same stream, so its values are the per-leaf path's values.
    Rebuilt when the number of sets, a parameter, a cached sigma or a leaf's form changes.  Injected eps take the per-leaf path."""

    def __init__(self, net):
        self.leaves = [m for m in net.modules() if isinstance(m, _BayesBase)]
        self.sig = None

    def usable(self, ctx):
        return (ctx is not None and ctx.eps is None and ctx.epoch_dev is None and bool(self.leaves) and ops.USE_X6
                and all(not m.training and not m.deterministic and "_eval_draw" in m.__dict__ for m in self.leaves))

    def _signature(self, ns):
        sig = [ns]
        for m in self.leaves:
            d = m._eval_draw
            sig += [d["mk"], d["w"], d["b"], m.mu_weight.data_ptr(), (m._sigma() if d["mk"] else m.rho_weight).data_ptr()]
            if m.bias:
                sig += [m.mu_bias.data_ptr(), m.rho_bias.data_ptr()]
        return tuple(sig)

    def _build(self, ns):
        dev = self.leaves[0].mu_weight.device
        rows, blks, off, self.views = [], [], 0, {}
        order = []
        for m in self.leaves:
            d = m._eval_draw
            if d["mk"] is not None:
                M, K = d["mk"]
                pe = ops.packed_elems(M, K, True)
                items = ns * ((M + 31) // 32) * ((K + 15) // 16) * 64
                rows.append([m.mu_weight.data_ptr(), m._sigma().data_ptr(), off, M * K, M | (K << 32), d["w"], items, ns * M * K])
                wv = (off, ns * pe, (ns, pe), (M, K))
                off += ns * pe
            else:
                n = m.mu_weight.numel()
                rows.append([m.mu_weight.data_ptr(), m.rho_weight.data_ptr(), off, n, 0, d["w"], (ns * n + 3) // 4, ns * n])
                wv = (off, ns * n, (ns,) + tuple(m.mu_weight.shape), None)
                off += (ns * n + 3) // 4 * 4
            bv = None
            if m.bias:
                n = m.mu_bias.numel()
                rows.append([m.mu_bias.data_ptr(), m.rho_bias.data_ptr(), off, n, 0, d["b"], (ns * n + 3) // 4, ns * n])
                bv = (off, ns * n, (ns,) + tuple(m.mu_bias.shape), None)
                off += (ns * n + 3) // 4 * 4
            order.append((m, wv, bv))
        for s_, r in enumerate(rows):
            blks += [[s_, b] for b in range((r[6] + 255) // 256)]
        if sorted(r[5] for r in rows) != list(range(1, len(rows) + 1)):
            return False                                 # the recorded forward did not draw every Bayesian tensor exactly once: stay leaf by leaf
        self.arena = torch.empty(off, device=dev, dtype=torch.float32)
        for m, wv, bv in order:
            w = self.arena[wv[0]:wv[0] + wv[1]].view(wv[2])
            if wv[3] is not None:
                w._bem_mk = wv[3]
            b = None if bv is None else self.arena[bv[0]:bv[0] + bv[1]].view(bv[2])
            self.views[id(m)] = (m._eval_draw["mk"], w, b)
        self.nblk, self.nrows = len(blks), len(rows)
        self.segs = torch.tensor(rows, dtype=torch.int64).to(dev)
        self.blks = torch.tensor(blks, dtype=torch.int32).to(dev)
        self.sig = self._signature(ns)
        return True

    def sample(self, ctx):
        ns = ctx.nsets
        if self.sig != self._signature(ns) and not self._build(ns):
            self.sig = None
            return
        base = (ctx.rank << 44) | (ctx.epoch << 20) | ctx.counter
        ops.bnn_ebank_sample(self, ctx.seed, base)
        ctx.counter += self.nrows
        ctx.bank = self

    def take(self, leaf, packed_mk, ns):
        mk, w, b = self.views[id(leaf)]
        if mk != (packed_mk if (packed_mk is not None and ops.USE_X6) else None):
            raise BemNativeError("EvalSampleBank: a leaf asked for its weights in another form than in the recorded forward")
        return w, b, ns


class Conv2dReparameterization(_BayesBase):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 sigma_init=0.05, decay=0.9998):
        super().__init__()
        self._init_common(sigma_init, decay, bias)
        ks = kernel_size if isinstance(kernel_size, (tuple, list)) else (kernel_size, kernel_size)
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride, self.padding, self.dilation, self.groups = stride, padding, dilation, groups
        shp = (out_channels, in_channels // groups, ks[0], ks[1])
        self.mu_weight = nn.Parameter(torch.empty(shp))
        self.rho_weight = nn.Parameter(torch.empty(shp))
        if bias:
            self.mu_bias = nn.Parameter(torch.empty(out_channels))
            self.rho_bias = nn.Parameter(torch.empty(out_channels))
        nn.init.kaiming_normal_(self.mu_weight, mode="fan_in", nonlinearity="leaky_relu")
        self.rho_weight.data.fill_(self._rho_init())
        if bias:
            self.mu_bias.data.zero_()
            self.rho_bias.data.fill_(self._rho_init())
        self._register_priors()
        self._is_dw = groups == in_channels and groups == out_channels and tuple(ks) == (3, 3)
        self._is_pw = groups == 1 and tuple(ks) == (1, 1)
        if not (self._is_dw or self._is_pw):
            raise NotImplementedError("Bayesian conv: only 1x1 dense and 3x3 depthwise occur on the BEM path")

    # pointwise interface
    def gemm_weights(self, B):
        if self.deterministic or not ops.USE_X6 or self.training:
            w, b, ns = self._sampled(B)
            bw = self.__dict__.get("_bank_wp")
            if self.training and bw is not None and bw[0] is self._sample_owner and not self.deterministic:
                return bw[1], b                      # packed by the bank's one launch for this forward
            return ops.pack_pw_weight(w.reshape(ns, self.out_channels, self.in_channels).contiguous()), b
        Wp, b, _ = self._sampled(B, (self.out_channels, self.in_channels))
        return Wp, b

    # depthwise interface
    def dw_weights(self, B):
        w, b, ns = self._sampled(B)
        return (w if ns > 1 else w[0]), (b if (b is None or ns > 1) else b[0])


class Linear2dReparameterization(_BayesBase):
    def __init__(self, in_features, out_features, bias=True, sigma_init=0.05, decay=0.9998):
        super().__init__()
        self._init_common(sigma_init, decay, bias)
        self.in_features, self.out_features = in_features, out_features
        self.mu_weight = nn.Parameter(torch.empty(out_features, in_features))
        self.rho_weight = nn.Parameter(torch.empty(out_features, in_features))
        if bias:
            self.mu_bias = nn.Parameter(torch.empty(out_features))
            self.rho_bias = nn.Parameter(torch.empty(out_features))
        nn.init.xavier_uniform_(self.mu_weight)
        self.rho_weight.data.fill_(self._rho_init())
        if bias:
            self.mu_bias.data.zero_()
            self.rho_bias.data.fill_(self._rho_init())
        self._register_priors()

    def gemm_weights(self, B):
        if self.deterministic or not ops.USE_X6 or self.training:
            w, b, ns = self._sampled(B)
            bw = self.__dict__.get("_bank_wp")
            if self.training and bw is not None and bw[0] is self._sample_owner and not self.deterministic:
                return bw[1], b
            return ops.pack_pw_weight(w.contiguous()), b
        Wp, b, _ = self._sampled(B, (self.out_features, self.in_features))
        return Wp, b


# ------------------------------------------------------------------------------------------------
# SS2D / gdMlp / VSSBlock  (basicsr/vmamba/models/vmamba.py:116-133, 438-716, 1241-1334)
# ------------------------------------------------------------------------------------------------
def _out_features(m):
    return m.out_features if hasattr(m, "out_features") else m.out_channels


class gdMlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0, channels_first=False):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.project_in = PwConv2d(in_features, hidden_features * 2)
        self.dwconv = DwConv2d(hidden_features * 2)
        self.project_out = PwConv2d(hidden_features, out_features)
        self.act = act_layer()
        self._cache = _Cache()

    def forward_fused(self, x, norm: LayerNorm2d):
        """x + project_out(GELU(h1) * h2), h = dwconv(project_in(LN(x)))."""
        B, C = x.shape[0], x.shape[1]
        Hd = self.project_in.out_channels // 2
        if isinstance(self.project_in, PwConv2d) and isinstance(self.dwconv, DwConv2d) and isinstance(self.project_out, PwConv2d) \
                and ops.gdmlp_x6_supported(C, Hd):
            # the whole branch in one kernel (bem_gdmlp_x6_f32): neither the 2Hd-channel nor the Hd-channel tensor reaches HBM
            pi, dw, po = self.project_in, self.dwconv, self.project_out

            def prep():
                perm = ops.gate_interleave(Hd, pi.weight.device)
                bg = pi.bias.detach()[perm].contiguous() if pi.bias is not None else torch.zeros(2 * Hd, device=pi.weight.device)
                return (ops.pack_pw_weight(pi.weight.detach().reshape(2 * Hd, C)[perm].contiguous(), x6=True), bg,
                        ops.dw_gate_params10(dw.weight.detach(), None if dw.bias is None else dw.bias.detach(), Hd),
                        ops.pack_pw_weight(po.weight.detach().reshape(po.out_channels, Hd).contiguous(), x6=True),
                        None if po.bias is None else po.bias.detach().contiguous())
            Wg, bg, w10, Wo, bo = self._cache.get("gdmlp_x6", [t for t in (pi.weight, pi.bias, dw.weight, dw.bias, po.weight, po.bias) if t is not None], prep)
            return ops.gdmlp_x6(x, norm.weight.detach(), norm.bias.detach(), norm.eps, Wg, bg, w10, Wo, bo, Hd)
        Wp, b = self.project_in.gemm_weights(B)
        t = ops.pw_gemm(x, Wp, _out_features(self.project_in), ln=(norm.weight.detach(), norm.bias.detach()),
                        ln_eps=norm.eps, bias=b)
        w, b = self.dwconv.dw_weights(B)
        g = ops.dwconv3x3(t, w, b, mode=2)
        Wp, b = self.project_out.gemm_weights(B)
        return ops.pw_gemm(g, Wp, _out_features(self.project_out), bias=b, res=x)


class SS2D(nn.Module):
    """forward_type 'v05_noz' only (the one every arch on this path selects, UNet_arch.py:219)."""

    def __init__(self, d_model=96, d_state=16, ssm_ratio=2.0, dt_rank="auto", act_layer=nn.SiLU, d_conv=3,
                 conv_bias=True, dropout=0.0, bias=False, dt_min=0.001, dt_max=0.1, dt_init="random", dt_scale=1.0,
                 dt_init_floor=1e-4, initialize="v0", forward_type="v2", channel_first=False, **kwargs):
        super().__init__()
        if forward_type != "v05_noz" or not channel_first or d_conv != 3 or initialize != "v0":
            raise NotImplementedError("SS2D: only forward_type='v05_noz', channel_first, d_conv=3, init v0 are on the BEM path")
        if int(d_state) != 1:
            raise NotImplementedError("SS2D: the fused HIP scan is specialised for d_state = 1 (every shipped option file)")
        d_inner = int(ssm_ratio * d_model)
        R = math.ceil(d_model / 16) if dt_rank == "auto" else dt_rank
        self.d_inner, self.dt_rank, self.d_state = d_inner, R, 1
        self.in_proj = Linear2d(d_model, d_inner, bias=bias)
        self.act = act_layer()
        self.conv2d = DwConv2d(d_inner, bias=conv_bias)
        K, N = 4, 1
        self.x_proj_weight = nn.Parameter(torch.stack([nn.Linear(d_inner, R + 2 * N, bias=False).weight.detach() for _ in range(K)], 0))
        self.out_proj = Linear2d(d_inner, d_model, bias=bias)
        # mamba_init.init_dt_A_D (vmamba.py:222-289)
        std = R ** -0.5 * dt_scale
        dtw, dtb = [], []
        for _ in range(K):
            w = torch.empty(d_inner, R).uniform_(-std, std)
            dt = torch.exp(torch.rand(d_inner) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min)).clamp(min=dt_init_floor)
            dtw.append(w)
            dtb.append(dt + torch.log(-torch.expm1(-dt)))
        self.dt_projs_weight = nn.Parameter(torch.stack(dtw, 0))
        self.dt_projs_bias = nn.Parameter(torch.stack(dtb, 0))
        self.A_logs = nn.Parameter(torch.log(torch.arange(1, N + 1, dtype=torch.float32)).view(1, -1).repeat(K * d_inner, 1).contiguous())
        self.Ds = nn.Parameter(torch.ones(K * d_inner))
        self.out_norm = LayerNorm2d(d_inner)
        self._cache = _Cache()

    def _scan_params(self):
        R = self.dt_rank

        def prep():
            xw = self.x_proj_weight.detach()                                # (4, R+2, C)
            # one x_proj GEMM over the row-major planes for all four directions: rows [dir 0 | dir 2 | dir 1 | dir 3]
            wall = ops.pack_pw_weight(torch.cat([xw[0], xw[2], xw[1], xw[3]], 0).contiguous())
            A = (-torch.exp(self.A_logs.detach().float())).reshape(-1).contiguous()
            return (wall, self.dt_projs_weight.detach().contiguous(), self.dt_projs_bias.detach().contiguous(), A,
                    self.Ds.detach().float().contiguous())
        return self._cache.get("scan", [self.x_proj_weight, self.dt_projs_weight, self.dt_projs_bias, self.A_logs, self.Ds], prep)

    def forward_fused(self, x, norm: LayerNorm2d):
        """x + out_proj(out_norm(merge(scan(SiLU(dw(in_proj(LN(x))))))))  (vmamba.py:700-716 + 547-698)."""
        B, C, H, W = x.shape
        Ci, R, L = self.d_inner, self.dt_rank, H * W
        wall, dtw, dtb, A, Ds = self._scan_params()
        front = SCAN_RM and ops.ss2d_scan_rm_supported(H, W, R) and ops.ss2d_front_supported(C, 4 * (R + 2)) and Ci == C \
            and type(self.in_proj) is Linear2d and type(self.conv2d) is DwConv2d and ops.USE_X6
        if front:
            # LayerNorm + in_proj + depthwise 3x3 + SiLU + x_proj in one kernel (bem_ss2d_front_x6_f32): the in_proj output stays in LDS
            Wpi, bi = self.in_proj.gemm_weights(B)
            w, bw = self.conv2d.dw_weights(B)
            xc, xd = ops.ss2d_front(x, norm.weight.detach(), norm.bias.detach(), norm.eps, Wpi, bi, w, bw, wall, 4 * (R + 2))
        else:
            Wp, b = self.in_proj.gemm_weights(B)
            t = ops.pw_gemm(x, Wp, Ci, ln=(norm.weight.detach(), norm.bias.detach()), ln_eps=norm.eps, bias=b)
            w, b = self.conv2d.dw_weights(B)
            xc = ops.dwconv3x3(t, w, b, mode=1)
        Wp, b = self.out_proj.gemm_weights(B)
        on = self.out_norm
        if SCAN_RM and ops.ss2d_scan_rm_supported(H, W, R):
            # row-major scan: no transposed copy of xc, y1 comes back row-major (the column orientation goes through LDS)
            if not front:
                xd = ops.pw_gemm(xc, wall, 4 * (R + 2))
            xd1 = ops.transpose_plane_slice(xd, 2 * (R + 2), 2 * (R + 2))
            y0, y1 = ops.ss2d_scan_rm(xc, xd.view(B, 4, R + 2, L)[:, :2], xd1.view(B, 2, R + 2, L), dtw, dtb, A, Ds)
            return ops.pw_gemm(y0, Wp, _out_features(self.out_proj), x2=y1, in_mode=1,
                               ln=(on.weight.detach(), on.bias.detach()), ln_eps=on.eps, bias=b, res=x)
        xcT = ops.transpose_planes(xc)
        # x_dbl of the column-major directions = the row-major GEMM's rows in transposed pixel order: 2 (R+2) planes to
        # transpose instead of a second GEMM pass over the C planes of xcT
        xd = ops.pw_gemm(xc, wall, 4 * (R + 2))                                   # (B, 4(R+2), H, W)
        xd1 = ops.transpose_plane_slice(xd, 2 * (R + 2), 2 * (R + 2))             # (B, 2(R+2), W, H)
        y0, y1 = ops.ss2d_scan(xc.view(B, Ci, L), xcT.view(B, Ci, L), xd.view(B, 4, R + 2, L)[:, :2], xd1.view(B, 2, R + 2, L),
                               dtw, dtb, A, Ds)
        y1r = ops.transpose_planes(y1.view(B, Ci, W, H))
        return ops.pw_gemm(y0.view(B, Ci, H, W), Wp, _out_features(self.out_proj), x2=y1r, in_mode=1,
                           ln=(on.weight.detach(), on.bias.detach()), ln_eps=on.eps, bias=b, res=x)

    def forward(self, x):
        raise BemNativeError("SS2D runs fused with its VSSBlock (norm prologue + residual epilogue); call the block")


class VSSBlock(nn.Module):
    def __init__(self, hidden_dim=0, drop_path=0.0, norm_layer=LayerNorm2d, channel_first=True, ssm_d_state=16,
                 ssm_ratio=2.0, ssm_dt_rank="auto", ssm_act_layer=nn.SiLU, ssm_conv=3, ssm_conv_bias=True,
                 ssm_drop_rate=0.0, ssm_init="v0", forward_type="v2", mlp_ratio=4.0, mlp_act_layer=nn.GELU,
                 mlp_drop_rate=0.0, mlp_type="mlp", use_checkpoint=False, post_norm=False, grid_size=None, **kwargs):
        super().__init__()
        if post_norm or grid_size or drop_path or mlp_type != "gdmlp" or not channel_first:
            raise NotImplementedError("VSSBlock: only pre-norm, gdmlp, channel-first, drop_path=0 are on the BEM path")
        self.norm = LayerNorm2d(hidden_dim)
        self.op = SS2D(d_model=hidden_dim, d_state=ssm_d_state, ssm_ratio=ssm_ratio, dt_rank=ssm_dt_rank,
                       act_layer=ssm_act_layer, d_conv=ssm_conv, conv_bias=ssm_conv_bias, dropout=ssm_drop_rate,
                       initialize=ssm_init, forward_type=forward_type, channel_first=channel_first)
        self.drop_path = nn.Identity()
        self.norm2 = LayerNorm2d(hidden_dim)
        self.mlp = gdMlp(in_features=hidden_dim, hidden_features=int(hidden_dim * mlp_ratio), act_layer=mlp_act_layer,
                         drop=mlp_drop_rate, channels_first=channel_first)

    def forward(self, x):
        _need_cuda(x)
        if grad_mode(self):
            return ag.vssblock(self, x)
        x = x.contiguous()
        x = self.op.forward_fused(x, self.norm)
        return self.mlp.forward_fused(x, self.norm2)


def make_vss_level(dim, num_block, d_state, ssm_ratio, mlp_ratio, mlp_type):
    return nn.Sequential(*[VSSBlock(hidden_dim=dim, drop_path=0, norm_layer=LayerNorm2d, channel_first=True,
                                    ssm_d_state=d_state, ssm_ratio=ssm_ratio, ssm_dt_rank="auto", ssm_act_layer=nn.SiLU,
                                    ssm_conv=3, ssm_conv_bias=False, ssm_drop_rate=0, ssm_init="v0",
                                    forward_type="v05_noz", mlp_ratio=mlp_ratio, mlp_act_layer=nn.GELU,
                                    mlp_drop_rate=0.0, mlp_type=mlp_type, use_checkpoint=False, post_norm=False)
                           for _ in range(num_block)])


def set_module_paths(root: nn.Module):
    """Record each leaf's dotted path (used as the key of injected epsilon draws)."""
    for name, m in root.named_modules():
        if hasattr(m, "module_path"):
            m.module_path = name

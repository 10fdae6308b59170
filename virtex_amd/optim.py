"""Optimizer tail of the reference training step, restated for the native modules
(/root/reference/scripts/pretrain_virtex.py:157-162 and virtex/factories.py:503-545):

    clip_grad_norm_(model.parameters(), 10.0)  ->  SGD(momentum .9, per-tensor lr / weight decay)
    ->  Lookahead(k = 5, alpha = .5)  ->  LinearWarmupCosineAnnealingLR.

Parameter grouping follows the reference exactly: one group per named parameter, weight decay 0
for names matching ``.*textual.(embedding|transformer).*(norm.*|bias)``, CNN_LR for names
containing ``cnn``.  `FusedPretrainOptimizer` (what bench.py drives) executes the whole tail with two HIP
kernels over flat buffers (csrc/optim.hip); `PretrainOptimizer` is the same arithmetic on stock
``torch.optim.SGD(foreach=True)`` for models without a flat gradient buffer.  Everything is device-side, no host
synchronisation happens inside ``step()``.  State dicts of both are indexed in ``model.named_parameters()``
order -- the order the reference's OptimizerFactory builds its one-tensor param groups in
(virtex/factories.py:529-533) -- so checkpoints interchange with the reference's Lookahead(SGD).
"""
import ctypes
import math
import re
import weakref
from typing import Iterable, List, Tuple

import torch

from .streams import branch_stream, wgrad_stream

NO_DECAY = r".*textual.(embedding|transformer).*(norm.*|bias)"

# ---- the device word every dropout kernel mixes into its seed (vtx_set_dropout_epoch): one per (loaded library, device),
# reference-counted.  The C side keeps ONE pointer per loaded library (one process drives one GPU: DESIGN.md 7), so a
# registration for a second device through the SAME library while the first is held is refused instead of silently re-pointing
# the first device's kernels.  Two libraries in one process (the HIP build and the tests' emulator build) have independent
# globals and therefore independent registrations: the key carries the library handle.
_epoch_words = {}     # (library handle, device) -> tensor (kept for the life of the process: 4 bytes; the address never dangles)
_epoch_refs = {}      # (library handle, device) -> number of optimizers with an enabled device schedule
# Every optimizer on the device-side schedule advances the word once per step of ITS OWN (N optimizers stepping in one process
# give N increments per iteration).  That is deliberate: a launch recording bakes its optimizer's increment into the recorded
# list, so "one owner advances" would silently stop the masks of a replay whose optimizer is not the owner, or freeze them when
# the owner is released.  The masks only have to CHANGE from step to step and be equal between the forward and the backward
# of one step, which any monotone word gives.


def _epoch_key(device):
    from . import _lib
    handle = _lib.lib()
    return (handle._handle, torch.device(device)), handle


def _epoch_acquire(device, ops):
    key, handle = _epoch_key(device)
    held = [k for k, n in _epoch_refs.items() if n > 0 and k[0] == key[0] and k[1] != key[1]]
    if held:
        raise RuntimeError(f"a device dropout epoch is registered for {held[0][1]}; the library holds one registration per "
                           f"process (one process per GPU) -- disable that optimizer's device schedule first")
    word = _epoch_words.get(key)
    if word is None:
        word = _epoch_words[key] = torch.zeros(1, dtype=torch.int32, device=device)
    if _epoch_refs.get(key, 0) == 0:
        ops.set_dropout_epoch(word)
    _epoch_refs[key] = _epoch_refs.get(key, 0) + 1
    return word, key, handle


def _epoch_release(key, handle):
    n = _epoch_refs.get(key, 0) - 1
    _epoch_refs[key] = max(n, 0)
    if n == 0:
        try:
            handle.vtx_set_dropout_epoch(ctypes.c_void_p(0))     # on the library that holds the registration
        except Exception:        # interpreter shutdown: the library may already be gone
            pass


def param_groups(named_parameters: Iterable[Tuple[str, torch.nn.Parameter]], cnn_lr=0.2, lr=0.001,
                 weight_decay=1e-4, no_decay=NO_DECAY) -> List[dict]:
    groups = []
    for name, p in named_parameters:
        wd = 0.0 if re.match(no_decay, name) else weight_decay
        groups.append({"params": [p], "lr": cnn_lr if "cnn" in name else lr, "weight_decay": wd})
    return groups


def lr_multiplier(step: int, total_steps: int, warmup_steps: int) -> float:
    """Linear warm-up then cos^2 decay (virtex/optim/lr_scheduler.py:174-183)."""
    if step < warmup_steps:
        return step / float(max(1, warmup_steps))
    frac = (step - warmup_steps) / (total_steps - warmup_steps)
    return max(0.0, math.cos(frac * (math.pi / 2)) ** 2)


class PretrainOptimizer:
    """SGD + Lookahead + warm-up/cosine schedule + global-norm clipping in one object."""

    def __init__(self, model: torch.nn.Module, cnn_lr=0.2, lr=0.001, weight_decay=1e-4, momentum=0.9,
                 clip_norm=10.0, lookahead_k=5, lookahead_alpha=0.5, total_steps=500000, warmup_steps=10000,
                 start_step=0):
        self.model = model
        groups = param_groups(model.named_parameters(), cnn_lr, lr, weight_decay)
        self.base_lrs = [g["lr"] for g in groups]
        self.sgd = torch.optim.SGD(groups, momentum=momentum, foreach=True)
        self.params = [g["params"][0] for g in self.sgd.param_groups]
        self.clip_norm, self.k, self.alpha = clip_norm, lookahead_k, lookahead_alpha
        self.slow = [p.detach().clone() for p in self.params]
        self.kc, self.step_idx = 0, start_step
        self.total_steps, self.warmup_steps = total_steps, warmup_steps
        self._set_lr()

    def _set_lr(self):
        mult = lr_multiplier(self.step_idx, self.total_steps, self.warmup_steps)
        for g, base in zip(self.sgd.param_groups, self.base_lrs):
            g["lr"] = base * mult

    def zero_grad(self):
        self.sgd.zero_grad(set_to_none=False)

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0):
        """`grad_scale` multiplies every gradient first (1/world_size after a SUM all-reduce)."""
        branch_stream.join(self.params[0].device)
        wgrad_stream.join(self.params[0].device)
        grads = [p.grad for p in self.params]
        if grad_scale != 1.0:
            torch._foreach_mul_(grads, grad_scale)
        if self.clip_norm is not None:
            torch.nn.utils.clip_grad_norm_(self.params, self.clip_norm, foreach=True)
        self.sgd.step()
        self.kc += 1
        if self.kc >= self.k:
            self.kc = 0
            data = [p.data for p in self.params]
            torch._foreach_mul_(data, self.alpha)
            torch._foreach_add_(data, self.slow, alpha=1.0 - self.alpha)
            torch._foreach_copy_(self.slow, data)
            for p in self.params:       # written through .data: tell the version-keyed caches
                torch.autograd.graph.increment_version(p)
        self.step_idx += 1
        self._set_lr()

    # -- checkpointing: the layout of torch.optim.SGD.state_dict(), which is what the reference's
    #    Lookahead.state_dict() returns (virtex/optim/lookahead.py:74-86) and CheckpointManager serialises
    #    (virtex/utils/checkpointing.py:112-123).  Like the reference, slow weights are NOT saved: loading
    #    re-seeds them from the current parameters.
    def state_dict(self):
        groups = []
        for i, (g, base) in enumerate(zip(self.sgd.param_groups, self.base_lrs)):
            d = {k: v for k, v in g.items() if k != "params"}
            d["initial_lr"] = base
            d["params"] = [i]
            groups.append(d)
        state = {i: {"momentum_buffer": self.sgd.state[p]["momentum_buffer"]}
                 for i, p in enumerate(self.params) if "momentum_buffer" in self.sgd.state.get(p, {})}
        return {"state": state, "param_groups": groups,
                "virtex_amd": {"step": self.step_idx, "k_counter": self.kc}}

    def load_state_dict(self, sd):
        for i, p in enumerate(self.params):
            st = sd["state"].get(i, sd["state"].get(str(i)))
            if st is not None and st.get("momentum_buffer") is not None:
                self.sgd.state[p]["momentum_buffer"] = st["momentum_buffer"].to(p.device, p.dtype).clone()
        extra = sd.get("virtex_amd", {})
        self.step_idx = int(extra.get("step", self.step_idx))
        self.kc = int(extra.get("k_counter", 0))
        with torch.no_grad():
            for s_, p in zip(self.slow, self.params):
                s_.copy_(p.data)
        self._set_lr()

    def load_slow_weights(self):
        """Evaluate on the slow weights (lookahead.py:104-118); undo with restore_fast_weights()."""
        self._backup = [p.detach().clone() for p in self.params]
        with torch.no_grad():
            for p, s_ in zip(self.params, self.slow):
                p.copy_(s_)          # p.copy_ (not p.data.copy_): bumps p._version, which the compute-weight caches key on

    def restore_fast_weights(self):
        with torch.no_grad():
            for p, b in zip(self.params, self._backup):
                p.copy_(b)
        del self._backup


class FusedPretrainOptimizer:
    """Same arithmetic as `PretrainOptimizer`, executed by two HIP kernels over flat buffers
    (csrc/optim.hip): parameters, gradients (the data-parallel buckets), momentum and Lookahead slow
    weights share one layout; per-parameter learning rate / weight decay come from a segment table.
    No host synchronisation: the clip coefficient is computed on the device from the squared norm."""

    def __init__(self, model: torch.nn.Module, buckets, cnn_lr=0.2, lr=0.001, weight_decay=1e-4, momentum=0.9,
                 clip_norm=10.0, lookahead_k=5, lookahead_alpha=0.5, total_steps=500000, warmup_steps=10000,
                 start_step=0, no_decay=NO_DECAY):
        from . import _lib, ops

        self.ops = ops
        self.buckets = buckets
        names = {p: n for n, p in model.named_parameters()}
        params = buckets.params
        # state dicts are indexed like the reference's param groups: model.named_parameters() order (frozen
        # parameters included, they simply carry no momentum); slot = position in the flat buffers
        self.named = [p for _, p in model.named_parameters()]
        self._slot_of = {p: i for i, p in enumerate(params)}
        dev = buckets.flat.device
        lib = _lib.lib()
        chunk = lib.vtx_optim_chunk_elems()
        self.flat_p = torch.empty_like(buckets.flat)
        offs, lens, segs, seg_lr, seg_wd = [], [], [], [], []
        off = 0
        with torch.no_grad():
            for si, p in enumerate(params):
                n = p.numel()
                slot = self.flat_p[off: off + n]
                if p.dim() == 4 and p.stride(1) == 1 and p.shape[1] > 1:     # (KO,R,S,C) physical
                    O, I, R, S = p.shape
                    view = slot.view(O, R, S, I).permute(0, 3, 1, 2)
                else:
                    view = slot.view(p.shape)
                view.copy_(p.data)
                p.data = view                      # the parameter now lives inside the flat buffer
                name = names[p]
                seg_lr.append(cnn_lr if "cnn" in name else lr)
                seg_wd.append(0.0 if re.match(no_decay, name) else weight_decay)
                for c0 in range(0, n, chunk):
                    offs.append(off + c0); lens.append(min(chunk, n - c0)); segs.append(si)
                off += n
        self.n_elems = int(sum(lens))
        self.chunk_off = torch.tensor(offs, dtype=torch.int64, device=dev)
        self.chunk_len = torch.tensor(lens, dtype=torch.int32, device=dev)
        self.chunk_seg = torch.tensor(segs, dtype=torch.int32, device=dev)
        self.seg_lr = torch.tensor(seg_lr, dtype=torch.float32, device=dev)
        self.seg_wd = torch.tensor(seg_wd, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_slow = self.flat_p.clone()
        self.partials = torch.empty(1024, dtype=torch.float32, device=dev)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.momentum, self.clip_norm, self.k, self.alpha = momentum, clip_norm, lookahead_k, lookahead_alpha
        self.kc, self.step_idx = 0, start_step
        self.dev = None              # device-side schedule state (enable_device_schedule)
        self.total_steps, self.warmup_steps = total_steps, warmup_steps
        self.cnn_lr, self.lr, self.weight_decay, self.no_decay = cnn_lr, lr, weight_decay, no_decay
        self.model_named = model.named_parameters

    def zero_grad(self):
        self.buckets.zero()

    def grad_norm(self) -> torch.Tensor:
        """Device tensor holding ||g||_2 of the last step (before grad_scale / clipping)."""
        return self.sumsq.sqrt()

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0):
        g = self.buckets.flat
        branch_stream.join(g.device)
        wgrad_stream.join(g.device)
        if self.clip_norm:
            self.ops.sumsq(g, self.partials, self.sumsq)
        if self.dev is not None:
            self._device_schedule_step(g, grad_scale)
            return
        self.kc += 1
        look = self.kc >= self.k
        if look:
            self.kc = 0
        mult = lr_multiplier(self.step_idx, self.total_steps, self.warmup_steps)
        self.ops.sgd_lookahead_step(self.flat_p, g, self.flat_m, self.flat_slow, self.chunk_off, self.chunk_len,
                                    self.chunk_seg, self.seg_lr, self.seg_wd, mult, self.momentum, grad_scale,
                                    self.sumsq, self.clip_norm, look, self.alpha, n_elems=self.n_elems)
        # the kernel wrote the parameters behind torch's back: bump their version counters so that everything
        # keyed on them (the eval-mode folded weights, autograd's saved-tensor checks) sees the update
        for p in self.buckets.params:
            torch.autograd.graph.increment_version(p)
        self.step_idx += 1

    # -- device-side schedule (what a captured hipGraph of the step needs: virtex_amd/graph.py).  The step index, the
    #    Lookahead counter, the LR multiplier and the dropout epoch live in device memory and are advanced by (captured)
    #    device operations; the by-value arguments of the eager path would be frozen at capture time.
    def enable_device_schedule(self):
        if self.dev is not None:
            return
        d = self.flat_p.device
        # The dropout epoch is ONE word per device, owned by this module and never freed while any optimizer holds a
        # registration (_epoch_acquire): the library dereferences the registered address from every dropout kernel of
        # every model, so the word must not share the lifetime of one optimizer, and a second optimizer's enable /
        # disable must not re-point or null what the first one's replays read.
        self.dev = {"step": torch.tensor([float(self.step_idx)], dtype=torch.float32, device=d),
                    "kc": torch.tensor([float(self.kc)], dtype=torch.float32, device=d),
                    "sched": torch.zeros(2, dtype=torch.float32, device=d),
                    "epoch": None}
        self.dev["epoch"], key, handle = _epoch_acquire(d, self.ops)
        self._epoch_release = weakref.finalize(self, _epoch_release, key, handle)

    def disable_device_schedule(self):
        if self.dev is None:
            return
        self.sync_host()
        self._epoch_release()            # detaches the finalizer; the registration goes when the LAST holder lets go
        self.dev = None

    def sync_host(self):
        """After graph replays: pull the device-side counters back into the host mirrors and tell autograd / the weight
        caches that the parameters have changed (one device synchronisation)."""
        if self.dev is not None:
            self.step_idx = int(round(self.dev["step"].item()))
            self.kc = int(round(self.dev["kc"].item()))
        self._touch()

    def _device_schedule_step(self, g, grad_scale):
        dv = self.dev
        st, kc, sched = dv["step"], dv["kc"], dv["sched"]
        warm = float(max(1, self.warmup_steps))
        span = float(max(1, self.total_steps - self.warmup_steps))
        # lr_multiplier() on the device: linear warm-up, then cos^2 decay (virtex/optim/lr_scheduler.py:174-183)
        cosm = torch.cos(((st - float(self.warmup_steps)) / span).clamp_(min=0.0) * (math.pi / 2)).square_()
        mult = torch.where(st < float(self.warmup_steps), st / warm, cosm)
        kc.add_(1.0)
        look = (kc >= float(self.k)).to(torch.float32)
        kc.mul_(1.0 - look)
        sched[0:1].copy_(mult)
        sched[1:2].copy_(look)
        self.ops.sgd_lookahead_step_dev(self.flat_p, g, self.flat_m, self.flat_slow, self.chunk_off, self.chunk_len,
                                        self.chunk_seg, self.seg_lr, self.seg_wd, sched, self.momentum, grad_scale,
                                        self.sumsq, self.clip_norm, self.alpha, n_elems=self.n_elems)
        st.add_(1.0)
        dv["epoch"].add_(1)
        capturing = self.flat_p.is_cuda and torch.cuda.is_current_stream_capturing()
        if not capturing:                # host mirrors (a replayed graph never runs this: sync_host() catches up)
            for p in self.buckets.params:
                torch.autograd.graph.increment_version(p)
            self.step_idx += 1

    # -- checkpointing: same torch.optim.SGD layout as PretrainOptimizer.state_dict() (momentum buffers are
    #    views of the flat buffer in each parameter's logical shape), so checkpoints move freely between
    #    the reference's Lookahead(SGD), PretrainOptimizer and this class.
    def _views(self, flat):
        out, off = [], 0
        for p in self.buckets.params:
            n = p.numel()
            slot = flat[off: off + n]
            if p.dim() == 4 and p.stride(1) == 1 and p.shape[1] > 1:
                O, I, R, S = p.shape
                out.append(slot.view(O, R, S, I).permute(0, 3, 1, 2))
            else:
                out.append(slot.view(p.shape))
            off += n
        return out

    def state_dict(self):
        if self.dev is not None:
            self.sync_host()
        mult = lr_multiplier(self.step_idx, self.total_steps, self.warmup_steps)
        names = {p: n for n, p in self.model_named()}
        views = self._views(self.flat_m)
        groups, state = [], {}
        for i, p in enumerate(self.named):
            name = names[p]
            lr = self.cnn_lr if "cnn" in name else self.lr
            wd = 0.0 if re.match(self.no_decay, name) else self.weight_decay
            groups.append({"lr": lr * mult, "momentum": self.momentum, "dampening": 0, "weight_decay": wd,
                           "nesterov": False, "maximize": False, "foreach": None, "differentiable": False,
                           "fused": None, "initial_lr": lr, "params": [i]})
            slot = self._slot_of.get(p)
            if slot is not None:
                state[i] = {"momentum_buffer": views[slot]}
        return {"state": state, "param_groups": groups,
                "virtex_amd": {"step": self.step_idx, "k_counter": self.kc}}

    @torch.no_grad()
    def load_state_dict(self, sd):
        if len(sd["param_groups"]) != len(self.named):
            raise ValueError(f"optimizer state has {len(sd['param_groups'])} parameter groups, the model has "
                             f"{len(self.named)} named parameters")
        views = self._views(self.flat_m)
        for i, p in enumerate(self.named):
            slot = self._slot_of.get(p)
            if slot is None:
                continue
            st = sd["state"].get(i, sd["state"].get(str(i)))
            buf = st.get("momentum_buffer") if st is not None else None
            if buf is not None:
                if tuple(buf.shape) != tuple(p.shape):
                    raise ValueError(f"momentum buffer {i} has shape {tuple(buf.shape)}, parameter {tuple(p.shape)}")
                views[slot].copy_(buf)
            else:
                views[slot].zero_()
        extra = sd.get("virtex_amd", {})
        self.step_idx = int(extra.get("step", self.step_idx))
        self.kc = int(extra.get("k_counter", 0))
        if self.dev is not None:
            self.dev["step"].fill_(float(self.step_idx)); self.dev["kc"].fill_(float(self.kc))
        self.flat_slow.copy_(self.flat_p)        # reference semantics: slow weights restart from the loaded ones

    def _touch(self):
        for p in self.buckets.params:
            torch.autograd.graph.increment_version(p)

    @torch.no_grad()
    def load_slow_weights(self):
        self._backup = self.flat_p.clone()
        self.flat_p.copy_(self.flat_slow)
        self._touch()

    @torch.no_grad()
    def restore_fast_weights(self):
        self.flat_p.copy_(self._backup)
        del self._backup
        self._touch()

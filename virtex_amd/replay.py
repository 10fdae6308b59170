"""Launch replay: the training step re-issued from a recorded launch list instead of re-running its Python.

The step of `scripts/pretrain_virtex.py:145-163` is ~1 100 kernel launches on three HIP streams; producing them costs the host
~10 ms of Python per step (tools/host_profile.py: autograd, argument marshalling, allocator calls) -- hidden behind the GPU at 256
images per step, the LIMIT at 64-128 (BASELINE configs 4 and 5).  A hipGraph of the step removes that cost but replays
SLOWER on ROCm 7.2 (profiles/r04_hipgraph_configs_2_4_5.txt: the captured side-stream branches run one after the other).
`StepReplay` keeps the eager launches -- same kernels, same three streams, same events -- and removes only the Python:

  record   one eager step runs with every C-ABI call (`_lib.call`: function pointer + ctypes arguments), every stream / event
           operation (event record / wait, stream switch, wait_stream) and every ATen operator that launches work (fills,
           copies, the few adds) appended to ONE ordered list; every tensor that took part stays alive, so device addresses in
           the recorded arguments stay valid;
  replay   the list is re-issued in order: ~1 400 prebuilt calls, no autograd, no allocation, no shape logic.

What makes a replay a faithful NEXT step rather than a repetition (the same three things a hipGraph needs, virtex_amd/graph.py):
dropout masks advance through the device epoch mixed into the seeds by the kernels, the LR multiplier and the Lookahead phase
are computed on the device (`FusedPretrainOptimizer.enable_device_schedule`), BatchNorm's counters live on the device.
Work the autograd ENGINE does by itself cannot be recorded as such, so none is left to it: (1) the one launch it made (the sum
of the two heads' gradients of the shared visual projection) is a Function of the model; (2) the one cross-stream edge it ordered
in C++ (the gradient of `loss_a + loss_b`, produced on the compute stream, consumed by the branch head on the branch stream) is an
explicit `wait_stream` in `models._LossSumFn`; (3) its stream guards -- every backward node runs on the stream of its forward,
switched without any Python setter -- are noticed by the recorder as an unexplained change of the current stream and recorded as
switches of their own (kernel launches carry their stream in their arguments, but `event.record()` and ATen operators go to
whatever stream is current when they are re-issued).  Construction VALIDATES the recording: made on the example batch X, it is
replayed on a second batch Y (rows rolled, images negated) from the state and dropout seeds an eager step on Y started from, and
must reproduce that step's loss, gradients and updated state; a recorded launch that runs before its producer reads what the
recording left in its buffers -- X-values -- and fails the comparison.  `tests/test_replay.py` also walks the list the way the
replay does and checks every op against the stream it saw when it was recorded.

Data parallelism (round 5): the gradient exchange is part of the list.  `distributed.GradientBuckets` records every bucket
all-reduce it issues (`torch.distributed.all_reduce(..., async_op=True)` on the communication stream, behind the recorded event
/ stream waits that fence it against the three compute streams) and the `wait()` that orders the streams after it; a replay
re-issues the collectives from the main thread in the recorded order -- the order every rank recorded, so ranks may even mix
replayed and eager steps.  world_size-2 gloo test on the emulator: tests/test_distributed_emu.py; RCCL with one forced rank on
the GPU: tests/test_distributed_gpu.py.
"""
import threading
from typing import Callable, Dict, List, Optional

import torch
from torch.utils._python_dispatch import TorchDispatchMode

from . import _lib

_VIEW_NAMES = {"view", "_unsafe_view", "reshape", "permute", "transpose", "t", "slice", "select", "detach", "alias", "expand",
               "as_strided", "unsqueeze", "squeeze", "split", "split_with_sizes", "unbind", "chunk", "narrow", "view_as",
               "unfold", "diagonal", "movedim", "swapaxes", "flatten", "unflatten", "lift_fresh", "_reshape_alias", "real"}
_NO_WORK = {"empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided", "_local_scalar_dense", "is_same_size",
            "sym_size", "sym_stride", "sym_numel", "stride", "size", "numel", "dim", "is_contiguous", "record_stream",
            "_has_compatible_shallow_copy_type", "is_pinned", "set_", "resize_"}


def _current_stream_key():
    """(stream_id, device_index, device_type) of the calling thread's current stream; None without a GPU."""
    get = getattr(torch._C, "_cuda_getCurrentStream", None)
    if get is None or not torch.cuda.is_available():
        return None
    return tuple(get(torch.cuda.current_device()))


def _set_stream_key(key):
    torch._C._cuda_setStream(stream_id=key[0], device_index=key[1], device_type=key[2])


class Recorder:
    """The ordered launch list of one step.  Appended to from the main thread and from autograd's worker thread (the backward
    pass runs there; the two never run at the same time)."""

    def __init__(self):
        self.ops: List[Callable[[], None]] = []
        self.labels: List[str] = []      # what each op is (C entry point / stream operation / ATen operator): tools/replay_dump.py
        self.at_stream: List[Optional[int]] = []   # id of the calling thread's current stream when the op was recorded (tests)
        self.keep = []                   # tensors / ctypes objects whose memory the recorded arguments point into
        self.lock = threading.Lock()
        self.counts = {"kernel": 0, "aten": 0, "stream": 0, "engine_switch": 0, "collective": 0}
        self.cur = _current_stream_key()         # the stream the replay will be on at this point of the list
        self.start = self.cur

    def add(self, kind, thunk, *keep, label="", sets_stream=False):
        with self.lock:
            # The autograd engine puts every node on the stream its forward ran on with a C++ stream guard: no Python setter
            # is called, so the recording never sees the switch.  Kernel launches carry their stream handle in their recorded
            # arguments, but `event.record()` and ATen operators go to whatever stream is CURRENT when they are re-issued:
            # an unexplained change of the calling thread's current stream becomes a recorded switch of its own.
            now = _current_stream_key()
            if now is not None:
                if not sets_stream and now != self.cur:
                    self.ops.append(lambda k=now: _set_stream_key(k))
                    self.labels.append(f"stream:engine switch to stream#{now[0]}")
                    self.at_stream.append(now[0])
                    self.counts["engine_switch"] += 1
                self.cur = now
            self.ops.append(thunk)
            self.at_stream.append(now[0] if now is not None else None)
            self.labels.append(f"{kind}:{label}")
            self.keep.extend(keep)
            self.counts[kind] += 1


_active: Optional[Recorder] = None
_tl = threading.local()


class explicit_ops:
    """Inside this context the ATen dispatch recorder records nothing: the caller adds its own thunks (`active().add`).  For
    code that may run either inside a traced backward (dispatch mode on) or from an autograd hook on the engine's thread
    (dispatch mode off) and must be recorded exactly once: the data-parallel engine's payload conversions."""

    def __enter__(self):
        _tl.explicit = getattr(_tl, "explicit", 0) + 1

    def __exit__(self, *exc):
        _tl.explicit -= 1
        return False


def active() -> Optional[Recorder]:
    return _active


# factory operators re-issued as a fill of the tensor the recording produced (their out= overloads take other arguments)
_FILLS = {
    "zeros": lambda o, a, k: (lambda: o.zero_()),
    "zeros_like": lambda o, a, k: (lambda: o.zero_()),
    "ones": lambda o, a, k: (lambda: o.fill_(1)),
    "ones_like": lambda o, a, k: (lambda: o.fill_(1)),
    "full": lambda o, a, k: (lambda v=(a[1] if len(a) > 1 else k["fill_value"]): o.fill_(v)),
    "full_like": lambda o, a, k: (lambda v=(a[1] if len(a) > 1 else k["fill_value"]): o.fill_(v)),
}
_out_cache = {}


def _out_variant(func):
    """The `out=` overload of a functional ATen overload -- same arguments plus ONE `out` tensor, same result -- or None.
    `o.copy_(f(...))` costs an allocation and a copy launch per replayed operator (87 copy launches per step in the round-4
    kernel trace); the out= form writes the recorded result tensor directly."""
    if func in _out_cache:
        return _out_cache[func]
    found = None
    try:
        packet = func.overloadpacket
        ov = func._overloadname
        for cand in (("out",) if ov in ("", "default") else (ov + "_out", "out")):
            op = getattr(packet, cand, None)
            if op is None:
                continue
            fa = [(a.name, str(a.type)) for a in func._schema.arguments]
            oa = [(a.name, str(a.type)) for a in op._schema.arguments if not a.is_out]
            n_out = sum(1 for a in op._schema.arguments if a.is_out)
            if fa == oa and n_out == 1 and [a.name for a in op._schema.arguments if a.is_out] == ["out"]:
                found = op
                break
    except Exception:
        found = None
    _out_cache[func] = found
    return found


class _RecordAten(TorchDispatchMode):
    """Every ATen operator that launches device work, as a thunk that repeats it INTO the tensors the recording produced."""

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        out = func(*args, **kwargs)
        rec = _active
        if rec is None or getattr(_tl, "explicit", 0):
            return out
        name = func._schema.name.split("::")[-1]
        if name in _VIEW_NAMES or name in _NO_WORK:
            return out
        flat_in = [a for a in list(args) + list(kwargs.values()) if isinstance(a, torch.Tensor)]
        outs = [o for o in (out if isinstance(out, (tuple, list)) else (out,)) if isinstance(o, torch.Tensor)]
        if not any(t.is_cuda for t in flat_in + outs) and flat_in + outs and not _lib.is_emulator():
            return out                                   # host-side tensor arithmetic: no device work to repeat
        if func._schema.is_mutable:                      # in-place / out= : repeat as it was
            rec.add("aten", lambda f=func, a=args, k=kwargs: f(*a, **k), args, kwargs, out, label=name)
        elif any(r.alias_info is not None for r in func._schema.returns):
            return out                                   # a view the table above does not know
        elif outs:                                       # functional: recompute, store into the recorded result
            single = len(outs) == 1 and isinstance(out, torch.Tensor)
            out_op = _out_variant(func) if single else None
            if single and name in _FILLS:                # factories: refill the recorded tensor (one launch)
                rec.add("aten", _FILLS[name](outs[0], args, kwargs), args, kwargs, out, label=name + ".refill")
            elif single and name in ("_to_copy", "clone") and isinstance(args[0], torch.Tensor):
                rec.add("aten", lambda o=outs[0], src=args[0]: o.copy_(src), args, kwargs, out, label=name + ".copy_")
            elif out_op is not None:                       # ... by the operator's own out= overload: no temporary, no copy launch
                rec.add("aten", lambda f=out_op, a=args, k=kwargs, o=outs[0]: f(*a, **k, out=o), args, kwargs, out,
                        label=name + ".out")
            elif len(outs) == 1:
                rec.add("aten", lambda f=func, a=args, k=kwargs, o=outs[0]: o.copy_(f(*a, **k)), args, kwargs, out, label=name)
            else:
                def thunk(f=func, a=args, k=kwargs, os_=outs):
                    r = f(*a, **k)
                    r = [x for x in (r if isinstance(r, (tuple, list)) else (r,)) if isinstance(x, torch.Tensor)]
                    for o, x in zip(os_, r):
                        o.copy_(x)
                rec.add("aten", thunk, args, kwargs, out, label=name)
        return out


def _describe(obj, args):
    """stream / event identities for the labels of recorded stream operations"""
    def one(o):
        if isinstance(o, torch.cuda.Stream):
            return f"stream#{o.stream_id}"
        if isinstance(o, torch.cuda.Event):
            return f"event@{id(o) & 0xffff:04x}"
        return type(o).__name__
    try:
        cur = torch.cuda.current_stream().stream_id
    except Exception:
        cur = "?"
    return f"{one(obj)}({', '.join(one(a) for a in args)}) [current stream#{cur}]"


def traced_backward(fn):
    """Decorator for the backward of the library's autograd Functions: autograd calls them on its own thread, where the
    recording dispatch mode of the main thread is not active."""
    def wrapper(ctx, *grads):
        if _active is not None:
            with _RecordAten():
                return fn(ctx, *grads)
        return fn(ctx, *grads)
    wrapper.__name__ = getattr(fn, "__name__", "backward")
    wrapper.__doc__ = fn.__doc__
    return wrapper


class _Patches:
    """Stream / event operations of torch.cuda recorded at the level every caller goes through."""

    def __init__(self, rec: Recorder):
        self.rec = rec
        self.saved = []
        self.tl = threading.local()

    def _wrap(self, owner, name, bound: bool):
        orig = getattr(owner, name)
        rec = self.rec

        tl = self.tl

        if bound:                        # (wait_stream is record_event + wait_event inside: only the outermost call is recorded)
            def wrapped(self_, *a, **k):
                depth = getattr(tl, "depth", 0)
                tl.depth = depth + 1
                try:
                    r = orig(self_, *a, **k)
                finally:
                    tl.depth = depth
                if depth == 0:
                    rec.add("stream", lambda s=self_, a=a, k=k: orig(s, *a, **k), self_, a, k,
                            label=f"{type(self_).__name__}.{name} {_describe(self_, a)}")
                return r
        else:
            def wrapped(*a, **k):
                depth = getattr(tl, "depth", 0)
                tl.depth = depth + 1
                try:
                    r = orig(*a, **k)
                finally:
                    tl.depth = depth
                if depth == 0:
                    rec.add("stream", lambda a=a, k=k: orig(*a, **k), a, k, label=f"{name} {k or a}", sets_stream=name == "_cuda_setStream")
                return r
        self.saved.append((owner, name, orig))
        setattr(owner, name, wrapped)

    def __enter__(self):
        if torch.cuda.is_available():
            self._wrap(torch.cuda.Event, "record", True)
            self._wrap(torch.cuda.Event, "wait", True)
            self._wrap(torch.cuda.Stream, "wait_event", True)
            self._wrap(torch.cuda.Stream, "wait_stream", True)
            if hasattr(torch._C, "_cuda_setStream"):
                self._wrap(torch._C, "_cuda_setStream", False)
                from . import streams                         # (streams.py holds its own reference to the raw setter)
                if getattr(streams, "_set_cur", None) is not None:
                    self.saved.append((streams, "_set_cur", streams._set_cur))
                    streams._set_cur = torch._C._cuda_setStream
        return self

    def __exit__(self, *exc):
        for owner, name, orig in reversed(self.saved):
            setattr(owner, name, orig)
        return False


def record(step_fn: Callable[[], torch.Tensor]):
    """Run `step_fn` once, recording it.  Returns (recorder, the step's result)."""
    global _active
    rec = Recorder()
    with _Patches(rec):
        _active = rec
        _lib.set_recorder(rec)
        try:
            with _RecordAten():
                out = step_fn()
        finally:
            _active = None
            _lib.set_recorder(None)
    return rec, out


def _run(rec: Recorder):
    """Re-issue a recording; the calling thread ends on the stream it was on (the list may end on another one)."""
    try:
        # no autograd: recorded tensors may still carry requires_grad / grad_fn from the recorded step, and a re-issued
        # `o.copy_(...)` with grad mode on would hang a new CopyBackwards node on them every replay (an unbounded graph)
        with torch.no_grad():
            for op in rec.ops:
                op()
    finally:
        if rec.start is not None and _current_stream_key() != rec.start:
            _set_stream_key(rec.start)


class StepReplay:
    """step = StepReplay(model, buckets, optimizer, example_batch); loss = step(batch)

    `validate`: compare one replayed step with one eager step from identical state (parameters, optimizer state, BatchNorm
    buffers, device counters) on the example batch, dropout off -- raises RuntimeError when anything differs."""

    def __init__(self, model: torch.nn.Module, buckets, optimizer, example_batch: Dict[str, torch.Tensor], warmup: int = 2,
                 validate: bool = True):
        self.model, self.buckets, self.opt = model, buckets, optimizer
        self.static = {k: v.clone() for k, v in example_batch.items()}
        optimizer.enable_device_schedule()
        for _ in range(max(1, warmup)):
            self._eager()
        self._sync()
        if validate:
            self._validate()
        self.rec, self.loss = record(self._eager)
        self._sync()
        self.replays = 0

    # ---- the step
    def _eager(self) -> torch.Tensor:
        self.buckets.zero(); self.buckets.begin()
        out = self.model(self.static)
        out["loss"].backward()
        self.opt.step(grad_scale=self.buckets.finish())
        return out["loss"].detach()

    def _replay(self):
        _run(self.rec)

    def __call__(self, batch: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
        if batch is not None:
            for k, v in self.static.items():
                if batch[k].shape != v.shape:
                    # a recording is a list of launches on FIXED shapes: collate to the recorded caption length
                    # (virtex_amd.data.collate_captions(..., pad_to=...)) or take the eager step for odd batches
                    raise ValueError(f"StepReplay: batch[{k!r}] has shape {tuple(batch[k].shape)}, the recording was made on "
                                     f"{tuple(v.shape)}")
                v.copy_(batch[k], non_blocking=True)
        self._replay()
        self.replays += 1
        return self.loss

    def sync(self):
        """Refresh the host-side mirrors after replays (optimizer step index, parameter version counters)."""
        self.opt.sync_host()

    def _sync(self):
        dev = next(self.model.parameters()).device
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    # ---- validation: eager step vs replayed step from the same state
    def _state(self):
        ts = [self.opt.flat_p, self.opt.flat_m, self.opt.flat_slow, self.opt.dev["step"], self.opt.dev["kc"], self.opt.dev["epoch"]]
        ts += [b for b in self.model.buffers()]
        return ts

    def _validate(self):
        """The local comparison, then the verdict of ALL ranks.  Whatever happens locally -- a mismatch, but also an exception
        on the way (out of memory while cloning the state, a failing restore) -- this rank still arrives at the verdict
        collective and contributes "no": a rank that left early would fall back to the eager step and issue bucket all-reduces
        while its peers sit in the one-element MIN all-reduce (mismatched collectives: a hang or silent corruption)."""
        err, bad = None, []
        try:
            bad = self._validate_local()
        except Exception as e:                               # reported after the collective
            err, bad = e, [f"{type(e).__name__}: {e}"]
        # Data parallel: every rank validated on its own batch, and every rank must take the same decision -- a rank that alone
        # falls back to the eager step issues a different number of gradient exchanges than its peers and the job hangs.  The
        # verdict is therefore the minimum over the ranks (one more collective, at the same point of every rank's sequence).
        peers_ok = True
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            dev = next(self.model.parameters()).device
            flag = torch.tensor([0.0 if bad else 1.0], device=dev if dev.type == "cuda" else "cpu")
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            peers_ok = bool(flag.item() > 0.5)
        if err is not None:
            raise RuntimeError(f"StepReplay: validation could not be completed on this rank ({bad[0]})") from err
        if bad:
            raise RuntimeError("StepReplay: the replayed step does not reproduce the eager step: " + "; ".join(bad[:4]))
        if not peers_ok:
            raise RuntimeError("StepReplay: the replayed step does not reproduce the eager step on another rank")

    def _validate_local(self):
        """-> list of differences (empty = the recording reproduces the eager step).  Recording on batch X, comparison on batch Y: a recorded launch that runs too early (a stream dependency the
        recording lost) reads what the recording left in its buffers -- X-values -- and that shows only when the reference
        values come from a different batch."""
        from .modules.textual_heads import dropout_seed_state
        saved = [t.clone() for t in self._state()]
        batch_x = {k: v.clone() for k, v in self.static.items()}
        seeds = dropout_seed_state()
        rec, loss_rec = record(self._eager)                  # the recording (batch X); its tensors are the replay's buffers
        self._sync()

        def restore():
            for t, s0 in zip(self._state(), saved):
                t.copy_(s0)
        restore()
        for k, v in self.static.items():                     # batch Y: the images / captions of X assigned to other rows, images negated
            y = torch.roll(batch_x[k], 1, 0) if v.dim() > 0 and v.shape[0] > 1 else batch_x[k]
            v.copy_(-y if (k == "image" and y.dtype.is_floating_point) else y)
        dropout_seed_state(seeds)                            # the eager reference draws the seeds the recording holds
        loss_a = self._eager().clone()                       # step A: eager on Y
        self._sync()
        after_a = [t.clone() for t in self._state()]
        grads_a = self.buckets.flat.clone()
        restore()
        _run(rec)                                            # step B: the recording replayed on Y from the same state
        self._sync()
        restored_batch = batch_x

        # the embedding's fp32 atomics reorder sums (1e-7 relative); with the bf16 gradient payload of the data-parallel engine
        # such a difference can cross a rounding boundary: one bf16 ulp (2^-8 relative) on single elements of the exchanged buffer
        # (and into the momentum / parameters it updates): the comparison is then one of norms at the payload's precision -- a
        # launch that ran before its producer differs by the whole tensor, not by rounding
        bf16_wire = getattr(self.buckets, "enabled", False) and getattr(self.buckets, "payload", "fp32") == "bf16"

        def close(a, b):
            if a.dtype.is_floating_point:
                if bf16_wire:
                    return (a.double() - b.double()).norm().item() <= 1e-3 * max(a.double().norm().item(), 1e-30)
                scale = max(a.abs().max().item(), 1e-30)
                return (a.double() - b.double()).abs().max().item() <= 2e-5 * scale
            return torch.equal(a, b)
        bad = []
        if not close(loss_a, loss_rec):
            bad.append(f"loss {loss_a.item()} vs {loss_rec.item()}")
        if not close(grads_a, self.buckets.flat):
            d = (grads_a - self.buckets.flat).abs().max().item()
            bad.append(f"gradients differ (max abs {d:.3e} of {grads_a.abs().max().item():.3e})")
        for i, (a, b) in enumerate(zip(after_a, self._state())):
            if not close(a, b):
                bad.append(f"state tensor {i} differs")
        self.validated = {"ops": len(rec.ops), **rec.counts}
        for k, v in self.static.items():
            v.copy_(restored_batch[k])
        restore()
        return bad

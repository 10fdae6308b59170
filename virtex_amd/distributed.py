"""Data-parallel engine for the bicaptioning step: one process per GPU, gradients exchanged
with RCCL over xGMI, overlapped with the backward pass.

Replaces what the reference gets from ``nn.parallel.DistributedDataParallel``
(/root/reference/scripts/pretrain_virtex.py:121-123) and the helpers of
/root/reference/virtex/utils/distributed.py (:82-112 process-group set-up, :115-160
``synchronize`` / ``average_across_processes``) -- re-designed for this model instead of being
a generic reducer:

* all 202 gradient tensors live in ONE flat fp32 buffer (``p.grad`` are views), cut into a few
  large buckets in reverse execution order -- first the text-side parameters (66 % of the bytes,
  final before the ResNet backward starts), then the ResNet from layer4 to conv1;
* a bucket is all-reduced (SUM) as soon as autograd has accumulated its last gradient: the
  collective is issued on a dedicated side HIP stream that waits on an event recorded on the
  compute stream, so it runs under the remaining backward kernels;
* xGMI is point-to-point (7 links x ~153 GB/s per GPU), so few large messages beat DDP's 25 MB
  default: the bucket size is a parameter (default 64 MB -> 5 collectives per step);
* ``finish()`` makes the compute stream wait for the side stream; the 1/world scaling is folded
  into the optimizer (``PretrainOptimizer.step(grad_scale=1/world)``), not a separate pass;
* DDP's default ``broadcast_buffers=True`` re-broadcasts rank 0's BatchNorm running statistics at
  every forward (C4).  In training mode those buffers are write-only (batch statistics normalise),
  so the per-step broadcast changes nothing a training step computes; it matters where the buffers
  are READ: validation and checkpoints.  ``broadcast_buffers(model)`` does that one flat broadcast
  -- call it before switching to eval and before saving -- and ``broadcast_parameters`` does it once
  at start-up, so every rank validates on rank 0's statistics exactly like the reference.

Works with any torch.distributed backend: ``nccl`` (= RCCL on ROCm) on GPUs, ``gloo`` on CPU for
the world_size-2 tests.
"""
import os
from typing import List, Optional

import torch
import torch.distributed as dist

from .streams import branch_stream, wgrad_stream


def _forced() -> Optional[str]:
    """VIRTEX_AMD_FORCE_DIST=nccl|gloo: run the whole data-parallel path (communicator, parameter broadcast, bucket
    all-reduces on the side stream, barriers) even with ONE rank -- how the RCCL plumbing is exercised on a 1-GPU box."""
    return os.environ.get("VIRTEX_AMD_FORCE_DIST") or None


def active() -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _forced() is not None)


def init_process_group(backend: Optional[str] = None) -> int:
    """Rendezvous from torchrun-style environment variables; returns the local rank."""
    if dist.is_initialized():
        return int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if _forced() is not None and world == 1:
        backend = _forced()
        os.environ.setdefault("RANK", "0")
    if world > 1 or _forced() is not None:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            # several ranks must agree on the port: the launcher's job (torchrun sets it); the reference's default
            # (tcp://127.0.0.1:23456, virtex/utils/distributed.py:20) is the fallback.  A single forced rank has nobody
            # to agree with: any free port, so that two such processes on one host never collide.
            os.environ["MASTER_PORT"] = str(_free_port()) if world == 1 else "23456"
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL's kernels share the CUs with three busy compute streams: VIRTEX_AMD_RCCL_MAX_CHANNELS caps the number of
        # channels (= workgroups) a collective may occupy (NCCL_MAX_NCHANNELS, read by RCCL at communicator creation)
        ch = os.environ.get("VIRTEX_AMD_RCCL_MAX_CHANNELS")
        if ch:
            os.environ.setdefault("NCCL_MAX_NCHANNELS", ch)
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        # A rank that never arrives at a collective must become an ERROR, not an endless wait: every collective of this
        # process group (the rendezvous included) carries an explicit deadline -- 600 s by default, VIRTEX_AMD_COLLECTIVE_TIMEOUT_S
        # (torch's own defaults: 600 s for nccl, 1800 s for gloo; not shorter by default because the first `import torch` on a
        # fresh box can take minutes and the ranks do not start together).  With nccl = RCCL the watchdog thread aborts the
        # communicator and tears the process down with the reason on stderr; gloo raises in the waiting thread.
        import datetime
        deadline = datetime.timedelta(seconds=float(os.environ.get("VIRTEX_AMD_COLLECTIVE_TIMEOUT_S", "600")))
        dist.init_process_group(backend=backend, rank=int(os.environ["RANK"]), world_size=world, timeout=deadline)
    return local_rank


def shutdown():
    """Tear the process group down (what torch asks for before the interpreter exits: a communicator left alive can leak its
    proxy threads or stall the exit of a multi-rank job)."""
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def ranks_seen(device=None) -> int:
    """All-reduce (SUM) of a one: how many ranks the process group's transport really connects.  1 without a group."""
    if not active():
        return 1
    one = torch.ones(1, dtype=torch.float32, device=device if device is not None else
                     (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    return int(round(one.item()))


def transport_description() -> dict:
    """Backend of the process group and, for nccl (= RCCL on ROCm), the library version torch was built against."""
    out = {"backend": dist.get_backend() if active() else None, "world_size": world_size()}
    if out["backend"] == "nccl":
        try:
            out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:          # a build without the binding: say so instead of guessing
            out["rccl_version"] = f"unavailable ({type(e).__name__})"
        out["hip"] = getattr(torch.version, "hip", None)
    return out


def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def synchronize():
    if active():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])   # pin the barrier's collective to this rank's GPU
        else:
            dist.barrier()


def average_across_processes(t):
    """Averages a tensor, or every tensor of a dict, across processes IN PLACE, like the reference helper
    (virtex/utils/distributed.py:141-160; its callers ignore the return value: scripts/pretrain_virtex.py:213).
    A dict costs one fused all-reduce instead of one per key.  The argument is also returned."""
    if not active():
        return t
    if isinstance(t, torch.Tensor):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t /= world_size()
        return t
    keys = sorted(t)
    flat = torch.cat([t[k].detach().float().reshape(-1) for k in keys])      # values of any shape, like the reference accepts
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= world_size()
    with torch.no_grad():
        o = 0
        for k in keys:
            n = t[k].numel()
            t[k].copy_(flat[o:o + n].to(t[k].dtype).reshape(t[k].shape))
            o += n
    return t


def _broadcast_flat(tensors, src):
    """One broadcast per dtype instead of one per tensor; results are written back through `copy_`, which bumps
    the tensors' version counters (the compute-weight caches are keyed on them)."""
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    with torch.no_grad():
        for dtype, ts in by_dtype.items():
            flat = torch.cat([t.detach().reshape(-1) for t in ts])
            dist.broadcast(flat, src=src)
            off = 0
            for t in ts:
                n = t.numel()
                t.copy_(flat[off: off + n].view(t.shape))
                off += n


def broadcast_parameters(model: torch.nn.Module, src: int = 0):
    """What DDP's constructor does once (C3): rank-0 parameters and buffers everywhere."""
    if not active():
        return
    _broadcast_flat(list(model.parameters()) + list(model.buffers()), src)


def broadcast_buffers(model: torch.nn.Module, src: int = 0):
    """Rank 0's buffers (BatchNorm running statistics, step counters) everywhere: what DDP's per-forward buffer
    broadcast (C4) amounts to at the two places the buffers are read -- before validation and before a checkpoint."""
    if not active():
        return
    _broadcast_flat(list(model.buffers()), src)


def execution_order(model: torch.nn.Module) -> List[torch.nn.Parameter]:
    """Unique parameters in the order their gradients become final during backward."""
    named = list(model.named_parameters())
    text = [p for n, p in named if not n.startswith("visual.")]
    cnn = [p for n, p in named if n.startswith("visual.")]
    return text + list(reversed(cnn))


class _RecordedHandle:
    """The work handle of a recorded all-reduce: `slot[0]` is the handle of the most recent issue (the recording's, then each
    replay's); wait() waits for it and, while recording, appends that wait to the list."""

    def __init__(self, slot):
        self.slot = slot

    def wait(self):
        from . import replay
        self.slot[0].wait()
        rec = replay.active()
        if rec is not None:
            rec.add("collective", lambda s=self.slot: s[0].wait(), label="all_reduce.wait")


class GradientBuckets:
    """payload: "fp32" (default) exchanges the flat gradient buffer itself; "bf16" rounds every bucket to bf16 first
    (half the xGMI bytes: 138.9 instead of 277.9 MB per step) and widens the summed result back into the fp32 buffer
    -- the sum of N bf16-rounded gradients carries a relative error of ~2^-9/sqrt(3) per element (tests/test_distributed.py)."""

    def __init__(self, model: torch.nn.Module, bucket_mb: float = 64.0, payload: Optional[str] = None,
                 tail_mb: Optional[float] = None):
        """tail_mb: the LAST bucket closes with the last gradient of the backward pass (the stem's), so its all-reduce
        is fully exposed: it is cut off as its own small bucket of at most `tail_mb` megabytes (default 8, never more
        than bucket_mb / 2; VIRTEX_AMD_DP_TAIL_MB) -- for ResNet-50 that is the stem + layer1 + layer2 (5.8 MB), while
        the bulk of the ResNet's gradient goes out with the previous bucket under the early stages' backward."""
        params = [p for p in execution_order(model) if p.requires_grad]
        self.params = params
        dev = params[0].device
        total = sum(p.numel() for p in params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        cap = int(bucket_mb * (1 << 20) / 4)
        if tail_mb is None:
            tail_mb = float(os.environ.get("VIRTEX_AMD_DP_TAIL_MB", "8"))
        tail_cap = int(min(tail_mb, bucket_mb / 2) * (1 << 20) / 4)
        tail_first, acc = len(params), 0            # index of the first parameter of the tail bucket
        for i in range(len(params) - 1, 0, -1):
            if acc + params[i].numel() > tail_cap:
                break
            acc += params[i].numel()
            tail_first = i
        self.buckets = []          # (start, end, n_params)
        self.bucket_of = {}
        off, bstart, bcount = 0, 0, 0
        for i, p in enumerate(params):
            if i == tail_first and bcount:          # close the bucket in front of the tail
                self.buckets.append((bstart, off, bcount))
                bstart, bcount = off, 0
            n = p.numel()
            if p.dim() == 4 and p.stride(1) == 1 and p.shape[1] > 1:
                # conv master weights are stored (KO,R,S,C): give the gradient the same layout
                O, I, R, S = p.shape
                view = self.flat[off: off + n].view(O, R, S, I).permute(0, 3, 1, 2)
            else:
                view = self.flat[off: off + n].view(p.shape)
            p.grad = view
            self.bucket_of[p] = len(self.buckets)
            off += n
            bcount += 1
            if off - bstart >= cap:
                self.buckets.append((bstart, off, bcount))
                bstart, bcount = off, 0
        if bcount:
            self.buckets.append((bstart, off, bcount))
        self.pending = [0] * len(self.buckets)
        self.exposed = []            # (event before, event after) the compute stream's wait for the collectives, per step
        # diagnostic only (bench.py sets it; VIRTEX_AMD_DP_MEASURE_EXPOSED=1): time the compute stream's wait in finish()
        self.measure_exposed = os.environ.get("VIRTEX_AMD_DP_MEASURE_EXPOSED", "0") == "1"
        self.early = set()           # parameters announced by gradsink.mark_ready() in the current step
        self.last_early = 0
        self.handles = []
        self.world = world_size()
        self.side = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self.enabled = active()
        self.payload = payload or os.environ.get("VIRTEX_AMD_DP_PAYLOAD", "fp32")
        if self.payload not in ("fp32", "bf16"):
            raise ValueError(f"gradient payload must be fp32 or bf16, got {self.payload}")
        self.wire = torch.empty(total, dtype=torch.bfloat16, device=dev) if (self.enabled and self.payload == "bf16") else None
        if self.enabled:
            # fires once per parameter per backward, also when the backward function accumulated into
            # p.grad itself and returned None (virtex_amd/gradsink.py)
            for p in params:
                p.register_post_accumulate_grad_hook(self._on_grad)
            from . import gradsink
            gradsink.register(params, self._on_ready)      # per instance: keyed by parameter
        self.begin()

    # ---------------------------------------------------------------------------------
    def begin(self):
        """Re-arm the per-bucket counters.  Runs at construction and at the end of every finish(), so a training
        loop that never calls it is still correct; calling it before each backward as well is harmless."""
        self.pending = [c for (_, _, c) in self.buckets]
        self.launched = [False] * len(self.buckets)
        self.handles = []
        self.widen = []              # bf16 payload: (handle, fp32 chunk, bf16 wire buffer) to widen back in finish()
        self.early = set()

    def zero(self):
        self.flat.zero_()

    def _on_ready(self, p):
        """gradsink.mark_ready: p's gradient kernels are enqueued (compute or weight-gradient stream)."""
        if p in self.bucket_of and p not in self.early:
            self.early.add(p)
            self._count(p)

    def _on_grad(self, p):
        if p in self.early:          # already counted when its backward function announced it
            return
        self._count(p)

    def _count(self, p):
        b = self.bucket_of[p]
        self.pending[b] -= 1
        if self.pending[b] < 0:
            raise RuntimeError("GradientBuckets: a parameter's gradient was announced twice in one step (a second "
                               "backward without finish() in between?)")
        if self.pending[b] == 0:
            self._launch(b)

    @staticmethod
    def _all_reduce(t):
        """One asynchronous SUM all-reduce.  While a launch recording is being made (virtex_amd.replay) the collective and
        the later wait() on its handle become ops of the recorded list: a replay re-issues them in the recorded order."""
        from . import replay
        h = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)
        rec = replay.active()
        if rec is None:
            return h
        slot = [h]
        rec.add("collective", lambda t=t, slot=slot: slot.__setitem__(0, dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)),
                t, label=f"all_reduce {t.numel()} x {str(t.dtype).split('.')[-1]}")
        return _RecordedHandle(slot)

    @staticmethod
    def _copy(dst, src):
        """dst.copy_(src) of the payload conversions: recorded exactly once, whichever thread / dispatch mode runs it."""
        from . import replay
        rec = replay.active()
        if rec is None:
            dst.copy_(src)
            return
        with replay.explicit_ops():
            dst.copy_(src)
        rec.add("aten", lambda d=dst, s_=src: d.copy_(s_), dst, src, label="copy_ (gradient payload)")

    def _reduce(self, s, e):
        """The collective of one bucket (runs on the communication stream when there is one)."""
        chunk = self.flat[s:e]
        if self.wire is None:
            return self._all_reduce(chunk)
        wire = self.wire[s:e]
        self._copy(wire, chunk)                             # fp32 -> bf16 (round to nearest even)
        h = self._all_reduce(wire)
        self.widen.append((h, chunk, wire))
        return h

    def _launch(self, b):
        self.launched[b] = True
        s, e, _ = self.buckets[b]
        chunk = self.flat[s:e]
        if self.side is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            wg = wgrad_stream.peek(chunk.device)
            with torch.cuda.stream(self.side):
                self.side.wait_event(ev)
                if wg is not None:              # weight gradients are produced on their own side stream
                    self.side.wait_stream(wg)
                bs = branch_stream.peek(chunk.device)
                if bs is not None:              # ... and one caption direction's backward runs on the branch stream
                    self.side.wait_stream(bs)
                self.handles.append(self._reduce(s, e))
        else:
            self.handles.append(self._reduce(s, e))

    def finish(self) -> float:
        """Wait for the outstanding collectives; returns the scale (1/world) the optimizer must
        apply to the summed gradients."""
        branch_stream.join(self.flat.device)
        wgrad_stream.join(self.flat.device)
        if not self.enabled:
            return 1.0
        for b in range(len(self.buckets)):
            if not self.launched[b]:    # buckets holding parameters that received no gradient this step
                self._launch(b)
        if self.side is not None:
            with torch.cuda.stream(self.side):      # handle.wait() orders the CURRENT stream after the collective
                for h in self.handles:
                    h.wait()
                for (_, chunk, wire) in self.widen:
                    self._copy(chunk, wire)         # bf16 sum -> the fp32 buffer the optimizer reads
            # how long the compute stream sits in this wait = the part of the gradient exchange the backward pass did NOT
            # hide; two events per step, read (and synchronised) only by comm_exposed_ms()
            cur = torch.cuda.current_stream()
            if self.measure_exposed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                cur.wait_stream(self.side)
                e1.record(cur)
                self.exposed.append((e0, e1))
                if len(self.exposed) > 4096:
                    del self.exposed[:2048]
            else:
                cur.wait_stream(self.side)
        else:
            for h in self.handles:
                h.wait()
            for (_, chunk, wire) in self.widen:
                self._copy(chunk, wire)
        self.last_early = len(self.early)    # how many gradients were announced from inside a backward node (diagnostic)
        self.begin()                    # armed for the next backward
        return 1.0 / self.world

    def comm_exposed_ms(self, last: Optional[int] = None, reset: bool = True) -> Optional[float]:
        """Mean time per step the compute stream waited in finish() for the outstanding all-reduces (GPU time between
        two events around the wait) over the last `last` recorded steps (all if None); None when nothing was recorded
        (single process, CPU).  Synchronises the device: call it outside the timed region."""
        ev = self.exposed[-last:] if last else self.exposed
        if not ev:
            return None
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
        if reset:
            self.exposed = []
        return ms


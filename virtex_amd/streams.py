"""The weight-gradient side stream.

Weight gradients are off the critical path of backward: nothing downstream reads them until the gradient
exchange / the optimizer.  The hand-written backward passes therefore enqueue their weight-gradient GEMMs (and
the bias-gradient column sums) on ONE side HIP stream per device, where they run concurrently with the
input-gradient chain on the compute stream.  One stream for all of them keeps every accumulation into a given
gradient buffer ordered.  Who waits for it:
  * the last backward node of each sub-graph (`_ResNetFn.backward`, `_EmbeddingFn.backward`) joins it into the
    compute stream, so ordinary consumers of `.grad` on the compute stream are safe;
  * the data-parallel engine makes its communication stream wait for it before every bucket all-reduce;
  * the optimizers join it before touching gradients.
"""
import os

import torch

_side_streams = {}
_fence_events = {}
_get_cur = getattr(torch._C, "_cuda_getCurrentStream", None)
_set_cur = getattr(torch._C, "_cuda_setStream", None)
_fast_ok = None      # None: the raw setters have not been probed yet; False: this torch build does not take them


def _probe_fast_path(device, stream):
    """The raw setters are private torch API (a 3-tuple from _cuda_getCurrentStream, keyword arguments of _cuda_setStream):
    verified ONCE with a set / get / restore round trip; any surprise switches the module to torch.cuda.stream() for good."""
    global _fast_ok
    try:
        prev = _get_cur(device.index)
        assert isinstance(prev, tuple) and len(prev) == 3
        _set_cur(stream_id=stream.stream_id, device_index=stream.device_index, device_type=stream.device_type)
        now = _get_cur(device.index)
        _set_cur(stream_id=prev[0], device_index=prev[1], device_type=prev[2])
        _fast_ok = tuple(now) == (stream.stream_id, stream.device_index, stream.device_type) and tuple(_get_cur(device.index)) == tuple(prev)
    except Exception:
        _fast_ok = False
    return _fast_ok


def _switch_stream(device, stream):
    """Make `stream` torch's current stream on `device`; returns what _restore_stream needs.  (Same effect as entering
    torch.cuda.stream(stream), without building Stream objects and device guards around it.)"""
    fast = _fast_ok
    if fast is None and _get_cur is not None and _set_cur is not None and device.index is not None \
            and device.index == torch.cuda.current_device():
        fast = _probe_fast_path(device, stream)
    if fast and device.index is not None and device.index == torch.cuda.current_device():
        prev = _get_cur(device.index)                    # (stream_id, device_index, device_type)
        _set_cur(stream_id=stream.stream_id, device_index=stream.device_index, device_type=stream.device_type)
        return prev
    ctx = torch.cuda.stream(stream)
    ctx.__enter__()
    return ctx


def _restore_stream(prev):
    if isinstance(prev, tuple):
        _set_cur(stream_id=prev[0], device_index=prev[1], device_type=prev[2])
    else:
        prev.__exit__(None, None, None)


_hip = None
_masked = []        # (handle, ExternalStream): masked streams live for the life of the process


def masked_stream(device, lo, hi):
    """A HIP stream whose kernels may only occupy compute units [lo, hi) of the device's 256 (hipExtStreamCreateWithCUMask),
    wrapped for torch.  Measurement aid (tools/ab_step.py `wgrad_cus=` / `compute_cus=`, tools/cu_mask_probe.py): partitions the
    chip between the compute stream and a side stream instead of letting their blocks share compute units.  Measured in round 4
    and NOT used: the mask is honoured (a GEMM scales with the CUs it is given, 128 CUs still copy at 5.2 of 5.6 TB/s), but the
    step runs 43-80 ms instead of 24 ms with any masked stream in it (profiles/r04_cu_masked_streams.txt)."""
    import ctypes
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
    words = (ctypes.c_uint32 * 8)()
    for b in range(lo, hi):
        words[b // 32] |= 1 << (b % 32)
    h = ctypes.c_void_p()
    with torch.cuda.device(device):
        err = _hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(8), words)
    if err != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask: error {err}")
    st = torch.cuda.ExternalStream(h.value, device=device)
    _masked.append((h, st))
    return st


class wgrad_stream:
    """Context manager: run weight-gradient GEMMs (off the critical path of backward: nothing downstream
    reads them until the optimizer) on a side HIP stream, concurrently with the input-gradient chain.
    Inputs are fenced with an event; their memory is kept alive for the side stream with record_stream;
    `join()` makes the main stream wait for everything issued so far."""
    enabled = os.environ.get("VIRTEX_AMD_WGRAD_STREAM", "1") != "0"

    def __init__(self, device, *inputs):
        self.device, self.inputs = device, inputs
        self.active = wgrad_stream.enabled and device.type == "cuda"

    @staticmethod
    def side(device):
        st = _side_streams.get(device)
        if st is None:
            st = torch.cuda.Stream(device=device)
            _side_streams[device] = st
        return st

    def __enter__(self):
        if self.active:
            side = wgrad_stream.side(self.device)
            # ~75 of these per step: the fence reuses ONE event per device (a wait captures the record in front of it, so
            # re-recording it for the next fence is safe) and the stream switch goes through the raw setters -- the
            # Event() + torch.cuda.stream() pair cost ~15 us of host time per use (tools/host_profile.py)
            ev = _fence_events.get(self.device)
            if ev is None:
                ev = _fence_events[self.device] = torch.cuda.Event()
            if self.device.index is not None and self.device.index == torch.cuda.current_device():
                ev.record()                               # torch's current stream on this device
            else:
                ev.record(torch.cuda.current_stream(self.device))
            side.wait_event(ev)
            for t in self.inputs:
                t.record_stream(side)
            self.prev = _switch_stream(self.device, side)
        return self

    def __exit__(self, *exc):
        if self.active:
            _restore_stream(self.prev)
        return False

    @staticmethod
    def join(device):
        if wgrad_stream.enabled and device.type == "cuda" and device in _side_streams:
            torch.cuda.current_stream(device).wait_stream(_side_streams[device])

    @staticmethod
    def peek(device):
        """The side stream of `device` if one has been created (None otherwise)."""
        return _side_streams.get(device) if wgrad_stream.enabled else None


_branch_streams = {}


class branch_stream:
    """Fork / join for work that is independent of the compute stream for a while: the projection shortcut of a
    Bottleneck (1x1 conv + BatchNorm, and their backward) runs here while the three-conv main branch runs on the
    compute stream -- HBM-bound BatchNorm passes of one branch under the MFMA-bound convolutions of the other.

        br = branch_stream(device, *tensors_it_reads)
        with br:            # side stream waits for the compute stream's work so far
            y = ...
        ...                 # compute stream carries on
        br.wait(y)          # compute stream waits for the branch; y may now be used (and freed) there
    """
    enabled = os.environ.get("VIRTEX_AMD_BRANCH_STREAM", "1") != "0"

    def __init__(self, device, *inputs):
        self.device, self.inputs = device, inputs
        self.active = branch_stream.enabled and device.type == "cuda"
        self.done = None
        self.start = None

    def mark(self):
        """Fix the fork point NOW (the branch will wait for the compute stream's work up to here, not up to
        the later `with`): lets the caller enqueue the compute stream's share first."""
        if self.active:
            self.start = torch.cuda.Event()
            self.start.record(torch.cuda.current_stream(self.device))
        return self

    @staticmethod
    def side(device):
        st = _branch_streams.get(device)
        if st is None:
            st = torch.cuda.Stream(device=device)
            _branch_streams[device] = st
        return st

    def __enter__(self):
        if self.active:
            side = branch_stream.side(self.device)
            ev = self.start
            if ev is None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
            side.wait_event(ev)
            for t in self.inputs:
                if t is not None:
                    t.record_stream(side)
            self.ctx = torch.cuda.stream(side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.active:
            self.done = torch.cuda.Event()
            self.done.record(torch.cuda.current_stream(self.device))
            self.ctx.__exit__(*exc)
        return False

    def wait(self, *outputs):
        if self.active and self.done is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(self.done)
            for t in outputs:
                if t is not None:
                    t.record_stream(cur)

    @staticmethod
    def join(device):
        if branch_stream.enabled and device.type == "cuda" and device in _branch_streams:
            torch.cuda.current_stream(device).wait_stream(_branch_streams[device])

    @staticmethod
    def peek(device):
        return _branch_streams.get(device) if branch_stream.enabled else None

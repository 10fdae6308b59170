"""Direct gradient accumulation for the hand-written backward passes.

autograd's contract is `param.grad += returned_gradient`, which costs one zero-filled temporary plus
one add kernel per parameter (~400 tiny launches per step here).  Every weight-gradient kernel of this
package already accumulates (`+=`) into the fp32 buffer it is given, so when a parameter owns a
suitable `.grad` buffer (e.g. a view into the data-parallel flat buffer) the backward functions write
into it directly and return None to autograd.  Parameters that are used more than once per step (the
tied embedding / output matrix, the shared visual projection ...) keep the autograd path, so
the gradient is complete when autograd visits the parameter.  autograd still runs the parameter's
post-accumulate-grad hooks when a backward function returns None for it (verified in
tests/test_model_parity.py), which is what the data-parallel engine keys its bucket launches on: by
then every kernel that writes the buffer has been enqueued on the compute stream.
"""
import torch


def target(p: torch.nn.Parameter, shape=None):
    """A contiguous fp32 view of p.grad with `shape` (default p.shape) to accumulate into, or None."""
    g = p.grad
    if g is None or g.dtype != torch.float32 or not p.requires_grad:
        return None
    if shape is None:
        return g if g.is_contiguous() else None
    if p.dim() == 4:                       # conv weight: want the (KO,R,S,C) physical layout
        v = g.permute(0, 2, 3, 1)
        return v if (v.is_contiguous() and tuple(v.shape) == tuple(shape)) else None
    return g if (g.is_contiguous() and tuple(g.shape) == tuple(shape)) else None


# A backward function that accumulates in place may tell the data-parallel engine "the gradient kernels of these
# parameters are enqueued" long before autograd visits the parameters (the whole ResNet backward is ONE autograd
# node: without this its 161 gradients would all become visible at its end and their all-reduce could not overlap
# with it).  The engine installs the callback; with no engine this is a no-op.
_ready_callbacks = {}        # id(parameter) -> (weakref to the parameter, engine callback)


def register(params, callback):
    """An engine (distributed.GradientBuckets) subscribes to the announcements of ITS parameters; several engines
    (several models in one process) coexist because the table is keyed by parameter, not global."""
    import weakref
    for p in params:
        _ready_callbacks[id(p)] = (weakref.ref(p), callback)


def mark_ready(params):
    """Only parameters whose gradient was accumulated IN PLACE (target() returned a buffer) may be announced."""
    if not _ready_callbacks:
        return
    for p in params:
        e = _ready_callbacks.get(id(p))
        if e is not None and e[0]() is p:
            e[1](p)

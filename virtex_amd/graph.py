"""The training step as ONE hipGraph.

The step of `scripts/pretrain_virtex.py:145-163` -- zero the gradients, forward, backward, gradient exchange, clip + SGD +
Lookahead + LR schedule -- is ~1 100 kernel launches on three HIP streams; enqueueing them costs the host 10.5 ms per step
(tools/host_profile.py), which is hidden behind the GPU at 256 images per step and is the LIMIT at 64-128 (BASELINE
configs 4 and 5).  `GraphedTrainStep` captures one step with `torch.cuda.CUDAGraph` (hipGraph underneath) -- the side
streams fork and join through events, so they become graph edges -- and replays it: one launch per step.

What had to leave the host for a replay to stay a faithful training step (by-value kernel arguments are frozen at capture):
  * the dropout seeds: every dropout-carrying kernel mixes a device word into its seed at entry (`Dropout::resolved`,
    csrc/vtx_common.h; `vtx_set_dropout_epoch`), incremented once per step on the device -- forward and backward of a step see
    the same value, successive replays different masks;
  * the LR multiplier and the Lookahead phase: computed by device operations from a device step counter and read by the
    optimizer kernel through a pointer (`vtx_sgd_lookahead_step_dev`, `FusedPretrainOptimizer.enable_device_schedule`).
BatchNorm's `num_batches_tracked` was already incremented by the finalize kernel.  Host-side mirrors (the optimizer's step
index, autograd version counters of the parameters, which the eval-mode weight cache is keyed on) are refreshed by `sync()`.

Single process only: the data-parallel all-reduces stay on the eager path (N > 1 keeps `bench.py`'s eager step).

Caller's duty (found the hard way, ROCm 7.2 / torch 2.10: hipStreamEndCapture segfaults otherwise): no output of an EAGER
step that still carries its autograd graph may be alive when the capture starts -- keep `loss.detach()`, not `loss`.
"""
import gc
from typing import Callable, Dict, Optional

import torch


class GraphedTrainStep:
    def __init__(self, model: torch.nn.Module, buckets, optimizer, example_batch: Dict[str, torch.Tensor], warmup: int = 3):
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise ValueError("GraphedTrainStep needs a GPU model")
        self.model, self.buckets, self.opt = model, buckets, optimizer
        self.static = {k: v.clone() for k, v in example_batch.items()}
        optimizer.enable_device_schedule()
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):                  # warm-up off the default stream (allocator pools, lazily set kernel
            for _ in range(max(1, warmup)):            # attributes, cached weight copies, workspaces): capture must allocate nothing new
                self._step()
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        gc.collect()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._step()
        self.replays = 0

    def _step(self):
        self.buckets.zero(); self.buckets.begin()
        out = self.model(self.static)
        out["loss"].backward()
        self.opt.step(grad_scale=self.buckets.finish())
        return out["loss"].detach()

    def __call__(self, batch: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
        """One training step on `batch` (copied into the graph's static input tensors; None: the tensors as they are).
        Returns the loss tensor of the step (device memory owned by the graph: read it before the next call)."""
        if batch is not None:
            for k, v in self.static.items():
                v.copy_(batch[k], non_blocking=True)
        self.graph.replay()
        self.replays += 1
        return self.loss

    def sync(self):
        """Refresh the host-side mirrors after replays (optimizer step index, parameter version counters)."""
        self.opt.sync_host()

"""Text-side share of the step: both heads (features -> decoder -> tied projection -> CE) forward + backward on
fixed backbone features, and the same split per phase."""
import sys, time, torch
sys.path.insert(0, ".")
import virtex_amd.factories as vf
from virtex_amd import synthetic

def timed(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3

B = 256
dev = torch.device("cuda:0")
model = vf.build_bicaptioning_model(compute_dtype=torch.bfloat16).to(dev).train()
batch = synthetic.synthetic_batch(B, dev)
feats = torch.randn(B, 7, 7, 2048, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2).requires_grad_()
for p in model.parameters():
    p.grad = torch.zeros_like(p, dtype=torch.float32)

def both(backward=True):
    model._refresh_compute_weights()
    l1 = model._head_loss(model.textual, feats, batch["caption_tokens"], batch["caption_lengths"])
    l2 = model._head_loss(model.backward_textual, feats, batch["noitpac_tokens"], batch["caption_lengths"])
    if backward:
        (l1 + l2).backward()

print(f"text heads fwd only : {timed(lambda: both(False)):.2f} ms")
print(f"text heads fwd + bwd: {timed(lambda: both(True)):.2f} ms")

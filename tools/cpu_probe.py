import os, sys, time, torch
sys.path.insert(0, ".")
from oracle import bicaptioning as port, synth
for threads in (32, 64, 128):
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    t0 = time.time(); model = port.build_model(dropout=0.1).train(); step = port.TrainStep(model, start_step=100)
    b = synth.synthetic_batch(16, seed=0); t1 = time.time()
    step(b); t2 = time.time(); step(b); t3 = time.time()
    print(f"threads {threads}: build {t1-t0:.1f}s first step {t2-t1:.1f}s second step {t3-t2:.2f}s -> {16/(t3-t2):.1f} img/s", flush=True)

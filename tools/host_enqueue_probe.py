#!/usr/bin/env python3
"""How long the host takes to ENQUEUE one training step (no synchronisation) vs how long the GPU takes to run it:
whether the step is bound by the launch rate.   python tools/host_enqueue_probe.py [config 2|3]"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, bench, mmvqa_amd
bench.CONFIG = int(sys.argv[1]) if len(sys.argv) > 1 else 2
from mmvqa_amd import synth
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = mmvqa_amd.Model(bench.make_args()); model.to(dev).train(); model.set_seed(1)
opt = mmvqa_amd.FusedAdam(model, lr=2e-5)
img, ids, seg, mask, tgt = synth.roco_batch(16, 32, 224, 30522, seed=1, device=dev)
model.tune(img, ids, seg, mask)
def step():
    logits = model(img, ids, seg, mask)
    loss, _, st = mmvqa_amd.mlm_loss(logits, tgt)
    loss.backward(); opt.step(zero_grad=True)
for _ in range(3): step()
torch.cuda.synchronize()
for trial in range(3):   # one step into an empty queue: the host's own cost, free of back-pressure
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"single step: host enqueue {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms")
for trial in range(3):
    t0 = time.perf_counter()
    for _ in range(10): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host enqueue {1e3*(t1-t0)/10:.2f} ms/step, total {1e3*(t2-t0)/10:.2f} ms/step")

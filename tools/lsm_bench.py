#!/usr/bin/env python3
"""vocabulary log-softmax + NLL forward / backward kernels at the bench shape (512 x 30522), for rocprofv3 --stats"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mmvqa_amd
torch.manual_seed(0)
x = (torch.randn(512, 30524, device="cuda") * 2)[:, :30522].requires_grad_(True)
t = torch.randint(0, 30522, (512,), device="cuda")
for _ in range(30):
    loss = mmvqa_amd.mlm_loss(x.view(16, 32, 30522), t.view(16, 32))[0]
    loss.backward()
torch.cuda.synchronize()
print("ok", float(loss))

#!/usr/bin/env python3
"""ProjectionLayer forward (projection.py:68-106) timing at the reference's shapes: B=256, K=5, D in {512, 3584, 5376}."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
dev = torch.device("cuda:0")
cfg = R.Config(); cfg.update(device=dev)
out = {}
for D in (512, 3584, 5376):
    layer = R.ProjectionLayer(cfg, D).eval()
    for B in (1, 256):
        x = torch.randn(B, 5, D, device=dev)
        with torch.no_grad():
            for _ in range(3): layer(x)
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(20): y = layer(x)
            t1.record(); torch.cuda.synchronize()
        out[f"D{D}_B{B}"] = round(t0.elapsed_time(t1) / 20, 4)
print(json.dumps(out))

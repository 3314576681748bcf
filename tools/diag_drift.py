#!/usr/bin/env python3
"""certificate statistics of the scan on a store that drifts (tests/test_gpu_reference_shapes.py::test_a_store_that_drifts...)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radad_retrievalaugmenteddeepfakeaudiodetection_amd import HipFlatIndex, _lib
gpu = torch.device("cuda:0")
def rows(row0, n, dim, seed):
    t = torch.empty((n, dim), device=gpu)
    _lib.check(_lib.load().radad_synth_rows(t.data_ptr(), row0, n, dim, seed, 0, _lib.stream_ptr(gpu)))
    return t
dim, k, B, n1 = 1024, 15, 256, 20000
for metric in (_lib.METRIC_L2, _lib.METRIC_COSINE):
    base_a = rows(0, 1, dim, 7101).abs() + 0.5
    base_b = rows(1, 1, dim, 7101).abs() * 2.0 + 0.1
    part = [base_a + 0.3 * rows(0, n1, dim, 7102), base_b + 0.3 * rows(n1, n1, dim, 7102), 30.0 * (base_a + 0.3 * rows(2 * n1, n1, dim, 7102))]
    idx = HipFlatIndex(dim, metric, 0)
    if len(sys.argv) > 1: idx.reserve(3 * n1)
    for step, r in enumerate(part):
        q = r[torch.arange(B, device=gpu) * 71 % n1] + 0.05 * rows(0, B, dim, 7110 + step)
        idx.add_device(r)
        for rep in range(5):
            idx.search_device(q, k)
            l = idx.last_launch()
            print(metric, step, rep, l["scan_kind"], l["certificate"], idx.plane_info())

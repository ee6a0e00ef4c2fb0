#!/usr/bin/env python3
"""Experiment: the CFG batch (B = 2) as ONE forward of the DiT against its two items as two B = 1 forwards on two HIP streams (separate workspaces, shared
weights) — would the HBM-bound passes of one item (AdaLN, QK-norm/RoPE) hide behind the other's GEMM / attention launches?  Timing only.
usage: two_stream_probe.py [reps] [layers]"""
import copy
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tokensgen_amd import rope as R  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 42
dev = torch.device("cuda", 0)
bf = torch.bfloat16
m1 = bench.build_model(dev, layers)
m2 = copy.copy(m1)
m2._ws = {}
g = torch.Generator(device=dev).manual_seed(0)
nf, C, H, W = 13, 16, 60, 90
x = torch.randn(2, nf, C, H, W, generator=g, device=dev).to(bf)
prompt = (torch.randn(2, 226, 4096, generator=g, device=dev) * 0.1).to(bf)
emb = torch.nn.functional.layer_norm(torch.randn(1, 5, 8, 12, 3072, generator=g, device=dev), (3072,)).permute(0, 1, 4, 2, 3).to(bf).repeat(2, 1, 1, 1, 1).contiguous()
rope = tuple(t.to(dev, torch.float32).contiguous() for t in R.rope_3d_crop(64, (0, 0, 0), (nf, 30, 45), (nf, 30, 45)))
f32 = np.float32
vr = R.rope_3d(64, np.arange(nf, dtype=f32), np.arange(30, dtype=f32), np.arange(45, dtype=f32), device=dev)
cr = R.rope_3d(64, np.linspace(1000, 1016.25, 5, dtype=f32), np.linspace(0, 30, 8, endpoint=False, dtype=f32), np.linspace(0, 45, 12, endpoint=False, dtype=f32), device=dev)
tt = torch.full((2, nf), 500, dtype=torch.int64, device=dev)


def fwd(m, lo, hi):
    return m(hidden_states=x[lo:hi], encoder_hidden_states=prompt[lo:hi], timestep=tt[lo:hi], image_rotary_emb=rope, vip_image_rotary_emb=vr,
             vip_condition_rotary_emb=cr, vip_encoder_hidden_states=emb[lo:hi], return_dict=False)[0]


def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def two():
    e = torch.cuda.Event(); e.record()
    s1.wait_event(e); s2.wait_event(e)
    with torch.cuda.stream(s1):
        fwd(m1, 0, 1)
    with torch.cuda.stream(s2):
        fwd(m2, 1, 2)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)


def seq():
    fwd(m1, 0, 1); fwd(m2, 1, 2)


for r in range(2):
    print("B=2 one forward      %.1f ms" % timed(lambda: fwd(m1, 0, 2)), flush=True)
    print("2 x B=1 sequential   %.1f ms" % timed(seq), flush=True)
    print("2 x B=1 two streams  %.1f ms" % timed(two), flush=True)

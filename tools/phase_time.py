#!/usr/bin/env python3
"""HIP-event timing of the four calls of one training step (A: seg OD, B: shape OD, C: seg OC, D: shape OC), B=32, 256x256.
    gpurun -- python tools/phase_time.py        (round 1: 27.8 / 16.7 / 26.3 / 16.7 ms)"""
import os, sys, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"wt-pse-code_amd")]
import bench
from wtpse_hip.step import TrainStep
from wtpse_hip.synth import make_batch, default_hparams
from wtpse_hip import ops
dev=torch.device("cuda:0"); hp=default_hparams(True); B=32
nets=bench.build_nets(hp,B//3,dev); ts=TrainStep(*nets,hp,dp=None)
image,od,oc=make_batch(B,256,256,dev,seed=1)
for _ in range(2): ts.step(image,od,oc)
torch.cuda.synchronize()
ev=lambda: torch.cuda.Event(enable_timing=True)
marks=[]
def mark(name):
    e=ev(); e.record(); marks.append((name,e))
orig_seg, orig_shape = ts._seg_call, ts._shape_call
def seg(*a, **k):
    mark("seg start"); r=orig_seg(*a, **k); mark("seg end"); return r
def shp(*a, **k):
    mark("shape start"); r=orig_shape(*a, **k); mark("shape end"); return r
ts._seg_call, ts._shape_call = seg, shp
mark("step start"); ts.step(image,od,oc); mark("step end")
torch.cuda.synchronize()
t0=marks[0][1]
for n,e in marks: print("%-12s %8.2f ms"%(n, t0.elapsed_time(e)))

#!/usr/bin/env python3
"""The three implementations at the bench geometry (256x128x128) side by side, as the Markdown table of DESIGN.md §4, from the committed recordings: the HIP
path (profiles/r05 + r06 snr_head_*.json, tools/snr_protocol_gpu.py), the reference's algorithm in float64 on aten GPU kernels (profiles/r06/
snr_head_aten_gpu_fp64_*.json, tests/diag/snr_protocol_aten_gpu.py) and the reference's own CPU runs (tests/golden/snr_bench_head_256x128x128.npz,
oracle/make_snr_spread.py).  Mean +- s.d. over runs of the SNR averaged over the 11 iterations up to the checkpoint; differences with their Welch s.e.

    python tools/snr_head_table.py
"""
import glob
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
z=np.load(ROOT+'/tests/golden/snr_bench_head_256x128x128.npz'); ref=z['snr'].astype(float); its=z['iterations'].astype(int)
def load(pats):
    runs=[]
    for p in pats:
        for f in sorted(glob.glob(ROOT+'/profiles/'+p)): runs+=json.load(open(f))['runs']
    n=min(len(r['snr']) for r in runs); return np.array([r['snr'][:n] for r in runs])
hip=load(['r05/snr_head_hip6.json','r05/snr_head_hip6_seeds6to11.json']+['r06/'+os.path.basename(f) for f in glob.glob(ROOT+'/profiles/r06/snr_head_*.json') if 'aten' not in f])
f64=load(['r06/snr_head_aten_gpu_fp64_*.json'])
w=lambda a,it:a[...,it-10:it+1].mean(-1)
fmt=lambda x:'%.2f +- %.2f (n = %d)'%(x.mean(),x.std(ddof=1),len(x)) if len(x)>1 else ('%.2f (n = 1)'%x[0] if len(x) else '-')
print('| iteration | HIP fp32 (n = %d) | float64, aten GPU kernels (n = %d) | reference, CPU fp32: seeds 0-2 | seeds 3-5 (round 6) | all reference seeds | HIP - reference | float64 - reference | HIP - float64 |'%(len(hip),len(f64)))
print('|---|---|---|---|---|---|---|---|---|')
for it in (100,150,220,300,400,500,599):
    h,f=w(hip,it),w(f64,it)
    a=np.array([w(ref[k],it) for k in range(3) if its[k]>it]); b=np.array([w(ref[k],it) for k in range(3,len(its)) if its[k]>it]); r=np.concatenate([a,b])
    se=lambda x,y:np.sqrt(x.var(ddof=1)/len(x)+y.var(ddof=1)/len(y))
    d=lambda x,y:'%+.2f (%.1f s.e.)'%(x.mean()-y.mean(),abs(x.mean()-y.mean())/se(x,y)) if len(x)>1 and len(y)>1 else '-'
    print('| %d | %.2f +- %.2f | %.2f +- %.2f | %s | %s | %s | %s | %s | %s |'%(it,h.mean(),h.std(ddof=1),f.mean(),f.std(ddof=1),fmt(a),' / '.join('%.2f'%v for v in b) if len(b) else '-',fmt(r),d(h,r),d(f,r),d(h,f)))

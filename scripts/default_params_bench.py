#!/usr/bin/env python3
"""One registration at a time with the library's DEFAULT parameters (m3dreg_default_params: point-to-plane pyramid 0.4 m -> 0.1 m), bucketing of both
clouds included, on a config-4 pair: python scripts/default_params_bench.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mandala_mapping_amd import abi, binding, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
R = binding.Registrar(None, device=0)
p = R.params if hasattr(R, "params") else None
src, tgt, Tg = synth.config4_pair(0, 3125)
def once():
    cs, ct = R.clouds([src, tgt], source_only=[True, False], wait=False)
    T, st = R.align(cs, ct)
    return T, st
for _ in range(5): once()
t = []
for _ in range(n):
    t0 = time.perf_counter(); T, st = once(); t.append(time.perf_counter() - t0)
t = np.array(t) * 1e3
E = np.asarray(T, np.float64) @ np.linalg.inv(Tg) if False else None
print("default parameters, 100 k pair, host payloads: median %.3f ms, min %.3f ms; iterations %d, status %d, n_corr %d" % (np.median(t), t.min(), st.iterations, st.status, st.n_corr))

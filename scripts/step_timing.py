import sys, time, ctypes as C
sys.path.insert(0, ".")
import numpy as np, torch
from mandala_mapping_amd import abi, binding, synth
from mandala_mapping_amd.pointcloud2 import encode_xyz
dev = torch.device("cuda", 0)
params = abi.Params.make(leaf=0.1, iterations=20, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
stream = torch.cuda.Stream(device=dev)
reg = binding.Registrar(params, device=0, stream=C.c_void_p(stream.cuda_stream))
payloads = []
for i in range(8):
    src, tgt, Tgt = synth.config4_pair(i)
    ms, mt = encode_xyz(src), encode_xyz(tgt)
    payloads.append((torch.frombuffer(bytearray(ms.data), dtype=torch.uint8).to(dev), ms.n, torch.frombuffer(bytearray(mt.data), dtype=torch.uint8).to(dev), mt.n))
torch.cuda.synchronize()
tb = ta = td = 0.0
for it in range(12):
    t0 = time.perf_counter()
    items = []
    for ds, ns, dt, nt in payloads:
        items += [(ds.data_ptr(), ns), (dt.data_ptr(), nt)]
    cl = reg.clouds_from_device(items)
    t1 = time.perf_counter()
    T, st = reg.align_batch([(cl[2*i], cl[2*i+1], None) for i in range(8)])
    t2 = time.perf_counter()
    del cl
    t3 = time.perf_counter()
    if it >= 2:
        tb += t1 - t0; ta += t2 - t1; td += t3 - t2
print("per step ms: bucketing %.3f  align %.3f  destroy %.3f" % (tb/10*1e3, ta/10*1e3, td/10*1e3))

"""A/B of the synchronous and the stream-ordered collective path on one GPU (world size 1, backend nccl)."""
import os, sys, time, pickle
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.cuda.set_device(0)
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
def init_pg():
    if os.environ.get("AB_NO_DEVICE_ID"):
        dist.init_process_group("nccl", rank=0, world_size=1)
    else:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
if not os.environ.get("AB_NO_PG") and not os.environ.get("AB_ENGINE_FIRST"):
    init_pg()
from bundle_adjustment_amd import distributed, engine, scene
cache = "/tmp/cfg4.pkl"
fp = pickle.load(open(cache, "rb")) if os.path.exists(cache) else scene.config("cfg4")
eng = engine.Engine(fp, image_range=None if os.environ.get('AB_NO_RANGE') else (0, fp.n_images), apply_shared=True)
eng.set_parameters(fp.values); s2 = fp.sigma2apriori
if os.environ.get('AB_ENGINE_FIRST'):
    init_pg()
dev = torch.device("cuda", 0)

def step(mode):
    if mode == "plain":
        eng.build(s2, 0.0); return eng.solve(False)
    eng.prepare_inverse(0); eng.accumulate(s2, 0.0)
    if mode == "sync":
        ptr, cnt = eng.reduce_buffer()
        buf = torch.as_tensor(distributed.DeviceArray(ptr, cnt), device=dev); dist.all_reduce(buf); torch.cuda.synchronize(dev)
    else:
        distributed.allreduce_engine_buffer(eng, dist, dev)
    eng.finalize(s2, 0.0)
    return eng.solve(False)

for mode in (("plain", "plain") if os.environ.get("AB_NO_PG") else ("plain", "sync", "async", "plain")):
    for _ in range(2): step(mode)
    torch.cuda.synchronize(); t = time.perf_counter(); acc = {}
    for _ in range(8):
        step(mode)
        for k, v in eng.timings().items(): acc[k] = acc.get(k, 0) + v / 8
    torch.cuda.synchronize()
    print(f"{mode:6s} wall={(time.perf_counter()-t)/8*1e3:6.2f} ms  " + " ".join(f"{k}={v:.2f}" for k, v in acc.items() if k in ("assembly","finalize","factor","solve","total")), flush=True)
eng.close()
if not os.environ.get('AB_NO_PG'): dist.destroy_process_group()

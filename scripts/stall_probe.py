"""Does the rare stall of the dataflow factorisation depend on how many HIP streams (hardware queues) the process holds?
    python scripts/stall_probe.py [engines held beforehand = 0] [passes = 4000] [scene = cfg3]
Creates K small engines at the same time (each takes eight streams from the library's pool, which keeps them for the life of the process),
closes them, then runs `passes` LM passes on one engine and reports jaicov_neq_kernel_stats()[6] (factorisations abandoned and repeated)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bundle_adjustment_amd import engine, scene
K = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
name = sys.argv[3] if len(sys.argv) > 3 else "cfg3"
if os.environ.get("STALL_PROBE_RCCL"):      # an RCCL communicator in the process (as after tests/test_gpu_parity.py::test_rccl_reduce_path_world1)
    import torch, torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29877")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    t = torch.ones(1 << 20, device="cuda", dtype=torch.float64)
    dist.all_reduce(t); torch.cuda.synchronize()
    if os.environ["STALL_PROBE_RCCL"] == "destroy":
        dist.destroy_process_group()
    print("RCCL communicator", "created and destroyed" if os.environ["STALL_PROBE_RCCL"] == "destroy" else "alive", flush=True)
small = scene.config("tiny_block")
held = [engine.Engine(small) for _ in range(K)]
for e in held:
    e.close()
fp = scene.config(name)
eng = engine.Engine(fp)
eng.set_parameters(fp.values)
t0 = time.time()
for i in range(n):
    eng.build(fp.sigma2apriori, 0.0)
    dx = eng.solve(False)
    assert np.isfinite(dx).all(), i
st = eng.kernel_stats()
eng.close()
print(f"streams of {K} engines in the pool, {n} passes at {name}: {st['flow_retries']} factorisations abandoned and repeated, {time.time() - t0:.1f} s", flush=True)

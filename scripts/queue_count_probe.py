"""Does the NUMBER OF HARDWARE QUEUES of the process matter to the dataflow factorisation?  K extra CU-masked streams (each is a hardware queue
of its own) are created and used once, then an ordinary problem of 94 block columns is factorised 20 times: abandoned factorisations and ms.
usage: queue_count_probe.py K [plain]   (one process per setting; GPU box)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bundle_adjustment_amd import engine, scene
K = int(sys.argv[1]); plain = len(sys.argv) > 2 and sys.argv[2] == "plain"
engine.load_library()
hip = C.CDLL("libamdhip64.so")
streams = []
buf = C.c_void_p(); assert hip.hipMalloc(C.byref(buf), 4096) == 0
for i in range(K):
    s = C.c_void_p()
    if plain:
        rc = hip.hipStreamCreateWithFlags(C.byref(s), 1)
    else:
        mask = (C.c_uint32 * 8)(*([0xFFFFFFFF] * 7 + [0x00FFFFFF if i % 2 else 0xFF000000]))
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, mask)
    assert rc == 0, rc
    assert hip.hipMemsetAsync(buf, 0, 4096, s) == 0 and hip.hipStreamSynchronize(s) == 0
    streams.append(s)
fp = scene.make_scene(140, 4000, 2000, dist=scene.DIST_RADIAL, weights="2x2", n_control=6)
eng = engine.Engine(fp, ordinary_group_elimination=1); eng.set_parameters(fp.values)
t = time.perf_counter()
for _ in range(20):
    eng.build(fp.sigma2apriori, 0.0); eng.solve(False)
ms = 1e3 * (time.perf_counter() - t) / 20
st = eng.kernel_stats()
print(f"{K} extra {'plain' if plain else 'CU-masked'} streams: {ms:.2f} ms per pass, abandoned factorisations {st['flow_retries']}", flush=True)
eng.close()

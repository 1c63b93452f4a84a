"""Stand-in for RCCL / another rank's traffic on the same GPU while the dataflow factorisation runs: streams
copies of `gb` GB buffers (HBM read + write) for `seconds` seconds in its own process, and reports the rate it reached.
    python scripts/hbm_hammer.py [seconds=120] [gb=2]"""
import sys, time
import torch
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
gb = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
n = int(gb * 2 ** 30 / 8)
a = torch.ones(n, dtype=torch.float64, device="cuda")
b = torch.empty_like(a)
torch.cuda.synchronize()
t0 = time.time(); it = 0
while time.time() - t0 < seconds:
    for _ in range(20):
        b.copy_(a); a.copy_(b)
    torch.cuda.synchronize()
    it += 40
el = time.time() - t0
print(f"hbm_hammer: {it} copies of {gb:.1f} GB in {el:.1f} s = {2 * gb * 1.073741824 * it / el / 1000:.2f} TB/s of HBM traffic", flush=True)

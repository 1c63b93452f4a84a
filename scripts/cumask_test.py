import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_amd import engine
L = engine.load_library()
L.jaicov_debug_cumask.argtypes = [C.POINTER(C.c_uint32), C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
def run(bits, label):
    w = [0]*8
    for b in bits: w[b//32] |= (1 << (b%32))
    m = (C.c_uint32 * 8)(*w); ms = C.c_double(); tf = C.c_double()
    rc = L.jaicov_debug_cumask(m, 512, 400, C.byref(ms), C.byref(tf))
    print(f"{label:34s} nbits={len(bits):3d} rc={rc} {ms.value:8.2f} ms {tf.value:6.2f} TF/s  ~{tf.value/77.6*256:6.1f} CUs", flush=True)
for k in (0, 1, 7, 8, 9, 31, 32, 33, 64, 255):
    run([k], f"single bit {k}")
run([0, 1], "bits 0,1"); run([0, 8], "bits 0,8"); run([0, 32], "bits 0,32"); run([0, 8, 16, 24], "0,8,16,24")
run(list(range(0, 256, 8)), "every 8th"); run(list(range(0, 64, 8)), "0..63 step 8"); run(list(range(8)), "0-7")
run(list(range(0, 16)), "0-15"); run([b for b in range(256) if b % 8 != 0], "all but every 8th")
run([b for b in range(256) if b % 32 >= 2], "drop bits 0,1 of each word")
print("---- reservation candidates")
run(list(range(248, 256)), "bits 248-255")
run(list(range(0, 248)), "bits 0-247")
run(list(range(240, 256)), "bits 240-255")
run(list(range(0, 240)), "bits 0-239")

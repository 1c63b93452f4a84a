// Store -> poll round trips between two workgroups (gfx950): what one hop of a flag chain costs, by cache scope and by placement.
//   hipcc --offload-arch=gfx950 -O3 -o hop_latency hop_latency.hip && ./hop_latency
// Workgroup b runs on XCD b % 8 (checked with XCC_ID).  Ping = workgroup 0, pong = workgroup `peer`: 8 -> same XCD, 1 -> the next XCD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
enum { SC1 = 0, SC0 = 1, BOTH = 2 };
template <int LOADK>
__device__ __forceinline__ unsigned long long ld(const unsigned long long *p) {
    if (LOADK == SC0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int STOREK>
__device__ __forceinline__ void st(unsigned long long *p, unsigned long long v) {
    if (STOREK == SC1 || STOREK == BOTH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (STOREK == SC0 || STOREK == BOTH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// flags: [0] ping -> pong, [16] pong -> ping (separate lines).  rounds round trips; out[0] = ticks (100 MHz), out[1], out[2] = XCC ids
template <int LOADK, int STOREK>
__global__ void pingpong(unsigned long long *flags, int peer, int rounds, long long *out) {
    const int b = blockIdx.x;
    if (b != 0 && b != peer) return;
    if (threadIdx.x != 0) return;
    const int xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 0xf;
    if (b == 0) {
        out[1] = xcc;
        int lost = 0;
        const long long t0 = wall_clock64();
        for (int r = 1; r <= rounds; r++) {
            st<STOREK>(flags, (unsigned long long)r);
            int spin = 0;
            while (ld<LOADK>(flags + 16) != (unsigned long long)r && ++spin < (1 << 13)) {}
            if (spin >= (1 << 13)) lost++;
        }
        out[0] = wall_clock64() - t0;
        out[3] = lost;
    } else {
        out[2] = xcc;
        for (int r = 1; r <= rounds; r++) {
            int spin = 0;
            while (ld<LOADK>(flags) != (unsigned long long)r && ++spin < (1 << 13)) {}
            st<STOREK>(flags + 16, (unsigned long long)r);
        }
    }
}
template <int LOADK, int STOREK>
static void run(const char *name, int peer) {
    unsigned long long *flags; long long *out, h[4];
    hipMalloc(&flags, 4096); hipMalloc(&out, 64);
    const int rounds = 300;
    for (int rep = 0; rep < 2; rep++) {
        hipMemset(flags, 0, 4096); hipMemset(out, 0, 64);
        hipLaunchKernelGGL((pingpong<LOADK, STOREK>), dim3(16), dim3(64), 0, 0, flags, peer, rounds, out);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s peer %2d  XCD %lld -> %lld : %7.3f us per hop (round trip / 2)\n", name, peer, h[1], h[2], (double)h[0] * 0.01 / rounds / 2.0);
    if (h[3]) printf("    (%lld of %d round trips timed out: the flag was never seen through this path)\n", h[3], rounds);
    fflush(stdout);
    hipFree(flags); hipFree(out);
}
int main() {
    for (int peer : {8, 1}) {
        run<SC1, SC1>("store sc1 (agent), load sc1 (agent)", peer);
        run<SC0, SC1>("store sc1, load sc0 (workgroup scope)", peer);
        run<SC0, BOTH>("store sc1 + sc0, load sc0", peer);
        run<SC0, SC0>("store sc0, load sc0", peer);
    }
    return 0;
}

// Exploration probe: can an error-free SLICED product on the low-precision matrix cores beat the fp64 matrix pipe for the
// trailing update C -= A B' of the dataflow Cholesky?  This file measures what the chip gives; scripts/ozaki_study.py has the numerical half
// (slices needed) and puts the two together.  Not part of the product, nothing links it.
//   hipcc --offload-arch=gfx950 -O3 -o ozaki_probe ozaki_probe.hip && ./ozaki_probe
//   1. issue rate of v_mfma_i32_16x16x64_i8 / i32_32x32x32_i8 / f32_16x16x32_bf16 / f64_16x16x4 from registers (independent chains)
//   2. one tile step of the sliced product from LDS: a 128 x 128 tile of C (4 waves x 64 x 64), K = 128, S slices of int8 per operand,
//      every slice pair (p, q) with p + q <= S + 1, int32 accumulators per scale group p + q flushed into fp64 accumulators once per
//      64 columns of K with the group's power of two and the row / column scales of that k-block -- everything a real kernel has to do
//      per step EXCEPT the global loads (the slices sit in LDS; same bytes per element as fp64 at S = 8).  Time per step, one and two
//      workgroups per CU where they fit, against the fp64 pipe's 14.3 us per 128^3 step.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef double v4d __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(int *out, int iters) {
    const int t = threadIdx.x;
    if (KIND == 0) {            // i8 16x16x64: 32 768 ops per instruction
        v4i a = {t, t + 1, t + 2, t + 3}, b = {t * 3, 7, t ^ 5, 11};
        v4i c[8];
        for (int i = 0; i < 8; i++) c[i] = v4i{0, 0, 0, 0};
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int i = 0; i < 8; i++) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[i], 0, 0, 0);
        int s = 0;
        for (int i = 0; i < 8; i++) s += c[i][0] + c[i][3];
        out[blockIdx.x * 256 + t] = s;
    } else if (KIND == 1) {     // i8 32x32x32: 65 536 ops per instruction
        v4i a = {t, t + 1, t + 2, t + 3}, b = {t * 3, 7, t ^ 5, 11};
        v16i c[4];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 16; j++) c[i][j] = 0;
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int i = 0; i < 4; i++) c[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[i], 0, 0, 0);
        int s = 0;
        for (int i = 0; i < 4; i++) s += c[i][0] + c[i][15];
        out[blockIdx.x * 256 + t] = s;
    } else if (KIND == 2) {     // bf16 16x16x32: 16 384 flops per instruction
        v8bf a, b;
        for (int i = 0; i < 8; i++) { a[i] = (__bf16)(float)((t + i) & 7); b[i] = (__bf16)(float)((t * 3 + i) & 3); }
        v4f c[8];
        for (int i = 0; i < 8; i++) c[i] = v4f{0, 0, 0, 0};
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int i = 0; i < 8; i++) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[i], 0, 0, 0);
        float s = 0;
        for (int i = 0; i < 8; i++) s += c[i][0];
        out[blockIdx.x * 256 + t] = (int)s;
    } else {                    // f64 16x16x4: 2 048 flops per instruction
        double a = 1.0 + t * 1e-3, b = 2.0 - t * 1e-3;
        v4d c[8];
        for (int i = 0; i < 8; i++) c[i] = v4d{0, 0, 0, 0};
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int i = 0; i < 8; i++) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
        double s = 0;
        for (int i = 0; i < 8; i++) s += c[i][0];
        out[blockIdx.x * 256 + t] = (int)s;
    }
}

// ---- the sliced tile step -----------------------------------------------------------------------------------------------------------
// LDS: A slices [S][2 k-blocks][128 rows][64 k] int8 = S x 16 KB, B the same: 2 S x 16 KB (S = 8: 256 KB -- too much: ONE k-block of 64
// resident per operand, 2 S x 8 KB = 128 KB at S = 8, one workgroup per CU; the k loop re-reads the same block: the probe times the
// arithmetic + LDS traffic of a step, not its staging).  Fragment of the 16x16x64 MFMA: lane l holds 16 consecutive k of row l & 15
// starting at k = 16 (l >> 4): one ds_read_b128.
template <int S>
__global__ __launch_bounds__(256) void tile_step_kernel(double *out, const double *scales, int steps) {
    extern __shared__ char lds[];
    char *As = lds, *Bs = lds + S * 8192;                 // [slice][row 0..127][k 0..63]
    double *rs = reinterpret_cast<double *>(lds + 2 * S * 8192), *cs = rs + 128;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wr = wave >> 1, wc = wave & 1;
    for (int i = t; i < 2 * S * 8192 / 4; i += 256) reinterpret_cast<int *>(lds)[i] = (i * 2654435761u) >> 7;
    if (t < 128) { rs[t] = scales[t]; cs[t] = scales[128 + t]; }
    __syncthreads();
    v4d facc[4][4];
    for (int x = 0; x < 4; x++) for (int y = 0; y < 4; y++) facc[x][y] = v4d{0, 0, 0, 0};
    const int l15 = lane & 15, l4 = lane >> 4;
    for (int st = 0; st < steps; st++)
        for (int kb = 0; kb < 2; kb++) {                  // K = 128 per step = two k-blocks of 64
            asm volatile("" ::: "memory");                // (the block in LDS is the same every time: the loads must not be hoisted out of the loop)
#pragma unroll
            for (int x = 0; x < 4; x++) {
                asm volatile("" ::: "memory");
                v4i a[S];
                const int row = 64 * wr + 16 * x + l15;
#pragma unroll
                for (int p = 0; p < S; p++) a[p] = *reinterpret_cast<const v4i *>(As + p * 8192 + row * 64 + 16 * l4);
                // row scales of this lane's four C rows (l4 + 4 r) in this k-block
                double rsv[4];
#pragma unroll
                for (int r = 0; r < 4; r++) rsv[r] = rs[64 * wr + 16 * x + l4 + 4 * r];
#pragma unroll
                for (int y = 0; y < 4; y++) {
                    asm volatile("" ::: "memory");
                    v4i b[S];
                    const int col = 64 * wc + 16 * y + l15;
#pragma unroll
                    for (int q = 0; q < S; q++) b[q] = *reinterpret_cast<const v4i *>(Bs + q * 8192 + col * 64 + 16 * l4);
                    const double csv = cs[col];
                    // groups g = p + q (0-based: 0 .. S - 1 kept: p + q <= S - 1): int32 accumulation, then one fp64 flush per group
#pragma unroll
                    for (int g = 0; g < S; g++) {
                        v4i acc = {0, 0, 0, 0};
#pragma unroll
                        for (int p = 0; p <= g; p++) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[p], b[g - p], acc, 0, 0, 0);
                        const double w = __builtin_ldexp(csv, -7 * g);
#pragma unroll
                        for (int r = 0; r < 4; r++) facc[x][y][r] = __builtin_fma((double)acc[r], w * rsv[r], facc[x][y][r]);
                    }
                }
            }
        }
    double s = 0;
    for (int x = 0; x < 4; x++) for (int y = 0; y < 4; y++) s += facc[x][y][0] + facc[x][y][3];
    out[blockIdx.x * 256 + t] = s;
}

template <int S>
static void run_tile(int cus, int wg_per_cu, double *d_out, const double *d_sc) {
    const int steps = 200;
    const size_t lds = (size_t)2 * S * 8192 + 256 * sizeof(double);
    if (lds * wg_per_cu > 160 * 1024) { printf("  S=%d: %d workgroups per CU do not fit the LDS (%zu KB each)\n", S, wg_per_cu, lds / 1024); return; }
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(tile_step_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int grid = cus * wg_per_cu;
    hipLaunchKernelGGL(tile_step_kernel<S>, dim3(grid), dim3(256), lds, 0, d_out, d_sc, 10);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(tile_step_kernel<S>, dim3(grid), dim3(256), lds, 0, d_out, d_sc, steps);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double us_step = 1e3 * ms / steps;                        // every workgroup does `steps` steps side by side
    const double pairs = S * (S + 1) / 2.0;
    const double tf_equiv = 2.0 * 128 * 128 * 128 * grid / (us_step * 1e-6) / 1e12;
    printf("  S=%d (%2.0f slice pairs), %d workgroup(s) per CU: %.2f us per 128^3 step and workgroup -> %.1f TFLOP/s fp64-equivalent on %d CUs "
           "(int8 MFMA at %.0f TOP/s)\n", S, pairs, wg_per_cu, us_step, tf_equiv, cus, tf_equiv * pairs);
}

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs\n", prop.name, cus);
    int *d_i;
    CHK(hipMalloc(&d_i, 4096 * 256 * sizeof(int)));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const char *names[4] = {"v_mfma_i32_16x16x64_i8", "v_mfma_i32_32x32x32_i8", "v_mfma_f32_16x16x32_bf16", "v_mfma_f64_16x16x4_f64"};
    const double ops[4] = {8 * 32768.0, 4 * 65536.0, 8 * 16384.0, 8 * 2048.0};
    for (int kind = 0; kind < 4; kind++) {
        const int blocks = 8 * cus, iters = kind == 3 ? 4000 : 20000;
        for (int rep = 0; rep < 2; rep++) {
            CHK(hipEventRecord(e0));
            if (kind == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(256), 0, 0, d_i, iters);
            if (kind == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(256), 0, 0, d_i, iters);
            if (kind == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(blocks), dim3(256), 0, 0, d_i, iters);
            if (kind == 3) hipLaunchKernelGGL(rate_kernel<3>, dim3(blocks), dim3(256), 0, 0, d_i, iters);
            CHK(hipEventRecord(e1));
            CHK(hipDeviceSynchronize());
            float ms = 0;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 1) printf("%-26s %8.1f T(FL)OP/s  (%d blocks x 4 waves x %d x %g ops, %.2f ms)\n", names[kind],
                                 ops[kind] * 4.0 * blocks * iters / (ms * 1e-3) / 1e12, blocks, iters, ops[kind], ms);
        }
    }
    double *d_out, *d_sc;
    CHK(hipMalloc(&d_out, (size_t)2 * cus * 256 * sizeof(double)));
    std::vector<double> sc(256);
    for (int i = 0; i < 256; i++) sc[i] = 1.0 / (1 << (i % 5));
    CHK(hipMalloc(&d_sc, 256 * sizeof(double)));
    CHK(hipMemcpy(d_sc, sc.data(), 256 * sizeof(double), hipMemcpyHostToDevice));
    printf("sliced 128^3 tile step from LDS (no global loads), fp64 pipe for comparison: 14.3 us per step and CU = 75 TFLOP/s at 2.29 GHz\n");
    for (int w = 1; w <= 2; w++) {
        run_tile<5>(cus, w, d_out, d_sc);
        run_tile<6>(cus, w, d_out, d_sc);
        run_tile<7>(cus, w, d_out, d_sc);
        run_tile<8>(cus, w, d_out, d_sc);
        run_tile<9>(cus, w, d_out, d_sc);
    }
    return 0;
}

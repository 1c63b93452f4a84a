// fp64 MFMA GEMM family for gfx950 (v_mfma_f64_16x16x4_f64), the workhorse of the factorisation, the inverse and the
// dense J'WJ contraction.  Row-major C.  One workgroup = 4 waves (2x2), 128x128 output tile, each wave 64x64 =
// 4x4 MFMA tiles (16 accumulator tiles x 4 f64 = 128 VGPRs), BK = 16 per LDS stage, register-staged double buffering.
//
// Operand layouts (what is contiguous in global memory):
//   A: LAY_KC  A(i,k) = A[i*lda + k]        A: LAY_XC  A(i,k) = A[k*lda + i]   (i.e. the transpose is stored)
//   B: LAY_KC  B(k,j) = B[j*ldb + k]        B: LAY_XC  B(k,j) = B[k*ldb + j]
// so  C = A.B^T of two row-major matrices is (KC,KC); C = A.B row-major is (KC,XC); C = A^T.B is (XC,XC).
//
// LDS image of both operands is k-major: S[k][x], row stride 144 doubles (1152 B == 128 mod 256), which makes the
// MFMA fragment reads (lanes 0-15 -> 16 consecutive x at k, lanes 16-31 -> same x at k+1, ...) hit all 64 banks once
// per 32-lane group: conflict-free ds_read_b64 (MI355X_MICROARCH.md, LDS table).
//
// f64 MFMA lane maps (cdna_hip_programming.md 3): A[i = l&15][k = l>>4], B[k = l>>4][j = l&15],
// C/D: col = l&15, row = (l>>4) + 4*reg.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

namespace jaicov {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

enum { LAY_KC = 0, LAY_XC = 1 };
enum { KMODE_FULL = 0, KMODE_LE_ROW = 1, KMODE_GE_ROW = 2, KMODE_GE_COL = 3 };

constexpr int GEMM_BK = 16;

struct GemmArgs {
    const double *A, *B;
    double *C;
    long lda, ldb, ldc;
    int M, N, K;         // multiples of 128 / 128 / 16
    double alpha, beta;  // C = alpha*A*B + beta*C
    int lower_only;      // square tile grid: skip tiles with tile_col > tile_row
    int kmode;           // restrict the k range per tile (triangular operands), see KMODE_*
    long strideA, strideB, strideC;  // batch strides (blockIdx.y)
    long strideA2, strideB2, strideC2;  // second batch dimension (blockIdx.z; gridDim.z = batch2, default 1)
    int batch_sum_limit;             // > 0: batches with blockIdx.y + blockIdx.z >= the limit do not exist (triangular batch sets)
    long long *trace;    // debug: per workgroup {start, loop begin, loop end, end} of the 100 MHz clock + hardware id
    const int2 *tile_map; // optional: workgroup -> (tile_row, tile_col), row < 0 = no tile (see xcd_tile_map)
    int n_map;            // entries of tile_map = workgroups to launch
};

// TM = 128: the throughput tile described above.  TM = 64: latency variant for launches that cannot fill the chip with
// 128-tiles (panel GEMMs of the factorisation): 4x the workgroups, each wave 32x32 = 2x2 MFMA tiles, LDS stride 80.
// TAG only gives a launch site its own kernel symbol, so that rocprofv3 lists it separately (TAG 1 = Cholesky trailing update).
template <int ALAY, int BLAY, int TM = 128, int TN = 128, int TAG = 0>
__global__ __launch_bounds__(256, TM * TN >= 8192 ? 2 : 4) void gemm_f64_kernel(GemmArgs g) {
    constexpr int GEMM_BM = TM, GEMM_BN = TN;
    constexpr int LDA_S = TM + 16, LDB_S = TN + 16;   // LDS row strides: == 16 mod 32 doubles -> conflict-free fragments
    constexpr int MTM = TM / 32, MTN = TN / 32;       // MFMA tiles per wave
    constexpr int EPA = TM / 16, EPB = TN / 16;       // doubles per thread and stage
    constexpr int WTM = TM / 2, WTN = TN / 2;         // wave tile
    constexpr int STAGE = GEMM_BK * (LDA_S + LDB_S);
    __shared__ double smem[2 * STAGE];  // [stage][A: k][x] [B: k][x]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    int tile_row, tile_col;
    if (g.tile_map) {
        const int2 rc = g.tile_map[blockIdx.x];
        if (rc.x < 0) return;
        tile_row = rc.x;
        tile_col = rc.y;
    } else {
        const int t = blockIdx.x;
        if (g.lower_only) {
            // t -> (row, col) over the lower triangle, row-major: t = row(row+1)/2 + col
            int r = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
            while ((long)(r + 1) * (r + 2) / 2 <= t) ++r;
            while ((long)r * (r + 1) / 2 > t) --r;
            tile_row = r;
            tile_col = t - r * (r + 1) / 2;
        } else {
            const int tn = g.N / GEMM_BN;
            tile_row = t / tn;
            tile_col = t - tile_row * tn;
        }
    }
    long long t_start = 0, t_loop0 = 0, t_loop1 = 0, c_loop0 = 0;
    if (g.trace) t_start = wall_clock64();
    const long m0 = (long)tile_row * GEMM_BM, n0 = (long)tile_col * GEMM_BN;
    int kbeg = 0, kend = g.K;
    if (g.kmode == KMODE_LE_ROW) kend = min(g.K, (tile_row + 1) * GEMM_BM);
    else if (g.kmode == KMODE_GE_ROW) kbeg = min(g.K, tile_row * GEMM_BM);
    else if (g.kmode == KMODE_GE_COL) kbeg = min(g.K, tile_col * GEMM_BN);

    if (g.batch_sum_limit > 0 && (int)(blockIdx.y + blockIdx.z) >= g.batch_sum_limit) return;
    const double *A = g.A + (long)blockIdx.y * g.strideA + (long)blockIdx.z * g.strideA2;
    const double *B = g.B + (long)blockIdx.y * g.strideB + (long)blockIdx.z * g.strideB2;
    double *C = g.C + (long)blockIdx.y * g.strideC + (long)blockIdx.z * g.strideC2;

    // ---- global -> register staging (8 doubles per operand per thread) -------------------------------------
    const double *ap, *bp;
    long astep, bstep;
    int a_lds, b_lds;  // LDS element offset of this thread's first element
    // Operand in KC layout (k contiguous in memory): 8 lanes fetch one 128-byte line = the 16 k of one row, so a wave
    // instruction touches 8 lines.  (One lane per row, 64 lines per instruction, kept the L1's address path busy for
    // the whole k-step: the loop ran 10 % faster with its global loads removed, and equally fast with this pattern.)
    // The thread's pair (k = 2 seg, 2 seg + 1) goes to LDS rows 2 seg + (seg & 1) and the other one of the pair, at
    // column row + 8 (seg >> 1 & 1) + 4 (seg >> 2): the 32 lanes of a half-wave (8 seg x 4 rows) then hit the 32
    // double-wide banks 16 (seg & 1) + 8 (seg >> 1 & 1) + 4 (seg >> 2) + row % 4 once each.  The k order inside a block
    // of 4 is free as long as A and B agree (ksw), and the column shift is the same for the 4 LDS rows of one MFMA
    // k-block (8 (ks & 1) + 4 (ks >> 1) below), so the fragment reads keep compile-time offsets.
    const int seg = tid & 7, rbase = tid >> 3;
    const int a_odd = (seg & 1) * LDA_S, b_odd = (seg & 1) * LDB_S;   // x (k = 2 seg) goes to LDS row 2 seg + (seg & 1), y to the other
    auto ksw = [](int k) { return (k & ~1) | ((k ^ (k >> 1)) & 1); };   // LDS row of k-in-tile
    if (ALAY == LAY_KC) {
        ap = A + (m0 + rbase) * g.lda + kbeg + 2 * seg;
        astep = GEMM_BK;
        a_lds = (2 * seg) * LDA_S + rbase + 8 * ((seg >> 1) & 1) + 4 * (seg >> 2);
    } else {
        const int kr = tid >> 4, ms = tid & 15;
        ap = A + (long)(kbeg + kr) * g.lda + m0 + 2 * ms;   // 16 lanes x 16 bytes = 2 lines of one k row per instruction
        astep = (long)GEMM_BK * g.lda;
        a_lds = ksw(kr) * LDA_S + 2 * ms;
    }
    if (BLAY == LAY_KC) {
        bp = B + (n0 + rbase) * g.ldb + kbeg + 2 * seg;
        bstep = GEMM_BK;
        b_lds = (2 * seg) * LDB_S + rbase + 8 * ((seg >> 1) & 1) + 4 * (seg >> 2);
    } else {
        const int kr = tid >> 4, ns = tid & 15;
        bp = B + (long)(kbeg + kr) * g.ldb + n0 + 2 * ns;
        bstep = (long)GEMM_BK * g.ldb;
        b_lds = ksw(kr) * LDB_S + 2 * ns;
    }

    d2_t ra[EPA / 2], rb[EPB / 2];
    auto gload = [&]() {
#pragma unroll
        for (int j = 0; j < EPA / 2; j++)
            ra[j] = *reinterpret_cast<const d2_t *>(ALAY == LAY_KC ? ap + (long)(32 * j) * g.lda : ap + 32 * j);
#pragma unroll
        for (int j = 0; j < EPB / 2; j++)
            rb[j] = *reinterpret_cast<const d2_t *>(BLAY == LAY_KC ? bp + (long)(32 * j) * g.ldb : bp + 32 * j);
        ap += astep;
        bp += bstep;
    };
    auto lstore = [&](int stage) {
        double *sa = smem + stage * STAGE + a_lds;
        double *sb = smem + stage * STAGE + GEMM_BK * LDA_S + b_lds;
        if (ALAY == LAY_KC) {
#pragma unroll
            for (int j = 0; j < EPA / 2; j++) {
                sa[32 * j + a_odd] = ra[j].x;
                sa[32 * j + LDA_S - a_odd] = ra[j].y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < EPA / 2; j++) *reinterpret_cast<d2_t *>(sa + 32 * j) = ra[j];
        }
        if (BLAY == LAY_KC) {
#pragma unroll
            for (int j = 0; j < EPB / 2; j++) {
                sb[32 * j + b_odd] = rb[j].x;
                sb[32 * j + LDB_S - b_odd] = rb[j].y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < EPB / 2; j++) *reinterpret_cast<d2_t *>(sb + 32 * j) = rb[j];
        }
    };

    // accumulators start from (beta/alpha)*C, so the epilogue is a pure store (alpha*acc): the C tile is fetched while
    // the first operand tiles (requested first) are still in flight instead of as a dependent read-modify-write at the end.
    const int nk = (kend - kbeg) / GEMM_BK;
    if (nk > 0) gload();
    const double alpha = g.alpha, beta = g.beta;
    double *cbase = C + (m0 + WTM * wr + (lane >> 4)) * g.ldc + n0 + WTN * wc + (lane & 15);
    d4_t acc[MTM][MTN];
    if (beta != 0.0) {
        const double bs = beta / alpha;
#pragma unroll
        for (int i = 0; i < MTM; i++)
#pragma unroll
            for (int j = 0; j < MTN; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc[i][j][r] = bs * cbase[(long)(16 * i + 4 * r) * g.ldc + 16 * j];
    } else {
#pragma unroll
        for (int i = 0; i < MTM; i++)
#pragma unroll
            for (int j = 0; j < MTN; j++) acc[i][j] = (d4_t){0.0, 0.0, 0.0, 0.0};
    }

    if (g.trace) { t_loop0 = wall_clock64(); c_loop0 = clock64(); }
    if (nk > 0) {
        lstore(0);
        __syncthreads();
        const int fa = (lane >> 4) * LDA_S + WTM * wr + (lane & 15);
        const int fb = (lane >> 4) * LDB_S + WTN * wc + (lane & 15);
        for (int kt = 0; kt < nk; kt++) {
            const int st = kt & 1;
            if (kt + 1 < nk && TAG < 2) gload();   // TAG 2, 3: timing experiments of jaicov_debug_gemm_trace (wrong results)
            const double *sa = smem + st * STAGE + fa;
            const double *sb = smem + st * STAGE + GEMM_BK * LDA_S + fb;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                double a[MTM], b[MTN];
#pragma unroll
                for (int i = 0; i < MTM; i++) a[i] = sa[(4 * ks) * LDA_S + 16 * i + (ALAY == LAY_KC ? 8 * (ks & 1) + 4 * (ks >> 1) : 0)];
#pragma unroll
                for (int j = 0; j < MTN; j++) b[j] = sb[(4 * ks) * LDB_S + 16 * j + (BLAY == LAY_KC ? 8 * (ks & 1) + 4 * (ks >> 1) : 0)];
#pragma unroll
                for (int i = 0; i < MTM; i++)
#pragma unroll
                    for (int j = 0; j < MTN; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            if (TAG == 3) continue;
            if (kt + 1 < nk) lstore(st ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue ------------------------------------------------------------------------------------------
    long long c_loop1 = 0;
    if (g.trace) { t_loop1 = wall_clock64(); c_loop1 = clock64(); }
#pragma unroll
    for (int i = 0; i < MTM; i++)
#pragma unroll
        for (int j = 0; j < MTN; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) cbase[(long)(16 * i + 4 * r) * g.ldc + 16 * j] = alpha * acc[i][j][r];
    if (g.trace && tid == 0) {
        long long *t = g.trace + 8 * ((long)blockIdx.y * gridDim.x + blockIdx.x);
        t[0] = t_start; t[1] = t_loop0; t[2] = t_loop1; t[3] = wall_clock64();
        t[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
        t[5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // XCC_ID
        t[6] = c_loop1 - c_loop0;   // shader-clock cycles of the k loop (against t[2]-t[1] at 100 MHz: the clock the loop ran at)
    }
}

// XCD-aware order of the lower-triangular tile grid (T tile rows).  Workgroups are dealt round-robin to the 8 XCDs
// (workgroup w runs on XCD w % 8), each XCD has its own 4 MB L2 and runs ~62 tiles at a time.  The map gives every XCD
// whole 8x8 super-tiles: the 64 tiles that are in flight together on one XCD share 8 A strips and 8 B strips and walk
// along k in step, so a strip is fetched from HBM once per super-tile instead of once per tile.
inline std::vector<int2> xcd_tile_map(int T) {
    constexpr int S = 8, X = 8;
    struct Super { int R, C, count; };
    std::vector<Super> sup;
    const int nS = (T + S - 1) / S;
    for (int R = 0; R < nS; R++)
        for (int C = 0; C <= R; C++) {
            int cnt = 0;
            for (int r = R * S; r < std::min(T, (R + 1) * S); r++)
                for (int c = C * S; c < std::min(T, (C + 1) * S); c++) cnt += c <= r;
            sup.push_back({R, C, cnt});
        }
    std::stable_sort(sup.begin(), sup.end(), [](const Super &a, const Super &b) { return a.count > b.count; });
    std::vector<std::vector<int2>> lists(X);
    for (const Super &s : sup) {
        int best = 0;
        for (int x = 1; x < X; x++)
            if (lists[x].size() < lists[best].size()) best = x;
        for (int r = s.R * S; r < std::min(T, (s.R + 1) * S); r++)
            for (int c = s.C * S; c < std::min(T, (s.C + 1) * S); c++)
                if (c <= r) lists[best].push_back(make_int2(r, c));
    }
    for (;;) {   // level the tails tile by tile: all XCDs finish within one tile of each other
        int lo = 0, hi = 0;
        for (int x = 1; x < X; x++) {
            if (lists[x].size() < lists[lo].size()) lo = x;
            if (lists[x].size() > lists[hi].size()) hi = x;
        }
        if (lists[hi].size() <= lists[lo].size() + 1) break;
        lists[lo].push_back(lists[hi].back());
        lists[hi].pop_back();
    }
    size_t len = 0;
    for (auto &l : lists) len = std::max(len, l.size());
    std::vector<int2> map(len * X, make_int2(-1, -1));
    for (int x = 0; x < X; x++)
        for (size_t q = 0; q < lists[x].size(); q++) map[q * X + x] = lists[x][q];
    return map;
}

// Launches that cannot give every CU two 128-tiles (the regime in which the 128-tile runs at its rate) take the
// 64-tile latency variant; `small_tiles` < 0 = that rule, 0 = never, 1 = always (rectangular KMODE_FULL calls only).
constexpr int GEMM_SMALL_TILE_LIMIT = 480;
constexpr int GEMM_TINY_TILE_LIMIT = 256;   // 64-tiles below which the 32-tile variant is used
inline hipError_t gemm_f64(hipStream_t s, int alay, int blay, const GemmArgs &g, int batch = 1, int small_tiles = -1, int tag = 0, int batch2 = 1) {
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    const int tm = g.M / 128, tn = g.N / 128;
    const int tiles = g.lower_only ? tm * (tm + 1) / 2 : tm * tn;
    dim3 grid(g.tile_map ? g.n_map : tiles, batch, batch2), block(256);
    if (g.lower_only && small_tiles == 1 && g.kmode == KMODE_FULL && alay == LAY_KC && blay == LAY_KC && g.M == g.N) {
        // lower-triangular grid in 64-tiles: 4x the workgroups, a quarter of the time each.  For the trailing update of
        // the factorisation's tail, where the panel chain on the other stream waits for its workgroups to retire.
        GemmArgs h = g;
        h.tile_map = nullptr; h.n_map = 0;
        const int t64 = g.M / 64;
        grid.x = t64 * (t64 + 1) / 2;
        if (tag == 1) hipLaunchKernelGGL((gemm_f64_kernel<LAY_KC, LAY_KC, 64, 64, 1>), grid, block, 0, s, h);
        else hipLaunchKernelGGL((gemm_f64_kernel<LAY_KC, LAY_KC, 64, 64>), grid, block, 0, s, h);
        return hipGetLastError();
    }
    const bool can_small = !g.lower_only && g.kmode == KMODE_FULL && alay == LAY_KC && blay == LAY_KC;
    if (can_small && (small_tiles > 0 || (small_tiles < 0 && tiles * batch < GEMM_SMALL_TILE_LIMIT))) {
        // even the 64-tile leaves most SIMDs idle when only a few block rows remain; a wave then spends its k-step in
        // 16 dependent-issue MFMAs.  The 32-tile (one MFMA tile per wave) cuts that to 4.
        const bool tiny = small_tiles < 0 && tiles * batch * 4 < GEMM_TINY_TILE_LIMIT;
        if (g.C == g.A || g.C == g.B) {   // in place (one column tile): keep the whole row of C in one workgroup
            if (tn != 1) return hipErrorInvalidValue;
            if (tiny) {
                grid.x = tiles * 4;
                hipLaunchKernelGGL((gemm_f64_kernel<LAY_KC, LAY_KC, 32, 128>), grid, block, 0, s, g);
            } else {
                grid.x = tiles * 2;
                hipLaunchKernelGGL((gemm_f64_kernel<LAY_KC, LAY_KC, 64, 128>), grid, block, 0, s, g);
            }
        } else if (tiny) {
            grid.x = tiles * 16;
            hipLaunchKernelGGL((gemm_f64_kernel<LAY_KC, LAY_KC, 32, 32>), grid, block, 0, s, g);
        } else {
            grid.x = tiles * 4;
            hipLaunchKernelGGL((gemm_f64_kernel<LAY_KC, LAY_KC, 64, 64>), grid, block, 0, s, g);
        }
        return hipGetLastError();
    }
    if (alay == LAY_KC && blay == LAY_KC && tag == 2) hipLaunchKernelGGL((gemm_f64_kernel<LAY_KC, LAY_KC, 128, 128, 2>), grid, block, 0, s, g);
    else if (alay == LAY_KC && blay == LAY_KC && tag == 3) hipLaunchKernelGGL((gemm_f64_kernel<LAY_KC, LAY_KC, 128, 128, 3>), grid, block, 0, s, g);
    else if (alay == LAY_KC && blay == LAY_KC && tag == 1) hipLaunchKernelGGL((gemm_f64_kernel<LAY_KC, LAY_KC, 128, 128, 1>), grid, block, 0, s, g);
    else if (alay == LAY_KC && blay == LAY_KC) hipLaunchKernelGGL((gemm_f64_kernel<LAY_KC, LAY_KC>), grid, block, 0, s, g);
    else if (alay == LAY_KC && blay == LAY_XC) hipLaunchKernelGGL((gemm_f64_kernel<LAY_KC, LAY_XC>), grid, block, 0, s, g);
    else if (alay == LAY_XC && blay == LAY_XC) hipLaunchKernelGGL((gemm_f64_kernel<LAY_XC, LAY_XC>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_f64_kernel<LAY_XC, LAY_KC>), grid, block, 0, s, g);
    return hipGetLastError();
}

}  // namespace jaicov

// riccati.hip -- constrained discrete Riccati backward recursion, dlqr(A,Bu,Bλ,G,Q,R,N) of src/control/lqr.jl:141-184 and its
// time-varying twin dlqr(mechanism, ...) of src/control/lqr_tracking.jl:73-122 (A,Bu,Bλ,G indexed by knot).
//
// Two launch shapes (RESIDENT: one workgroup per problem with P in LDS; TILED: every step tiled over the device), see below.
// The sweep is sequential in k, so all parallelism inside a problem is in the dense algebra of one step.  The mx x mx products (P [A' | D], Abar' (P Abar)) run on the
// fp64 matrix cores (v_mfma_f64_16x16x4_f64, 16x16 tiles per wavefront); the linear solves are in-kernel LUs with partial
// pivoting (Julia's `\` on a square matrix) followed by one-column-per-thread substitution.
// Statement-by-statement correspondence with lqr.jl is marked with the line numbers.
#include "cclqr_internal.h"
#include <math.h>
#include <type_traits>

namespace cclqr {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) double lds_double;   // explicit LDS address space: ds_read/ds_write instead of flat_*
#define RIC_THREADS 512
#define RIC_WAVES (RIC_THREADS / 64)

// C (M x N, ldc) = beta * C + alpha * op(A) (M x K) * B (K x N);  op(A) = A' when TA (A stored K x M).  Whole workgroup.
// Each wavefront owns a 32x32 block of C = 2x2 tiles of v_mfma_f64_16x16x4_f64, so every operand fragment feeds two MFMAs,
// and k is unrolled by two so that eight loads are in flight before the first MFMA of an iteration.
// fragment maps: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15], C/D: col = lane&15, row = (lane>>4) + 4*reg
// (cdna_hip_programming.md §3).
template <bool TA>
__device__ void wg_gemm(int M, int N, int K, double alpha, const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb,
                        double beta, double* __restrict__ C, int ldc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int bm = (M + 31) >> 5, bn = (N + 31) >> 5;
    for (int blk = wave; blk < bm * bn; blk += RIC_WAVES) {
        const int i0 = (blk / bn) << 5, j0 = (blk % bn) << 5;
        v4d acc[2][2] = {{{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}, {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}};
        const bool iok[2] = {(i0 + li) < M, (i0 + 16 + li) < M}, jok[2] = {(j0 + li) < N, (j0 + 16 + li) < N};
        for (int k0 = 0; k0 < K; k0 += 8) {
            double a[2][2], b[2][2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int k = k0 + 4 * u + lk;
                const bool kok = k < K;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int i = i0 + 16 * h + li, j = j0 + 16 * h + li;
                    a[u][h] = (kok && iok[h]) ? (TA ? A[(size_t)k * lda + i] : A[(size_t)i * lda + k]) : 0.0;
                    b[u][h] = (kok && jok[h]) ? B[(size_t)k * ldb + j] : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int hi = 0; hi < 2; hi++)
#pragma unroll
                    for (int hj = 0; hj < 2; hj++) acc[hi][hj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][hi], b[u][hj], acc[hi][hj], 0, 0, 0);
        }
#pragma unroll
        for (int hi = 0; hi < 2; hi++)
#pragma unroll
            for (int hj = 0; hj < 2; hj++) {
                if (!jok[hj]) continue;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = i0 + 16 * hi + lk + 4 * r;
                    if (row < M) {
                        double* c = C + (size_t)row * ldc + j0 + 16 * hj + li;
                        *c = (beta == 0.0 ? 0.0 : beta * *c) + alpha * acc[hi][hj][r];
                    }
                }
            }
    }
    __syncthreads();
}

// in-place LU with partial pivoting of the n x n matrix A (row major, lda); piv[c] = pivot row of column c.  *sing set if a pivot is 0.
template <typename MP>
__device__ void wg_lu(int n, MP A, int lda, int* piv, int* sing) {
    const int tid = threadIdx.x;
    for (int c = 0; c < n; c++) {
        // pivot search by the first wavefront
        if (tid < 64) {
            double best = -1.0; int bi = c;
            for (int r = c + tid; r < n; r += 64) { double v = fabs(A[(size_t)r * lda + c]); if (v > best) { best = v; bi = r; } }
            for (int o = 32; o > 0; o >>= 1) {
                double ov = __shfl_xor(best, o, 64); int oi = __shfl_xor(bi, o, 64);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            if (tid == 0) { piv[c] = bi; if (!(best > 0.0)) *sing = 1; }
        }
        __syncthreads();
        const int p = piv[c];
        if (p != c)
            for (int j = tid; j < n; j += RIC_THREADS) { double t = A[(size_t)c * lda + j]; A[(size_t)c * lda + j] = A[(size_t)p * lda + j]; A[(size_t)p * lda + j] = t; }
        __syncthreads();
        const double inv = 1.0 / A[(size_t)c * lda + c];
        for (int r = c + 1 + tid; r < n; r += RIC_THREADS) A[(size_t)r * lda + c] *= inv;
        __syncthreads();
        // rank-1 update of the trailing block on a 16 x 32 thread grid (no per-element index division)
        {
            const int tr = tid >> 5, tc = tid & 31;
            for (int r = c + 1 + tr; r < n; r += RIC_THREADS / 32) {
                const double l = A[(size_t)r * lda + c];
                MP Ar = A + (size_t)r * lda;
                MP Ac = A + (size_t)c * lda;
                for (int j = c + 1 + tc; j < n; j += 32) Ar[j] -= l * Ac[j];
            }
        }
        __syncthreads();
    }
}
// solve (LU) X = B for nrhs columns, B (n x nrhs, ldb) in place; one column per thread.
// With Xs != nullptr the columns are processed in batches of CB staged in LDS (Xs[i*CB + thread]: conflict-free), so the
// n^2 multiply-adds per column read LDS instead of re-reading B from global memory.
template <typename MP>
__device__ void wg_lu_solve(int n, MP LU, int lda, const int* piv, double* B, int ldb, int nrhs, MP Xs, int CB) {
    if (CB > 0) {
        const int tid = threadIdx.x;
        for (int c0 = 0; c0 < nrhs; c0 += CB) {
            const int j = c0 + tid;
            if (tid < CB && j < nrhs) {
                MP x = Xs + tid;
                for (int i = 0; i < n; i++) x[(size_t)i * CB] = B[(size_t)i * ldb + j];
                for (int c = 0; c < n; c++) { int p = piv[c]; if (p != c) { double t = x[(size_t)c * CB]; x[(size_t)c * CB] = x[(size_t)p * CB]; x[(size_t)p * CB] = t; } }
                // dot products unrolled by 8 with 4 accumulators: 16 LDS loads are in flight before the first multiply-add
                for (int i = 1; i < n; i++) {
                    MP Li = LU + (size_t)i * lda;
                    double s[4] = {x[i * CB], 0.0, 0.0, 0.0};
                    int r = 0;
                    for (; r + 8 <= i; r += 8) {
                        double lv[8], xv[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) { lv[u] = Li[r + u]; xv[u] = x[(r + u) * CB]; }
#pragma unroll
                        for (int u = 0; u < 8; u++) s[u & 3] -= lv[u] * xv[u];
                    }
                    for (; r < i; r++) s[0] -= Li[r] * x[r * CB];
                    x[i * CB] = (s[0] + s[1]) + (s[2] + s[3]);
                }
                for (int i = n - 1; i >= 0; i--) {
                    MP Li = LU + (size_t)i * lda;
                    double s[4] = {x[i * CB], 0.0, 0.0, 0.0};
                    int r = i + 1;
                    for (; r + 8 <= n; r += 8) {
                        double lv[8], xv[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) { lv[u] = Li[r + u]; xv[u] = x[(r + u) * CB]; }
#pragma unroll
                        for (int u = 0; u < 8; u++) s[u & 3] -= lv[u] * xv[u];
                    }
                    for (; r < n; r++) s[0] -= Li[r] * x[r * CB];
                    x[i * CB] = ((s[0] + s[1]) + (s[2] + s[3])) / Li[i];
                }
                for (int i = 0; i < n; i++) B[(size_t)i * ldb + j] = x[(size_t)i * CB];
            }
            __syncthreads();
        }
        return;
    }
    for (int j = threadIdx.x; j < nrhs; j += RIC_THREADS) {
        for (int c = 0; c < n; c++) { int p = piv[c]; if (p != c) { double t = B[(size_t)c * ldb + j]; B[(size_t)c * ldb + j] = B[(size_t)p * ldb + j]; B[(size_t)p * ldb + j] = t; } }
        for (int i = 1; i < n; i++) { double s = B[(size_t)i * ldb + j]; for (int r = 0; r < i; r++) s -= LU[(size_t)i * lda + r] * B[(size_t)r * ldb + j]; B[(size_t)i * ldb + j] = s; }
        for (int i = n - 1; i >= 0; i--) {
            double s = B[(size_t)i * ldb + j];
            for (int r = i + 1; r < n; r++) s -= LU[(size_t)i * lda + r] * B[(size_t)r * ldb + j];
            B[(size_t)i * ldb + j] = s / LU[(size_t)i * lda + i];
        }
    }
    __syncthreads();
}

// The recursion runs in its PROJECTED form: with E = (G Bλ)^-1 G Bu and F = (G Bλ)^-1 G A,
//   D  = Bu - Bλ E            (lqr.jl:151, the reference's D)
//   A' = A  - Bλ F            (dynamics projected onto the constraint manifold)
// the second block row of M Kk = b (lqr.jl:154-160) gives Kλ = F - E Ku, and substituting it into the first block row leaves
//   (R + D' P D) Ku = D' P A' ,   Abar = A - Bu Ku - Bλ Kλ = A' - D Ku        (lqr.jl:160,169)
// i.e. the same Kk and Abar as the reference's (mu+ml)-square solve, at the cost of a mu-square one; E, F, D, A' do not depend on P,
// so a time-invariant problem computes them once.  Only Ku is stored by the reference (lqr.jl:162-164), Kλ is never formed.

#ifdef CCLQR_PROFILE
enum { RP_PA, RP_GAIN, RP_UPD, RP_PP, RP_NORM, RP_STEPS, RP_WAVE_W = 8, RP_WAVE_PP = 16, RP_N = 24 };      // RP_WAVE_*: per wavefront (block 0), cycles inside its own W / Pkp1 tiles
static __device__ unsigned long long g_rprof[RP_N];
#define RWAVE(c, t0_) do { if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) g_rprof[(c) + (threadIdx.x >> 6)] += __builtin_readcyclecounter() - (t0_); } while (0)
#define RNOW() __builtin_readcyclecounter()
#define RSTAMP(c) do { if (threadIdx.x == 0 && blockIdx.x == 0) { unsigned long long t1_ = __builtin_readcyclecounter(); g_rprof[c] += t1_ - rt0; rt0 = t1_; } } while (0)
#else
#define RSTAMP(c)
#define RWAVE(c, t0_)
#define RNOW() 0ull
#endif

#define RIC_MU_REG 7   // the mu x mu system of a backward step is solved in registers up to this many inputs (8: spills under the 256-register budget)
#define RIC_LDS_M 96   // G Bλ (ml x ml) is kept in LDS for its pivoted LU when ml <= 96 (72 KB)

// =====================================================================================================================
// TILED path: one backward step = three launches over a grid of 32x32 tiles (x problems), so a single large problem uses
// the whole device instead of one CU, and a batch of problems fills it with tiles rather than with whole recursions:
//   ric_project_kernel   (once per knot)   [A' | D]                                             -> AD
//   ric_pa_kernel        W = Pk [A' | D]                                                        (tiles of W)
//   ric_gain_update_kernel   break test of the previous step; S = R + D'PkD; Ku = S \ D'PkA' -> K[k];
//                        Abar = A' - D Ku ; Pk Abar = W_A' - W_D Ku                              (row blocks)
//   ric_pn_kernel        Pkp1 = Q + Ku'RKu + Abar'(Pk Abar); per-tile |Pk - Pkp1|^2             (tiles of P)
// Each tile is computed by four wavefronts that split the k range and are summed in a fixed order through LDS; the norm is
// summed per tile and then over tiles in index order, so results do not depend on scheduling.  A problem that has met the
// break criterion (lqr.jl:172) raises its `stop` flag and its later launches return at once.
#define TILE_THREADS 256
struct RicGrid {
    int nprob, mx, mu, ml, N, nlin, na, tm, tn;
    double tol;
    const double *A, *Bu, *Bl, *G, *Q, *R;
    double *AD, *W, *Abar, *P, *Ku, *KRK, *part, *TSp, *scratch, *K;
    int *stop, *kbreak, *status;
    long long kpad;    // doubles between the gain tables of consecutive problems in K (per-instance controller tables: capi.hip gain_row_overrun)
    int keep_last;     // 1: only the gain of the last executed backward step is kept (K [nprob][mu][mx] = Ku[1] after the back-fill of
                       // lqr.jl:179-181 = the one gain LQR{T,Inf} keeps, lqr.jl:40-43); 0: the whole table K [nprob][N-1][mu][mx]
    int bf16_terms;    // 0: fp64 MFMA (parity mode); 1..3: the two mx^3 products of a backward step on bf16 MFMA with fp32 accumulation,
                       // every fp64 operand split into that many bf16 terms (measured-error mode, BASELINE configs[3]; tiled path only)
    unsigned char pp_mask[8];   // resident kernel, register-fragment form: bit ct of pp_mask[w] = wavefront w computes the 16x16 tile (its own
                                // row strip, column tile ct) of the SYMMETRIC Pkp1 and mirrors it (ric_pp_assign)
};

template <bool LDSM>
__global__ __launch_bounds__(RIC_THREADS) void ric_project_kernel(RicGrid a, int lds_cols) {
    extern __shared__ double lds_M[];
    typedef typename std::conditional<LDSM, lds_double*, double*>::type MP;
    __shared__ int sing;
    const int knot = blockIdx.x, prob = blockIdx.y, tid = threadIdx.x;
    const int mx = a.mx, mu = a.mu, ml = a.ml, na = a.na;
    const size_t lin = (size_t)prob * a.nlin + knot;
    const double *A = a.A + lin * mx * mx, *Bu = a.Bu + lin * mx * mu, *Bl = a.Bl + lin * mx * ml, *G = a.G + lin * ml * mx;
    double* AD = a.AD + lin * mx * na;
    double* sc = a.scratch + lin * (((size_t)ml * ml + (size_t)ml * na + 3) & ~(size_t)1);
    double* GBlg = sc;
    double* X = sc + (((size_t)ml * ml + 1) & ~(size_t)1);
    int* piv = (int*)(a.scratch + (size_t)a.nprob * a.nlin * (((size_t)ml * ml + (size_t)ml * na + 3) & ~(size_t)1)) + lin * (ml + 2);
    MP GBl = LDSM ? (MP)lds_M : (MP)GBlg;
    MP Xs = LDSM ? (MP)lds_M + (size_t)ml * ml : (MP) nullptr;
    if (tid == 0) sing = 0;
    const int tr = tid >> 5, tc = tid & 31;
    for (int i = tr; i < mx; i += RIC_THREADS / 32) {
        for (int j = tc; j < mx; j += 32) AD[(size_t)i * na + j] = A[(size_t)i * mx + j];
        for (int j = tc; j < mu; j += 32) AD[(size_t)i * na + mx + j] = Bu[(size_t)i * mu + j];
        if (knot == 0) for (int j = tc; j < mx; j += 32) a.P[((size_t)prob * mx + i) * mx + j] = a.Q[(size_t)i * mx + j];   // Pk = Q  lqr.jl:147
    }
    __syncthreads();
    if (ml > 0) {
        wg_gemm<false>(ml, ml, mx, 1.0, G, mx, Bl, ml, 0.0, GBlg, ml);                   // G*Bλ                         lqr.jl:155
        wg_gemm<false>(ml, na, mx, 1.0, G, mx, AD, na, 0.0, X, na);                      // [G*A | G*Bu]                 lqr.jl:158,154
        if (LDSM) { for (int e = tid; e < ml * ml; e += RIC_THREADS) GBl[e] = GBlg[e]; __syncthreads(); }
        wg_lu<MP>(ml, GBl, ml, piv, &sing);
        if (sing) { if (tid == 0) { a.status[prob] = CCLQR_ESINGULAR_; a.stop[prob] = 1; a.kbreak[prob] = knot + 1; } return; }
        wg_lu_solve<MP>(ml, GBl, ml, piv, X, na, na, Xs, LDSM ? lds_cols : 0);           // X = (G Bλ)^-1 G [A | Bu]
        wg_gemm<false>(mx, na, ml, -1.0, Bl, ml, X, na, 1.0, AD, na);                    // [A' | D]                     lqr.jl:151
    }
}

// 32x32 tile of op(A)' B summed over k, split over the four wavefronts of the workgroup (wave w takes the k-groups of 4 with
// index = w mod 4), then reduced through LDS in wave order.  la(k, h) / lb(k, h) return operand elements A[k][i0+16h+li] / B[k][j0+16h+li]
// (already masked to 0 outside the matrix).  On return red[e], e = row*32 + col, holds the tile; all threads have synchronised.
template <class FA, class FB>
__device__ inline void tile_mfma_splitk(int K, FA la, FB lb, double (*red)[1024]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    v4d acc[2][2] = {{{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}, {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}};
    const int ngroups = (K + 3) >> 2;
    // CH k-groups per wavefront and pass (4 CH loads in flight before the first MFMA): passes of 8 while they are full, then of 2
    auto pass = [&](auto chtag, int g0) {
        constexpr int CH = decltype(chtag)::value;
        double av[CH][2], bv[CH][2];
#pragma unroll
        for (int u = 0; u < CH; u++) {
            const int k = 4 * (g0 + 4 * u) + lk;
            const bool kok = (g0 + 4 * u) < ngroups && k < K;
#pragma unroll
            for (int h = 0; h < 2; h++) { av[u][h] = kok ? la(k, h) : 0.0; bv[u][h] = kok ? lb(k, h) : 0.0; }
        }
#pragma unroll
        for (int u = 0; u < CH; u++)
#pragma unroll
            for (int hi = 0; hi < 2; hi++)
#pragma unroll
                for (int hj = 0; hj < 2; hj++) acc[hi][hj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][hi], bv[u][hj], acc[hi][hj], 0, 0, 0);
    };
    // every pass is one round trip to L2 (its loads are issued together, its MFMAs wait for them), so the tail is covered by the
    // FEWEST passes -- a masked pass of 8 or 4 rather than a string of passes of 2 (mx = 204: 13 k-groups per wavefront = 8 + 8
    // masked instead of 8 + 2 + 2 + 2; the masked groups cost idle MFMAs, not latency)
    int g0 = wave;
    for (; g0 + 4 * 7 < ngroups; g0 += 4 * 8) pass(std::integral_constant<int, 8>{}, g0);
    while (g0 < ngroups) {
        const int left = (ngroups - g0 + 3) >> 2;        // k-groups this wavefront still has
        if (left > 4) { pass(std::integral_constant<int, 8>{}, g0); g0 += 4 * 8; }
        else if (left > 2) { pass(std::integral_constant<int, 4>{}, g0); g0 += 4 * 4; }
        else { pass(std::integral_constant<int, 2>{}, g0); g0 += 4 * 2; }
    }
#pragma unroll
    for (int hi = 0; hi < 2; hi++)
#pragma unroll
        for (int hj = 0; hj < 2; hj++)
#pragma unroll
            for (int r = 0; r < 4; r++) red[wave][(16 * hi + lk + 4 * r) * 32 + 16 * hj + li] = acc[hi][hj][r];
    __syncthreads();
}

// ---- measured-error mode (BASELINE configs[3] "dense Riccati on MFMA bf16 -> fp32 accumulate"): the same 32x32 split-k tile on
// v_mfma_f32_16x16x16_bf16.  Every fp64 operand x is split on the fly into NS bf16 terms x ~ t0 + t1 + t2 (t0 = bf16(x),
// t1 = bf16(x - t0), ...), the products t_i(A) t_j(B) with i + j < NS are accumulated in fp32 (NS = 1: one product = plain bf16;
// 2: three; 3: six) and the tile goes back to fp64 after the wave reduction.  What limits the accuracy is the term count for
// NS <= 2 and the fp32 accumulator for NS = 3; the error of the resulting gains and the time are REPORTED (tools/gpu_riccati_bf16.py,
// tests/test_gpu_setup.py::test_riccati_bf16_split_mode_error_is_measured), the fp64 path stays the parity mode.
typedef short v4s __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ short bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (short)(u >> 16);
}
__device__ __forceinline__ float bf16_to_float(short b) { return __uint_as_float(((unsigned)(unsigned short)b) << 16); }
template <int NS>
__device__ __forceinline__ void bf16_split(double x, short* t) {
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const short b = bf16_rne((float)x);
        t[s] = b;
        x -= (double)bf16_to_float(b);
    }
}
template <int NS, class FA, class FB>
__device__ inline void tile_mfma_splitk_bf16(int K, FA la, FB lb, double (*red)[1024]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    v4f acc[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
    const int ngroups = (K + 15) >> 4;                     // k-groups of 16: lane (li, lk) holds k = 16 g + 4 lk + 0..3
    for (int g = wave; g < ngroups; g += 4) {
        v4s at[2][NS], bt[2][NS];
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int k = 16 * g + 4 * lk + j;
                short ta[NS], tb[NS];
                bf16_split<NS>(k < K ? la(k, h) : 0.0, ta);
                bf16_split<NS>(k < K ? lb(k, h) : 0.0, tb);
#pragma unroll
                for (int s = 0; s < NS; s++) { at[h][s][j] = ta[s]; bt[h][s][j] = tb[s]; }
            }
#pragma unroll
        for (int hi = 0; hi < 2; hi++)
#pragma unroll
            for (int hj = 0; hj < 2; hj++)
#pragma unroll
                for (int sa = NS - 1; sa >= 0; sa--)        // smallest products first
#pragma unroll
                    for (int sb = NS - 1; sb >= 0; sb--)
                        if (sa + sb < NS) acc[hi][hj] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(at[hi][sa], bt[hj][sb], acc[hi][hj], 0, 0, 0);
    }
#pragma unroll
    for (int hi = 0; hi < 2; hi++)
#pragma unroll
        for (int hj = 0; hj < 2; hj++)
#pragma unroll
            for (int r = 0; r < 4; r++) red[wave][(16 * hi + 4 * lk + r) * 32 + 16 * hj + li] = (double)acc[hi][hj][r];
    __syncthreads();
}
// dispatch on the precision mode of the launch
template <class FA, class FB>
__device__ inline void tile_product(int bf16_terms, int K, FA la, FB lb, double (*red)[1024]) {
    if (bf16_terms == 0) tile_mfma_splitk(K, la, lb, red);
    else if (bf16_terms == 1) tile_mfma_splitk_bf16<1>(K, la, lb, red);
    else if (bf16_terms == 2) tile_mfma_splitk_bf16<2>(K, la, lb, red);
    else tile_mfma_splitk_bf16<3>(K, la, lb, red);
}

// has problem `prob` already stopped, or does the step before this one (k+1) meet the break criterion?  (uniform over the workgroup)
__device__ inline bool ric_stopped(const RicGrid& a, int prob, int k, int ntile_p, int* flag) {
    if (threadIdx.x < 64) {             // first wavefront: strided partial sums, then a butterfly -- the same order in every workgroup
        int st = a.stop[prob];
        if (!st && k < a.N - 1) {
            const double* part = a.part + ((size_t)((k + 1) & 1) * a.nprob + prob) * ntile_p;
            double tot = 0.0;
            for (int t = threadIdx.x; t < ntile_p; t += 64) tot += part[t];
            for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
            if (sqrt(tot) < a.tol) st = 2;                                            // if norm(Pk-Pkp1) < 1e-5  break   lqr.jl:172-174
        }
        if (threadIdx.x == 0) *flag = st;
    }
    __syncthreads();
    return *flag != 0;
}

__global__ __launch_bounds__(TILE_THREADS) void ric_pa_kernel(RicGrid a, int k) {
    __shared__ double red[4][1024];
    extern __shared__ double Dl[];      // D rows of this tile, [32][mu]
    __shared__ int flag;
    const int prob = blockIdx.y, tid = threadIdx.x, lane = tid & 63, li = lane & 15;
    const int mx = a.mx, na = a.na;
    if (ric_stopped(a, prob, k, a.tm * a.tm, &flag)) return;
    const int i0 = (blockIdx.x / a.tn) << 5, j0 = (blockIdx.x % a.tn) << 5;
    const double* P = a.P + ((size_t)((a.N - 1 - k) & 1) * a.nprob + prob) * mx * mx;
    const double* AD = a.AD + ((size_t)prob * a.nlin + (a.nlin > 1 ? k - 1 : 0)) * mx * na;
    double* W = a.W + (size_t)prob * mx * na;
    const bool iok[2] = {i0 + li < mx, i0 + 16 + li < mx}, jok[2] = {j0 + li < na, j0 + 16 + li < na};
    const int mu = a.mu, ti = blockIdx.x / a.tn;
    for (int t = tid; t < 32 * mu; t += TILE_THREADS) { const int r = t / mu, q = t % mu; Dl[t] = (i0 + r < mx) ? AD[(size_t)(i0 + r) * na + mx + q] : 0.0; }
    tile_product(a.bf16_terms, mx,
        [&](int kk, int h) { return iok[h] ? P[(size_t)kk * mx + i0 + 16 * h + li] : 0.0; },      // Pk symmetric
        [&](int kk, int h) { return jok[h] ? AD[(size_t)kk * na + j0 + 16 * h + li] : 0.0; }, red);
    for (int e = tid; e < 1024; e += TILE_THREADS) {
        const int i = i0 + (e >> 5), j = j0 + (e & 31);
        const double v = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
        red[0][e] = v;
        if (i < mx && j < na) W[(size_t)i * na + j] = v;
    }
    __syncthreads();
    // this tile's rows of D' W (summed over the row tiles, in order, by the gain kernel)
    for (int t = tid; t < mu * 32; t += TILE_THREADS) {
        const int q = t >> 5, c = t & 31;
        if (j0 + c >= na) continue;
        double sacc = 0.0;
#pragma unroll 8
        for (int r = 0; r < 32; r++) sacc += Dl[r * mu + q] * red[0][r * 32 + c];
        a.TSp[(((size_t)prob * a.tm + ti) * mu + q) * na + j0 + c] = sacc;
    }
}

// S = R + D' Pk D ; Ku = S \ (D' Pk A')   (= the first mu rows of M \ b, lqr.jl:152-164), then
// Abar = A' - D Kuk (= A-Bu*Kuk-Bλ*Kλk, lqr.jl:169) and Pk Abar = W_A' - W_D Kuk for RU rows per workgroup.
// Every workgroup of a problem derives the same Ku from the tile partials of D'W (mu x mu solve: cheaper than a launch);
// workgroup 0 publishes it (K[k], Ku, R Ku) and owns the stop flag.
#define RU 16
__global__ __launch_bounds__(TILE_THREADS) void ric_gain_update_kernel(RicGrid a, int k) {
    extern __shared__ double gl[];      // TS [mu][na] | S [mu][mu] | piv [mu]
    __shared__ int flag, sing;
    const int prob = blockIdx.y, tid = threadIdx.x;
    const int mx = a.mx, mu = a.mu, na = a.na;
    const bool lead = blockIdx.x == 0;
    if (ric_stopped(a, prob, k, a.tm * a.tm, &flag)) {
        if (lead && tid == 0 && flag == 2) { a.stop[prob] = 1; a.kbreak[prob] = k + 1; }
        return;
    }
    double* TS = gl;
    double* S = gl + (size_t)mu * na;
    int* piv = (int*)(S + (size_t)mu * mu);
    if (tid == 0) sing = 0;
    for (int e = tid; e < mu * na; e += TILE_THREADS) {
        const double* tp = a.TSp + (size_t)prob * a.tm * mu * na + e;
        double sacc = 0.0;
        for (int ti = 0; ti < a.tm; ti++) sacc += tp[(size_t)ti * mu * na];
        TS[e] = sacc;
    }
    __syncthreads();
    for (int e = tid; e < mu * mu; e += TILE_THREADS) S[e] = a.R[e] + TS[(size_t)(e / mu) * na + mx + e % mu];
    __syncthreads();
    for (int c = 0; c < mu; c++) {                       // LU with partial pivoting, mu <= 32
        if (tid == 0) {
            double best = -1.0; int bi = c;
            for (int r = c; r < mu; r++) { double v = fabs(S[r * mu + c]); if (v > best) { best = v; bi = r; } }
            piv[c] = bi;
            if (!(best > 0.0)) sing = 1;
        }
        __syncthreads();
        if (sing) break;
        const int p = piv[c];
        if (p != c && tid < mu) { double t = S[c * mu + tid]; S[c * mu + tid] = S[p * mu + tid]; S[p * mu + tid] = t; }
        __syncthreads();
        if (tid > c && tid < mu) S[tid * mu + c] /= S[c * mu + c];
        __syncthreads();
        for (int e = tid; e < (mu - c - 1) * (mu - c - 1); e += TILE_THREADS) {
            const int r = c + 1 + e / (mu - c - 1), j = c + 1 + e % (mu - c - 1);
            S[r * mu + j] -= S[r * mu + c] * S[c * mu + j];
        }
        __syncthreads();
    }
    if (sing) { if (lead && tid == 0) { a.status[prob] = CCLQR_ESINGULAR_; a.stop[prob] = 1; a.kbreak[prob] = k; } return; }
    double* Ku = a.Ku + (size_t)prob * mu * mx;
    double* KRK = a.KRK + (size_t)prob * mu * mx;
    double* Kout = a.K + (a.keep_last ? (size_t)prob : ((size_t)prob * (a.N - 1) + (k - 1))) * mu * mx + (size_t)prob * a.kpad;
    for (int j = tid; j < mx; j += TILE_THREADS) {
        for (int c = 0; c < mu; c++) { const int p = piv[c]; if (p != c) { double t = TS[(size_t)c * na + j]; TS[(size_t)c * na + j] = TS[(size_t)p * na + j]; TS[(size_t)p * na + j] = t; } }
        for (int i = 1; i < mu; i++) { double sacc = TS[(size_t)i * na + j]; for (int r = 0; r < i; r++) sacc -= S[i * mu + r] * TS[(size_t)r * na + j]; TS[(size_t)i * na + j] = sacc; }
        for (int i = mu - 1; i >= 0; i--) {
            double sacc = TS[(size_t)i * na + j];
            for (int r = i + 1; r < mu; r++) sacc -= S[i * mu + r] * TS[(size_t)r * na + j];
            TS[(size_t)i * na + j] = sacc / S[i * mu + i];
        }
        if (lead) {
            for (int q = 0; q < mu; q++) { const double v = TS[(size_t)q * na + j]; Ku[(size_t)q * mx + j] = v; Kout[(size_t)q * mx + j] = v; }   // Ku[k][i] = Kk[i:i,:]  lqr.jl:162-164
            for (int q = 0; q < mu; q++) { double sacc = 0.0; for (int r = 0; r < mu; r++) sacc += a.R[q * mu + r] * TS[(size_t)r * na + j]; KRK[(size_t)q * mx + j] = sacc; }
        }
    }
    __syncthreads();
    const double* AD = a.AD + ((size_t)prob * a.nlin + (a.nlin > 1 ? k - 1 : 0)) * mx * na;
    double* W = a.W + (size_t)prob * mx * na;
    double* Abar = a.Abar + (size_t)prob * mx * mx;
    // D and Pk D rows of this block into LDS (S, piv are dead now), then all loads of a column before the arithmetic
    double* DP = S;                     // [RU][2 mu]; S has mu*mu doubles, the tail of the dynamic allocation covers the rest
    const int i_base = blockIdx.x * RU;
    for (int t = tid; t < RU * mu; t += TILE_THREADS) {
        const int r = t / mu, q = t % mu, i = i_base + r;
        DP[r * 2 * mu + q] = (i < mx) ? AD[(size_t)i * na + mx + q] : 0.0;
        DP[r * 2 * mu + mu + q] = (i < mx) ? W[(size_t)i * na + mx + q] : 0.0;
    }
    __syncthreads();
    for (int j = tid; j < mx; j += TILE_THREADS) {
        double ab[RU], pw[RU];
#pragma unroll
        for (int r = 0; r < RU; r++) {
            const int i = i_base + r;
            ab[r] = (i < mx) ? AD[(size_t)i * na + j] : 0.0;
            pw[r] = (i < mx) ? W[(size_t)i * na + j] : 0.0;
        }
        for (int q = 0; q < mu; q++) {
            const double kq = TS[(size_t)q * na + j];
#pragma unroll
            for (int r = 0; r < RU; r++) { ab[r] -= DP[r * 2 * mu + q] * kq; pw[r] -= DP[r * 2 * mu + mu + q] * kq; }
        }
#pragma unroll
        for (int r = 0; r < RU; r++) {
            const int i = i_base + r;
            if (i < mx) { Abar[(size_t)i * mx + j] = ab[r]; W[(size_t)i * na + j] = pw[r]; }
        }
    }
}

// Pkp1 = Q + Kuk'*R*Kuk + Abar'*(Pk*Abar) (lqr.jl:170), tiles of P; the per-tile |Pk - Pkp1|^2 feeds the break test
__global__ __launch_bounds__(TILE_THREADS) void ric_pn_kernel(RicGrid a, int k) {
    __shared__ double red[4][1024];
    extern __shared__ double kt[];      // KuI [mu][32] | KRKJ [mu][32]
    __shared__ double wsum[4];
    __shared__ int flag;
    const int prob = blockIdx.y, tid = threadIdx.x, lane = tid & 63, li = lane & 15;
    const int mx = a.mx, mu = a.mu, na = a.na;
    if (a.stop[prob]) return;           // the gain kernel of this step has already evaluated the break test
    (void)flag;
    const int i0 = (blockIdx.x / a.tm) << 5, j0 = (blockIdx.x % a.tm) << 5;
    const double* P = a.P + ((size_t)((a.N - 1 - k) & 1) * a.nprob + prob) * mx * mx;
    double* Pn = a.P + ((size_t)((a.N - k) & 1) * a.nprob + prob) * mx * mx;
    const double* W = a.W + (size_t)prob * mx * na;
    const double* Ku = a.Ku + (size_t)prob * mu * mx;
    const double* KRK = a.KRK + (size_t)prob * mu * mx;
    double *KuI = kt, *KRKJ = kt + mu * 32;
    for (int e = tid; e < mu * 32; e += TILE_THREADS) {
        const int q = e >> 5, c = e & 31;
        KuI[e] = (i0 + c < mx) ? Ku[(size_t)q * mx + i0 + c] : 0.0;
        KRKJ[e] = (j0 + c < mx) ? KRK[(size_t)q * mx + j0 + c] : 0.0;
    }
    const double* Abar = a.Abar + (size_t)prob * mx * mx;
    const bool iok[2] = {i0 + li < mx, i0 + 16 + li < mx}, jok[2] = {j0 + li < mx, j0 + 16 + li < mx};
    tile_product(a.bf16_terms, mx,
        [&](int kk, int h) { return iok[h] ? Abar[(size_t)kk * mx + i0 + 16 * h + li] : 0.0; },
        [&](int kk, int h) { return jok[h] ? W[(size_t)kk * na + j0 + 16 * h + li] : 0.0; }, red);
    double acc = 0.0;
    for (int e = tid; e < 1024; e += TILE_THREADS) {
        const int ii = e >> 5, jj = e & 31, i = i0 + ii, j = j0 + jj;
        if (i < mx && j < mx) {
            double v = a.Q[(size_t)i * mx + j];
            for (int q = 0; q < mu; q++) v += KuI[q * 32 + ii] * KRKJ[q * 32 + jj];
            v += ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
            Pn[(size_t)i * mx + j] = v;
            const double d = P[(size_t)i * mx + j] - v;
            acc += d * d;
        }
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) wsum[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) a.part[((size_t)(k & 1) * a.nprob + prob) * (a.tm * a.tm) + blockIdx.x] = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
}

// =====================================================================================================================
// RESIDENT path (problems whose P and W = P [A'|D] fit one CU's LDS, mx up to 96): one 512-thread workgroup per problem
// runs the whole sweep with P and W resident in LDS and the [A'|D] / Abar MFMA fragments in registers (ric_wave_tiles).  Per step:
//   W = Pk [A' | D]                       16x16 MFMA tiles, A operand from LDS, B operand from the wavefront's register fragment
//   S = R + D'W_D ; Ku = S \ D'W_A'       mu x mu system in registers (gain_in_registers), pivoted LU in LDS as the fallback      lqr.jl:152-164
//   W_A' -= W_D Ku (= Pk Abar) in LDS ; Abar = A' - D Ku as a register fragment                       lqr.jl:169
//   Pkp1 = Q + Ku'RKu + Abar' (Pk Abar)   tiles again, written over Pk in LDS while |Pk - Pkp1|^2 is summed   lqr.jl:170-176
// No flags, no launches per step, no global memory operand inside a step but Q and the gain that leaves: this is the shape for many
// small problems (config 4 with a setpoint per instance).
// 16x16 tile of sum_k a(k) b(k) by one wavefront, eight k-groups (16 operand loads) per pass.  Measured alternatives that were
// slower under the 256-register budget of a 512-thread workgroup: all of a tile's operands in one pass, strips of tiles sharing the
// global operand's fragments, two register buffers with the next pass prefetched (spills in every case).
template <class FA, class FB>
__device__ inline v4d wave_tile16(int K, FA la, FB lb) {
    const int lk = (threadIdx.x & 63) >> 4;
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    const int ngroups = (K + 3) >> 2;
    constexpr int CH = 8;
    for (int g0 = 0; g0 < ngroups; g0 += CH) {
        double av[CH], bv[CH];
#pragma unroll
        for (int u = 0; u < CH; u++) {
            const int k = 4 * (g0 + u) + lk;
            const bool kok = k < K;
            av[u] = kok ? la(k) : 0.0;
            bv[u] = kok ? lb(k) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < CH; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
    }
    return acc;
}

// Ku = (R + D'PkD) \ (D'Pk A') for ONE column j, entirely in registers (lqr.jl:152-164 in the projected form).  S = R + D'PkD is
// symmetric positive definite for every valid LQR (R > 0, Pk >= 0), so it is factorised WITHOUT pivoting: every thread reads S (MU^2
// broadcast LDS reads: all lanes read the same addresses), runs the LU itself with compile-time register indices, and the threads that
// own a column of D'Pk A' substitute.  No barrier and no LDS round trip between the pivot steps -- the pivoted LDS version below
// (riccati_resident_kernel's general path) is one wavefront stepping through three fenced LDS passes per pivot with a 512-thread
// barrier on either side, 28 % of a backward step at mu = 7.  Returns false, having written nothing, when a pivot is not positive
// (S not positive definite: the caller falls back to the pivoted LU, which also detects a singular S); the decision is a function of
// S alone, hence uniform over the workgroup.  Also writes R Ku for the Ku'RKu term of lqr.jl:170.
template <int MU>
__device__ __forceinline__ bool gain_in_registers(int j, bool owner, int mx, int na, const lds_double* S, const lds_double* Rl, const lds_double* TS,
                                                  lds_double* Ku, lds_double* KRK, double* Kdst) {
    double Sr[MU][MU], x[MU];
    bool ok = true;
#pragma unroll
    for (int r = 0; r < MU; r++)
#pragma unroll
        for (int cc = 0; cc < MU; cc++) Sr[r][cc] = S[r * MU + cc];
#pragma unroll
    for (int c = 0; c < MU; c++) {
        ok = ok && (Sr[c][c] > 0.0);
#ifdef RIC_IEEE_DIV
        const double pinv = 1.0 / Sr[c][c];
#else
        // the pivot's reciprocal (v_rcp_f64 + two Newton steps, ~1 ulp: cclqr_dev.h fast_rcp) is kept in the pivot's place and MULTIPLIES in the back
        // substitution: 14 IEEE division sequences (~ 100 dependent cycles each) sat on the critical path of every backward step between two barriers
        const double pinv = fast_rcp(Sr[c][c]);
        Sr[c][c] = pinv;
#endif
#pragma unroll
        for (int r = c + 1; r < MU; r++) Sr[r][c] *= pinv;
#pragma unroll
        for (int r = c + 1; r < MU; r++)
#pragma unroll
            for (int cc = c + 1; cc < MU; cc++) Sr[r][cc] -= Sr[r][c] * Sr[c][cc];
    }
    if (!ok || !owner) return ok;
#pragma unroll
    for (int r = 0; r < MU; r++) x[r] = TS[r * na + j];
#pragma unroll
    for (int i = 1; i < MU; i++)
#pragma unroll
        for (int r = 0; r < i; r++) x[i] -= Sr[i][r] * x[r];
#pragma unroll
    for (int i = MU - 1; i >= 0; i--) {
#pragma unroll
        for (int r = i + 1; r < MU; r++) x[i] -= Sr[i][r] * x[r];
#ifdef RIC_IEEE_DIV
        x[i] = x[i] / Sr[i][i];
#else
        x[i] = x[i] * Sr[i][i];
#endif
    }
#pragma unroll
    for (int q = 0; q < MU; q++) { Ku[q * mx + j] = x[q]; Kdst[(size_t)q * mx + j] = x[q]; }   // Ku[k][i] = Kk[i:i,:]  lqr.jl:162-164
#pragma unroll
    for (int q = 0; q < MU; q++) {
        double sacc = 0.0;
#pragma unroll
        for (int r = 0; r < MU; r++) sacc += Rl[q * MU + r] * x[r];
        KRK[q * mx + j] = sacc;
    }
    return true;
}

// The same tile with the operands addressed by pointer and stride (no per-load lambdas and predicates) and DOUBLE-BUFFERED: the loads of
// the next four k-groups are in flight while the MFMAs of the current four issue, so a wavefront no longer stops for a full LDS / L2
// round trip in front of every batch of MFMAs (with two wavefronts per SIMD the partner covers part of it; the single-buffered form ran at
// a third of the matrix core's rate).  pa / pb point at the lane's element of k-group 0 (row lk of the group, the lane's column clamped
// into range by the caller: rows / columns past the edge compute garbage that the caller does not store); ga / gb = elements between
// consecutive k-groups.  K must be a multiple of 4 (mx = 12 nb always is; other sizes take wave_tile16).
template <class PA, class PB>
__device__ inline v4d wave_tile16_db(int ngroups, PA pa, int ga, PB pb, int gb) {
    constexpr int CH = 4;
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    double a0[CH], b0[CH], a1[CH], b1[CH];
#pragma unroll
    for (int u = 0; u < CH; u++) { const bool ok = u < ngroups; a0[u] = ok ? pa[(size_t)u * ga] : 0.0; b0[u] = ok ? pb[(size_t)u * gb] : 0.0; }
    for (int g0 = 0; g0 < ngroups; g0 += 2 * CH) {
#pragma unroll
        for (int u = 0; u < CH; u++) { const int g = g0 + CH + u; const bool ok = g < ngroups; a1[u] = ok ? pa[(size_t)g * ga] : 0.0; b1[u] = ok ? pb[(size_t)g * gb] : 0.0; }
#pragma unroll
        for (int u = 0; u < CH; u++) if (g0 + u < ngroups) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b0[u], acc, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < CH; u++) { const int g = g0 + 2 * CH + u; const bool ok = g < ngroups; a0[u] = ok ? pa[(size_t)g * ga] : 0.0; b0[u] = ok ? pb[(size_t)g * gb] : 0.0; }
#pragma unroll
        for (int u = 0; u < CH; u++) if (g0 + CH + u < ngroups) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b1[u], acc, 0, 0, 0);
    }
    return acc;
}

// Tile ownership of the resident kernel.  Every wavefront works on ONE 16-column block `col` of [A' | D] for the whole sweep, so that the
// block's MFMA fragment (A'|D)[k][col*16 + lane] -- the B operand of W = Pk [A'|D] -- stays in its registers across all backward steps of a
// time-invariant problem, and Abar[k][col*16 + lane] = A' - D Ku, the A operand of Abar'(Pk Abar), is formed from it in registers: no MFMA
// operand comes from global memory any more (the L2 round trips of the streamed A'|D and of the Abar scratch kept the matrix cores at a
// third of their rate).  The C <= 8 column blocks are dealt to the 8 wavefronts; blocks that get several wavefronts split the T row tiles
// (of W; column tiles of Pkp1) among them.  For the Sawyer shape (C = 6 blocks): wavefronts 0-3 take blocks 0-3 whole, 4/5 halves of block
// 4, 6/7 halves of block 5 -- wavefronts w and w + 4 share a SIMD, which then has 9 tiles whichever pair it hosts.
__device__ inline void ric_wave_tiles(int wave, int C, int T, int* col, int* lo, int* hi) {
    *col = -1; *lo = 0; *hi = 0;
    if (C > RIC_WAVES) return;                           // (never: the resident path is chosen for na <= 128 only)
    const int base = RIC_WAVES / C, extra = RIC_WAVES % C;   // the LAST `extra` blocks get one wavefront more
    int w = 0;
    for (int c = 0; c < C; c++) {
        const int n = base + (c >= C - extra ? 1 : 0);
        if (wave >= w && wave < w + n) { const int p = wave - w; *col = c; *lo = (T * p) / n; *hi = (T * (p + 1)) / n; return; }
        w += n;
    }
}

// the (inputs, states) shapes that have a register-fragment instantiation of riccati_resident_kernel (launch_riccati picks it): the BASELINE mechanisms
static bool ric_resident_is_frag(int mx, int mu) { return (mu == 1 && (mx == 12 || mx == 24 || mx == 48)) || (mu == 7 && mx == 84); }
#define RIC_YB 128      // doubles of a wavefront's private transposition buffer for Y = Abar' W_D (16 rows x 8 inputs), register-fragment form only
size_t ric_resident_lds_bytes(int mx, int mu) {
    const size_t na = (size_t)mx + mu;
    return ((size_t)mx * mx + mx * na + (size_t)mx * mu + 2 * (size_t)mu * mx + mu * na + 2 * (size_t)mu * mu + 2 * RIC_WAVES + 2 +
            (ric_resident_is_frag(mx, mu) ? (size_t)RIC_WAVES * RIC_YB : 0)) * sizeof(double) + (mu + 2) * sizeof(int);
}

// MUT: the number of inputs as a compile-time constant (1 .. RIC_MU_REG: the mu x mu system is solved in registers, gain_in_registers), or 0:
// any mu, pivoted LU in LDS
// NGT: mx / 4 as a compile-time constant (the [A'|D] / Abar MFMA fragments live in registers, ric_wave_tiles), or 0: any mx, the operands
// stream from L2 through the double-buffered tiles
// BF: 0 = fp64 MFMA (parity mode); 1..3 = the measured-error mode of BASELINE configs[3] on THIS kernel ("dense Riccati on MFMA bf16 -> fp32
// accumulate"): the two mx^3 products run on v_mfma_f32_16x16x16_bf16 with fp32 accumulation, every fp64 operand split into BF bf16 terms
// (products t_i t_j with i + j < BF).  The register fragment is split once per step, the LDS operand on the fly per tile; Pk, W, the gain
// solve and the rank-mu update stay fp64.  Only with NGT > 0.
template <int BF, int NGF>
struct RicBf16Frag {
    static constexpr int NKB = (NGF + 3) / 4;        // k-blocks of 16: lane (li, lk) holds k = 16 kb + 4 t + lk, t = 0..3 (same map for both operands)
    v4s t[NKB][BF > 0 ? BF : 1];
    __device__ __forceinline__ void split(const double* frag) {
#pragma unroll
        for (int kb = 0; kb < NKB; kb++)
#pragma unroll
            for (int u = 0; u < 4; u++) {
                short ts[BF > 0 ? BF : 1];
                bf16_split<(BF > 0 ? BF : 1)>((4 * kb + u) < NGF ? frag[(4 * kb + u) < NGF ? 4 * kb + u : 0] : 0.0, ts);
#pragma unroll
                for (int s_ = 0; s_ < (BF > 0 ? BF : 1); s_++) t[kb][s_][u] = ts[s_];
            }
    }
};
// one 16x16 tile with the register fragment `F` as one operand and an LDS column (element of k-group g at p[g * gs]) as the other;
// REG_IS_A: the fragment is the A operand (Pkp1 tiles) or the B operand (W tiles)
template <int BF, int NGF, bool REG_IS_A>
__device__ __forceinline__ v4d ric_tile_bf16(const RicBf16Frag<BF, NGF>& F, const lds_double* p, int gs) {
    constexpr int NS = BF > 0 ? BF : 1;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < RicBf16Frag<BF, NGF>::NKB; kb++) {
        v4s o[NS];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            short ts[NS];
            bf16_split<NS>((4 * kb + u) < NGF ? p[(4 * kb + u) * gs] : 0.0, ts);
#pragma unroll
            for (int s_ = 0; s_ < NS; s_++) o[s_][u] = ts[s_];
        }
#pragma unroll
        for (int sa = NS - 1; sa >= 0; sa--)            // smallest products first
#pragma unroll
            for (int sb = NS - 1; sb >= 0; sb--)
                if (sa + sb < NS)
                    acc = REG_IS_A ? __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(F.t[kb][sa], o[sb], acc, 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(o[sa], F.t[kb][sb], acc, 0, 0, 0);
    }
    v4d r = {(double)acc[0], (double)acc[1], (double)acc[2], (double)acc[3]};
    return r;
}

template <int MUT, int NGT, int BF = 0>
__global__ __launch_bounds__(RIC_THREADS) void riccati_resident_kernel(RicGrid a) {
    extern __shared__ double rl[];
#ifdef CCLQR_PROFILE
    unsigned long long rt0 = __builtin_readcyclecounter();
#endif
    __shared__ int sing;
    const int prob = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int mx = NGT > 0 ? 4 * NGT : a.mx, mu = MUT > 0 ? MUT : a.mu, na = mx + mu, N = a.N;
    if (a.stop[prob]) return;            // G Bλ was singular in the projection (status already set)
    lds_double* P = (lds_double*)rl;
    lds_double* W = P + (size_t)mx * mx;
    lds_double* Dl = W + (size_t)mx * na;
    lds_double* Ku = Dl + (size_t)mx * mu;
    lds_double* KRK = Ku + (size_t)mu * mx;
    lds_double* TS = KRK + (size_t)mu * mx;
    lds_double* S = TS + (size_t)mu * na;
    lds_double* Rl = S + (size_t)mu * mu;
    lds_double* red = Rl + (size_t)mu * mu;
    lds_double* Yb = red + 2 * RIC_WAVES + 2 + (size_t)wave * RIC_YB;      // (register-fragment form) this wavefront's transposition buffer
    int* piv = (int*)(rl + ((size_t)mx * mx + (size_t)mx * na + (size_t)mx * mu + 2 * (size_t)mu * mx + (size_t)mu * na + 2 * (size_t)mu * mu + 2 * RIC_WAVES + 2 +
                            (NGT > 0 ? (size_t)RIC_WAVES * RIC_YB : 0)));
    static_assert(NGT == 0 || (MUT >= 1 && MUT <= 8), "the register-fragment form keeps Y = Abar' W_D of a row strip as 16 x 8 doubles");
    double* Kout = a.K + (size_t)prob * ((size_t)(a.keep_last ? 1 : (N > 1 ? N - 1 : 0)) * mu * mx + a.kpad);
    if (tid == 0) sing = 0;
    for (int e = tid; e < mx * mx; e += RIC_THREADS) P[e] = a.Q[e];       // Pk = Q                                  lqr.jl:147
    for (int e = tid; e < mu * mu; e += RIC_THREADS) Rl[e] = a.R[e];
    __syncthreads();
    const int t16m = (mx + 15) >> 4, t16n = (na + 15) >> 4;
    int col, tlo, thi;                   // this wavefront's column block of [A'|D] and its share of the row tiles (ric_wave_tiles)
    ric_wave_tiles(wave, t16n, t16m, &col, &tlo, &thi);
    const int cj = col >= 0 ? col * 16 + li : 0;
    const bool cj_ok = col >= 0 && cj < na;               // the lane's column of [A'|D] exists
    const int cjc = cj_ok ? cj : na - 1;                  // clamped: lanes past the edge compute garbage that is never stored
    const bool pp_col = col >= 0 && col * 16 < mx;        // the block holds columns of A' (not only of D): it owns a row strip of Pkp1
    const int cic = (cj < mx) ? cj : mx - 1;
    // (A'|D)[4 g + lk][cj]: B operand of the W tiles; after the gain it becomes Abar[4 g + lk][cj], the A operand of the Pkp1 tiles.  Fetched at
    // the top of every step (one batch of mx/4 loads per lane, L2-resident; keeping a second copy for time-invariant models costs 48
    // registers the 256-register budget of a 512-thread workgroup does not have)
    constexpr int NGF = NGT > 0 ? NGT : 1;
    double frag[NGF];
    RicBf16Frag<BF, NGF> fsplit;         // the fragment's bf16 terms (BF > 0 only)
    double* Abar = a.Abar + (size_t)prob * mx * mx;      // scratch of the streaming form (NGT = 0)
    int k = 0, status = 0;
    // the step's operands from L2 -- D into LDS, the wavefront's [A'|D] fragment into registers -- are fetched at the END of the step before (as soon
    // as the Pkp1 tiles have consumed the Abar the fragment registers held), so that their round trip runs under the norm reduction and the barriers
    auto fetch_operands = [&](int kk) {
        const double* ADk = a.AD + ((size_t)prob * a.nlin + (a.nlin > 1 ? kk - 1 : 0)) * mx * na;
        for (int e = tid; e < mx * mu; e += RIC_THREADS) Dl[e] = ADk[(size_t)(e / mu) * na + mx + e % mu];
        if (NGT > 0) {
#pragma unroll
            for (int g = 0; g < NGF; g++) frag[g] = col >= 0 ? ADk[(size_t)(4 * g + lk) * na + cjc] : 0.0;
        }
    };
    // The END-of-step fetch may only touch what is the wavefront's own: the fp64 register-fragment form has no barrier between the update phase
    // (which reads D's LDS copy) and the Pkp1 tiles, so a fast wavefront must not overwrite Dl while a slow one still reads the step's D.  The
    // fragment goes to registers at once; the next D -- it only changes for time-varying problems (nlin > 1: TrackingLQR) -- waits in registers
    // (ND per thread) and is stored behind the norm's barrier.
    constexpr bool FRAG64 = NGT > 0 && BF == 0;
    constexpr int ND = FRAG64 ? (4 * NGT * (MUT > 0 ? MUT : 1) + RIC_THREADS - 1) / RIC_THREADS : 1;
    double dnext[ND];
    auto fetch_next = [&](int kk) {
        const double* ADk = a.AD + ((size_t)prob * a.nlin + (a.nlin > 1 ? kk - 1 : 0)) * mx * na;
        if (a.nlin > 1) {
#pragma unroll
            for (int u = 0; u < ND; u++) { const int e = tid + u * RIC_THREADS; dnext[u] = e < mx * mu ? ADk[(size_t)(e / mu) * na + mx + e % mu] : 0.0; }
        }
#pragma unroll
        for (int g = 0; g < NGF; g++) frag[g] = col >= 0 ? ADk[(size_t)(4 * g + lk) * na + cjc] : 0.0;
    };
    if (N - 1 >= 1 && BF == 0) fetch_operands(N - 1);
    for (k = N - 1; k >= 1; k--) {                                        // for outer k=N-1:-1:1                    lqr.jl:150
        const double* AD = a.AD + ((size_t)prob * a.nlin + (a.nlin > 1 ? k - 1 : 0)) * mx * na;
        if (BF > 0) {                    // (the bf16 measured-error modes fetch at the top of the step: no registers to carry the fragment across the norm)
            fetch_operands(k);
            if (NGT > 0) fsplit.split(frag);
        }
        if (NGT > 0) {
            // W = Pk [A' | D]   (Pk symmetric): the wavefront's column block, its share of the row tiles; A operand from LDS, B from registers
            const unsigned long long rw0 = RNOW(); (void)rw0;
            if (col >= 0) {
                for (int rt = tlo; rt < thi; rt++) {
                    const int i0 = rt << 4;
                    const int ic = (i0 + li < mx) ? i0 + li : mx - 1;
                    const lds_double* pa = P + lk * mx + ic;
                    v4d acc = {0.0, 0.0, 0.0, 0.0};
                    if (BF > 0) acc = ric_tile_bf16<BF, NGF, false>(fsplit, pa, 4 * mx);
                    else {
#pragma unroll
                        for (int g = 0; g < NGF; g++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[g * 4 * mx], frag[g], acc, 0, 0, 0);
                    }
                    if (cj_ok) {       // accumulator rows of a lane: lk + 4 r (v_mfma_f64_16x16x4) / 4 lk + r (v_mfma_f32_16x16x16_bf16)
#pragma unroll
                        for (int r = 0; r < 4; r++) { const int row = i0 + (BF > 0 ? 4 * lk + r : lk + 4 * r); if (row < mx) W[row * na + cj] = acc[r]; }
                    }
                }
            }
            RWAVE(RP_WAVE_W, rw0);
        } else {
            // W = Pk [A' | D]   (Pk symmetric)
            for (int tile = wave; tile < t16m * t16n; tile += RIC_WAVES) {
                const int i0 = (tile / t16n) << 4, j0 = (tile % t16n) << 4;
                const bool iok = i0 + li < mx, jok = j0 + li < na;
                const int ic = iok ? i0 + li : mx - 1, jc = jok ? j0 + li : na - 1;
                const v4d acc = wave_tile16_db(mx >> 2, P + lk * mx + ic, 4 * mx, AD + (size_t)lk * na + jc, 4 * na);
                if (jok) {
#pragma unroll
                    for (int r = 0; r < 4; r++) { const int row = i0 + lk + 4 * r; if (row < mx) W[row * na + j0 + li] = acc[r]; }
                }
            }
        }
        __syncthreads();
        RSTAMP(RP_PA);
        // TS = D' W = [D' Pk A' | D' Pk D] ; S = R + D' Pk D
        if (mu <= 16) {            // on the matrix core: D' (mu rows of a 16-row tile) against the column tiles of W
            for (int tile = wave; tile < t16n; tile += RIC_WAVES) {
                const int j0 = tile << 4;
                const bool iok = li < mu, jok = j0 + li < na;
                const int ic = iok ? li : mu - 1, jc = jok ? j0 + li : na - 1;
                const v4d acc = wave_tile16_db(mx >> 2, Dl + lk * mu + ic, 4 * mu, W + lk * na + jc, 4 * na);
                if (jok) {
                    const int j = j0 + li;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int row = lk + 4 * r;
                        if (row < mu) {
                            TS[row * na + j] = acc[r];
#ifndef RIC_S_SEPARATE
                            if (j >= mx) S[row * mu + (j - mx)] = Rl[row * mu + (j - mx)] + acc[r];      // S = R + D'PkD (lqr.jl:152-153) leaves with the tile that holds D'W_D: no pass, no barrier of its own
#endif
                        }
                    }
                }
            }
        } else {
            for (int e = tid; e < mu * na; e += RIC_THREADS) {
                const int q = e / na, j = e - q * na;
                double sacc = 0.0;
                for (int i = 0; i < mx; i++) sacc += Dl[i * mu + q] * W[i * na + j];
                TS[e] = sacc;
            }
        }
        __syncthreads();
#ifndef RIC_S_SEPARATE
        if (mu > 16)
#endif
        {
            for (int e = tid; e < mu * mu; e += RIC_THREADS) S[e] = Rl[e] + TS[(e / mu) * na + mx + e % mu];       // S = R + D'PkD   lqr.jl:152-153
            __syncthreads();
        }
        bool in_regs = false;
        if (MUT > 0) {                 // mu x mu system in registers (mu <= 7: every BASELINE config), see gain_in_registers
            double* Kdst = Kout + (a.keep_last ? 0 : (size_t)(k - 1) * mu * mx);
            in_regs = gain_in_registers<(MUT > 0 ? MUT : 1)>(tid, tid < mx, mx, na, S, Rl, TS, Ku, KRK, Kdst);
        }
        if (in_regs) {
            __syncthreads();
        } else {
            if (wave == 0) {                                     // LU with partial pivoting (mu <= 32) by ONE wavefront: LDS is in order per
                for (int c = 0; c < mu; c++) {                   // wavefront, so fences replace the workgroup barriers
                    int bi = c;
                    {
                        double best = -1.0;
                        for (int r = c; r < mu; r++) { double v = fabs(S[r * mu + c]); if (v > best) { best = v; bi = r; } }   // every lane: same scan
                        if (!(best > 0.0)) { if (lane == 0) sing = 1; break; }
                    }
                    if (lane == 0) piv[c] = bi;
                    if (bi != c && lane < mu) { double t = S[c * mu + lane]; S[c * mu + lane] = S[bi * mu + lane]; S[bi * mu + lane] = t; }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    const double pinv = 1.0 / S[c * mu + c];
                    if (lane > c && lane < mu) S[lane * mu + c] *= pinv;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    const int w = mu - c - 1;
                    for (int e = lane; e < w * w; e += 64) {
                        const int r = c + 1 + e / w, j = c + 1 + e % w;
                        S[r * mu + j] -= S[r * mu + c] * S[c * mu + j];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
            }
            __syncthreads();
            if (sing) { status = CCLQR_ESINGULAR_; break; }
            for (int j = tid; j < mx; j += RIC_THREADS) {         // Ku = S \ (D' Pk A'), one column per thread
                for (int c = 0; c < mu; c++) { const int p = piv[c]; if (p != c) { double t = TS[c * na + j]; TS[c * na + j] = TS[p * na + j]; TS[p * na + j] = t; } }
                for (int i = 1; i < mu; i++) { double sacc = TS[i * na + j]; for (int r = 0; r < i; r++) sacc -= S[i * mu + r] * TS[r * na + j]; TS[i * na + j] = sacc; }
                for (int i = mu - 1; i >= 0; i--) {
                    double sacc = TS[i * na + j];
                    for (int r = i + 1; r < mu; r++) sacc -= S[i * mu + r] * TS[r * na + j];
                    TS[i * na + j] = sacc / S[i * mu + i];
                }
                for (int q = 0; q < mu; q++) { const double v = TS[q * na + j]; Ku[q * mx + j] = v; Kout[(a.keep_last ? 0 : (size_t)(k - 1) * mu * mx) + (size_t)q * mx + j] = v; }   // lqr.jl:162-164 (keep_last: one slot, the last step's gain stays)
                for (int q = 0; q < mu; q++) { double sacc = 0.0; for (int r = 0; r < mu; r++) sacc += Rl[q * mu + r] * TS[r * na + j]; KRK[q * mx + j] = sacc; }
            }
            __syncthreads();
        }
        RSTAMP(RP_GAIN);
        if (NGT > 0) {
            // Pk Abar = W_A' - W_D Kuk (in LDS) ; Abar = A' - D Kuk (= A-Bu*Kuk-Bλ*Kλk, lqr.jl:169): the wavefront's column strip of it, as the
            // MFMA fragment Abar[4 g + lk][cj], from the A' fragment it already holds (never written to memory)
            // W_A' -= W_D Ku on the matrix core: a rank-mu update, ceil(mu / 4) MFMAs per 16x16 tile with the tile itself as the accumulator.
            // fp64 form (BF = 0): NOT done -- the Pkp1 tiles below take Abar'(Pk Abar) = Abar' W_A' - (Abar' W_D) Ku with Y = Abar' W_D of the wavefront's
            // own row strip (21 MFMAs, no other wavefront's data), which removes this read-modify-write of all of W_A' and the barrier behind it
            if (BF > 0)
            for (int tile = wave; tile < t16m * t16m; tile += RIC_WAVES) {
                const int i0 = (tile / t16m) << 4, j0 = (tile % t16m) << 4;
                const bool iok = i0 + li < mx, jok = j0 + li < mx;
                const int ic = iok ? i0 + li : mx - 1, jc = jok ? j0 + li : mx - 1;
                v4d acc;
#pragma unroll
                for (int r = 0; r < 4; r++) { const int row = i0 + lk + 4 * r; acc[r] = W[(row < mx ? row : mx - 1) * na + jc]; }
                for (int q0 = 0; q0 < mu; q0 += 4) {
                    const int q = q0 + lk;
                    const double av = q < mu ? -W[ic * na + mx + q] : 0.0, bv = q < mu ? Ku[q * mx + jc] : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
                if (jok) {
#pragma unroll
                    for (int r = 0; r < 4; r++) { const int row = i0 + lk + 4 * r; if (row < mx) W[row * na + j0 + li] = acc[r]; }
                }
            }
            if (pp_col) {
                // Abar fragment = A' fragment - D Ku, also a rank-mu update on the matrix core: the fragment's element (k = 4 g + lk, column cj)
                // IS accumulator register r = g - 4 T of the 16-row tile T = g / 4 in v_mfma_f64_16x16x4's output layout (row lk + 4 r)
                constexpr int NT = (NGF + 3) / 4;
#pragma unroll
                for (int T = 0; T < NT; T++) {
                    v4d acc;
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[r] = (4 * T + r) < NGF ? frag[(4 * T + r) < NGF ? 4 * T + r : 0] : 0.0;
                    // (row 16 T + li of D, unclamped: the rows past mx of the last tile read on into Ku's region of the image and feed accumulator rows
                    // that belong to no fragment element -- one base address + immediate offsets instead of a clamped address per tile)
                    const int dr = 16 * T + li;
                    for (int q0 = 0; q0 < mu; q0 += 4) {
                        const int q = q0 + lk;
                        const double av = q < mu ? -Dl[dr * mu + q] : 0.0, bv = q < mu ? Ku[q * mx + cic] : 0.0;
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; r++) if ((4 * T + r) < NGF) frag[4 * T + r] = acc[r];
                }
                if (BF > 0) fsplit.split(frag);
                if (BF == 0) {       // Y = Abar' W_D of this row strip, through the wavefront's own buffer into A-operand layout: Yb[i][q], i < 16, q < 8
                    v4d yacc = {0.0, 0.0, 0.0, 0.0};
                    const lds_double* pwd = W + lk * na + mx + (li < mu ? li : mu - 1);
#pragma unroll
                    for (int g = 0; g < NGF; g++) yacc = __builtin_amdgcn_mfma_f64_16x16x4f64(frag[g], pwd[g * 4 * na], yacc, 0, 0, 0);
                    if (li < 8) {
#pragma unroll
                        for (int r = 0; r < 4; r++) Yb[(lk + 4 * r) * 8 + li] = yacc[r];
                    }
                }
            }
        } else {
            // Pk Abar = W_A' - W_D Kuk (in LDS) ; Abar = A' - D Kuk (= A-Bu*Kuk-Bλ*Kλk, lqr.jl:169) -> global scratch
            for (int i = tid >> 5; i < mx; i += RIC_THREADS / 32) {
                for (int j0 = tid & 31; j0 < mx; j0 += 128) {          // four columns of a row per pass: their A' loads are in flight together
                    double ab[4], pw[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int j = j0 + 32 * u;
                        ab[u] = j < mx ? AD[(size_t)i * na + j] : 0.0;
                        pw[u] = j < mx ? W[i * na + j] : 0.0;
                    }
                    for (int q = 0; q < mu; q++) {
                        const double dq = Dl[i * mu + q], wq = W[i * na + mx + q];
#pragma unroll
                        for (int u = 0; u < 4; u++) { const int j = j0 + 32 * u; const double kq = j < mx ? Ku[q * mx + j] : 0.0; ab[u] -= dq * kq; pw[u] -= wq * kq; }
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) { const int j = j0 + 32 * u; if (j < mx) { Abar[(size_t)i * mx + j] = ab[u]; W[i * na + j] = pw[u]; } }
                }
            }
        }
        if (!(NGT > 0 && BF == 0)) __syncthreads();      // (fp64 register-fragment form: everything since the gain has been the wavefront's own)
        RSTAMP(RP_UPD);
        // Pkp1 = Q + Kuk'*R*Kuk + Abar'*(Pk*Abar), over Pk; |Pk - Pkp1|^2 on the way                                lqr.jl:170-176
        double nacc = 0.0;
        const unsigned long long rp0 = RNOW(); (void)rp0;
        if (NGT > 0) {
            if (pp_col) {        // tiles (row strip i0 = col * 16, column tile ct) of the SYMMETRIC Pkp1 that ric_pp_assign dealt to this wavefront, each
                                 // mirrored into (ct, col): A operand from registers (Abar fragment), B operand Pk Abar from LDS
                const int i0 = col << 4;
                // (the bf16 measured-error modes keep the whole strip, unmirrored: they sit at exactly 256 registers)
                const unsigned ppm = BF > 0 ? (((1u << thi) - 1u) & ~((1u << tlo) - 1u)) : a.pp_mask[wave];
#pragma unroll 1
                for (int ct = 0; ct < t16m; ct++) {      // (not unrolled: six tiles' worth of hoisted LDS addresses cost the kernel its 256-register budget)
                    if (!((ppm >> ct) & 1u)) continue;
                    const int j0 = ct << 4;
                    const bool jok = j0 + li < mx, offdiag = BF == 0 && ct != col;
                    const int jc = jok ? j0 + li : mx - 1;
                    const lds_double* pb = W + lk * na + jc;
                    double qv[4];                     // Q of the tile's outputs, fetched BEFORE the MFMAs: its L2 round trip hides under them
                    if (BF == 0) {
#pragma unroll
                        for (int r = 0; r < 4; r++) { const int i = i0 + lk + 4 * r; qv[r] = a.Q[(size_t)(i < mx ? i : mx - 1) * mx + jc]; }
                    }
                    v4d acc = {0.0, 0.0, 0.0, 0.0};
                    if (BF > 0) {                     // (measured-error mode: Q is fetched behind the tile -- the split fragments leave no registers to hold it)
                        acc = ric_tile_bf16<BF, NGF, true>(fsplit, pb, 4 * na);
#pragma unroll
                        for (int r = 0; r < 4; r++) { const int i = i0 + 4 * lk + r; qv[r] = a.Q[(size_t)(i < mx ? i : mx - 1) * mx + jc]; }
                    } else {
                        // Q + Kuk'*(R*Kuk) + Abar'*(Pk*Abar) in ONE accumulation chain: Q is the initial accumulator, the rank-mu term
                        // ceil(mu / 4) more k-groups (A = Ku[q][i], B = (R Ku)[q][j])
#pragma unroll
                        for (int r = 0; r < 4; r++) acc[r] = qv[r];
                        const int ia = (i0 + li < mx) ? i0 + li : mx - 1;
                        for (int q0 = 0; q0 < mu; q0 += 4) {
                            const int q = q0 + lk;
                            const double av = q < mu ? Ku[q * mx + ia] : 0.0, bv = q < mu ? KRK[q * mx + jc] : 0.0;
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                        }
                        for (int q0 = 0; q0 < mu; q0 += 4) {      // - (Abar' W_D) Ku: the rank-mu part of Abar'(Pk Abar), W_A' itself is the B operand below
                            const int q = q0 + lk;
                            const double av = q < mu ? -Yb[li * 8 + q] : 0.0, bv = q < mu ? Ku[q * mx + jc] : 0.0;
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                        }
#pragma unroll
                        for (int g = 0; g < NGF; g++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(frag[g], pb[g * 4 * na], acc, 0, 0, 0);
                    }
                    if (jok) {
                        const int j = j0 + li;
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int i = i0 + (BF > 0 ? 4 * lk + r : lk + 4 * r);
                            if (i < mx) {
                                double v = acc[r];
                                if (BF > 0) { v += qv[r]; for (int q = 0; q < mu; q++) v += Ku[q * mx + i] * KRK[q * mx + j]; }
                                const double d = P[i * mx + j] - v;
                                nacc += d * d;
                                P[i * mx + j] = v;     // Pk = Pkp1 (lqr.jl:176); after a break nothing reads Pk again
                                if (offdiag) { const double dm = P[j * mx + i] - v; nacc += dm * dm; P[j * mx + i] = v; }      // the mirrored entry
                            }
                        }
                    }
                }
            }
        } else {
            for (int tile = wave; tile < t16m * t16m; tile += RIC_WAVES) {
                const int i0 = (tile / t16m) << 4, j0 = (tile % t16m) << 4;
                const bool iok = i0 + li < mx, jok = j0 + li < mx;
                const int ic = iok ? i0 + li : mx - 1, jc = jok ? j0 + li : mx - 1;
                const v4d acc = wave_tile16_db(mx >> 2, Abar + (size_t)lk * mx + ic, 4 * mx, W + lk * na + jc, 4 * na);
                if (jok) {
                    const int j = j0 + li;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int i = i0 + lk + 4 * r;
                        if (i < mx) {
                            double v = a.Q[(size_t)i * mx + j] + acc[r];
                            for (int q = 0; q < mu; q++) v += Ku[q * mx + i] * KRK[q * mx + j];
                            const double d = P[i * mx + j] - v;
                            nacc += d * d;
                            P[i * mx + j] = v;     // Pk = Pkp1 (lqr.jl:176); after a break nothing reads Pk again
                        }
                    }
                }
            }
        }
        RWAVE(RP_WAVE_PP, rp0);
        RSTAMP(RP_NORM);       // (diagnostic build: wavefront 0's own Pkp1 tiles end here; RP_PP below = operand prefetch, norm reduction, barrier waits)
        if (k > 1 && BF == 0) {
            if (FRAG64) fetch_next(k - 1);         // (the fragment was last read by this wavefront's tiles above; D waits in registers)
            else fetch_operands(k - 1);            // (streaming form: a barrier separates the update phase, D's last reader, from the tiles)
        }
        for (int o = 32; o > 0; o >>= 1) nacc += __shfl_xor(nacc, o, 64);
        if (lane == 0) red[wave] = nacc;
        __syncthreads();
        if (FRAG64 && k > 1 && a.nlin > 1) {       // every wavefront is past its update phase: the next knot's D may land (visible behind the next barrier)
#pragma unroll
            for (int u = 0; u < ND; u++) { const int e = tid + u * RIC_THREADS; if (e < mx * mu) Dl[e] = dnext[u]; }
        }
        double tot = 0.0;
        for (int q = 0; q < RIC_WAVES; q++) tot += red[q];
        __syncthreads();
        RSTAMP(RP_PP);
#ifdef CCLQR_PROFILE
        if (tid == 0 && blockIdx.x == 0) g_rprof[RP_STEPS] += 1;
#endif
        if (sqrt(tot) < a.tol) break;                                     // if norm(Pk-Pkp1) < 1e-5  break          lqr.jl:172-174
    }
    if (status == 0) {
        if (k < 1 && N - 1 >= 1) k = 1;   // Julia: after a completed loop the outer k holds its last value
        if (N - 1 < 1) k = 0;
        __syncthreads();
        for (int k2 = k - 1; k2 >= 1 && !a.keep_last; k2--) {             // Ku[k2] = Ku[k2+1]                       lqr.jl:179-181
            for (int e = tid; e < mu * mx; e += RIC_THREADS) Kout[(size_t)(k2 - 1) * mu * mx + e] = Kout[(size_t)k2 * mu * mx + e];
            __syncthreads();
        }
    }
    if (tid == 0) { a.kbreak[prob] = k; a.status[prob] = status; }
}

// Ku[k2] = Ku[k2+1] below the break index (lqr.jl:179-181) and the loop variable's final value
__global__ void ric_backfill_kernel(RicGrid a) {
    const int prob = blockIdx.x, tid = threadIdx.x;
    const int mu = a.mu, mx = a.mx, N = a.N;
    __shared__ int kb;
    if (tid == 0) {
        int k = a.stop[prob] ? a.kbreak[prob] : (N - 1 >= 1 ? 1 : 0);
        if (!a.stop[prob] && N - 1 >= 1) {
            // the test after the last step (k = 1) is a break at k = 1 as well: same index, nothing to do
        }
        a.kbreak[prob] = k;
        kb = k;
    }
    __syncthreads();
    if (a.status[prob] != 0 || a.keep_last) return;
    double* Kout = a.K + (size_t)prob * ((size_t)(N > 1 ? N - 1 : 0) * mu * mx + a.kpad);
    for (int k2 = kb - 1; k2 >= 1; k2--) {
        for (int e = tid; e < mu * mx; e += blockDim.x) Kout[(size_t)(k2 - 1) * mu * mx + e] = Kout[(size_t)k2 * mu * mx + e];
        __syncthreads();
    }
}


size_t ric_grid_work_doubles(int nprob, int mx, int mu, int ml, int N, int time_varying) {
    const size_t nlin = time_varying ? (size_t)(N > 1 ? N - 1 : 1) : 1, na = (size_t)mx + mu, tm = (mx + 31) / 32;
    const size_t sc = (((size_t)ml * ml + (size_t)ml * na + 3) & ~(size_t)1);
    return (size_t)nprob * (nlin * mx * na + mx * na + 3 * (size_t)mx * mx + 2 * (size_t)mu * mx + 2 * tm * tm + tm * mu * na + nlin * sc + nlin * (ml + 2) + 8) + 64;
}

// P and W in one CU's LDS, and whole k-groups of four for the double-buffered tiles (mx = 12 nb always is a multiple of 4)
static bool ric_resident_fits(const RicArgs& a) {
    return ric_resident_lds_bytes(a.mx, a.mu) <= 158 * 1024 && (a.mx & 3) == 0 && (a.mx + a.mu + 15) / 16 <= RIC_WAVES;
}
// resident (one workgroup per problem, P and W in LDS) whenever it fits; otherwise the tiled three-launch step
// shapes for which the bf16 measured-error mode exists on the resident kernel (the register-fragment specialisations)
static bool ric_resident_has_bf16(const RicArgs& a) { return (a.mu == 7 && a.mx == 84) || (a.mu == 1 && a.mx == 24); }
static bool ric_use_tiled(const RicArgs& a) {
    if (!ric_resident_fits(a)) return true;
    if (a.bf16_terms > 0 && !ric_resident_has_bf16(a)) return true;   // elsewhere the measured-error mode exists on the tiled path only
    const int path = a.path;      // 0 auto, 1 LDS-resident workgroup per problem, 2 tiled
    if (path != 0) return path == 2;
    // measured crossover: a single 84..96-state problem is faster spread over the device (41 vs 59 us per step), small problems
    // and large batches are faster resident (mx 48: 14 vs 20 us; 1024 x mx 84: 0.54 vs 0.74 ms per step)
    return a.mx >= 64 && a.nprob < 128;
}

size_t ric_total_work_doubles(const RicArgs& a) { return ric_grid_work_doubles(a.nprob, a.mx, a.mu, a.ml, a.N, a.time_varying); }

// Pkp1 = Q + Ku'RKu + Abar'(Pk Abar) is symmetric, so of its T x T tiles of 16 x 16 only the T (T + 1) / 2 with column tile >= row tile are
// computed (and mirrored): 21 instead of 36 for the Sawyer's 84 states.  A tile's A operand is the Abar fragment of the wavefront that owns the
// tile's ROW strip (ric_wave_tiles), so the pair {a, b} can go to an owner of strip a (as tile (a, b)) or of strip b (as (b, a)): dealt greedily so
// that the four SIMDs (wavefronts w and w + 4 share one) carry equal numbers of tiles -- Sawyer: 5 / 5 / 6 / 5 tiles per SIMD instead of 9 each.
static void ric_pp_assign(int T, int C, unsigned char* mask) {
    const int Wv = RIC_WAVES;
    int first[16], cnt[16], wl[RIC_WAVES] = {0};
    for (int w = 0; w < Wv; w++) mask[w] = 0;
    if (C > Wv || C > 16 || T > 8 || T > C) return;
    {
        const int base = Wv / C, extra = Wv % C;       // = ric_wave_tiles: the LAST `extra` column blocks get one wavefront more
        int w = 0;
        for (int c = 0; c < C; c++) { cnt[c] = base + (c >= C - extra ? 1 : 0); first[c] = w; w += cnt[c]; }
    }
    struct Item { int a, b, owners; } items[36];
    int n = 0;
    for (int a = 0; a < T; a++)
        for (int b = a; b < T; b++) items[n++] = {a, b, cnt[a] + (b != a ? cnt[b] : 0)};
    for (int i = 1; i < n; i++)                         // stable insertion sort: the pairs with the fewest possible owners first
        for (int j = i; j > 0 && items[j].owners < items[j - 1].owners; j--) { const Item t = items[j]; items[j] = items[j - 1]; items[j - 1] = t; }
    for (int i = 0; i < n; i++) {
        int bw = -1, bc = 0, bs = 1 << 30, bl = 1 << 30;
        for (int side = 0; side < (items[i].b != items[i].a ? 2 : 1); side++) {
            const int strip = side ? items[i].b : items[i].a, colt = side ? items[i].a : items[i].b;
            for (int w = first[strip]; w < first[strip] + cnt[strip]; w++) {
                int sl = 1;
                for (int x = w % 4; x < Wv; x += 4) sl += wl[x];          // tiles on the wavefront's SIMD if it takes this one
                const int l = wl[w] + 1;
                if (sl < bs || (sl == bs && l < bl)) { bs = sl; bl = l; bw = w; bc = colt; }
            }
        }
        wl[bw]++;
        mask[bw] |= (unsigned char)(1u << bc);
    }
}

hipError_t launch_riccati(const RicArgs& a, hipStream_t stream) {
    if (a.nprob <= 0) return hipSuccess;
    RicGrid g;
    ric_pp_assign((a.mx + 15) / 16, (a.mx + a.mu + 15) / 16, g.pp_mask);
    g.nprob = a.nprob; g.mx = a.mx; g.mu = a.mu; g.ml = a.ml; g.N = a.N; g.nlin = a.time_varying ? (a.N > 1 ? a.N - 1 : 1) : 1;
    g.na = a.mx + a.mu; g.tm = (a.mx + 31) / 32; g.tn = (g.na + 31) / 32; g.tol = a.tol; g.bf16_terms = a.bf16_terms; g.keep_last = a.keep_last; g.kpad = a.kpad;
    g.A = a.A; g.Bu = a.Bu; g.Bl = a.Bl; g.G = a.G; g.Q = a.Q; g.R = a.R; g.K = a.K; g.kbreak = a.kbreak; g.status = a.status; g.stop = a.stop;
    const size_t np = a.nprob, nlin = g.nlin, mx = a.mx, na = g.na, mu = a.mu, ml = a.ml;
    double* o = a.work;
    g.AD = o; o += np * nlin * mx * na;
    g.W = o; o += np * mx * na;
    g.Abar = o; o += np * mx * mx;
    g.P = o; o += 2 * np * mx * mx;
    g.Ku = o; o += np * mu * mx;
    g.KRK = o; o += np * mu * mx;
    g.part = o; o += 2 * np * g.tm * g.tm;
    g.TSp = o; o += np * g.tm * mu * na;
    g.scratch = o;
    hipError_t e = hipMemsetAsync(a.stop, 0, np * sizeof(int), stream);
    if (e == hipSuccess) e = hipMemsetAsync(a.status, 0, np * sizeof(int), stream);
    if (e == hipSuccess) e = hipMemsetAsync(a.kbreak, 0, np * sizeof(int), stream);
    if (e != hipSuccess) return e;
    // [A' | D] of every knot of every problem, in parallel
    size_t lds = 0; int cols = 0;
    if (ml <= RIC_LDS_M && ml > 0) {
        const size_t budget = 150 * 1024;                 // of the 160 KB per CU; the rest is static LDS
        size_t c = (budget - ml * ml * sizeof(double)) / (ml * sizeof(double));
        if (c > RIC_THREADS) c = RIC_THREADS;
        if (c > na) c = na;
        cols = (int)c;
        lds = (ml * ml + ml * c) * sizeof(double);        // G Bλ for the pivoted LU + one batch of right-hand-side columns
    }
    if (lds > 0) {
        e = set_max_dynamic_lds_once((const void*)ric_project_kernel<true>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(ric_project_kernel<true>, dim3(g.nlin, a.nprob), dim3(RIC_THREADS), lds, stream, g, cols);
    } else {
        hipLaunchKernelGGL(ric_project_kernel<false>, dim3(g.nlin, a.nprob), dim3(RIC_THREADS), 0, stream, g, 0);
    }
    if (!ric_use_tiled(a)) {
        const size_t rl = ric_resident_lds_bytes(a.mx, a.mu);
        typedef void (*ResKernel)(RicGrid);
        // specialised (inputs, states / 4) shapes: the BASELINE mechanisms; every other shape takes the generic forms
        static const ResKernel by_mu[RIC_MU_REG + 1] = {riccati_resident_kernel<0, 0>, riccati_resident_kernel<1, 0>, riccati_resident_kernel<2, 0>, riccati_resident_kernel<3, 0>,
                                                        riccati_resident_kernel<4, 0>, riccati_resident_kernel<5, 0>, riccati_resident_kernel<6, 0>, riccati_resident_kernel<7, 0>};
        ResKernel kern = by_mu[(a.mu >= 1 && a.mu <= RIC_MU_REG) ? a.mu : 0];
        const int ng4 = a.mx >> 2;
        if (a.mu == 1 && ng4 == 3) kern = riccati_resident_kernel<1, 3>;          // pendulum (mx 12)
        else if (a.mu == 1 && ng4 == 6) kern = riccati_resident_kernel<1, 6>;     // cartpole, acrobot (mx 24)
        else if (a.mu == 1 && ng4 == 12) kern = riccati_resident_kernel<1, 12>;   // triple cartpole (mx 48)
        else if (a.mu == 7 && ng4 == 21) kern = riccati_resident_kernel<7, 21>;   // Sawyer (mx 84)
        if (a.bf16_terms > 0) {      // ric_use_tiled has checked that the shape is one of these
            if (a.mu == 7) kern = a.bf16_terms == 1 ? riccati_resident_kernel<7, 21, 1> : (a.bf16_terms == 2 ? riccati_resident_kernel<7, 21, 2> : riccati_resident_kernel<7, 21, 3>);
            else kern = a.bf16_terms == 1 ? riccati_resident_kernel<1, 6, 1> : (a.bf16_terms == 2 ? riccati_resident_kernel<1, 6, 2> : riccati_resident_kernel<1, 6, 3>);
        }
        e = set_max_dynamic_lds_once((const void*)kern, rl);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(a.nprob), dim3(RIC_THREADS), rl, stream, g);
        return hipGetLastError();
    }
    const size_t lds_gain = (mu * na + mu * mu + 2 * RU * mu) * sizeof(double) + (mu + 2) * sizeof(int), lds_pn = 2 * mu * 32 * sizeof(double);
    if (lds_gain > 48 * 1024) {
        e = set_max_dynamic_lds_once((const void*)ric_gain_update_kernel, lds_gain);
        if (e != hipSuccess) return e;
    }
    for (int k = a.N - 1; k >= 1; k--) {                                 // for outer k=N-1:-1:1                    lqr.jl:150
        hipLaunchKernelGGL(ric_pa_kernel, dim3(g.tm * g.tn, a.nprob), dim3(TILE_THREADS), 32 * mu * sizeof(double), stream, g, k);
        hipLaunchKernelGGL(ric_gain_update_kernel, dim3((a.mx + RU - 1) / RU, a.nprob), dim3(TILE_THREADS), lds_gain, stream, g, k);
        hipLaunchKernelGGL(ric_pn_kernel, dim3(g.tm * g.tm, a.nprob), dim3(TILE_THREADS), lds_pn, stream, g, k);
    }
    hipLaunchKernelGGL(ric_backfill_kernel, dim3(a.nprob), dim3(TILE_THREADS), 0, stream, g);
    return hipGetLastError();
}

#ifdef CCLQR_PROFILE
extern "C" int cclqr_ric_prof_read(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rprof), sizeof(unsigned long long) * RP_N);
    if (e == hipSuccess && reset) { unsigned long long z[RP_N] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_rprof), z, sizeof(z)); }
    return e == hipSuccess ? RP_N : -1;
}
#endif

}  // namespace cclqr

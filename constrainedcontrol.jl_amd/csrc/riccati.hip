// riccati.hip -- constrained discrete Riccati backward recursion, dlqr(A,Bu,Bλ,G,Q,R,N) of src/control/lqr.jl:141-184 and its
// time-varying twin dlqr(mechanism, ...) of src/control/lqr_tracking.jl:73-122 (A,Bu,Bλ,G indexed by knot).
//
// One workgroup (8 wavefronts) per independent problem, persistent over k = N-1 ... 1: the sweep is sequential in k, so all
// parallelism inside a problem is in the dense algebra of one step.  The mx x mx products (P [A' | D], Abar' (P Abar)) run on the
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
__device__ void wg_lu(int n, MP A, int lda, int* piv, int* sing, double* red_v, int* red_i) {
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
            if (tid == 0) { piv[c] = bi; if (!(best > 0.0)) *sing = 1; red_v[0] = best; }
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
    (void)red_i;
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

// Workspace of one problem.  The recursion runs in its PROJECTED form: with E = (G Bλ)^-1 G Bu and F = (G Bλ)^-1 G A,
//   D  = Bu - Bλ E            (lqr.jl:151, the reference's D)
//   A' = A  - Bλ F            (dynamics projected onto the constraint manifold)
// the second block row of M Kk = b (lqr.jl:154-160) gives Kλ = F - E Ku, and substituting it into the first block row leaves
//   (R + D' P D) Ku = D' P A' ,   Abar = A - Bu Ku - Bλ Kλ = A' - D Ku        (lqr.jl:160,169)
// i.e. the same Kk and Abar as the reference's (mu+ml)-square solve, at the cost of a mu-square one; E, F, D, A' do not depend on P,
// so a time-invariant problem computes them once.  Only Ku is stored by the reference (lqr.jl:162-164), Kλ is never formed.
struct RicWork {
    double *GBl, *X, *AD, *W, *TS, *S, *Ku, *Abar, *P, *Pn, *KRK;
    int* piv;
};
__host__ __device__ inline size_t ric_carve(int mx, int mu, int ml, double* base, RicWork* w) {
    const size_t na = (size_t)mx + mu;
    size_t o = 0;
    auto take = [&](size_t n) { double* p = base ? base + o : nullptr; o += (n + 1) & ~(size_t)1; return p; };
    double *GBl = take((size_t)ml * ml), *X = take((size_t)ml * na), *AD = take((size_t)mx * na), *W = take((size_t)mx * na),
           *TS = take((size_t)mu * na), *S = take((size_t)mu * mu), *Ku = take((size_t)mu * mx), *Abar = take((size_t)mx * mx),
           *P = take((size_t)mx * mx), *Pn = take((size_t)mx * mx), *KRK = take((size_t)mu * mx), *piv = take((size_t)ml + mu + 16);
    if (w) { w->GBl = GBl; w->X = X; w->AD = AD; w->W = W; w->TS = TS; w->S = S; w->Ku = Ku; w->Abar = Abar; w->P = P; w->Pn = Pn;
             w->KRK = KRK; w->piv = (int*)piv; }
    return o;
}
size_t ric_work_doubles(int mx, int mu, int ml) { return ric_carve(mx, mu, ml, nullptr, nullptr); }

#ifdef CCLQR_PROFILE
enum { RP_PRE, RP_PA, RP_GAIN, RP_UPD, RP_PP, RP_NORM, RP_STEPS, RP_N };
static __device__ unsigned long long g_rprof[RP_N];
#define RSTAMP(c) do { if (threadIdx.x == 0 && blockIdx.x == 0) { unsigned long long t1_ = __builtin_readcyclecounter(); g_rprof[c] += t1_ - rt0; rt0 = t1_; } } while (0)
#else
#define RSTAMP(c)
#endif

#define RIC_LDS_M 96   // G Bλ (ml x ml) is kept in LDS for its pivoted LU when ml <= 96 (72 KB)

template <bool LDSM>
__global__ __launch_bounds__(RIC_THREADS) void riccati_kernel(RicArgs a) {
    extern __shared__ double lds_M[];
    typedef typename std::conditional<LDSM, lds_double*, double*>::type MP;
#ifdef CCLQR_PROFILE
    unsigned long long rt0 = __builtin_readcyclecounter();
#endif
    __shared__ double red_v[RIC_WAVES];
    __shared__ int red_i[4];
    __shared__ int sing;
    const int prob = blockIdx.x, tid = threadIdx.x;
    const int mx = a.mx, mu = a.mu, ml = a.ml, N = a.N, na = mx + mu;
    RicWork w;
    ric_carve(mx, mu, ml, a.work + (size_t)prob * ric_carve(mx, mu, ml, nullptr, nullptr), &w);
    MP GBl = LDSM ? (MP)lds_M : (MP)w.GBl;
    MP Xs = LDSM ? (MP)lds_M + (size_t)ml * ml : (MP) nullptr;
    const int CB = LDSM ? a.lds_cols : 0;
    const size_t nlin = a.time_varying ? (size_t)(N - 1) : 1;
    const double* Ab = a.A + (size_t)prob * nlin * mx * mx;
    const double* Bub = a.Bu + (size_t)prob * nlin * mx * mu;
    const double* Blb = a.Bl + (size_t)prob * nlin * mx * ml;
    const double* Gb = a.G + (size_t)prob * nlin * ml * mx;
    double* Kout = a.K + (size_t)prob * (N > 1 ? N - 1 : 0) * mu * mx;
    if (tid == 0) sing = 0;
    for (int e = tid; e < mx * mx; e += RIC_THREADS) w.P[e] = a.Q[e];   // Pk = Q                                  lqr.jl:147
    __syncthreads();
    double* P = w.P;
    double* Pn = w.Pn;
    const int tr = tid >> 5, tc = tid & 31;   // 16 x 32 thread grid for the elementwise passes (no per-element index division)
    int k = 0, status = 0;
    for (k = N - 1; k >= 1; k--) {                                       // for outer k=N-1:-1:1                    lqr.jl:150
        const size_t li = a.time_varying ? (size_t)(k - 1) : 0;
        const double *A = Ab + li * mx * mx, *Bu = Bub + li * mx * mu, *Bl = Blb + li * mx * ml, *G = Gb + li * ml * mx;
        if (a.time_varying || k == N - 1) {
            // AD = [A' | D] = [A | Bu] - Bλ (G Bλ)^-1 G [A | Bu]                                                    lqr.jl:151,154-155,158
            for (int i = tr; i < mx; i += RIC_THREADS / 32) {
                for (int j = tc; j < mx; j += 32) w.AD[(size_t)i * na + j] = A[(size_t)i * mx + j];
                for (int j = tc; j < mu; j += 32) w.AD[(size_t)i * na + mx + j] = Bu[(size_t)i * mu + j];
            }
            __syncthreads();
            if (ml > 0) {
                wg_gemm<false>(ml, ml, mx, 1.0, G, mx, Bl, ml, 0.0, w.GBl, ml);          // M22 = G*Bλ                  lqr.jl:155
                wg_gemm<false>(ml, na, mx, 1.0, G, mx, w.AD, na, 0.0, w.X, na);          // [G*A | G*Bu]               lqr.jl:158,154
                if (LDSM) { for (int e = tid; e < ml * ml; e += RIC_THREADS) GBl[e] = w.GBl[e]; __syncthreads(); }
                wg_lu<MP>(ml, GBl, ml, w.piv, &sing, red_v, red_i);
                if (!sing) {
                    wg_lu_solve<MP>(ml, GBl, ml, w.piv, w.X, na, na, Xs, CB);            // X = [F | E]
                    wg_gemm<false>(mx, na, ml, -1.0, Bl, ml, w.X, na, 1.0, w.AD, na);
                }
            }
        }
        if (sing) { status = CCLQR_ESINGULAR_; break; }
        RSTAMP(RP_PRE);
        // W = Pk [A' | D]   (Pk symmetric)
        wg_gemm<true>(mx, na, mx, 1.0, P, mx, w.AD, na, 0.0, w.W, na);
        RSTAMP(RP_PA);
        // TS = D' W = [D' Pk A' | D' Pk D] ;  S = R + D' Pk D ;  Ku = S \ (D' Pk A')                               lqr.jl:152-160
        wg_gemm<true>(mu, na, mx, 1.0, w.AD + mx, na, w.W, na, 0.0, w.TS, na);
        for (int e = tid; e < mu * mu; e += RIC_THREADS) w.S[e] = a.R[e] + w.TS[(size_t)(e / mu) * na + mx + e % mu];
        for (int e = tid; e < mu * mx; e += RIC_THREADS) w.Ku[e] = w.TS[(size_t)(e / mx) * na + e % mx];
        __syncthreads();
        wg_lu<double*>(mu, w.S, mu, w.piv + ml + 4, &sing, red_v, red_i);
        if (sing) { status = CCLQR_ESINGULAR_; break; }
        wg_lu_solve<double*>(mu, w.S, mu, w.piv + ml + 4, w.Ku, mx, mx, nullptr, 0);
        const double* Ku = w.Ku;
        for (int e = tid; e < mu * mx; e += RIC_THREADS) Kout[(size_t)(k - 1) * mu * mx + e] = Ku[e];   // Ku[k][i] = Kk[i:i,:]  lqr.jl:162-164
        // KRK = R Kuk (mu x mx)
        for (int e = tid; e < mu * mx; e += RIC_THREADS) {
            int i = e / mx, c = e % mx; double s = 0.0;
            for (int q = 0; q < mu; q++) s += a.R[i * mu + q] * Ku[(size_t)q * mx + c];
            w.KRK[e] = s;
        }
        RSTAMP(RP_GAIN);
        // Abar = A' - D Kuk (= A-Bu*Kuk-Bλ*Kλk, lqr.jl:169) ;  W <- Pk Abar = Pk A' - (Pk D) Kuk ;  Pn = Q
        for (int i = tr; i < mx; i += RIC_THREADS / 32) {
            const double* ADi = w.AD + (size_t)i * na;
            double* Wi = w.W + (size_t)i * na;
            for (int j = tc; j < mx; j += 32) {
                double ab = ADi[j], pw = Wi[j];
                for (int q = 0; q < mu; q++) { const double kq = Ku[(size_t)q * mx + j]; ab -= ADi[mx + q] * kq; pw -= Wi[mx + q] * kq; }
                w.Abar[(size_t)i * mx + j] = ab;
                Wi[j] = pw;
                Pn[(size_t)i * mx + j] = a.Q[(size_t)i * mx + j];
            }
        }
        __syncthreads();
        RSTAMP(RP_UPD);
        // Pkp1 = Q + Kuk'*R*Kuk + Abar'*Pk*Abar                                                                     lqr.jl:170
        wg_gemm<true>(mx, mx, mu, 1.0, Ku, mx, w.KRK, mx, 1.0, Pn, mx);
        wg_gemm<true>(mx, mx, mx, 1.0, w.Abar, mx, w.W, na, 1.0, Pn, mx);
        RSTAMP(RP_PP);
        // if norm(Pk-Pkp1) < 1e-5  break                                                                            lqr.jl:172-174
        double acc = 0.0;
        for (int e = tid; e < mx * mx; e += RIC_THREADS) { double d = P[e] - Pn[e]; acc += d * d; }
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if ((tid & 63) == 0) red_v[tid >> 6] = acc;
        __syncthreads();
        double tot = 0.0;
        for (int q = 0; q < RIC_WAVES; q++) tot += red_v[q];
        __syncthreads();
        RSTAMP(RP_NORM);
#ifdef CCLQR_PROFILE
        if (tid == 0 && blockIdx.x == 0) g_rprof[RP_STEPS] += 1;
#endif
        if (sqrt(tot) < a.tol) break;
        double* tmp = P; P = Pn; Pn = tmp;                                                               // Pk = Pkp1  lqr.jl:176
    }
    if (status == 0) {
        if (k < 1 && N - 1 >= 1) k = 1;   // Julia: after a completed loop the outer k holds its last value
        if (N - 1 < 1) k = 0;
        __syncthreads();
        for (int k2 = k - 1; k2 >= 1; k2--) {                                                    // Ku[k2] = Ku[k2+1]  lqr.jl:179-181
            for (int e = tid; e < mu * mx; e += RIC_THREADS) Kout[(size_t)(k2 - 1) * mu * mx + e] = Kout[(size_t)k2 * mu * mx + e];
            __syncthreads();
        }
    }
    if (tid == 0) { a.kbreak[prob] = k; a.status[prob] = status; }
}

hipError_t launch_riccati(const RicArgs& a, hipStream_t stream) {
    if (a.nprob <= 0) return hipSuccess;
    const int m = a.ml, na = a.mx + a.mu;
    RicArgs a2 = a;
    size_t lds = 0;
    a2.lds_cols = 0;
    if (m <= RIC_LDS_M && m > 0) {
        const size_t budget = 150 * 1024;                 // of the 160 KB per CU; the rest is static LDS
        size_t cols = (budget - (size_t)m * m * sizeof(double)) / ((size_t)m * sizeof(double));
        if (cols > RIC_THREADS) cols = RIC_THREADS;
        if (cols > (size_t)na) cols = na;
        a2.lds_cols = (int)cols;
        lds = ((size_t)m * m + (size_t)m * cols) * sizeof(double);   // G Bλ for the pivoted LU + one batch of right-hand-side columns
    }
    if (lds > 0) {
        hipError_t e = hipFuncSetAttribute((const void*)riccati_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(riccati_kernel<true>, dim3(a.nprob), dim3(RIC_THREADS), lds, stream, a2);
    } else {
        hipLaunchKernelGGL(riccati_kernel<false>, dim3(a.nprob), dim3(RIC_THREADS), 0, stream, a2);
    }
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

// riccati.hip -- constrained discrete Riccati backward recursion, dlqr(A,Bu,Bλ,G,Q,R,N) of src/control/lqr.jl:141-184 and its
// time-varying twin dlqr(mechanism, ...) of src/control/lqr_tracking.jl:73-122 (A,Bu,Bλ,G indexed by knot).
//
// One workgroup (8 wavefronts) per independent problem, persistent over k = N-1 ... 1: the sweep is sequential in k, so all
// parallelism inside a problem is in the dense algebra of one step.  The mx x mx products (Abar = A - Bu Ku - Bλ Kλ,
// P Abar, Abar' (P Abar)) run on the fp64 matrix cores (v_mfma_f64_16x16x4_f64, 16x16 tiles per wavefront); M \ b is an
// in-kernel LU with partial pivoting (Julia's `\` on a square matrix) followed by one-column-per-thread substitution.
// Statement-by-statement correspondence with lqr.jl is marked with the line numbers.
#include "cclqr_internal.h"
#include <math.h>

namespace cclqr {

typedef double v4d __attribute__((ext_vector_type(4)));
#define RIC_THREADS 512
#define RIC_WAVES (RIC_THREADS / 64)

// C (M x N, ldc) = beta * C + alpha * op(A) (M x K) * B (K x N);  op(A) = A' when TA (A stored K x M).  Whole workgroup.
// fragment maps of v_mfma_f64_16x16x4_f64: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15],
// C/D: col = lane&15, row = (lane>>4) + 4*reg  (cdna_hip_programming.md §3).
template <bool TA>
__device__ void wg_gemm(int M, int N, int K, double alpha, const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb,
                        double beta, double* __restrict__ C, int ldc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int tm = (M + 15) >> 4, tn = (N + 15) >> 4;
    for (int tile = wave; tile < tm * tn; tile += RIC_WAVES) {
        const int i0 = (tile / tn) << 4, j0 = (tile % tn) << 4;
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        const bool iok = (i0 + li) < M, jok = (j0 + li) < N;
        for (int k0 = 0; k0 < K; k0 += 4) {
            const int k = k0 + lk;
            double a = 0.0, b = 0.0;
            if (k < K) {
                if (iok) a = TA ? A[(size_t)k * lda + i0 + li] : A[(size_t)(i0 + li) * lda + k];
                if (jok) b = B[(size_t)k * ldb + j0 + li];
            }
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
        if (jok) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = i0 + lk + 4 * r;
                if (row < M) {
                    double* c = C + (size_t)row * ldc + j0 + li;
                    *c = (beta == 0.0 ? 0.0 : beta * *c) + alpha * acc[r];
                }
            }
        }
    }
    __syncthreads();
}

// in-place LU with partial pivoting of the n x n matrix A (row major, lda); piv[c] = pivot row of column c.  *sing set if a pivot is 0.
__device__ void wg_lu(int n, double* A, int lda, int* piv, int* sing, double* red_v, int* red_i) {
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
        const int w = n - c - 1;
        for (int e = tid; e < w * w; e += RIC_THREADS) {
            int r = c + 1 + e / w, j = c + 1 + e % w;
            A[(size_t)r * lda + j] -= A[(size_t)r * lda + c] * A[(size_t)c * lda + j];
        }
        __syncthreads();
    }
    (void)red_i;
}
// solve (LU) X = B for nrhs columns, B (n x nrhs, ldb) in place; one column per thread
__device__ void wg_lu_solve(int n, const double* LU, int lda, const int* piv, double* B, int ldb, int nrhs) {
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

struct RicWork {
    double *GBl, *GBlT, *GBu, *Yt, *BlT, *BuT, *D, *GA, *DtP, *Mm, *bb, *Abar, *T, *P, *Pn, *KRK;
    int* piv;
};
__host__ __device__ inline size_t ric_carve(int mx, int mu, int ml, double* base, RicWork* w) {
    const size_t m = mu + ml;
    size_t o = 0;
    auto take = [&](size_t n) { double* p = base ? base + o : nullptr; o += (n + 1) & ~(size_t)1; return p; };
    double *GBl = take((size_t)ml * ml), *GBlT = take((size_t)ml * ml), *GBu = take((size_t)ml * mu), *Yt = take((size_t)ml * mx),
           *BlT = take((size_t)ml * mx), *BuT = take((size_t)mu * mx), *D = take((size_t)mx * mu), *GA = take((size_t)ml * mx),
           *DtP = take((size_t)mu * mx), *Mm = take(m * m), *bb = take(m * mx), *Abar = take((size_t)mx * mx), *T = take((size_t)mx * mx),
           *P = take((size_t)mx * mx), *Pn = take((size_t)mx * mx), *KRK = take((size_t)mu * mx), *piv = take(m + ml + 2);
    if (w) { w->GBl = GBl; w->GBlT = GBlT; w->GBu = GBu; w->Yt = Yt; w->BlT = BlT; w->BuT = BuT; w->D = D; w->GA = GA; w->DtP = DtP; w->Mm = Mm;
             w->bb = bb; w->Abar = Abar; w->T = T; w->P = P; w->Pn = Pn; w->KRK = KRK; w->piv = (int*)piv; }
    return o;
}
size_t ric_work_doubles(int mx, int mu, int ml) { return ric_carve(mx, mu, ml, nullptr, nullptr); }

__global__ __launch_bounds__(RIC_THREADS) void riccati_kernel(RicArgs a) {
    __shared__ double red_v[RIC_WAVES];
    __shared__ int red_i[4];
    __shared__ int sing;
    const int prob = blockIdx.x, tid = threadIdx.x;
    const int mx = a.mx, mu = a.mu, ml = a.ml, m = mu + ml, N = a.N;
    RicWork w;
    ric_carve(mx, mu, ml, a.work + (size_t)prob * ric_carve(mx, mu, ml, nullptr, nullptr), &w);
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
    int k = 0, status = 0;
    for (k = N - 1; k >= 1; k--) {                                       // for outer k=N-1:-1:1                    lqr.jl:150
        const size_t li = a.time_varying ? (size_t)(k - 1) : 0;
        const double *A = Ab + li * mx * mx, *Bu = Bub + li * mx * mu, *Bl = Blb + li * mx * ml, *G = Gb + li * ml * mx;
        if (a.time_varying || k == N - 1) {
            // D = Bu - Bλ/(G*Bλ)*G*Bu                                                                              lqr.jl:151
            for (int e = tid; e < ml * mx; e += RIC_THREADS) { int r = e / mx, c = e % mx; w.BlT[e] = Bl[(size_t)c * ml + r]; }
            for (int e = tid; e < mu * mx; e += RIC_THREADS) { int r = e / mx, c = e % mx; w.BuT[e] = Bu[(size_t)c * mu + r]; }
            for (int e = tid; e < mx * mu; e += RIC_THREADS) w.D[e] = Bu[e];
            __syncthreads();
            if (ml > 0) {
                wg_gemm<false>(ml, ml, mx, 1.0, G, mx, Bl, ml, 0.0, w.GBl, ml);   // M22 = G*Bλ                         lqr.jl:155
                wg_gemm<false>(ml, mu, mx, 1.0, G, mx, Bu, mu, 0.0, w.GBu, mu);   // M21 = G*Bu                         lqr.jl:154
                wg_gemm<false>(ml, mx, mx, 1.0, G, mx, A, mx, 0.0, w.GA, mx);     // G*A (lower block of b)             lqr.jl:158
                for (int e = tid; e < ml * ml; e += RIC_THREADS) { int r = e / ml, c = e % ml; w.GBlT[e] = w.GBl[(size_t)c * ml + r]; }
                for (int e = tid; e < ml * mx; e += RIC_THREADS) w.Yt[e] = w.BlT[e];
                __syncthreads();
                wg_lu(ml, w.GBlT, ml, w.piv + m + 1, &sing, red_v, red_i);          // (G Bλ)' Y' = Bλ'
                wg_lu_solve(ml, w.GBlT, ml, w.piv + m + 1, w.Yt, mx, mx);
                wg_gemm<true>(mx, mu, ml, -1.0, w.Yt, mx, w.GBu, mu, 1.0, w.D, mu);
            }
        }
        if (sing) { status = CCLQR_ESINGULAR_; break; }
        // M = [R + D'PBu  D'PBλ; G*Bu  G*Bλ] ; b = [D'*Pk; G]*A                                                    lqr.jl:152-158
        wg_gemm<true>(mu, mx, mx, 1.0, w.D, mu, P, mx, 0.0, w.DtP, mx);
        for (int e = tid; e < mu * mu; e += RIC_THREADS) w.Mm[(size_t)(e / mu) * m + e % mu] = a.R[e];
        for (int e = tid; e < ml * mu; e += RIC_THREADS) w.Mm[(size_t)(mu + e / mu) * m + e % mu] = w.GBu[e];
        for (int e = tid; e < ml * ml; e += RIC_THREADS) w.Mm[(size_t)(mu + e / ml) * m + mu + e % ml] = w.GBl[e];
        for (int e = tid; e < ml * mx; e += RIC_THREADS) w.bb[(size_t)mu * mx + e] = w.GA[e];
        __syncthreads();
        wg_gemm<false>(mu, mu, mx, 1.0, w.DtP, mx, Bu, mu, 1.0, w.Mm, m);
        if (ml > 0) wg_gemm<false>(mu, ml, mx, 1.0, w.DtP, mx, Bl, ml, 0.0, w.Mm + mu, m);
        wg_gemm<false>(mu, mx, mx, 1.0, w.DtP, mx, A, mx, 0.0, w.bb, mx);
        // Kk = M\b                                                                                                  lqr.jl:160
        wg_lu(m, w.Mm, m, w.piv, &sing, red_v, red_i);
        if (sing) { status = CCLQR_ESINGULAR_; break; }
        wg_lu_solve(m, w.Mm, m, w.piv, w.bb, mx, mx);
        for (int e = tid; e < mu * mx; e += RIC_THREADS) Kout[(size_t)(k - 1) * mu * mx + e] = w.bb[e];   // Ku[k][i] = Kk[i:i,:]  lqr.jl:162-164
        // Abar = A-Bu*Kuk-Bλ*Kλk                                                                                    lqr.jl:169
        for (int e = tid; e < mx * mx; e += RIC_THREADS) w.Abar[e] = A[e];
        // KRK = R Kuk (mu x mx), tiny
        for (int e = tid; e < mu * mx; e += RIC_THREADS) {
            int i = e / mx, c = e % mx; double s = 0.0;
            for (int q = 0; q < mu; q++) s += a.R[i * mu + q] * w.bb[(size_t)q * mx + c];
            w.KRK[e] = s;
        }
        for (int e = tid; e < mx * mx; e += RIC_THREADS) Pn[e] = a.Q[e];
        __syncthreads();
        wg_gemm<true>(mx, mx, mu, -1.0, w.BuT, mx, w.bb, mx, 1.0, w.Abar, mx);
        if (ml > 0) wg_gemm<true>(mx, mx, ml, -1.0, w.BlT, mx, w.bb + (size_t)mu * mx, mx, 1.0, w.Abar, mx);
        // Pkp1 = Q + Kuk'*R*Kuk + Abar'*Pk*Abar                                                                     lqr.jl:170
        wg_gemm<true>(mx, mx, mu, 1.0, w.bb, mx, w.KRK, mx, 1.0, Pn, mx);
        wg_gemm<true>(mx, mx, mx, 1.0, P, mx, w.Abar, mx, 0.0, w.T, mx);    // Pk Abar (Pk symmetric)
        wg_gemm<true>(mx, mx, mx, 1.0, w.Abar, mx, w.T, mx, 1.0, Pn, mx);
        // if norm(Pk-Pkp1) < 1e-5  break                                                                            lqr.jl:172-174
        double acc = 0.0;
        for (int e = tid; e < mx * mx; e += RIC_THREADS) { double d = P[e] - Pn[e]; acc += d * d; }
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if ((tid & 63) == 0) red_v[tid >> 6] = acc;
        __syncthreads();
        double tot = 0.0;
        for (int q = 0; q < RIC_WAVES; q++) tot += red_v[q];
        __syncthreads();
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
    hipLaunchKernelGGL(riccati_kernel, dim3(a.nprob), dim3(RIC_THREADS), 0, stream, a);
    return hipGetLastError();
}

}  // namespace cclqr

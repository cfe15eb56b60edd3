// cclqr_treereg.h -- the register-resident design of cclqr_chain.h for BRANCHING trees (rollout_treereg.hip; round 4).
//
// Lane t < nb of an instance's lane group owns link t (= body t and the joint that hangs it off its parent) and keeps the link in
// registers exactly as the chain kernel does (LinkC, sparse Jacobians, cclqr_chain.h).  What a tree changes:
//   * the parent link is an arbitrary lane with a smaller index and a body may carry up to CCLQR_MAXK child joints: parent data arrive by
//     an indexed lane read (ds_bpermute) instead of the wave shift, and what the children send to their parent is SUMMED over the child list;
//   * joints that share their parent body couple pairwise in S = G_v D^-1 G_k' (siblings): lane j also builds S_{j,s} = W_a[j] Gk_a[s]'
//     for each sibling s, into the two 5x5 blocks per sibling pair that the general-tree layout has always had (cclqr_dev.h);
//   * the solve is the no-fill elimination of cclqr_dev.h (ph_tree_elim / ph_tree_back: a link goes after its subtree and its larger
//     siblings), but SCHEDULED: links whose neighbourhoods do not meet are eliminated in the same step by different 8-lane groups, and a
//     lane reads what it has to do in a step -- LDS offsets only -- from a per-(step, lane) record the host wrote (TreeRegDev).
// LDS per link: the chain kernel's 150 doubles (Schur blocks 75, R 5, DL 5, sparse G_k 39, LAM 5, D_R^-1 9, invariants 6, C 6) + 50 per
// sibling pair, against the 300 + 50 of the LDS-resident tree kernel: two instances of a 9..16-body tree per wavefront.
// Blocks are ROW-major here (element (r, q) at 5 r + q), as in the LDS-resident tree kernel.
//
// Every function is __host__ __device__ so that tests/emu/emu_treereg.cpp runs the identical arithmetic lane by lane on the CPU (test
// infrastructure only).  Cross-lane inputs are explicit arguments: the kernel fills them by ds_bpermute, the emulator by reading the other lane.
#pragma once
#include "cclqr_chain.h"

namespace cclqr {

#ifndef TR_G16_MAXLINKS
#define TR_G16_MAXLINKS 8      // (experiment switch: 16 = four instances of a 9..16-link tree per wavefront)
#endif
#define TR_MAXSTEP 64      // steps of a schedule (<= links)
#define TR_LANES 64        // lanes per instance at most (33 .. 64 links: the whole wavefront is one instance, round 5)
// what one lane does in one step of the elimination / of the back substitution (all LDS offsets relative to the instance's image):
//   elimination, lane (group g of link l, column c < 5):  z = S_ll^-1 S_{l,x_g}[:, c];  S_{x_g', x_g}[:, c] -= S_{x_g', l} z for every g';  z -> S_{l,x_g}[:, c]
//                lane (group 0, c == 5):                  y = S_ll^-1 r_l;               r_{x_g'} -= S_{x_g', l} y;                          y -> r_l
//       o0 = S_ll, o1 = the right-hand side (stride ctl >> 8), a[g'] = S_{x_g', l}, b[g'] = the target (same stride)
//   back substitution, lane (link l, row r < 5):  dl_l[r] = y_l[r] - sum_g Z_{l,x_g}[r, :] dl_{x_g}
//       o0 = y_l[r], o1 = dl_l[r], a[g] = row r of Z_{l,x_g}, b[g] = dl_{x_g}
// ctl = number of neighbours nn (low byte) | stride << 8 | 1 << 16 when the lane has work in this step
struct alignas(16) TrRec { int o0, o1, ctl, pad; int a[CCLQR_MAXK], b[CCLQR_MAXK]; };
HD bool trrec_on(const TrRec& K) { return (K.ctl >> 16) != 0; }

// tables of the register-resident tree kernel, in device memory right behind the mechanism's MechDev (capi.hip allocates both in one piece)
struct TreeRegDev {
    int lanes, nbp, nss;          // lanes per instance / links the image is laid out for / sibling blocks, all as the kernel will be launched
    int maxchild;                 // largest child count of a body
    int sib_n[CCLQR_MAXL], sib_link[CCLQR_MAXL][CCLQR_MAXK - 1], sib_off[CCLQR_MAXL][CCLQR_MAXK - 1];   // the other joints on link l's parent body, offset of S_{l, sib}
    int maxsib;
    int ne_steps, nb_steps;
    TrRec el[TR_MAXSTEP][TR_LANES], bk[TR_MAXSTEP][TR_LANES];
};
// the tables sit behind the mechanism's MechDev in ONE device allocation (capi.hip cclqr_mech_create), at the next 16-byte boundary
HD size_t treereg_offset() { return (sizeof(MechDev) + 15) & ~(size_t)15; }
HD const TreeRegDev* treereg_of(const MechDev* M) { return (const TreeRegDev*)((const char*)M + treereg_offset()); }

// LDS image: make_chain_layout + the sibling blocks.  nbp = links the image is laid out for (compile time in the kernel), nss = 2 x sibling pairs
HD Lay make_treereg_layout(int nbp, int nss) {
    Lay L = make_chain_layout(nbp);
    L.SS = L.total + 1;            // (make_chain_layout's total is odd: even start)
    L.total = (L.SS + 25 * nss) | 1;
    return L;
}

// the owned link's place in the tree (registers; loaded once per launch)
struct TreeL {
    int par;                       // parent link (own link when the parent is the origin)
    int nchild, child[CCLQR_MAXK]; // child links of the owned BODY (own link in the unused entries)
    int nsib, sib[CCLQR_MAXK - 1], siboff[CCLQR_MAXK - 1];   // the other joints on the parent body, and the LDS offset of the block S_{own, sib}
};
HD void tree_load(TreeL& T, const MechDev* M, const TreeRegDev* R, int t, int nb) {
    const bool on = t < nb;
    const int l = on ? t : 0;
    const int p = M->parent[l];
    T.par = (on && p >= 0) ? p : t;
    T.nchild = on ? M->nchild[l] : 0;
#pragma unroll
    for (int k = 0; k < CCLQR_MAXK; k++) T.child[k] = (k < T.nchild) ? M->child[l][k] : t;
    T.nsib = on ? R->sib_n[l] : 0;
#pragma unroll
    for (int k = 0; k < CCLQR_MAXK - 1; k++) { T.sib[k] = (k < T.nsib) ? R->sib_link[l][k] : l; T.siboff[k] = (k < T.nsib) ? R->sib_off[l][k] : 0; }
}

// one 5x5 block W Gk_a[x]' (W = this link's child-side rows W_b when `bside`, else its parent-side rows W_a) -> row-major at `oblk`:
//   element (r, q) = w[r] . PA_x[q]  -/+  sx (wXT[r] . XT_x[q])      (Gk_a = (-XT | PA); W_b = (sxb wXT | wPB), W_a = (-sxa wXT | wPA))
HD void tr_side_block(int x, double sxs, const Lay& Y, double* L, const double (*wXT)[3], const double (*w)[3], int oblk) {
#pragma unroll
    for (int q = 0; q < 5; q++) {
        const int o = Y.GKA + GKSZ * x + gk_row(q), ob = q < 3 ? 3 : 0;
        double kx[3] = {0, 0, 0}, ka[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (q < 3) kx[i] = L[o + i];
            ka[i] = L[o + ob + 3 + i];
        }
#pragma unroll
        for (int r = 0; r < 5; r++) {
            double v = w[r][0] * ka[0] + w[r][1] * ka[1] + w[r][2] * ka[2];
            if (r < 3 && q < 3) v += sxs * (wXT[r][0] * kx[0] + wXT[r][1] * kx[1] + wXT[r][2] * kx[2]);
            L[oblk + 5 * r + q] = v;
        }
    }
}

// ---- Schur complement blocks of link j, built by its own lane straight into LDS (the tree version of ck_schur_rows), ROW-major:
//   S_jj = W_b[j] Gk_b[j]' + W_a[j] Gk_a[j]'          -> SJJ[j]
//   S_jp = W_a[j] Gk_b[p]'                             -> SJP[j]        (p = parent link, when there is one)
//   S_jc = W_b[j] Gk_a[c]'                             -> SPJ[c]        (every child c: "S_{parent,child}" of the child's slot)
//   S_js = W_a[j] Gk_a[s]'                             -> siboff[k]     (every sibling s: the joints on the parent body couple pairwise)
//   r_j  = g_j - W_b d_j - W_a d_p                     -> R[j]          (pd = residual of the parent body, from the parent lane)
// maxchild / maxsib: the mechanism's largest counts (wave-uniform loop bounds)
HD void tr_schur_rows(const LinkC& c, const TreeL& T, int j, bool store, int maxchild, int maxsib, const Lay& Y, double* L, const double (*wXT)[3],
                      const double (*wPB)[3], const double (*wPA)[3], const double* g, const double* d, const double* pd) {
    const double sx = c.sxb + c.sxa;
    const int jp = T.par;
#pragma unroll
    for (int q = 0; q < 5; q++) {
        const int o = gk_row(q), ob = q < 3 ? 3 : 0;   // offset of PB inside the row
        double kx[3] = {0, 0, 0}, kpx[3] = {0, 0, 0}, kb[3], ka[3], kpb[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (q < 3) { kx[i] = L[Y.GKA + GKSZ * j + o + i]; kpx[i] = L[Y.GKA + GKSZ * jp + o + i]; }
            kb[i] = L[Y.GKA + GKSZ * j + o + ob + i]; ka[i] = L[Y.GKA + GKSZ * j + o + ob + 3 + i];
            kpb[i] = L[Y.GKA + GKSZ * jp + o + ob + i];
        }
        double ojj[5], ojp[5];
#pragma unroll
        for (int r = 0; r < 5; r++) {
            const double bb = wPB[r][0] * kb[0] + wPB[r][1] * kb[1] + wPB[r][2] * kb[2];
            const double aa = wPA[r][0] * ka[0] + wPA[r][1] * ka[1] + wPA[r][2] * ka[2];
            const double ajp = wPA[r][0] * kpb[0] + wPA[r][1] * kpb[1] + wPA[r][2] * kpb[2];
            if (r < 3 && q < 3) {
                const double xx = wXT[r][0] * kx[0] + wXT[r][1] * kx[1] + wXT[r][2] * kx[2];
                const double xjp = wXT[r][0] * kpx[0] + wXT[r][1] * kpx[1] + wXT[r][2] * kpx[2];
                ojj[r] = sx * xx + bb + aa; ojp[r] = ajp - c.sxa * xjp;
            } else { ojj[r] = bb + aa; ojp[r] = ajp; }
        }
        if (store) {
#pragma unroll
            for (int r = 0; r < 5; r++) L[Y.SJJ + 25 * j + 5 * r + q] = ojj[r];
            if (c.has_a()) {
#pragma unroll
                for (int r = 0; r < 5; r++) L[Y.SJP + 25 * j + 5 * r + q] = ojp[r];
            }
        }
        SCHED_FENCE();
    }
    // real loops (one copy of the block code) over lists that ROTATE through fixed registers: an index the compiler cannot resolve would put
    // the lists into scratch memory
    {
        int c0 = T.child[0], c1 = T.child[1], c2 = T.child[2], c3 = T.child[3];
#pragma unroll 1
        for (int k = 0; k < maxchild; k++) {
            if (store && k < T.nchild) tr_side_block(c0, -c.sxb, Y, L, wXT, wPB, Y.SPJ + 25 * c0);
            c0 = c1; c1 = c2; c2 = c3;
        }
        int s0 = T.sib[0], s1 = T.sib[1], s2 = T.sib[2], o0 = T.siboff[0], o1 = T.siboff[1], o2 = T.siboff[2];
#pragma unroll 1
        for (int k = 0; k < maxsib; k++) {
            if (store && k < T.nsib) tr_side_block(s0, c.sxa, Y, L, wXT, wPA, o0);
            s0 = s1; s1 = s2; o0 = o1; o1 = o2;
        }
    }
    if (store) {
#pragma unroll
        for (int r = 0; r < 5; r++) {
            const double bd = wPB[r][0] * d[3] + wPB[r][1] * d[4] + wPB[r][2] * d[5];
            const double ad = wPA[r][0] * pd[3] + wPA[r][1] * pd[4] + wPA[r][2] * pd[5];
            double rr = g[r] - bd - ad;
            if (r < 3) {
                const double xd = wXT[r][0] * d[0] + wXT[r][1] * d[1] + wXT[r][2] * d[2];
                const double xa = wXT[r][0] * pd[0] + wXT[r][1] * pd[1] + wXT[r][2] * pd[2];
                rr = g[r] - (c.sxb * xd + bd) - (ad - c.sxa * xa);
            }
            L[Y.R + 5 * j + r] = rr;
        }
    }
}

// ---- one step of the scheduled elimination / back substitution for this lane (record K of the step; see TrRec)
HD void tr_elim(const TrRec& K, double* L) {
    if (!trrec_on(K)) return;
    const int nn = K.ctl & 0xff, st = (K.ctl >> 8) & 0xff;
    double lu[25], zy[5];
#pragma unroll
    for (int e = 0; e < 25; e++) lu[e] = L[K.o0 + e];
#pragma unroll
    for (int r = 0; r < 5; r++) zy[r] = L[K.o1 + st * r];
    lu5_factor(lu);
    lu5_solve(lu, zy);
    int a0 = K.a[0], a1 = K.a[1], a2 = K.a[2], a3 = K.a[3], b0 = K.b[0], b1 = K.b[1], b2 = K.b[2], b3 = K.b[3];      // rotate (see tr_schur_rows)
#pragma unroll 1
    for (int gp = 0; gp < nn; gp++) {
        double sxl[25], tg[5];
#pragma unroll
        for (int i = 0; i < 25; i++) sxl[i] = L[a0 + i];
#pragma unroll
        for (int r = 0; r < 5; r++) tg[r] = L[b0 + st * r];
#pragma unroll
        for (int r = 0; r < 5; r++) tg[r] -= sxl[5 * r] * zy[0] + sxl[5 * r + 1] * zy[1] + sxl[5 * r + 2] * zy[2] + sxl[5 * r + 3] * zy[3] + sxl[5 * r + 4] * zy[4];
#pragma unroll
        for (int r = 0; r < 5; r++) L[b0 + st * r] = tg[r];
        a0 = a1; a1 = a2; a2 = a3; b0 = b1; b1 = b2; b2 = b3;
    }
#pragma unroll
    for (int r = 0; r < 5; r++) L[K.o1 + st * r] = zy[r];
}
HD void tr_back(const TrRec& K, double* L) {
    if (!trrec_on(K)) return;
    const int nn = K.ctl & 0xff;
    double acc = L[K.o0];
    int a0 = K.a[0], a1 = K.a[1], a2 = K.a[2], a3 = K.a[3], b0 = K.b[0], b1 = K.b[1], b2 = K.b[2], b3 = K.b[3];
#pragma unroll 1
    for (int g = 0; g < nn; g++) {
        acc -= L[a0] * L[b0] + L[a0 + 1] * L[b0 + 1] + L[a0 + 2] * L[b0 + 2] + L[a0 + 3] * L[b0 + 3] + L[a0 + 4] * L[b0 + 4];
        a0 = a1; a1 = a2; a2 = a3; b0 = b1; b1 = b2; b2 = b3;
    }
    L[K.o1] = acc;
}

// lanes per instance and links of the image the kernel is instantiated for: 16 lanes when the mechanism has at most 8 links and no link
// with more than two neighbours left at its elimination (tree8 = 8 x the largest neighbour count, MechDev::tree), else 32
HD int treereg_lanes(int nb, int tree8) { return nb > 32 ? 64 : ((nb <= TR_G16_MAXLINKS && tree8 <= 16) ? 16 : 32); }
HD int treereg_layout_links(int nb, int tree8) {
    if (nb > 32) return nb <= 48 ? 48 : 64;
    const int n = nb <= 4 ? 4 : (nb <= 8 ? 8 : (nb <= 16 ? (nb + 1) / 2 * 2 : (nb <= 24 ? 24 : 32)));       // 10, 12, 14, 16: a 14-link image lets four workgroups share a CU
    return (treereg_lanes(nb, tree8) == 32 && n < 8) ? 8 : n;       // the instantiations of rollout_treereg.hip
}

}  // namespace cclqr

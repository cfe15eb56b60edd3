// cclqr_treereg_tables.h -- host side of cclqr_treereg.h: sibling lists, block offsets in the register-resident tree image and the
// SCHEDULE of the no-fill elimination (which links go in the same step, which 8-lane group does what), written as per-(step, lane) records.
#pragma once
#include "cclqr_treereg.h"
#include "cclqr_internal.h"
#include <string>
#include <vector>
#include <algorithm>
#include <string.h>

namespace cclqr {

// rollout_treereg.hip; a.M must point at the device image [MechDev | TreeRegDev] (treereg_of)
size_t treereg_lds_bytes(int nb, int tree8, int npairs);
hipError_t launch_rollout_treereg(const RolloutArgs& a, int nb, int tree8, int npairs, int extra, int newton_mode, hipStream_t stream);

// R for the mechanism H (H.tree != 0: build_mech_tables, cclqr_tables.h).  Returns false with `err` set when the mechanism does not fit
// the kernel (more than TR_LANES links).
static inline bool build_treereg_tables(const MechDev& H, TreeRegDev& R, std::string& err) {
    memset(&R, 0, sizeof(R));
    const int nb = H.nb;
    if (nb > TR_LANES) { err = "more links than lanes"; return false; }
    const int G = treereg_lanes(nb, H.tree), nbp = treereg_layout_links(nb, H.tree);
    R.lanes = G; R.nbp = nbp; R.nss = 2 * H.npairs;
    const Lay Y = make_treereg_layout(nbp, R.nss);
    auto pair_of = [&](int i, int j) { for (int q = 0; q < H.npairs; q++) if (H.pair_i[q] == i && H.pair_j[q] == j) return q; return -1; };
    // offset of the block S_{xr, xc} (both around one body, or one the joint of the body the other hangs off): cclqr_tables.h, in THIS layout
    auto block = [&](int xr, int xc) {
        if (xr == xc) return Y.SJJ + 25 * xr;
        if (H.parent[xc] == xr) return Y.SPJ + 25 * xc;       // S_{parent, child}
        if (H.parent[xr] == xc) return Y.SJP + 25 * xr;       // S_{child, parent}
        const int q = xr < xc ? pair_of(xr, xc) : pair_of(xc, xr);
        return Y.SS + 25 * (2 * q + (xr < xc ? 0 : 1));        // siblings: S_ij (i < j) then S_ji
    };
    for (int l = 0; l < nb; l++) {
        if (H.nchild[l] > R.maxchild) R.maxchild = H.nchild[l];
        const int a = H.parent[l];
        if (a < 0) continue;
        for (int x = 0; x < H.nchild[a]; x++) {
            const int s = H.child[a][x];
            if (s == l) continue;
            R.sib_link[l][R.sib_n[l]] = s; R.sib_off[l][R.sib_n[l]] = block(l, s); R.sib_n[l]++;
        }
        if (R.sib_n[l] > R.maxsib) R.maxsib = R.sib_n[l];
    }
    // neighbours left when l is eliminated (its subtree and its larger siblings are gone): the parent joint and the smaller siblings
    std::vector<std::vector<int>> N(nb);
    for (int l = 0; l < nb; l++) {
        const int a = H.parent[l];
        if (a < 0) continue;
        N[l].push_back(a);
        for (int x = 0; x < H.nchild[a]; x++) if (H.child[a][x] < l) N[l].push_back(H.child[a][x]);
        if ((int)N[l].size() > CCLQR_MAXK || 8 * (int)N[l].size() > G) { err = "internal: more neighbours in the elimination than lane groups"; return false; }
    }
    const int slots = G / 8;
    // ---- elimination schedule: a link is ready once every link that has it as a neighbour is gone; links of one step must not share a
    // link of their neighbourhoods {l} + N(l) (they would update the same blocks)
    {
        // Among the ready links the one with the longest chain of eliminations still behind it goes first (tail[l] = 1 + the longest tail of its
        // neighbours, which are all smaller-numbered): the schedule's length is what a Newton iteration pays, and taking ready links by number alone
        // leaves a critical chain waiting behind a link that could go any time (the benchmark's 14-body tree: 9 steps -> 8; its critical chain is 7)
        std::vector<int> tail(nb, 1), by_tail(nb);
        for (int l = 0; l < nb; l++) {
            for (int x : N[l]) if (tail[x] + 1 > tail[l]) tail[l] = tail[x] + 1;
            by_tail[l] = l;
        }
        std::stable_sort(by_tail.begin(), by_tail.end(), [&](int p, int q) { return tail[p] != tail[q] ? tail[p] > tail[q] : p > q; });
        std::vector<int> gone(nb, 0);
        int left = nb, s = 0;
        while (left > 0) {
            if (s >= TR_MAXSTEP) { err = "internal: elimination schedule too long"; return false; }
            std::vector<int> touched(nb, 0), chosen;
            int used = 0;
            for (int l : by_tail) {
                if (gone[l]) continue;
                bool ready = true;
                for (int m = 0; m < nb && ready; m++)
                    if (!gone[m] && m != l)
                        for (int x : N[m]) if (x == l) ready = false;
                if (!ready) continue;
                const int need = N[l].empty() ? 1 : (int)N[l].size();
                if (used + need > slots) continue;
                bool clash = touched[l] != 0;
                for (int x : N[l]) if (touched[x]) clash = true;
                if (clash) continue;
                touched[l] = 1;
                for (int x : N[l]) touched[x] = 1;
                const int nn = (int)N[l].size();
                for (int g = 0; g < need; g++) {
                    const int q = used + g;
                    for (int c = 0; c < 6; c++) {
                        TrRec& K = R.el[s][8 * q + c];
                        const bool isy = g == 0 && c == 5;
                        if (!(isy || (g < nn && c < 5))) continue;
                        const int st = isy ? 1 : 5;
                        K.ctl = nn | (st << 8) | (1 << 16);
                        K.o0 = Y.SJJ + 25 * l;
                        K.o1 = isy ? Y.R + 5 * l : block(l, N[l][g]) + c;
                        for (int gp = 0; gp < nn; gp++) {
                            K.a[gp] = block(N[l][gp], l);
                            K.b[gp] = isy ? Y.R + 5 * N[l][gp] : block(N[l][gp], N[l][g]) + c;
                        }
                    }
                }
                used += need;
                chosen.push_back(l);
            }
            if (chosen.empty()) { err = "internal: elimination schedule stalled"; return false; }
            for (int l : chosen) { gone[l] = 1; left--; }
            s++;
        }
        R.ne_steps = s;
    }
    // ---- back substitution schedule: a link is ready once all its neighbours are solved; one 8-lane group (5 rows) per link
    {
        std::vector<int> solved(nb, 0);
        int left = nb, s = 0;
        while (left > 0) {
            if (s >= TR_MAXSTEP) { err = "internal: back substitution schedule too long"; return false; }
            std::vector<int> chosen;
            for (int l = 0; l < nb && (int)chosen.size() < slots; l++) {
                if (solved[l]) continue;
                bool ready = true;
                for (int x : N[l]) if (!solved[x]) ready = false;
                if (!ready) continue;
                const int q = (int)chosen.size(), nn = (int)N[l].size();
                for (int r = 0; r < 5; r++) {
                    TrRec& K = R.bk[s][8 * q + r];
                    K.ctl = nn | (1 << 16);
                    K.o0 = Y.R + 5 * l + r; K.o1 = Y.DL + 5 * l + r;
                    for (int g = 0; g < nn; g++) { K.a[g] = block(l, N[l][g]) + 5 * r; K.b[g] = Y.DL + 5 * N[l][g]; }
                }
                chosen.push_back(l);
            }
            if (chosen.empty()) { err = "internal: back substitution schedule stalled"; return false; }
            for (int l : chosen) { solved[l] = 1; left--; }
            s++;
        }
        R.nb_steps = s;
    }
    return true;
}

}  // namespace cclqr

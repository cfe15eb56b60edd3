// cclqr_tables.h -- host-only: validation of a cclqr_mech_desc / cclqr_ctrl_desc and the index permutations between the
// caller's body/joint numbering and the kernels' breadth-first link order.  No dynamics arithmetic happens here.
#pragma once
#include "../../include/cclqr.h"
#include "cclqr_internal.h"
#include "cclqr_loop.h"
#include <math.h>
#include <string.h>
#include <string>
#include <vector>

namespace cclqr {

// two unit rows orthogonal to the unit axis (same deterministic choice as the oracle; any basis gives the same trajectory)
static inline void orth_rows(const double* a, double* V12) {
    int e = 0;
    double best = fabs(a[0]);
    for (int i = 1; i < 3; i++) if (fabs(a[i]) < best) { best = fabs(a[i]); e = i; }
    double v1[3] = {0, 0, 0};
    v1[e] = 1.0;
    double d = a[e];
    for (int i = 0; i < 3; i++) v1[i] -= d * a[i];
    double n = sqrt(v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2]);
    for (int i = 0; i < 3; i++) v1[i] /= n;
    V12[0] = v1[0]; V12[1] = v1[1]; V12[2] = v1[2];
    V12[3] = a[1] * v1[2] - a[2] * v1[1]; V12[4] = a[2] * v1[0] - a[0] * v1[2]; V12[5] = a[0] * v1[1] - a[1] * v1[0];
}


// joint constants that do not depend on the link order: unit axis, vertices, conj(qoffset), row selectors and row kinds.
// Revolute = Translational3 + Rotational2, Prismatic = Translational2 + Rotational3, FixedOrientation = Rotational3 padded with two
// null rows in front (the row layout of a prismatic joint with nothing selected translationally)
static inline int fill_joint_consts(const cclqr_mech_desc* d, int j, MechDev& H, int l, std::string& err) {
    double ax[3] = {d->axis[3 * j], d->axis[3 * j + 1], d->axis[3 * j + 2]};
    const double n = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    if (n == 0.0) { err = "zero joint axis"; return CCLQR_EINVAL; }
    for (int i = 0; i < 3; i++) { ax[i] /= n; H.axis[l][i] = ax[i]; H.p1[l][i] = d->p1[3 * j + i]; H.p2[l][i] = d->p2[3 * j + i]; }
    H.qoc[l][0] = d->qoff[4 * j];
    for (int i = 1; i < 4; i++) H.qoc[l][i] = -d->qoff[4 * j + i];
    double V12[6];
    orth_rows(ax, V12);
    const int nt = (d->type[j] == CCLQR_REVOLUTE) ? 3 : 2;
    H.rotmask[l] = 0;
    for (int r = 0; r < 5; r++) {
        const bool rot = r >= nt;
        const int q = rot ? r - nt : r, nrows = rot ? 5 - nt : nt;
        if (rot) H.rotmask[l] |= 1 << r;
        for (int i = 0; i < 3; i++) H.sel[l][r][i] = (nrows == 3) ? (i == q ? 1.0 : 0.0) : V12[3 * q + i];
        if (d->type[j] == CCLQR_FIXED_ORIENTATION && !rot)
            for (int i = 0; i < 3; i++) H.sel[l][r][i] = 0.0;
    }
    H.type[l] = d->type[j];
    return CCLQR_OK;
}

// Mechanism whose constraint graph has cycles (or a FixedOrientation constraint): bodies and joints keep the caller's order
static inline int build_loop_tables(const cclqr_mech_desc* d, cclqr_mech* m, std::string& err) {
    const int nb = d->nb, nj = d->ne;
    if (nb > CCLQR_LOOP_MAXB || nj > CCLQR_LOOP_MAXJ) { err = "closed-loop mechanisms: up to 8 bodies and 12 joints"; return CCLQR_EUNSUPPORTED; }
    memset(&m->host, 0, sizeof(MechDev));
    m->nb = nb; m->nj = nj;
    MechDev& H = m->host;
    H.nb = nb; H.nj = nj; H.loop = 1; H.dt = d->dt; H.g = d->g;
    for (int b = 0; b < nb; b++) {
        m->link_of_body[b] = b; H.perm[b] = b;
        H.m[b] = d->mass[b];
        if (!(H.m[b] > 0)) { err = "non-positive mass"; return CCLQR_EINVAL; }
        for (int i = 0; i < 9; i++) H.J[b][i] = d->inertia[9 * b + i];
    }
    std::vector<int> seen(nb, 0);
    for (int j = 0; j < nj; j++) {
        const int a = d->parent[j], b = d->child[j];
        if (b < 0 || b >= nb || a < -1 || a >= nb || a == b) { err = "joint references a body out of range"; return CCLQR_EINVAL; }
        if (d->type[j] < CCLQR_REVOLUTE || d->type[j] > CCLQR_FIXED_ORIENTATION) { err = "unknown joint type"; return CCLQR_EINVAL; }
        m->link_of_joint[j] = j; H.jperm[j] = j;
        H.parent[j] = a; H.jchild[j] = b; H.childl[j] = -1;
        int rc = fill_joint_consts(d, j, H, j, err);
        if (rc != CCLQR_OK) return rc;
        for (int side = 0; side < 2; side++) {
            const int body = side ? a : b;
            if (body < 0) continue;
            if (H.inc_n[body] >= CCLQR_MAXI) { err = "more than 8 joints around one body"; return CCLQR_EUNSUPPORTED; }
            H.inc_j[body][H.inc_n[body]] = j; H.inc_side[body][H.inc_n[body]] = side; H.inc_n[body]++;
            seen[body] = 1;
        }
    }
    for (int b = 0; b < nb; b++)
        if (!seen[b]) { err = "a body without any joint"; return CCLQR_EINVAL; }
    return CCLQR_OK;
}

// fills m->host, m->nb, m->link_of_*; returns CCLQR_OK or an error code with a message in err
static inline int build_mech_tables(const cclqr_mech_desc* d, cclqr_mech* m, std::string& err) {
    const int nb = d->nb;
    if (nb < 1 || d->ne < nb) { err = "need ne >= nb >= 1 (every body hangs off at least one joint)"; return CCLQR_EINVAL; }
    if (!(d->dt > 0)) { err = "dt must be positive"; return CCLQR_EINVAL; }
    {   // closed loops: more joints than bodies, a body that is the child of two joints, or a FixedOrientation constraint
        bool loop = d->ne != nb;
        std::vector<int> cnt(nb > 0 ? nb : 1, 0);
        for (int j = 0; j < d->ne && !loop; j++) {
            const int b = d->child[j];
            if (b >= 0 && b < nb && ++cnt[b] > 1) loop = true;
            if (d->type[j] == CCLQR_FIXED_ORIENTATION) loop = true;
        }
        if (loop) return build_loop_tables(d, m, err);
    }
    m->nj = nb;
    if (nb > CCLQR_MAXL) { err = "more than 64 bodies"; return CCLQR_EUNSUPPORTED; }
    std::vector<int> pj(nb, -1), nchild(nb, 0);
    for (int j = 0; j < nb; j++) {
        int a = d->parent[j], b = d->child[j];
        if (b < 0 || b >= nb || a < -1 || a >= nb || a == b) { err = "joint references a body out of range"; return CCLQR_EINVAL; }
        if (pj[b] != -1) { err = "a body is the child of two joints (closed loop)"; return CCLQR_EUNSUPPORTED; }
        pj[b] = j;
        if (a >= 0) nchild[a]++;
        if (d->type[j] != CCLQR_REVOLUTE && d->type[j] != CCLQR_PRISMATIC) { err = "unknown joint type"; return CCLQR_EINVAL; }
    }
    bool tree = false;
    int npairs_total = 0;
    for (int b = 0; b < nb; b++) {
        if (nchild[b] > CCLQR_MAXK) { err = "a body with more than 4 child joints is not supported"; return CCLQR_EUNSUPPORTED; }
        if (nchild[b] > 1) { tree = true; npairs_total += nchild[b] * (nchild[b] - 1) / 2; }
    }
    if (npairs_total > CCLQR_MAXP) { err = "too many sibling joint pairs"; return CCLQR_EUNSUPPORTED; }
    // link order: depth first from the origin, children in the caller's joint order; the first child of a body continues its
    // chain (link l+1), further children start new chains.  A forest of chains (no body with two child joints) therefore keeps
    // the chain-contiguous numbering the two-front sweep relies on.
    std::vector<int> bfs, cstart, clen;
    {
        std::vector<int> stack;
        for (int j = nb - 1; j >= 0; j--)
            if (d->parent[j] == -1) stack.push_back(d->child[j]);
        int prev = -2;
        while (!stack.empty()) {
            int b = stack.back();
            stack.pop_back();
            if ((int)bfs.size() >= nb) { bfs.push_back(b); break; }
            const int par = d->parent[pj[b]];
            if (!(prev >= 0 && par == prev)) {   // does not continue the previous link: a new chain starts here
                if (!cstart.empty()) clen.push_back((int)bfs.size() - cstart.back());
                cstart.push_back((int)bfs.size());
            }
            bfs.push_back(b);
            prev = b;
            for (int k = nb - 1; k >= 0; k--)
                if (d->parent[k] == b) stack.push_back(d->child[k]);
        }
        if (!cstart.empty()) clen.push_back((int)bfs.size() - cstart.back());
    }
    if ((int)bfs.size() != nb) { err = "mechanism is not a tree rooted at the origin"; return CCLQR_EINVAL; }

    memset(&m->host, 0, sizeof(MechDev));
    m->nb = nb;
    MechDev& H = m->host;
    H.nb = nb; H.dt = d->dt; H.g = d->g;
    for (int l = 0; l < nb; l++) m->link_of_body[bfs[l]] = l;
    for (int l = 0; l < nb; l++) {
        int b = bfs[l], j = pj[b];
        m->link_of_joint[j] = l;
        H.perm[l] = b;
        H.jperm[l] = j;
        H.parent[l] = d->parent[j] >= 0 ? m->link_of_body[d->parent[j]] : -1;
        H.childl[l] = -1;
        H.type[l] = d->type[j];
        H.m[l] = d->mass[b];
        for (int i = 0; i < 9; i++) H.J[l][i] = d->inertia[9 * b + i];
        double ax[3] = {d->axis[3 * j], d->axis[3 * j + 1], d->axis[3 * j + 2]};
        double n = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
        if (n == 0.0 || !(H.m[l] > 0)) { err = "zero joint axis or non-positive mass"; return CCLQR_EINVAL; }
        for (int i = 0; i < 3; i++) { ax[i] /= n; H.axis[l][i] = ax[i]; H.p1[l][i] = d->p1[3 * j + i]; H.p2[l][i] = d->p2[3 * j + i]; }
        H.qoc[l][0] = d->qoff[4 * j];
        for (int i = 1; i < 4; i++) H.qoc[l][i] = -d->qoff[4 * j + i];
        double V12[6];
        orth_rows(ax, V12);
        const int nt = (d->type[j] == CCLQR_REVOLUTE) ? 3 : 2;   // Revolute = Translational3 + Rotational2, Prismatic = Translational2 + Rotational3
        H.rotmask[l] = 0;
        for (int r = 0; r < 5; r++) {
            bool rot = r >= nt;
            int q = rot ? r - nt : r, nrows = rot ? 5 - nt : nt;
            if (rot) H.rotmask[l] |= 1 << r;
            for (int i = 0; i < 3; i++) H.sel[l][r][i] = (nrows == 3) ? (i == q ? 1.0 : 0.0) : V12[3 * q + i];
        }
    }
    for (int l = 0; l < nb; l++)
        if (H.parent[l] >= 0) {
            const int a = H.parent[l];
            if (H.nchild[a] == 0) H.childl[a] = l;
            H.child[a][H.nchild[a]++] = l;
        }
    H.tree = tree ? 1 : 0;
    H.nchains = (int)cstart.size();
    H.start_mask = 0; H.end_mask = 0;
    for (int c = 0; c < H.nchains; c++) {
        H.chain_start[c] = cstart[c]; H.chain_len[c] = clen[c];
        H.start_mask |= 1ull << cstart[c];
        H.end_mask |= 1ull << (cstart[c] + clen[c] - 1);
    }
    for (int l = 0; l < nb; l++) {
        if (H.parent[l] >= l) { err = "internal: a link precedes its parent"; return CCLQR_EINVAL; }
        if (!tree && H.parent[l] != (((H.start_mask >> l) & 1ull) ? -1 : l - 1)) { err = "internal: link order is not chain-contiguous"; return CCLQR_EINVAL; }
    }
    if (tree) {
        // sibling pairs and the elimination program (links in reverse order; when l goes, what is left around its parent body
        // is the parent's own joint and the siblings with a smaller index)
        const Lay Y = make_layout(nb, 2 * npairs_total);
        auto pair_of = [&](int i, int j) { for (int q = 0; q < H.npairs; q++) if (H.pair_i[q] == i && H.pair_j[q] == j) return q; return -1; };
        for (int a = 0; a < nb; a++)
            for (int x = 0; x < H.nchild[a]; x++)
                for (int y = x + 1; y < H.nchild[a]; y++) { H.pair_i[H.npairs] = H.child[a][x]; H.pair_j[H.npairs] = H.child[a][y]; H.npairs++; }
        // offset of the block S_{x', x} among two neighbours of an eliminated link (both hang off the same body, or one is its joint)
        auto block = [&](int xr, int xc) {
            if (xr == xc) return Y.SJJ + 25 * xr;
            if (H.parent[xc] == xr) return Y.SPJ + 25 * xc;       // S_{parent, child}
            if (H.parent[xr] == xc) return Y.SJP + 25 * xr;       // S_{child, parent}
            const int q = xr < xc ? pair_of(xr, xc) : pair_of(xc, xr);
            return Y.SS + 25 * (2 * q + (xr < xc ? 0 : 1));        // siblings: S_ij (i < j) then S_ji
        };
        for (int l = 0; l < nb; l++) {
            int nn = 0;
            const int a = H.parent[l];
            if (a >= 0) {
                H.el_x[l][nn++] = a;
                for (int x = 0; x < H.nchild[a]; x++)
                    if (H.child[a][x] < l) H.el_x[l][nn++] = H.child[a][x];
            }
            if (nn > CCLQR_MAXK) { err = "internal: too many neighbours in the elimination"; return CCLQR_EINVAL; }
            H.el_nn[l] = nn;
            if (8 * nn > H.tree) H.tree = 8 * nn;
            for (int g = 0; g < nn; g++) {
                H.el_lx[l][g] = block(l, H.el_x[l][g]);
                H.el_xl[l][g] = block(H.el_x[l][g], l);
                for (int gp = 0; gp < nn; gp++) H.el_t[l][gp][g] = block(H.el_x[l][gp], H.el_x[l][g]);
            }
        }
    }
    return CCLQR_OK;
}

// host images of the controller tables in internal link order (K columns, setpoints, controlled links, friction)
struct CtrlHostTables {
    CtrlDev H;
    std::vector<double> K, zd, Fd;
};
static inline int build_ctrl_tables(const cclqr_mech* m, const cclqr_ctrl_desc* d, CtrlHostTables& T, std::string& err) {
    const int nb = m->nb, mx = 12 * nb;
    const int nj = m->nj;
    if (d->mu < 0 || d->mu > nj) { err = "Missmatched length for constraints"; return CCLQR_EINVAL; }
    if (d->nsp < 1 || !d->zd) { err = "Missmatched length for bodies"; return CCLQR_EINVAL; }
    if (d->K && d->nK < 1) { err = "gain table without entries"; return CCLQR_EINVAL; }
    CtrlDev& H = T.H;
    memset(&H, 0, sizeof(H));
    H.mu = d->mu; H.nK = d->K ? d->nK : 0; H.N = d->N; H.nsp = d->nsp; H.noise_scale = d->noise_scale;
    for (int i = 0; i < d->mu; i++) {
        int j = d->ctrl_joint[i];
        if (j < 0 || j >= nj) { err = "controlled joint out of range"; return CCLQR_EINVAL; }
        H.cj[i] = m->link_of_joint[j];
    }
    // closed loops: the friction / noise law of examples/trackingLQR_triple_cartpole.jl:93-111 and the PID law (pid.jl) are taken per joint in the
    // caller's joint order (a loop mechanism has more joints than bodies)
    if (d->fric)
        for (int j = 0; j < (m->host.loop ? nj : nb); j++) { H.fric[m->link_of_joint[j]] = d->fric[j]; if (d->fric[j] != 0.0) H.has_fric = 1; }
    H.noise_philox = d->noise_philox ? 1 : 0;
    H.noise_key0 = (unsigned)(d->noise_seed & 0xffffffffu) ^ (unsigned)(d->noise_seed >> 32);
    for (int i = 0; i < d->npid; i++) {
        int j = d->pid_joint ? d->pid_joint[i] : -1;
        if (j < 0 || j >= nj || !d->pid_P || !d->pid_I || !d->pid_D || !d->pid_goal) { err = "PID joint out of range"; return CCLQR_EINVAL; }
        int l = m->link_of_joint[j];
        H.pid_on[l] = 1; H.has_pid = 1;
        H.pid_P[l] = d->pid_P[i]; H.pid_I[l] = d->pid_I[i]; H.pid_D[l] = d->pid_D[i]; H.pid_goal[l] = d->pid_goal[i];
    }
    if (d->n_ctrl < 0) { err = "negative number of controller tables"; return CCLQR_EINVAL; }
    const size_t nc = d->n_ctrl > 1 ? (size_t)d->n_ctrl : 1;      // controller tables: one shared, or one per instance
    H.n_ctrl = (int)nc;
    T.zd.resize(nc * d->nsp * nb * 13);
    for (size_t s = 0; s < nc * d->nsp; s++)
        for (int l = 0; l < nb; l++) memcpy(&T.zd[(s * nb + l) * 13], d->zd + (s * nb + m->host.perm[l]) * 13, 13 * sizeof(double));
    if (d->K) {
        T.K.resize(nc * d->nK * d->mu * mx);
        for (size_t row = 0; row < nc * d->nK * d->mu; row++)
            for (int l = 0; l < nb; l++) memcpy(&T.K[row * mx + 12 * l], d->K + row * mx + 12 * m->host.perm[l], 12 * sizeof(double));
    }
    if (d->Fd && d->mu > 0) T.Fd.assign(d->Fd, d->Fd + nc * d->nsp * d->mu);
    if (nc > 1) {
        H.K_stride = (long long)d->nK * d->mu * mx; H.zd_stride = (long long)d->nsp * nb * 13; H.Fd_stride = (long long)d->nsp * d->mu;
    }
    return CCLQR_OK;
}

}  // namespace cclqr

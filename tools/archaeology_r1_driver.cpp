// round-1 archaeology driver: the committed round-1 (40d496b) phase functions through their own emulator with a NaN-poisoned, exact-size LDS image,
// under ASAN/UBSAN, at the three shapes whose GPU instantiations misbehaved in round 1's uncommitted variants
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "include/cclqr.h"
extern "C" int emu_rollout(const cclqr_mech_desc* md, const cclqr_ctrl_desc* cd, int64_t n_inst, int steps, int k0, const double* z0,
                           const double* noise, double* traj, double* zT, int* status, int G_override);
static int chain(int n, int steps) {
    int nb = n + 1;
    std::vector<double> mass(nb), J(9 * nb, 0.0), p1(3 * nb, 0.0), p2(3 * nb, 0.0), ax(3 * nb, 0.0), qo(4 * nb, 0.0);
    std::vector<int32_t> par(nb), ch(nb), ty(nb);
    for (int b = 0; b < nb; b++) {
        double x = 0.1, y = b ? 0.1 : 0.5, z = b ? 1.0 : 0.1, m = b ? 1.0 : 0.5;
        mass[b] = m; J[9 * b] = m / 12 * (y * y + z * z); J[9 * b + 4] = m / 12 * (x * x + z * z); J[9 * b + 8] = m / 12 * (x * x + y * y);
        par[b] = b - 1; ch[b] = b; ty[b] = b ? 0 : 1; qo[4 * b] = 1.0;
        if (b == 0) ax[1] = 1.0; else { ax[3 * b] = 1.0; p2[3 * b + 2] = -0.5; if (b > 1) p1[3 * b + 2] = 0.5; }
    }
    cclqr_mech_desc md = {nb, nb, 0.01, -9.81, mass.data(), J.data(), par.data(), ch.data(), ty.data(), p1.data(), p2.data(), ax.data(), qo.data()};
    std::vector<double> K((size_t)(steps + 2) * 12 * nb, 0.01), zd(13 * nb, 0.0), z0(13 * nb, 0.0), traj((size_t)steps * 13 * nb), zT(13 * nb);
    for (int b = 0; b < nb; b++) { zd[13 * b + 3] = 1; z0[13 * b + 3] = 1; if (b) { zd[13 * b + 2] = b - 0.5; z0[13 * b + 2] = b - 0.5; } }
    z0[1] = 0.2; for (int b = 1; b < nb; b++) z0[13 * b + 1] = 0.2;
    int32_t cj = 0;
    cclqr_ctrl_desc cd = {};
    cd.mu = 1; cd.ctrl_joint = &cj; cd.nK = steps + 2; cd.N = steps + 3; cd.K = K.data(); cd.nsp = 1; cd.zd = zd.data();
    int st = 0;
    int rc = emu_rollout(&md, &cd, 1, steps, 1, z0.data(), nullptr, traj.data(), zT.data(), &st, 0);
    bool fin = true; for (double v : zT) fin = fin && std::isfinite(v);
    for (double v : traj) fin = fin && std::isfinite(v);
    printf("chain %2d bodies: rc %d status %d finite %d zT[1] %.9f\n", nb, rc, st, (int)fin, zT[1]);
    return (rc == 0 && st > 0 && fin) ? 0 : 1;
}
static int dual(int steps) {     // dual-pole cart: cart (prismatic y) with two poles on it -> a branching tree, rollout_kernel<16, true> in round 1
    const int nb = 3;
    double mass[3] = {0.5, 1.0, 0.6}, J[27] = {0};
    double dims[3][3] = {{0.1, 0.5, 0.1}, {0.1, 0.1, 1.0}, {0.1, 0.1, 0.6}};
    for (int b = 0; b < 3; b++) { double x = dims[b][0], y = dims[b][1], z = dims[b][2], m = mass[b]; J[9 * b] = m / 12 * (y * y + z * z); J[9 * b + 4] = m / 12 * (x * x + z * z); J[9 * b + 8] = m / 12 * (x * x + y * y); }
    int32_t par[3] = {-1, 0, 0}, ch[3] = {0, 1, 2}, ty[3] = {1, 0, 0};
    double p1[9] = {0, 0, 0, 0, 0.1, 0, 0, -0.1, 0}, p2[9] = {0, 0, 0, 0, 0, -0.5, 0, 0, -0.3}, ax[9] = {0, 1, 0, 1, 0, 0, 1, 0, 0}, qo[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    cclqr_mech_desc md = {nb, nb, 0.01, -9.81, mass, J, par, ch, ty, p1, p2, ax, qo};
    std::vector<double> K((size_t)(steps + 2) * 12 * nb, 0.02), zd(13 * nb, 0.0), z0(13 * nb, 0.0), traj((size_t)steps * 13 * nb), zT(13 * nb);
    // poses: cart at y = 0.1; poles upright on their vertices, tilted by a small angle about x
    double ang[3] = {0, 0.05, -0.08};
    for (int b = 0; b < 3; b++) { zd[13 * b + 3] = 1; z0[13 * b + 3] = cos(ang[b] / 2); z0[13 * b + 4] = sin(ang[b] / 2); }
    z0[1] = 0.1;
    double len[3] = {0, 0.5, 0.3}, off[3] = {0, 0.1, -0.1};
    for (int b = 1; b < 3; b++) { zd[13 * b + 1] = off[b]; zd[13 * b + 2] = len[b]; z0[13 * b + 1] = 0.1 + off[b] - len[b] * sin(ang[b]); z0[13 * b + 2] = len[b] * cos(ang[b]); }
    int32_t cj = 0;
    cclqr_ctrl_desc cd = {};
    cd.mu = 1; cd.ctrl_joint = &cj; cd.nK = steps + 2; cd.N = steps + 3; cd.K = K.data(); cd.nsp = 1; cd.zd = zd.data();
    int st = 0;
    int rc = emu_rollout(&md, &cd, 1, steps, 1, z0.data(), nullptr, traj.data(), zT.data(), &st, 0);
    bool fin = true; for (double v : zT) fin = fin && std::isfinite(v);
    printf("dual-pole cart (tree): rc %d status %d finite %d zT[1] %.9f\n", rc, st, (int)fin, zT[1]);
    return (rc == 0 && st > 0 && fin) ? 0 : 1;
}
int main() {
    int bad = 0;
    bad += dual(40);          // rollout_kernel<16, true>
    bad += chain(7, 40);      // 8 bodies: rollout_kernel<32, false>
    bad += chain(16, 30);     // 17 bodies: rollout_kernel<64, false>
    bad += chain(1, 60);      // 2 bodies
    printf(bad ? "ARCHAEOLOGY_FAIL\n" : "ARCHAEOLOGY_OK\n");
    return bad;
}

// standalone driver: cartpole n_links (arg1), runs emu_chain_rollout under ASAN/UBSAN;  "loop <file> <steps>": a closed-loop mechanism
// (tables, pose and inputs written by the test as a flat array of doubles) through emu_loop_rollout
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <string>
#include "../../include/cclqr.h"
extern "C" int emu_chain_rollout(const cclqr_mech_desc* md, const cclqr_ctrl_desc* cd, int64_t n_inst, int steps, int k0, const double* z0,
                                 const double* noise, double* traj, double* zT, int* status, int G_override);
extern "C" int emu_loop_rollout(const cclqr_mech_desc* md, const cclqr_ctrl_desc* cd, int64_t n_inst, int steps, int k0, const double* z0,
                                double* lam, double* traj, double* zT, int* status);
static int run_loop(const char* path, int steps) {
    FILE* f = fopen(path, "rb");
    if (!f) return 2;
    std::vector<double> v;
    double x;
    while (fread(&x, sizeof(double), 1, f) == 1) v.push_back(x);
    fclose(f);
    size_t o = 0;
    const int nb = (int)v[o++], ne = (int)v[o++];
    const double dt = v[o++], g = v[o++];
    auto take = [&](size_t n) { std::vector<double> a(v.begin() + o, v.begin() + o + n); o += n; return a; };
    auto takei = [&](size_t n) { std::vector<int32_t> a(n); for (size_t i = 0; i < n; i++) a[i] = (int32_t)v[o + i]; o += n; return a; };
    std::vector<double> mass = take(nb), J = take(9 * nb);
    std::vector<int32_t> par = takei(ne), ch = takei(ne), ty = takei(ne);
    std::vector<double> p1 = take(3 * ne), p2 = take(3 * ne), ax = take(3 * ne), qo = take(4 * ne), z0 = take(13 * nb), Fd = take(2);
    cclqr_mech_desc md = {nb, ne, dt, g, mass.data(), J.data(), par.data(), ch.data(), ty.data(), p1.data(), p2.data(), ax.data(), qo.data()};
    std::vector<double> zd = z0, traj((size_t)steps * 13 * nb), zT(13 * nb), lam(5 * ne, 0.0), K(2 * 12 * nb, 0.01);
    int32_t cj[2] = {0, 1};
    cclqr_ctrl_desc cd = {};
    cd.mu = 2; cd.ctrl_joint = cj; cd.nK = 1; cd.N = 0; cd.K = K.data(); cd.nsp = 1; cd.zd = zd.data(); cd.Fd = Fd.data();
    int st = 0;
    int rc = emu_loop_rollout(&md, &cd, 1, steps, 1, z0.data(), lam.data(), traj.data(), zT.data(), &st);
    printf("rc %d status %d zT[1] %.6f\n", rc, st, zT[1]);
    return 0;
}
int main(int argc, char** argv) {
    if (argc > 3 && std::string(argv[1]) == "loop") return run_loop(argv[2], atoi(argv[3]));
    int n = argc > 1 ? atoi(argv[1]) : 1, nb = n + 1, steps = argc > 2 ? atoi(argv[2]) : 20;
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
    int rc = emu_chain_rollout(&md, &cd, 1, steps, 1, z0.data(), nullptr, traj.data(), zT.data(), &st, 0);
    printf("rc %d status %d zT[1] %.6f\n", rc, st, zT[1]);
    return 0;
}

/* c_abi_cartpole.c -- the C ABI of include/cclqr.h used from plain C, no Python: the cartpole of examples/lqr_cartpole.jl
 * (cart Box(0.1,0.5,0.1,0.5) on Prismatic(origin,cart,ey); pole Box(0.1,0.1,1,1) on Revolute(cart,pole,ex; p2=-[0,0,0.5]);
 * xd = [0,0,0], [0,0,0.5]; Q = I12 per body, R = 1, horizon 10 s, g = -9.81, dt = 0.01) through
 *   cclqr_mech_create -> cclqr_linearize -> cclqr_riccati -> cclqr_ctrl_create -> cclqr_rollout
 * for a small batch of start states.  Prints kbreak, |K|max and the final states (tests/test_gpu_setup.py compares them with the
 * Python mirror and the oracle).
 *   gcc -O2 -I include examples/c_abi_cartpole.c -o c_abi_cartpole -L constrainedcontrol.jl_amd -lcclqr -lm */
#include "cclqr.h"
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(call) do { int rc_ = (call); if (rc_ != CCLQR_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, cclqr_last_error()); return 1; } } while (0)

static void box_inertia(double x, double y, double z, double m, double *J) {
    memset(J, 0, 9 * sizeof(double));
    J[0] = m / 12.0 * (y * y + z * z); J[4] = m / 12.0 * (x * x + z * z); J[8] = m / 12.0 * (x * x + y * y);
}

/* what a foreign-language shim does at load time (julia/CCLQR.jl check_abi): the library's own view of the structs against the caller's */
static int check_abi(void) {
    int32_t lay[CCLQR_ABI_LAYOUT_LEN];
    if (cclqr_version() != CCLQR_ABI_VERSION) { fprintf(stderr, "ABI version %d, header %d\n", cclqr_version(), CCLQR_ABI_VERSION); return 1; }
    if (cclqr_abi_layout(lay, CCLQR_ABI_LAYOUT_LEN) != CCLQR_ABI_LAYOUT_LEN) return 1;
    const int32_t mine[] = {(int32_t)sizeof(cclqr_mech_desc), (int32_t)offsetof(cclqr_mech_desc, qoff), (int32_t)sizeof(cclqr_ctrl_desc), (int32_t)offsetof(cclqr_ctrl_desc, n_ctrl),
                            (int32_t)sizeof(cclqr_riccati_opts), (int32_t)sizeof(cclqr_rollout_opts), (int32_t)offsetof(cclqr_rollout_opts, flags)};
    const int at[] = {0, 13, 14, 33, 34, 39, 46};
    for (int i = 0; i < 7; i++)
        if (lay[at[i]] != mine[i]) { fprintf(stderr, "layout entry %d: library %d, caller %d\n", at[i], lay[at[i]], mine[i]); return 1; }
    printf("abi %d layout ok\n", cclqr_version());
    return 0;
}

int main(int argc, char **argv) {
    if (check_abi()) return 1;
    if (argc > 1 && !strcmp(argv[1], "--abi")) return 0;      /* (no GPU needed up to here) */
    const int n_inst = argc > 1 ? atoi(argv[1]) : 4, steps = 1000, nb = 2, mx = 24, ml = 10, mu = 1, N = 1000;
    const double dt = 0.01;
    double mass[2] = {0.5, 1.0}, inertia[18];
    box_inertia(0.1, 0.5, 0.1, 0.5, inertia);
    box_inertia(0.1, 0.1, 1.0, 1.0, inertia + 9);
    int32_t parent[2] = {-1, 0}, child[2] = {0, 1}, type[2] = {CCLQR_PRISMATIC, CCLQR_REVOLUTE};
    double p1[6] = {0, 0, 0, 0, 0, 0}, p2[6] = {0, 0, 0, 0, 0, -0.5}, axis[6] = {0, 1, 0, 1, 0, 0}, qoff[8] = {1, 0, 0, 0, 1, 0, 0, 0};
    cclqr_mech_desc md = {nb, nb, dt, -9.81, mass, inertia, parent, child, type, p1, p2, axis, qoff};
    cclqr_mech *mech = NULL;
    CHECK(cclqr_mech_create(&md, &mech));

    /* setpoint: cart at the origin, pole upright (COM at z = 0.5) */
    double zd[26];
    memset(zd, 0, sizeof zd);
    zd[3] = 1.0; zd[13 + 2] = 0.5; zd[13 + 3] = 1.0;
    int32_t ctrl_joint[1] = {0};
    double Fd[1] = {0.0};
    double *A = calloc(mx * mx, 8), *Bu = calloc(mx * mu, 8), *Bl = calloc(mx * ml, 8), *G = calloc(ml * mx, 8);
    CHECK(cclqr_linearize(mech, 1, zd, mu, ctrl_joint, Fd, A, Bu, Bl, G));
    double *Q = calloc(mx * mx, 8), R[1] = {dt};
    for (int i = 0; i < mx; i++) Q[i * mx + i] = dt;                    /* Q*dt, R*dt (lqr.jl:18-19) */
    double *K = calloc((size_t)(N - 1) * mu * mx, 8);
    int32_t kbreak = 0;
    CHECK(cclqr_riccati(1, mx, mu, ml, A, Bu, Bl, G, Q, R, N, 1e-5, K, &kbreak));
    double kmax = 0.0;
    for (int i = 0; i < (N - 1) * mx; i++) if (fabs(K[i]) > kmax) kmax = fabs(K[i]);
    printf("kbreak %d Kmax %.12e\n", kbreak, kmax);

    cclqr_ctrl_desc cd;
    memset(&cd, 0, sizeof cd);
    cd.mu = mu; cd.ctrl_joint = ctrl_joint; cd.nK = N - 1; cd.N = N; cd.K = K; cd.nsp = 1; cd.zd = zd; cd.Fd = Fd;
    cclqr_ctrl *ctrl = NULL;
    CHECK(cclqr_ctrl_create(mech, &cd, &ctrl));

    /* start states: cart at y0, pole tilted by phi about x (pole COM 0.5 above the cart along the pole's axis) */
    double *z0 = calloc((size_t)n_inst * 26, 8), *zT = calloc((size_t)n_inst * 26, 8);
    int32_t *status = calloc(n_inst, sizeof(int32_t));
    for (int n = 0; n < n_inst; n++) {
        const double y0 = -0.4 + 0.8 * n / (n_inst > 1 ? n_inst - 1 : 1), phi = 0.05 + 0.25 * n / (n_inst > 1 ? n_inst - 1 : 1);
        double *z = z0 + (size_t)n * 26;
        z[1] = y0; z[3] = 1.0;
        z[13 + 1] = y0 - 0.5 * sin(phi); z[13 + 2] = 0.5 * cos(phi); z[13 + 3] = cos(phi / 2); z[13 + 4] = sin(phi / 2);
    }
    CHECK(cclqr_rollout(mech, ctrl, n_inst, steps, 1, z0, NULL, NULL, zT, status));
    for (int n = 0; n < n_inst; n++) {
        printf("inst %d status %d zT", n, status[n]);
        for (int i = 0; i < 26; i++) printf(" %.15e", zT[(size_t)n * 26 + i]);
        printf("\n");
    }
    cclqr_ctrl_destroy(ctrl);
    cclqr_mech_destroy(mech);
    return 0;
}

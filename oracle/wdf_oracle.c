/*
 * wdf_oracle.c -- CPU ORACLE, TEST INFRASTRUCTURE ONLY (see wdf_oracle.h for the rules,
 * the reference citations and the parity-pin status).
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared -fPIC -> oracle/liboracle.so)
 */
#define _GNU_SOURCE
#include "wdf_oracle.h"

#include <complex.h>
#include <float.h>
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---- f64 ------------------------------------------------------------------------ */
#define REAL double
#define XREAL double
#define SUF(n) n##_f64
#define RE(x) ((double)(x))
#define IM(x) (0.0)
#define R_EXP exp
#define R_LOG log
#define R_TANH tanh
#define R_EPS DBL_EPSILON
#include "wdf_oracle_impl.inc"
#include "wdf_oracle_clipper.inc"
#undef REAL
#undef XREAL
#undef SUF
#undef RE
#undef IM
#undef R_EXP
#undef R_LOG
#undef R_TANH
#undef R_EPS

/* ---- f32 ------------------------------------------------------------------------ */
#define REAL float
#define XREAL float
#define SUF(n) n##_f32
#define RE(x) ((double)(x))
#define IM(x) (0.0)
#define R_EXP expf
#define R_LOG logf
#define R_TANH tanhf
#define R_EPS FLT_EPSILON
#include "wdf_oracle_impl.inc"
#include "wdf_oracle_clipper.inc"
#undef REAL
#undef XREAL
#undef SUF
#undef RE
#undef IM
#undef R_EXP
#undef R_LOG
#undef R_TANH
#undef R_EPS

/* ---- c64: complex-step derivative oracle ---------------------------------------- */
#define REAL double _Complex
#define XREAL double
#define SUF(n) n##_c64
#define RE(x) (creal(x))
#define IM(x) (cimag(x))
#define R_EXP cexp
#define R_LOG clog
#define R_TANH ctanh
#define R_EPS DBL_EPSILON
#include "wdf_oracle_impl.inc"
#undef REAL
#undef XREAL
#undef SUF
#undef RE
#undef IM
#undef R_EXP
#undef R_LOG
#undef R_TANH
#undef R_EPS

/* ---- exported wrappers ----------------------------------------------------------- */
double oracle_wright_omega_f64(double x) { int it; return omega_core_f64(x, &it); }
float  oracle_wright_omega_f32(float x)  { int it; return omega_core_f32(x, &it); }
double oracle_wright_omega_ext_f64(double x, int* n_iter) { return omega_core_f64(x, n_iter); }
float  oracle_wright_omega_ext_f32(float x, int* n_iter)  { return omega_core_f32(x, n_iter); }

void oracle_wright_omega_vec_f64(const double* x, double* w, int64_t n)
{
    int64_t i; int it;
    for (i = 0; i < n; ++i) w[i] = omega_core_f64(x[i], &it);
}
void oracle_wright_omega_vec_f32(const float* x, float* w, int64_t n)
{
    int64_t i; int it;
    for (i = 0; i < n; ++i) w[i] = omega_core_f32(x[i], &it);
}

double oracle_diode_pair_f64(double a, double R, double Is, double Vt, double nabla,
                             int n_up, int n_down)
{
    return diode_pair_core_f64(a, R, Is, Vt * nabla /* diode_pretraining.py:43 */, n_up, n_down, NULL);
}
float oracle_diode_pair_f32(float a, float R, float Is, float Vt, float nabla,
                            int n_up, int n_down)
{
    return diode_pair_core_f32(a, R, Is, Vt * nabla, n_up, n_down, NULL);
}

static int check_program(const orc_circuit* c, const orc_node* nd)
{
    int i;
    if (!c || !nd || c->n_nodes <= 0 || c->n_nodes > ORC_MAX_NODES) return -1;
    if (c->top < 0 || c->top >= c->n_nodes || c->probe < 0 || c->probe >= c->n_nodes) return -1;
    for (i = 0; i < c->n_nodes; ++i) {
        if (nd[i].c0 >= i || nd[i].c1 >= i) return -1;   /* post-order */
        if (nd[i].type < ORC_NODE_RESISTOR || nd[i].type > ORC_NODE_INVERTER) return -1;
    }
    if (c->root_kind == ORC_ROOT_MLP) {
        if (c->mlp_n_layers < 1 || c->mlp_n_layers > ORC_MAX_MLP_LAYERS) return -1;
        for (i = 0; i <= c->mlp_n_layers; ++i)
            if (c->mlp_sizes[i] < 1 || c->mlp_sizes[i] > 64) return -1;
        if (c->mlp_sizes[0] != 2 || c->mlp_sizes[c->mlp_n_layers] != 1) return -1;
    }
    return 0;
}

int oracle_tree_fwd_f64(const orc_circuit* c, const orc_node* nodes, const double* theta,
                        const double* x, double* y, const double* z0, double* zT,
                        int64_t B, int64_t T)
{
    int64_t b; int rc = check_program(c, nodes), err = 0;
    if (rc) return rc;
#pragma omp parallel for schedule(static)
    for (b = 0; b < B; ++b) {
        int e = tree_run_one_f64(c, nodes, theta, x + b * T * c->n_in, y + b, NULL, B,
                                 z0 ? z0 + b * c->n_nodes : NULL, zT ? zT + b * c->n_nodes : NULL, T);
        if (e) err = e;
    }
    return err;
}

int oracle_tree_fwd_f32(const orc_circuit* c, const orc_node* nodes, const float* theta,
                        const float* x, float* y, const float* z0, float* zT,
                        int64_t B, int64_t T)
{
    int64_t b; int rc = check_program(c, nodes), err = 0;
    if (rc) return rc;
#pragma omp parallel for schedule(static)
    for (b = 0; b < B; ++b) {
        int e = tree_run_one_f32(c, nodes, theta, x + b * T * c->n_in, y + b, NULL, B,
                                 z0 ? z0 + b * c->n_nodes : NULL, zT ? zT + b * c->n_nodes : NULL, T);
        if (e) err = e;
    }
    return err;
}

int oracle_tree_dtheta_c64(const orc_circuit* c, const orc_node* nodes, const double* theta,
                           int k, const double* x, double* y, double* dy,
                           int64_t B, int64_t T)
{
    enum { MAXP = 4096 };
    double _Complex th[MAXP];
    int64_t b; int i, n_theta = 0, rc = check_program(c, nodes), err = 0;
    double h;
    if (rc) return rc;
    /* number of theta entries = 1 + the highest index the program references */
    for (i = 0; i < c->n_nodes; ++i)
        if (nodes[i].param + 1 > n_theta) n_theta = nodes[i].param + 1;
    if (c->root_kind == ORC_ROOT_DIODE_PAIR) {
        if (c->p_is + 1 > n_theta) n_theta = c->p_is + 1;
        if (c->p_nvt + 1 > n_theta) n_theta = c->p_nvt + 1;
    }
    if (c->root_kind == ORC_ROOT_MLP) {
        int n = c->mlp_off;
        for (i = 0; i < c->mlp_n_layers; ++i) n += c->mlp_sizes[i] * c->mlp_sizes[i + 1] + c->mlp_sizes[i + 1];
        if (n > n_theta) n_theta = n;
    }
    if (n_theta > MAXP || k < 0 || k >= n_theta) return -3;
    for (i = 0; i < n_theta; ++i) th[i] = theta[i];
    h = 1e-30 * (fabs(theta[k]) > 0.0 ? fabs(theta[k]) : 1.0);
    th[k] = theta[k] + h * I;
#pragma omp parallel for schedule(static)
    for (b = 0; b < B; ++b) {
        int e = tree_run_one_c64(c, nodes, th, x + b * T * c->n_in, y + b, dy + b, B, NULL, NULL, T);
        if (e) err = e;
    }
    if (!err) {
        int64_t n = B * T, j;
        for (j = 0; j < n; ++j) dy[j] /= h;
    }
    return err;
}

int oracle_clipper_fwd_f64(const double* th, double fs, int n_up, int n_down, const double* x,
                           const double* r, double* y, int64_t B, int64_t T, int n_threads)
{ return clipper_run_f64(th, fs, n_up, n_down, x, r, NULL, NULL, y, NULL, NULL, 0, B, T, n_threads); }

int oracle_clipper_fwd_f32(const float* th, double fs, int n_up, int n_down, const float* x,
                           const float* r, float* y, int64_t B, int64_t T, int n_threads)
{ return clipper_run_f32(th, fs, n_up, n_down, x, r, NULL, NULL, y, NULL, NULL, 0, B, T, n_threads); }

int oracle_clipper_fwd_bwd_f64(const double* th, double fs, int n_up, int n_down, const double* x,
                               const double* r, const double* gy, double* y, double* g4,
                               int64_t B, int64_t T, int n_threads)
{ return clipper_run_f64(th, fs, n_up, n_down, x, r, gy, NULL, y, g4, NULL, 1, B, T, n_threads); }

int oracle_clipper_fwd_bwd_f32(const float* th, double fs, int n_up, int n_down, const float* x,
                               const float* r, const float* gy, float* y, double* g4,
                               int64_t B, int64_t T, int n_threads)
{ return clipper_run_f32(th, fs, n_up, n_down, x, r, gy, NULL, y, g4, NULL, 1, B, T, n_threads); }

int oracle_clipper_mse_step_f32(const float* th, double fs, int n_up, int n_down, const float* x,
                                const float* target, float* y, double* g4, double* loss,
                                int64_t B, int64_t T, int n_threads)
{ return clipper_run_f32(th, fs, n_up, n_down, x, NULL, NULL, target, y, g4, loss, 1, B, T, n_threads); }

int oracle_clipper_mse_step_f64(const double* th, double fs, int n_up, int n_down, const double* x,
                                const double* target, double* y, double* g4, double* loss,
                                int64_t B, int64_t T, int n_threads)
{ return clipper_run_f64(th, fs, n_up, n_down, x, NULL, NULL, target, y, g4, loss, 1, B, T, n_threads); }

static double asym_resid(double v, double a, double Rp, double Is1, double V1, double Is2, double V2)
{
    return v + Rp * (Is1 * expm1(v / V1) - Is2 * expm1(-v / V2)) - a;
}

double oracle_asym_root_f64(double a, double Rp, double Is1, double V1, double Is2, double V2)
{
    /* the residual is strictly increasing in v and the solution lies between 0 and a */
    double lo = a < 0.0 ? a : 0.0, hi = a < 0.0 ? 0.0 : a, v = 0.5 * (lo + hi);
    int it;
    for (it = 0; it < 200; ++it) {
        const double f = asym_resid(v, a, Rp, Is1, V1, Is2, V2);
        const double fp = 1.0 + Rp * (Is1 / V1 * exp(v / V1) + Is2 / V2 * exp(-v / V2));
        double vn;
        if (f > 0.0) hi = v; else lo = v;
        vn = v - f / fp;
        if (!(vn > lo && vn < hi)) vn = 0.5 * (lo + hi);        /* Newton left the bracket: bisect */
        if (fabs(vn - v) <= 1e-16 * (fabs(vn) + 1e-300) || hi - lo <= 0.0) { v = vn; break; }
        v = vn;
    }
    return v - Rp * (Is1 * expm1(v / V1) - Is2 * expm1(-v / V2));
}

int oracle_clipper_asym_fwd_f64(const double* th, double fs, const double* x, double* y, int64_t B, int64_t T)
{
    const double Is1 = th[0], V1 = th[1], Is2 = th[2], V2 = th[3], R = th[4], C = th[5];
    const double G1 = 1.0 / R, G2 = C * (2.0 * fs), G = G1 + G2, Rp = 1.0 / G, p = G1 / G;
    int64_t b;
#pragma omp parallel for schedule(static)
    for (b = 0; b < B; ++b) {
        double z = 0.0;
        int64_t t;
        for (t = 0; t < T; ++t) {
            const double b_diff = z - x[b * T + t];             /* tf_wdf.py:185-192 */
            const double b_temp = -p * b_diff;
            const double a = z + b_temp;
            const double br = oracle_asym_root_f64(a, Rp, Is1, V1, Is2, V2);
            const double zn = br + b_temp;                      /* tf_wdf.py:179-183 */
            y[t * B + b] = 0.5 * (zn + z);                      /* tf_wdf.py:8-10 */
            z = zn;
        }
    }
    return 0;
}

int oracle_max_threads(void) { return omp_get_max_threads(); }

/*
 * rt2d_oracle.c -- CPU restatement of the reference's 2D ray path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker for the HIP kernels: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product (ray-tracer_amd/) never does.
 *
 * It follows the reference (ahernot/ray-tracer) line by line, one ray at a time, in scalar fp64 with
 * -ffp-contract=off, reading the same piecewise-polynomial tables the device reads:
 *   orc_propagate  <- Ray2D.propagate main loop            ray.py:137-259 (closure fit_to_bounds ray.py:84-107)
 *   orc_post       <- Ray2D.propagate post-pass            ray.py:263-287
 *   orc_gain       <- Ray2D.populate_NEW                   ray.py:339-369, model_absorption.py:19-46,
 *                                                          model_reflection.py:11-46
 *   orc_heatmap_add<- commented Simulation2D.heatmap body  simulation.py:120-131 + functions.py:12-19
 * Deviations from the reference, all at the 1e-15 level and covered by the stated tolerances:
 *   - interp1d(kind='quadratic') (BSpline evaluation) is replaced by the equivalent power-basis pieces;
 *   - fsolve on the (linear) rebound equation ray.py:177,200 is replaced by its closed-form root;
 *   - np.power(c, 3) is c*c*c; pow(y, 0.5) is sqrt(y); complex division + abs is hypot/hypot.
 * Parity is pinned by tests/golden/ fixtures generated from the reference itself (oracle/gen_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t order;      /* 2 linear, 3 quadratic */
    int32_t n_pieces;
    int32_t oob;        /* 0 extrapolate, 1 clamp, 2 error */
    int32_t reserved;
    const double* breaks;
    const double* coef;
} OrcTable;

typedef struct {
    double range_min[2];
    double range_max[2];
    OrcTable c, dz_c, floor, ceil, dx_floor, dx_ceil, impedance;
    OrcTable absorb[6];
} OrcEnv;

typedef struct {
    double dx_max, dz_max;
    int32_t n_steps_max, n_rebounds_max;
    int32_t backprop, has_target;
    double target[2];
    int32_t rebound_cap, reserved;
    double power_cutoff, cutoff_freq;
} OrcParams;

/* ---- table lookup: binary search for the piece, Horner without contraction ---------------------- */
static int orc_piece(const OrcTable* t, double x) {
    int lo = 0, hi = t->n_pieces;           /* find last j with breaks[j] <= x, clipped to [0, n-1] */
    if (!(x >= t->breaks[0])) return 0;
    while (hi - lo > 1) {
        int mid = (lo + hi) / 2;
        if (t->breaks[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

static double orc_eval(const OrcTable* t, double x, int* err) {
    if (t->oob == 2 && !(x >= t->breaks[0] && x <= t->breaks[t->n_pieces])) { if (err) *err = 1; return NAN; }
    if (t->oob == 1) {
        if (x < t->breaks[0]) x = t->breaks[0];
        if (x > t->breaks[t->n_pieces]) x = t->breaks[t->n_pieces];
    }
    int j = orc_piece(t, x);
    double d = x - t->breaks[j];
    const double* c = t->coef + (size_t)j * t->order;
    if (t->order == 3) return (c[0] * d + c[1]) * d + c[2];
    return c[0] * d + c[1];
}

double orc_table_eval(const OrcTable* t, double x, int32_t* err) { int e = 0; double v = orc_eval(t, x, &e); if (err) *err = e; return v; }

/* ---- fit_to_bounds closure (ray.py:84-107); x_dir is the enclosing scope's current value --------- */
static int orc_fit(const OrcEnv* env, double x_prev, double* x_new, double z_prev, double* z_new,
                   double dx_z, double x_dir) {
    int out = 0;
    if (*x_new < env->range_min[0]) {
        *x_new = env->range_min[0];
        *z_new = -1 * dx_z * (*x_new - x_prev) + z_prev;
        out = 1;
    } else if (*x_new > env->range_max[0]) {
        *x_new = env->range_max[0];
        *z_new = dx_z * (*x_new - x_prev) + z_prev;
        out = 2;
    }
    if (*z_new < env->range_min[1]) {
        *z_new = env->range_min[1];
        *x_new = x_dir * (*z_new - z_prev) / dx_z + x_prev;
        out = 3;
    } else if (*z_new > env->range_max[1]) {
        *z_new = env->range_max[1];
        *x_new = x_dir * (*z_new - z_prev) / dx_z + x_prev;
        out = 4;
    }
    return out;
}

#define ORC_MARGIN(xx, zz) do { \
    margin = fmin(margin, fmin(fmin(fabs((xx) - env->range_min[0]), fabs((xx) - env->range_max[0])), \
                               fmin(fabs((zz) - env->range_min[1]), fabs((zz) - env->range_max[1])))); } while (0)

static double orc_sign(double v) { return (v > 0) ? 1.0 : ((v < 0) ? -1.0 : v); }   /* np.sign */

/*
 * One ray.  X/Z receive up to cap points (the caller sizes cap >= n_steps_max + 1).
 * init_slope / init_mult: if use_init != 0 they replace 1/tan(pi/2 - angle) and -(c0/sin(pi/2 - angle))^2.
 * Rebound log arrays hold rebound_cap entries.  Returns the per-ray status (0, -1, -2).
 */
int32_t orc_propagate(const OrcEnv* env, const double* src, double angle_in, const OrcParams* p,
                      int32_t use_init, double init_slope, double init_mult,
                      double* X, double* Z, int64_t cap,
                      int32_t* n_pts_out, int32_t* stop_out, int32_t* n_reb_out,
                      int32_t* reb_step, uint8_t* reb_surface, double* reb_velocity, double* reb_angle,
                      double* reb_px, double* reb_pz, double* margin_out) {
    int32_t status = 0, stop = 0, n_reb = 0;
    double margin = INFINITY;   /* smallest distance [m] by which any branch decision of the loop was taken */
    int64_t n_pts = 1;
    double x = src[0], z = src[1];
    X[0] = x; Z[0] = z;

    const double dx_z_max = fabs(p->dz_max / p->dx_max);                 /* ray.py:132 */
    double angle = -1 * angle_in + (M_PI / 2);                          /* ray.py:138 */
    double dx_z = use_init ? init_slope : 1 / tan(angle);                /* ray.py:139 */
    double x_dir = 1.;
    double kx = x_dir, kz = dx_z;
    double c = orc_eval(&env->c, 0.0, NULL);                             /* ray.py:145: c at the surface */
    double mult;
    if (use_init) mult = init_mult;
    else { double q = c / sin(angle); mult = -1 * (q * q); }             /* ray.py:146 */

    for (int32_t i = 0; i < p->n_steps_max; ++i) {
        /* step sizing, ray.py:153-154 */
        if (fabs(dx_z) > dx_z_max) { double s = p->dz_max / fabs(dx_z); kx *= s; kz *= s; }
        else { kx *= p->dx_max; kz *= p->dx_max; }
        double x_new = x + kx, z_new = z + kz;                           /* ray.py:157 */

        ORC_MARGIN(x_new, z_new);
        /* horizontal bounds first, ray.py:162-167 */
        { double xt = x_new, zt = z_new;
          int out = orc_fit(env, x, &xt, z, &zt, dx_z, x_dir);
          if (out == 1 || out == 2) { stop = out; break; } }

        double z_floor = orc_eval(&env->floor, x_new, NULL);             /* ray.py:172-173 */
        double z_ceil = orc_eval(&env->ceil, x_new, NULL);
        margin = fmin(margin, fmin(fabs(z_new - z_floor), fabs(z_new - z_ceil)));

        int rebound = 0; uint8_t surface = 0;
        if (z_new < z_floor) {                                           /* ray.py:176-181 */
            double fx = orc_eval(&env->floor, x, NULL);
            x_new = x + (fx - z) / dx_z;                                 /* root of floor(x) - dx_z*(x1-x) - z */
            z_new = z_floor;
            int e = 0; double s = orc_eval(&env->dx_floor, x_new, &e);
            if (e) { status = -1; break; }
            double u0 = 1., u1 = s, n0 = -1 * u1, n1 = u0;
            double ku = kx * u0 + kz * u1, kn = kx * n0 + kz * n1;
            double rx = ku * u0 - kn * n0, rz = ku * u1 - kn * n1;
            kx = rx; kz = rz;
            rebound = 1; surface = 0;
        } else if (z_new > z_ceil) {                                     /* ray.py:199-204 */
            double fx = orc_eval(&env->ceil, x, NULL);
            x_new = x + (fx - z) / dx_z;
            z_new = z_ceil;
            int e = 0; double s = orc_eval(&env->dx_ceil, x_new, &e);
            if (e) { status = -1; break; }
            double u0 = 1., u1 = s, n0 = u1, n1 = -1 * u0;
            double ku = kx * u0 + kz * u1, kn = kx * n0 + kz * n1;
            double rx = ku * u0 - kn * n0, rz = ku * u1 - kn * n1;
            kx = rx; kz = rz;
            rebound = 1; surface = 1;
        }
        if (rebound) {                                                   /* ray.py:184-195 / 207-219 */
            ORC_MARGIN(x_new, z_new);
            double r = kz / kx;
            double ang = 1 / sqrt(1 + r * r);
            if (n_reb < p->rebound_cap) {
                reb_step[n_reb] = i + 1; reb_surface[n_reb] = surface; reb_velocity[n_reb] = c;
                reb_angle[n_reb] = ang; reb_px[n_reb] = x; reb_pz[n_reb] = z;
            } else status = -2;
            n_reb += 1;
            if (p->n_rebounds_max > -1 && n_reb > p->n_rebounds_max) {
                orc_fit(env, x, &x_new, z, &z_new, dx_z, x_dir);
                if (n_pts < cap) { X[n_pts] = x_new; Z[n_pts] = z_new; }
                n_pts += 1;
                stop = 5; break;
            }
        }

        /* all bounds, ray.py:225-230 */
        { double xt = x_new, zt = z_new;
          int out = orc_fit(env, x, &xt, z, &zt, dx_z, x_dir);
          if (out) { stop = out; break; } }

        /* accept the point, ray.py:235-247 */
        if (n_pts < cap) { X[n_pts] = x_new; Z[n_pts] = z_new; }
        n_pts += 1;
        x = x_new; z = z_new;
        c = orc_eval(&env->c, z_new, NULL);
        double dz_c = orc_eval(&env->dz_c, z_new, NULL);
        x_dir = orc_sign(kx);
        dx_z = kz / kx;
        double dxdx_z = mult * dz_c / (c * c * c);
        dx_z += dxdx_z * kx;
        kx = x_dir; kz = dx_z;

        if (!p->backprop && x_dir < 0) { stop = 6; break; }              /* ray.py:252-255 */
        if (i == p->n_steps_max - 1) stop = 7;                            /* ray.py:257-259 */
    }
    *n_pts_out = (int32_t)n_pts; *stop_out = stop; *n_reb_out = n_reb;
    if (margin_out) *margin_out = margin;
    return status;
}

/* ---- post-pass (ray.py:263-287) --------------------------------------------------------------------- */
typedef struct { double x, z; int32_t idx; } OrcPt;
static int orc_cmp_pt(const void* a, const void* b) {
    const OrcPt* p = (const OrcPt*)a; const OrcPt* q = (const OrcPt*)b;
    if (p->x < q->x) return -1; if (p->x > q->x) return 1;
    return (p->idx < q->idx) ? -1 : (p->idx > q->idx);               /* stable, as interp1d's mergesort */
}

/* dL, L, C, dT, T: n_pts entries each (dL[0] = dT[0] = 0 by convention; reference arrays are one shorter).
 * range4 = {min x, min z, max x, max z}.  target may be NULL. */
void orc_post(const OrcEnv* env, const double* X, const double* Z, int32_t n_pts, const double* target,
              double* dL, double* L, double* C, double* dT, double* T, double* range4,
              int32_t* step_target, double* T_target, double* dist_to_target) {
    double mnx = X[0], mnz = Z[0], mxx = X[0], mxz = Z[0];
    for (int32_t s = 0; s < n_pts; ++s) {
        if (X[s] < mnx) mnx = X[s]; if (X[s] > mxx) mxx = X[s];
        if (Z[s] < mnz) mnz = Z[s]; if (Z[s] > mxz) mxz = Z[s];
        C[s] = orc_eval(&env->c, Z[s], NULL);                            /* ray.py:273 */
    }
    range4[0] = mnx; range4[1] = mnz; range4[2] = mxx; range4[3] = mxz;
    dL[0] = 0; L[0] = 0; dT[0] = 0; T[0] = 0;
    for (int32_t s = 1; s < n_pts; ++s) {
        double dx = X[s] - X[s - 1], dz = Z[s] - Z[s - 1];
        dL[s] = sqrt(dx * dx + dz * dz);                                 /* ray.py:271 */
        L[s] = L[s - 1] + dL[s];                                         /* ray.py:272 */
        dT[s] = dL[s] / C[s - 1];                                        /* ray.py:274 */
        T[s] = T[s - 1] + dT[s];                                         /* ray.py:275 */
    }
    if (!target) return;
    int32_t best = 0; double bd = fabs(X[0] - target[0]);                /* ray.py:278, first minimum */
    for (int32_t s = 1; s < n_pts; ++s) { double d = fabs(X[s] - target[0]); if (d < bd) { bd = d; best = s; } }
    *step_target = best; *T_target = T[best];
    /* calc_z = interp1d(X, Z, 'linear', bounds_error=True) on the x-sorted points (ray.py:282-287) */
    double tx = target[0];
    if (n_pts < 2 || !(tx >= mnx && tx <= mxx)) { *dist_to_target = NAN; return; }
    OrcPt* pts = (OrcPt*)malloc(sizeof(OrcPt) * (size_t)n_pts);
    for (int32_t s = 0; s < n_pts; ++s) { pts[s].x = X[s]; pts[s].z = Z[s]; pts[s].idx = s; }
    qsort(pts, (size_t)n_pts, sizeof(OrcPt), orc_cmp_pt);
    int32_t idx = 0; while (idx < n_pts && pts[idx].x < tx) ++idx;       /* searchsorted side='left' */
    if (idx < 1) idx = 1; if (idx > n_pts - 1) idx = n_pts - 1;
    double slope = (pts[idx].z - pts[idx - 1].z) / (pts[idx].x - pts[idx - 1].x);
    double zt = slope * (tx - pts[idx - 1].x) + pts[idx - 1].z;
    free(pts);
    *dist_to_target = fabs(target[1] - zt);
}

/* ---- gain (ray.py:339-369) ---------------------------------------------------------------------------- */
static const double ORC_A1 = 0.106, ORC_A3 = 0.00049;                   /* model_absorption.py:34-35 */
static const double ORC_WAVE_HEIGHT_RMS = 2., ORC_DZ_SEDIMENT = 2.;      /* preferences.py:11-12 */

double orc_absorption(const OrcEnv* env, double f_hz, double z) {        /* model_absorption.py:19-46 */
    double f1 = orc_eval(&env->absorb[0], z, NULL), f2 = orc_eval(&env->absorb[1], z, NULL);
    double A2 = orc_eval(&env->absorb[2], z, NULL), E1 = orc_eval(&env->absorb[3], z, NULL);
    double E2 = orc_eval(&env->absorb[4], z, NULL), E3 = orc_eval(&env->absorb[5], z, NULL);
    double f = f_hz / 1000;
    double ff = f * f;
    double alpha_1 = ORC_A1 * (f1 * ff) / (f1 * f1 + ff);
    double alpha_2 = A2 * (f2 * ff) / (f2 * f2 + ff);
    double alpha_3 = ORC_A3 * ff;
    return alpha_1 * E1 + alpha_2 * E2 + alpha_3 * E3;
}

double orc_refcoef_surface(double wavelength, double angle) {           /* model_reflection.py:11-18 */
    double a = 2 * M_PI / wavelength * ORC_WAVE_HEIGHT_RMS * sin(angle);
    return exp(-2 * (a * a));
}

double orc_refcoef_sediment(double wavelength, double Zp0, double Zp1, double Zp2) {   /* model_reflection.py:21-46 */
    double k1 = 2 * M_PI / wavelength;
    double r01 = (Zp1 - Zp0) / (Zp1 + Zp0);
    double r12 = (Zp2 - Zp1) / (Zp2 + Zp1);
    double th = -2 * k1 * ORC_DZ_SEDIMENT;
    double er = r12 * cos(th), ei = r12 * sin(th);
    double nr = r01 + er, ni = ei;
    double dr = 1 + r01 * er, di = r01 * ei;
    return hypot(nr, ni) / hypot(dr, di);
}

/* G: n_pts entries.  dL as produced by orc_post (dL[s] = segment ending at s). */
void orc_gain(const OrcEnv* env, const double* Z, const double* dL, int32_t n_pts, double freq,
              int32_t n_reb_logged, const int32_t* reb_step, const uint8_t* reb_surface,
              const double* reb_velocity, const double* reb_angle, const double* reb_pz,
              double Zp1, double Zp2, double* G) {
    G[0] = 0;
    for (int32_t s = 1; s < n_pts; ++s) {
        double dG = -1 * orc_absorption(env, freq, Z[s - 1]) / 1000 * dL[s];   /* ray.py:346 */
        G[s] = G[s - 1] + dG;                                                 /* ray.py:347 */
    }
    for (int32_t r = 0; r < n_reb_logged; ++r) {                              /* ray.py:350-362 */
        double wavelength = reb_velocity[r] / freq, refcoef;
        if (reb_surface[r] == 0) refcoef = orc_refcoef_sediment(wavelength, orc_eval(&env->impedance, reb_pz[r], NULL), Zp1, Zp2);
        else refcoef = orc_refcoef_surface(wavelength, reb_angle[r]);
        double add = 10 * log10(refcoef);
        for (int32_t s = reb_step[r] + 1; s < n_pts; ++s) G[s] += add;
    }
}

/* ---- heat map (simulation.py:120-131, functions.py:12-19) --------------------------------------------- */
static int64_t orc_floordiv(int64_t a, int64_t b) { int64_t q = a / b; if ((a % b != 0) && ((a < 0) != (b < 0))) --q; return q; }

/* Adds one ray to H (nx*nz, row-major).  Returns the number of points whose flat index is out of range
 * (the reference would raise IndexError); those points are skipped. */
int64_t orc_heatmap_add(const double* X, const double* Z, const double* G, int32_t n_pts,
                        const int32_t* res, const int32_t* shape, double* H) {
    int64_t nx = shape[0], nz = shape[1], size = nx * nz, dropped = 0;
    double* mask = (double*)calloc((size_t)size, sizeof(double));           /* functions.py:12-13 */
    for (int32_t s = 0; s < n_pts; ++s) {
        int64_t ix = orc_floordiv((int64_t)X[s], res[0]);                   /* simulation.py:123-125 */
        int64_t iz = orc_floordiv((int64_t)Z[s], -1 * (int64_t)res[1]);
        int64_t flat = ix * nz + iz;                                        /* functions.py:16 */
        if (flat < 0) flat += size;
        if (flat < 0 || flat >= size) { ++dropped; continue; }
        double tmult = pow(10, G[s] / 10);                                  /* ray.py:317 */
        mask[flat] = pow(tmult, -0.1);                                      /* simulation.py:129, last write wins */
    }
    for (int64_t c = 0; c < size; ++c) H[c] += mask[c];                     /* simulation.py:131 */
    free(mask);
    return dropped;
}

/* simulation.py:133-138: out is (nz, nx) = heatmap_plot.T */
void orc_heatmap_finalize(const double* H, const int32_t* shape, double cutoff, double reduction_power, double* out) {
    int64_t nx = shape[0], nz = shape[1];
    double mx = H[0];
    for (int64_t c = 1; c < nx * nz; ++c) if (H[c] > mx || isnan(H[c])) mx = H[c];   /* np.max propagates NaN */
    for (int64_t ix = 0; ix < nx; ++ix) for (int64_t iz = 0; iz < nz; ++iz) {
        double v = H[ix * nz + iz] / mx;
        if (v > cutoff) v = cutoff;
        if (reduction_power != 1) v = pow(v, reduction_power);
        out[iz * nx + ix] = v;
    }
}

/* ---- whole-fan driver for the CPU baseline (bench.py cpu_baseline): cast + post + gain(freq) + heat map,
 * one ray after the other as Simulation2D.cast / heatmap do (simulation.py:89-94, 120-131).  The heat map
 * shape comes from the caller (a first call with H = NULL returns the plot range in range4).
 * Returns the number of ray-steps (appended points). */
int64_t orc_run_fan(const OrcEnv* env, const double* src, const double* angles, int64_t n, const OrcParams* p,
                    double freq, double Zp1, double Zp2, const int32_t* res, const int32_t* shape, double* H,
                    double* range4, int64_t* n_errors) {
    const int64_t cap = (int64_t)p->n_steps_max + 2;
    double* X = (double*)malloc(sizeof(double) * cap), *Z = (double*)malloc(sizeof(double) * cap);
    double* dL = (double*)malloc(sizeof(double) * cap), *L = (double*)malloc(sizeof(double) * cap);
    double* C = (double*)malloc(sizeof(double) * cap), *dT = (double*)malloc(sizeof(double) * cap);
    double* T = (double*)malloc(sizeof(double) * cap), *G = (double*)malloc(sizeof(double) * cap);
    const int rc = p->rebound_cap > 0 ? p->rebound_cap : 1;
    int32_t* rs = (int32_t*)malloc(sizeof(int32_t) * rc); uint8_t* rsurf = (uint8_t*)malloc(rc);
    double* rv = (double*)malloc(sizeof(double) * rc), *ra = (double*)malloc(sizeof(double) * rc);
    double* rpx = (double*)malloc(sizeof(double) * rc), *rpz = (double*)malloc(sizeof(double) * rc);
    int64_t steps = 0, errors = 0;
    double mn[2] = {0, 0}, mx[2] = {0, 0};                                /* ray.py:441-442 */
    for (int64_t r = 0; r < n; ++r) {
        int32_t n_pts, stop, n_reb; double margin, rng[4];
        int32_t st = orc_propagate(env, src, angles[r], p, 0, 0, 0, X, Z, cap, &n_pts, &stop, &n_reb,
                                   rs, rsurf, rv, ra, rpx, rpz, &margin);
        if (st == -1) { ++errors; continue; }
        int32_t stt; double Tt, dd;
        orc_post(env, X, Z, n_pts, p->has_target ? p->target : NULL, dL, L, C, dT, T, rng, &stt, &Tt, &dd);
        steps += n_pts - 1;
        if (rng[0] < mn[0]) mn[0] = rng[0]; if (rng[1] < mn[1]) mn[1] = rng[1];
        if (rng[2] > mx[0]) mx[0] = rng[2]; if (rng[3] > mx[1]) mx[1] = rng[3];
        if (H) {
            int nl = n_reb < rc ? n_reb : rc;
            orc_gain(env, Z, dL, n_pts, freq, nl, rs, rsurf, rv, ra, rpz, Zp1, Zp2, G);
            orc_heatmap_add(X, Z, G, n_pts, res, shape, H);
        }
    }
    if (range4) { range4[0] = mn[0]; range4[1] = mn[1]; range4[2] = mx[0]; range4[3] = mx[1]; }
    if (n_errors) *n_errors = errors;
    free(X); free(Z); free(dL); free(L); free(C); free(dT); free(T); free(G);
    free(rs); free(rsurf); free(rv); free(ra); free(rpx); free(rpz);
    return steps;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Checker for an identity the HIP stepping kernel relies on (rt2d_div_by_const, csrc/rt2d_kernels.hip): for b > 0 and
 * y = RN(1/b), two FMA correction steps on q0 = a*y give the IEEE quotient a/b bit for bit.  Returns the number of
 * samples (n random a spread over 2^-200 .. 2^200, both signs) on which the two differ: must be 0. */
int64_t orc_check_div_by_const(double b, int64_t n, uint64_t seed) {
    const double y = 1.0 / b;
    uint64_t s = seed ? seed : 88172645463325252ull;
    int64_t bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const uint64_t m = s;
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const int e = (int)(s % 400) - 200;
        double a = ldexp((double)(m >> 11) * 0x1p-53 + 0.5, e);
        if (m & 1) a = -a;
        const double q0 = a * y;
        const double r0 = fma(-q0, b, a);
        const double q1 = fma(r0, y, q0);
        const double r1 = fma(-q1, b, a);
        const double q2 = fma(r1, y, q1);
        bad += (q2 != a / b);
    }
    return bad;
}

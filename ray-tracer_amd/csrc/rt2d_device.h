// rt2d_device.h -- device-side environment tables and their lookup (gfx950).
//
// All fields of the environment live in ONE array of doubles (the "blob"), written once by
// rt2d_env_create.  The tables the stepping loop reads (c/dc-dz, floor, ceiling, both slope tables)
// come first so a workgroup can stage exactly that prefix into LDS; the gain tables (impedance,
// absorption factors) follow.  A table is piecewise polynomial in power basis about the left
// break of each piece and is evaluated by Horner WITHOUT contraction, the same expression as
// PPTable.__call__ on the host (ray-tracer_amd/pptable.py) and orc_eval in the oracle.
//
// Storage: one RECORD per piece, [break_j, coefficients of piece j ...], `stride` doubles long, followed by
// a last record that holds only break_n.  A lookup guesses the piece from the uniform-grid hint and fetches
// the whole record plus the next break in ONE batch of independent LDS reads; the guess is then verified
// against the two breaks (so the piece is exactly the one a binary search finds) and only a wrong guess
// (ends of the grid, non-uniform tables) takes the corrective loop.  This keeps a lookup at one LDS round
// trip on the critical path of a ray step instead of three or four dependent ones.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT2D_WAVE 64

struct DevTable {
    int32_t off;          // offset (in doubles) of record 0 inside the blob (or inside its section)
    int32_t n_pieces;
    int32_t stride;       // doubles per record = 1 + coefficients per piece
    int32_t oob;          // RT2D_OOB_*
    int32_t uniform;      // 1: break_j ~ x0 + j*h for 1 <= j < n  ->  O(1) guess, verified against the breaks
    int32_t pad;
    double  x0;
    double  inv_h;
    double  lo, hi;       // break_0, break_n
};

struct DevEnv {
    double range_min[2];
    double range_max[2];
    double c_surface;         // c(0), ray.py:145
    DevTable cz;              // fused: {c: q,l,k ; dz_c: q,l,k}, stride 7
    DevTable floor, ceil;     // linear, stride 3
    DevTable dx_floor, dx_ceil;  // quadratic, stride 4
    DevTable imp;             // quadratic, stride 4
    DevTable absorb;          // fused 6 quadratics, stride 19
    int32_t step_doubles;     // length of the blob prefix needed by the stepping kernels
    int32_t cz_doubles;       // length of the prefix that holds only the c/dc-dz table
    int32_t gain_off;         // start of the gain section (impedance + absorption); its offsets are relative to it
    int32_t gain_doubles;     // its length
    int32_t total_doubles;
    int32_t pad;
};

// first guess of the piece: exact for interior pieces of a uniform grid, anything in [0, n-1] otherwise
__device__ __forceinline__ int rt2d_guess(const DevTable& t, double x) {
    double g = (x - t.x0) * t.inv_h;
    g = fmin(fmax(g, 0.0), (double)(t.n_pieces - 1));     // NaN -> 0 through fmax
    return (int)g;
}

// Piece index: last j with break_j <= x, clipped to [0, n-1]; starts from j and walks.
template <typename TabPtr>
__device__ __forceinline__ int rt2d_fix_piece(TabPtr tab, const DevTable& t, double x, int j) {
    const int n = t.n_pieces;
    if (t.uniform) {
        while (j > 0 && x < tab[t.off + j * t.stride]) --j;
        while (j < n - 1 && x >= tab[t.off + (j + 1) * t.stride]) ++j;
    } else {
        int lo = 0, hi = n;
        if (!(x >= tab[t.off])) return 0;
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (tab[t.off + mid * t.stride] <= x) lo = mid; else hi = mid; }
        j = lo;
    }
    return j;
}

// Fetches record j (STRIDE doubles) and the next break into r[0..STRIDE] for the piece that contains x.
template <int STRIDE, typename TabPtr>
__device__ __forceinline__ void rt2d_lookup(TabPtr tab, const DevTable& t, double x, double (&r)[STRIDE + 1]) {
    int j = rt2d_guess(t, x);
    {
        auto p = tab + t.off + j * STRIDE;
#pragma unroll
        for (int k = 0; k <= STRIDE; ++k) r[k] = p[k];
    }
    // bitwise on purpose: with && / || the compiler sinks the loads of r[1..] behind a branch on r[0], a second LDS
    // round trip on the critical path
    const bool ok = ((x >= r[0]) | (j == 0)) & ((x < r[STRIDE]) | (j == t.n_pieces - 1));
    if (!ok || !t.uniform) {
        j = rt2d_fix_piece(tab, t, x, j);
        auto p = tab + t.off + j * STRIDE;
#pragma unroll
        for (int k = 0; k <= STRIDE; ++k) r[k] = p[k];
    }
}

// Applies the clamp policy; returns false when the table's policy is ERROR and x is outside it.
__device__ __forceinline__ bool rt2d_domain(const DevTable& t, double& x) {
    if (t.oob == 2) return (x >= t.lo) && (x <= t.hi);
    if (t.oob == 1) { if (x < t.lo) x = t.lo; if (x > t.hi) x = t.hi; }
    return true;
}

template <typename TabPtr>
__device__ __forceinline__ double rt2d_eval_lin(TabPtr tab, const DevTable& t, double x) {
    rt2d_domain(t, x);
    double r[4];
    rt2d_lookup<3>(tab, t, x, r);
    double d = x - r[0];
    return r[1] * d + r[2];
}

// Two linear tables at the same abscissa (floor and surface under a ray point): both records are fetched
// before either guess is verified, so the two lookups share one LDS round trip.
template <typename TabPtr>
__device__ __forceinline__ void rt2d_eval_lin2(TabPtr tab, const DevTable& ta, const DevTable& tb, double x,
                                               double& ya, double& yb) {
    double xa = x, xb = x;
    rt2d_domain(ta, xa);
    rt2d_domain(tb, xb);
    int ja = rt2d_guess(ta, xa), jb = rt2d_guess(tb, xb);
    double ra[4], rb[4];
    {
        auto pa = tab + ta.off + ja * 3;
        auto pb = tab + tb.off + jb * 3;
#pragma unroll
        for (int k = 0; k < 4; ++k) { ra[k] = pa[k]; rb[k] = pb[k]; }
    }
    const bool oka = ((xa >= ra[0]) | (ja == 0)) & ((xa < ra[3]) | (ja == ta.n_pieces - 1)) & (ta.uniform != 0);
    const bool okb = ((xb >= rb[0]) | (jb == 0)) & ((xb < rb[3]) | (jb == tb.n_pieces - 1)) & (tb.uniform != 0);
    if (!(oka && okb)) {
        ja = rt2d_fix_piece(tab, ta, xa, ja);
        jb = rt2d_fix_piece(tab, tb, xb, jb);
        auto pa = tab + ta.off + ja * 3;
        auto pb = tab + tb.off + jb * 3;
#pragma unroll
        for (int k = 0; k < 4; ++k) { ra[k] = pa[k]; rb[k] = pb[k]; }
    }
    double da = xa - ra[0], db = xb - rb[0];
    ya = ra[1] * da + ra[2];
    yb = rb[1] * db + rb[2];
}

// quadratic k of a fused record r (record layout: break, then 3 coefficients per fused table)
__device__ __forceinline__ double rt2d_quad_of(const double* r, double d, int k) {
    return (r[1 + 3 * k] * d + r[2 + 3 * k]) * d + r[3 + 3 * k];
}

template <typename TabPtr>
__device__ __forceinline__ double rt2d_eval_quad(TabPtr tab, const DevTable& t, double x, bool* ok) {
    bool in = rt2d_domain(t, x);
    if (ok) *ok = in;
    if (!in) return __builtin_nan("");
    double r[5];
    rt2d_lookup<4>(tab, t, x, r);
    return rt2d_quad_of(r, x - r[0], 0);
}

// c(z) and dc/dz(z) from the fused sound-speed table
template <typename TabPtr>
__device__ __forceinline__ void rt2d_eval_cz(TabPtr tab, const DevTable& t, double z, double& c, double& dz_c) {
    double r[8];
    rt2d_lookup<7>(tab, t, z, r);
    double d = z - r[0];
    c = rt2d_quad_of(r, d, 0);
    dz_c = rt2d_quad_of(r, d, 1);
}

template <typename TabPtr>
__device__ __forceinline__ double rt2d_eval_c(TabPtr tab, const DevTable& t, double z) {
    double c, dz_c;
    rt2d_eval_cz(tab, t, z, c, dz_c);
    return c;
}

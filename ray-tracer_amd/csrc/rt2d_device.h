// rt2d_device.h -- device-side environment tables and their lookup (gfx950).
//
// All fields of the environment live in ONE array of doubles (the "blob"), written once by
// rt2d_env_create.  The tables the stepping loop reads (c/dc-dz, floor, ceiling, both slope tables)
// come first so a workgroup can stage exactly that prefix into LDS; the gain tables (impedance,
// absorption factors) follow.  A table is piecewise polynomial in power basis about the left
// break of each piece and is evaluated by Horner WITHOUT contraction, the same expression as
// PPTable.__call__ on the host (ray-tracer_amd/pptable.py) and orc_eval in the oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT2D_WAVE 64

struct DevTable {
    int32_t off_breaks;   // offset (in doubles) of breaks[n_pieces+1] inside the blob
    int32_t off_coef;     // offset of coef[n_pieces*stride]
    int32_t n_pieces;
    int32_t stride;       // doubles per piece (order, or the sum of orders of fused tables)
    int32_t oob;          // RT2D_OOB_*
    int32_t uniform;      // 1: breaks[j] ~ x0 + j*h for 1 <= j < n  ->  O(1) guess, verified against breaks
    double  x0;
    double  inv_h;
    double  lo, hi;       // breaks[0], breaks[n_pieces]
};

struct DevEnv {
    double range_min[2];
    double range_max[2];
    double c_surface;         // c(0), ray.py:145
    DevTable cz;              // fused: coef = {c: q,l,k ; dz_c: q,l,k}, stride 6
    DevTable floor, ceil;     // linear, stride 2
    DevTable dx_floor, dx_ceil;  // quadratic, stride 3
    DevTable imp;             // quadratic, stride 3
    DevTable absorb;          // fused 6 quadratics, stride 18
    int32_t step_doubles;     // length of the blob prefix needed by the stepping kernels
    int32_t gain_off;         // start of the gain section (impedance + absorption)
    int32_t gain_doubles;     // its length
    int32_t total_doubles;
};

// Piece index: last j with breaks[j] <= x, clipped to [0, n-1].  TabPtr is an LDS or a global pointer.
template <typename TabPtr>
__device__ __forceinline__ int rt2d_piece(TabPtr tab, const DevTable& t, double x) {
    const int n = t.n_pieces;
    auto br = tab + t.off_breaks;
    int j;
    if (t.uniform) {
        double g = (x - t.x0) * t.inv_h;
        g = fmin(fmax(g, 0.0), (double)(n - 1));    // NaN -> 0 through fmax
        j = (int)g;
        while (j > 0 && x < br[j]) --j;
        while (j < n - 1 && x >= br[j + 1]) ++j;
    } else {
        int lo = 0, hi = n;
        if (!(x >= br[0])) return 0;
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (br[mid] <= x) lo = mid; else hi = mid; }
        j = lo;
    }
    return j;
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
    int j = rt2d_piece(tab, t, x);
    double d = x - tab[t.off_breaks + j];
    auto c = tab + t.off_coef + j * t.stride;
    return c[0] * d + c[1];
}

// quadratic at coefficient slot k (k-th fused table) of piece j
template <typename TabPtr>
__device__ __forceinline__ double rt2d_quad_at(TabPtr tab, const DevTable& t, int j, double d, int k) {
    auto c = tab + t.off_coef + j * t.stride + 3 * k;
    return (c[0] * d + c[1]) * d + c[2];
}

template <typename TabPtr>
__device__ __forceinline__ double rt2d_eval_quad(TabPtr tab, const DevTable& t, double x, bool* ok) {
    bool in = rt2d_domain(t, x);
    if (ok) *ok = in;
    if (!in) return __builtin_nan("");
    int j = rt2d_piece(tab, t, x);
    double d = x - tab[t.off_breaks + j];
    return rt2d_quad_at(tab, t, j, d, 0);
}

// rt2d_kernels.hip -- hand-written gfx950 kernels of the 2D acoustic ray path + their C ABI (include/rt2d.h).
//
//   rt2d_cast_kernel      one lane = one ray; the Ray2D.propagate loop (reference ray.py:137-259) with the
//                         scalar post-pass (ray.py:263-287) folded in; environment tables staged in LDS;
//                         trajectories stored step-major per 64-ray tile (one 512-B line per wave per step).
//   rt2d_post_kernel      per-point dL, L, C, dT, T (ray.py:271-275).
//   rt2d_gain_kernel      Ray2D.populate_NEW for F frequencies (ray.py:339-369), 1/2/4/8 frequencies per lane.
//   rt2d_dist_kernel      distance to target from stored trajectories (ray.py:282-287).
//   rt2d_heatmap_kernel   Simulation2D.heatmap accumulation (simulation.py:120-131, functions.py:12-19):
//                         one wave per ray, last-write-wins per cell through an LDS hash, f64 atomics into H.
//   rt2d_hm_* kernels     normalise / clip / power / transpose (simulation.py:133-138).
//
// Compiled with -ffp-contract=off: every expression is evaluated exactly as written, which is what makes the
// trajectories comparable bit for bit with the CPU oracle (oracle/rt2d_oracle.c).
#include <hip/hip_runtime.h>
#include <type_traits>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rt2d.h"
#include "rt2d_device.h"

// ------------------------------------------------------------------------------------------------------
// error plumbing
static thread_local std::string g_err;
static int32_t fail(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
static int32_t fail(const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return -1;
}
#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail("%s: %s", #expr, hipGetErrorString(e_)); } while (0)

struct rt2d_env_s {
    DevEnv dev;
    double* blob;       // device
    int device;
    int n_cu;           // compute units of that device
    bool ceil_is_const; // the surface table is one constant (slopes 0, equal values) ...
    double ceil_const;  // ... this one
    std::vector<double> host_blob;
    // per frequency: -alpha(f, z) / 1000 as ONE polynomial per depth piece (alpha_table_for), built on first use
    struct AlphaTab { double* dev = nullptr; int32_t n_pieces = 0; bool ok = false; double max_rel_err = 0.0; };
    std::map<uint64_t, AlphaTab> alpha;
    std::mutex alpha_mu;
};

// ------------------------------------------------------------------------------------------------------
// heat-map cell arithmetic and deposit (shared by the fused stepping kernel and the heat-map kernels)
__device__ __forceinline__ int64_t rt2d_floordiv(int64_t a, int64_t b) {
    int64_t q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
    return q;
}

// Cell indices of a point: ix = int(x) // res_x, iz = int(z) // (-res_z) (simulation.py:123-125: C truncation, then
// Python floor division).  64-bit integer division costs ~150 instructions on this hardware and there are two per
// point, so coordinates below 2^31 take a reciprocal multiply instead: for integers |t| < 2^31 and 1 <= b < 2^20,
// floor((t + 0.5) * RN(1/b)) == t // b exactly -- (t + 0.5) / b is at least 0.5 / b away from every integer, and the
// two roundings move the product by less than 2^-21 / b (tests/test_host.py::test_half_offset_floor_division).
struct HeatGeom { int32_t res_x, res_z; double rx, rz, inv_rx, inv_rz, lim; };     // lim: |coordinate| bound of the fast division

__device__ __forceinline__ double rt2d_fdiv_pos(double t, double inv_b) { return floor((t + 0.5) * inv_b); }     // t // b

__device__ __forceinline__ void rt2d_cell_xz(const HeatGeom& g, double x, double z, int64_t& ix, int64_t& iz) {
    if (fabs(x) < g.lim && fabs(z) < g.lim) {
        ix = (int64_t)(int32_t)rt2d_fdiv_pos(trunc(x), g.inv_rx);
        iz = (int64_t)(int32_t)rt2d_fdiv_pos(-trunc(z), g.inv_rz);          // a // (-b) == (-a) // b
    } else {
        ix = rt2d_floordiv((int64_t)x, g.res_x);
        iz = rt2d_floordiv((int64_t)z, -1 * (int64_t)g.res_z);
    }
}

// The same cell in 32-bit arithmetic, for callers that treat coordinates beyond 2^31 as an error (returns false there;
// their geometry has res < 2^20)
__device__ __forceinline__ bool rt2d_cell_i32(const HeatGeom& g, double x, double z, int& ix, int& iz) {
    ix = (int)rt2d_fdiv_pos(trunc(x), g.inv_rx);
    iz = (int)rt2d_fdiv_pos(-trunc(z), g.inv_rz);
    return fabs(x) < 2147483648.0 && fabs(z) < 2147483648.0;
}

// flat cell index of a point, or -1 when it falls outside the map (functions.py:16; one negative wrap as NumPy does)
__device__ __forceinline__ int64_t rt2d_cell(const HeatGeom& g, double x, double z, int64_t nz, int64_t size) {
    int64_t ix, iz;
    rt2d_cell_xz(g, x, z, ix, iz);
    int64_t flat = ix * nz + iz;
    if (flat < 0) flat += size;
    if (flat < 0 || flat >= size) return -1;
    return flat;
}

// vals = (10**(G/10))**-0.1 (ray.py:317, simulation.py:129).  While 10**(G/10) is a normal double this is 10**(-G/100)
// to ~1e-15 relative (the heat-map tolerance is 1e-12) and one exp2 replaces two pows; outside that range the two
// pows are evaluated as written so that underflow to 0 -> inf behaves as in the reference.
// 2^t for |t| < 1000: t = n + f with n = rint(t), |f| <= 1/2 (the subtraction is exact); 2^f by the Taylor series of
// e^(f ln 2) to degree 13 (truncation 3e-18 relative at |f| = 1/2) in FMA Horner form, then the exponent is added.
// About 1e-16 relative, 19 instructions; the library exp2 spends ~110 on special cases this caller never meets.
__device__ __forceinline__ double rt2d_exp2(double t) {
    const double n = rint(t);
    const double f = t - n;
    double p = 0x1.816193166d0f9p-40;
    p = fma(p, f, 0x1.c3bd650fc2986p-36);
    p = fma(p, f, 0x1.e8cac7351bb25p-32);
    p = fma(p, f, 0x1.e4cf5158b8ecap-28);
    p = fma(p, f, 0x1.b5253d395e7c4p-24);
    p = fma(p, f, 0x1.62c0223a5c824p-20);
    p = fma(p, f, 0x1.ffcbfc588b0c7p-17);
    p = fma(p, f, 0x1.430912f86c787p-13);
    p = fma(p, f, 0x1.5d87fe78a6731p-10);
    p = fma(p, f, 0x1.3b2ab6fba4e77p-7);
    p = fma(p, f, 0x1.c6b08d704a0c0p-5);
    p = fma(p, f, 0x1.ebfbdff82c58fp-3);
    p = fma(p, f, 0x1.62e42fefa39efp-1);
    p = fma(p, f, 1.0);
    return ldexp(p, (int)n);
}

// The same polynomial with every Horner step written as a three-address v_fma_f64: inside a loop the compiler keeps the
// thirteen coefficients in vector registers (no fp64 literals in VOP3 on gfx9) and then forms each step as a copy of the
// coefficient plus a two-address v_fmac_f64 -- twelve copies per evaluation.  Same operations, same roundings.
__device__ __forceinline__ double rt2d_fma3(double a, double b, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ double rt2d_exp2_3addr(double t) {
    const double n = rint(t);
    const double f = t - n;
    double p = 0x1.816193166d0f9p-40;
    p = rt2d_fma3(p, f, 0x1.c3bd650fc2986p-36);
    p = rt2d_fma3(p, f, 0x1.e8cac7351bb25p-32);
    p = rt2d_fma3(p, f, 0x1.e4cf5158b8ecap-28);
    p = rt2d_fma3(p, f, 0x1.b5253d395e7c4p-24);
    p = rt2d_fma3(p, f, 0x1.62c0223a5c824p-20);
    p = rt2d_fma3(p, f, 0x1.ffcbfc588b0c7p-17);
    p = rt2d_fma3(p, f, 0x1.430912f86c787p-13);
    p = rt2d_fma3(p, f, 0x1.5d87fe78a6731p-10);
    p = rt2d_fma3(p, f, 0x1.3b2ab6fba4e77p-7);
    p = rt2d_fma3(p, f, 0x1.c6b08d704a0c0p-5);
    p = rt2d_fma3(p, f, 0x1.ebfbdff82c58fp-3);
    p = rt2d_fma3(p, f, 0x1.62e42fefa39efp-1);
    p = fma(p, f, 1.0);
    return ldexp(p, (int)n);
}

__device__ __forceinline__ bool rt2d_heat_val_is_plain(double g) { return g > -3000.0 && g < 3000.0; }
__device__ __forceinline__ double rt2d_heat_val_plain(double g) { return rt2d_exp2(g * -0.033219280948873623); }   // -log2(10)/100
__device__ __forceinline__ double rt2d_heat_val_pows(double g) {
    double tmult = pow(10.0, g / 10);
    return pow(tmult, -0.1);
}
// Far outside that range no pow is needed either: 10^(G/10) has overflowed to inf for G > 3090 (value inf^-0.1 = 0) and
// underflowed to 0 -- below half the smallest denormal, 10^-323.6 -- for G < -3240 (value 0^-0.1 = inf), infinities
// included.  This is the common case at high frequency: the surface coefficient exp(-2 (k h sin a)^2) is 1e-300 or 0
// from the first surface rebound on, and every later point of the ray carries G of minus thousands of dB or -inf
// (ray.py:350-362).  Only the bands in between take the two pows.
__device__ __forceinline__ bool rt2d_heat_val_is_inf(double g) { return g < -3240.0 || g > 3090.0; }
__device__ __forceinline__ double rt2d_heat_val_of_inf(double g) { return (g < 0.0) ? INFINITY : 0.0; }
__device__ __forceinline__ double rt2d_heat_val(double g) {
    if (rt2d_heat_val_is_plain(g)) return rt2d_heat_val_plain(g);
    if (rt2d_heat_val_is_inf(g)) return rt2d_heat_val_of_inf(g);
    return rt2d_heat_val_pows(g);
}


// Wave votes straight from the compare's lane mask (the library forms go through a 0/1 register and a second compare)
__device__ __forceinline__ bool rt2d_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ __forceinline__ bool rt2d_all(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0ull; }

// Adds val into H[cell] for every lane that calls with want = true.  Neighbouring lanes are neighbouring angles and
// mostly sit in the same cell at the same step: runs of adjacent depositing lanes with equal cells are summed first
// (segmented inclusive scan) and the last lane of each run issues ONE atomic.  Callable from divergent code: only the
// lanes that call take part, and nothing is read from a lane outside the caller's own run.
// value of `v` in the lane CTRL points at (DPP: row_shr:n = 0x110 + n inside rows of 16 lanes, row_bcast:15 = 0x142 and
// row_bcast:31 = 0x143 across rows); lanes without a source read 0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double rt2d_dpp(double v) {
    // mov_dpp with bound_ctrl: lanes without a source read 0 and there is no previous value to preserve (a row that
    // ROW_MASK leaves out gets an unspecified value: the callers below never use it there)
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(unsigned)b, CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(unsigned)(b >> 32), CTRL, ROW_MASK, 0xf, true);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

// The run structure of one deposit (which adjacent depositing lanes share a cell) is computed once and can carry several
// values: the maps of several frequencies bin the same points into the same cells.
struct DepositRuns {
    bool want, tail;
    int dist, reach, head_pos, lane;
    __device__ __forceinline__ bool init(bool want_, int cell, int lane_) {
        want = want_; lane = lane_;
        const unsigned long long act = __ballot(want);
        if (act == 0) return false;
        const int c_prev = __builtin_amdgcn_update_dpp(0, cell, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        const bool prev_in = lane > 0 && ((act >> (lane - 1)) & 1ull);
        const bool head = want && !(prev_in && c_prev == cell);
        const unsigned long long heads = __ballot(head);
        head_pos = 63 - __clzll((long long)(heads & (~0ull >> (63 - lane))));    // head of this lane's run
        dist = lane - head_pos;
        reach = want ? min(lane & 15, dist) : -1;
        tail = want && (lane == RT2D_WAVE - 1 || !((act >> (lane + 1)) & 1ull) || ((heads >> (lane + 1)) & 1ull));
        return true;
    }
    // Segmented inclusive scan, all in the vector ALU: four shifts inside each row of 16 lanes, then the two row
    // broadcasts that carry a run across rows.  A lane only adds a value that comes from its own run: `reach` is how
    // far below itself a lane may look inside its row, -1 for lanes that do not deposit.  The last lane of a run holds
    // the run's sum.
    __device__ __forceinline__ double sum(double val) const {
        double v = want ? val : 0.0;
        { const double u = rt2d_dpp<0x111, 0xf>(v); v += (reach >= 1) ? u : 0.0; }
        { const double u = rt2d_dpp<0x112, 0xf>(v); v += (reach >= 2) ? u : 0.0; }
        { const double u = rt2d_dpp<0x114, 0xf>(v); v += (reach >= 4) ? u : 0.0; }
        { const double u = rt2d_dpp<0x118, 0xf>(v); v += (reach >= 8) ? u : 0.0; }
        { const double u = rt2d_dpp<0x142, 0xa>(v);                                  // rows 1 and 3 <- lane 15 / 47
          v += (want && (lane & 16) && dist > (lane & 15)) ? u : 0.0; }
        { const double u = rt2d_dpp<0x143, 0xc>(v);                                  // rows 2 and 3 <- lane 31
          v += (want && lane >= 32 && head_pos <= 31) ? u : 0.0; }
        return v;
    }
};

__device__ __forceinline__ void rt2d_deposit(double* H, bool want, int cell, double val, int lane) {
    DepositRuns runs;
    if (!runs.init(want, cell, lane)) return;
    const double v = runs.sum(val);
    if (runs.tail) atomicAdd(&H[cell], v);
}

// The same for callers whose wave runs with ALL 64 lanes enabled (lanes without a deposit pass want = false): when every
// depositing lane holds the same cell -- neighbouring rays of a sorted fan, most steps -- the run bookkeeping drops
// out and the wave's sum is one unconditional scan (18 instructions instead of ~55); lane 63 issues the atomic.
__device__ __forceinline__ void rt2d_deposit_full(double* H, bool want, int cell, double val, int lane) {
    const unsigned long long wm = __builtin_amdgcn_ballot_w64(want);
    if (wm == 0ull) return;
    const int c0 = __builtin_amdgcn_readlane(cell, (int)__builtin_ctzll(wm));
    if (__builtin_amdgcn_ballot_w64(want & (cell == c0)) == wm) {
        double v = want ? val : 0.0;
        v += rt2d_dpp<0x111, 0xf>(v);
        v += rt2d_dpp<0x112, 0xf>(v);
        v += rt2d_dpp<0x114, 0xf>(v);
        v += rt2d_dpp<0x118, 0xf>(v);                 // lane 15 of each row: the row's sum
        v += rt2d_dpp<0x142, 0xa>(v);                 // lanes 31 and 63: two rows (rows 0 and 2 hold nothing of use from here on)
        v += rt2d_dpp<0x143, 0xc>(v);                 // lane 63: the wave
        if (lane == RT2D_WAVE - 1) atomicAdd(&H[c0], v);
        return;
    }
    rt2d_deposit(H, want, cell, val, lane);
}

// "Last write wins" for a ray whose heat-map column never decreases and whose points all fall inside the map, fed
// point by point in path order (functions.py:16 semantics without storing the ray).  Such a ray can only come back to a
// cell while it is still in the same column, so the memory it needs is local to that column: 8 slots per lane
// (direct-mapped by row & 7, in LDS: slot k of this lane at slot[k * 64]).  The value of a cell is deposited the moment
// the ray LEAVES the cell -- every lane of a wave leaves at most one cell per step, so one run-summed deposit per step
// serves the whole wave -- and the slot of the row remembers what was deposited; if the ray comes back to the row inside
// the same column and leaves it again, only the difference to the remembered value is deposited, which leaves the last
// visit's value in the map (to one rounding).  A ray that breaks a precondition (column decreases, point outside the
// map, more than 8 rows inside one column) stops being binned and is flagged.
struct BinTarget { double* H; HeatGeom geo; int32_t nx, nz; double sign; int32_t dbg; };

struct RayBinner {
    int cur_ix, cur_iz, col_lo, col_hi;
    unsigned occ;                 // rows (by slot) of the current column whose deposited value is remembered
    double cur_g;                 // gain of the latest point in the current cell
    bool ok, violated;
    double* slot;

    // deposits the cell this lane is leaving (want); stay: the ray stays in the column, so the row must be remembered
    // FULL: the caller's wave has all 64 lanes enabled (rt2d_deposit_full)
    template <bool FULL = false>
    __device__ __forceinline__ void leave(const BinTarget& t, bool want, bool stay, int lane) {
        if (t.dbg & 2) return;
        double v = 0.0, prev = 0.0;
        const int k = cur_iz & 7;
        if (want) {
            v = t.sign * ((t.dbg & 4) ? cur_g : rt2d_heat_val(cur_g));      // ray.py:317, simulation.py:129
            if (occ & (1u << k)) prev = slot[k * RT2D_WAVE];
            if (stay) { slot[k * RT2D_WAVE] = v; occ |= 1u << k; }
        }
        const int cell = cur_ix * t.nz + cur_iz;
        // never the difference of two infinities: a gain of -inf (a surface coefficient that underflowed, ray.py:350-362)
        // gives v = prev = +inf on a row the ray comes back to, and the cell already holds the +inf the reference puts there
        const double d = (v == prev) ? 0.0 : v - prev;
        if (t.dbg & 1) { if (want) atomicAdd(&t.H[cell], d); }
        else if (FULL) rt2d_deposit_full(t.H, want, cell, d, lane);
        else rt2d_deposit(t.H, want, cell, d, lane);
    }
    // first point of the ray (gain g)
    __device__ __forceinline__ void start(const BinTarget& t, bool enabled, double* slot_, double x, double z, double g) {
        slot = slot_; occ = 0u; cur_g = g; cur_ix = INT32_MIN; cur_iz = col_lo = col_hi = 0; ok = enabled; violated = false;
        if (!enabled) return;
        int ix, iz;
        if (rt2d_cell_i32(t.geo, x, z, ix, iz) && (unsigned)ix < (unsigned)t.nx && (unsigned)iz < (unsigned)t.nz) {
            cur_ix = ix; cur_iz = col_lo = col_hi = iz;
        } else { ok = false; violated = true; }
    }
    // next point of the ray; callable from divergent code (the deposit only involves the calling lanes)
    // (live = false: the lane has no point this time and only takes part in the wave's deposit)
    template <bool FULL = false>
    __device__ __forceinline__ void point(const BinTarget& t, double x, double z, double g, int lane, bool live = true) {
        int ix, iz;
        const bool in_map = rt2d_cell_i32(t.geo, x, z, ix, iz) && (unsigned)ix < (unsigned)t.nx && (unsigned)iz < (unsigned)t.nz;
        if (FULL && t.dbg == 0) {
            // The common step of a sorted fan: every lane that bins a point enters a column further on and remembers no row
            // of the column it leaves (dx_max = the cell width: a new column every step).  Then nothing of the slot and row
            // bookkeeping below applies: the cell left is deposited as it is and the new cell becomes the current one.
            const bool binning = live & ok;
            if (rt2d_all(!binning | (in_map & (ix > cur_ix) & (occ == 0u)))) {
                double v = 0.0;
                if (binning) v = t.sign * rt2d_heat_val(cur_g);                  // ray.py:317, simulation.py:129
                rt2d_deposit_full(t.H, binning, cur_ix * t.nz + cur_iz, v, lane);
                if (binning) { cur_ix = ix; cur_iz = iz; col_lo = iz; col_hi = iz; cur_g = g; }
                return;
            }
        }
        if (live && ok && (!in_map || ix < cur_ix)) { ok = false; violated = true; }
        const bool newcol = live && ok && ix != cur_ix;
        const bool newrow = live && ok && !newcol && iz != cur_iz;
        leave<FULL>(t, newcol | newrow, newrow, lane);
        if (newcol) { cur_ix = ix; cur_iz = iz; col_lo = iz; col_hi = iz; occ = 0u; }
        else if (newrow) {
            cur_iz = iz; col_lo = min(col_lo, iz); col_hi = max(col_hi, iz);
            if (col_hi - col_lo >= 8) { ok = false; violated = true; }      // two rows would share a slot
        }
        if (live && ok) cur_g = g;
    }
    // after the last point: the cell the ray ends in
    template <bool FULL = false>
    __device__ __forceinline__ void finish(const BinTarget& t, int lane) { leave<FULL>(t, ok, false, lane); }
};

// Points of ray r that its tile has room for: every kernel that READS trajectories counts a ray's points through this, so
// that a cast into a provisional layout whose room a ray outgrew (its n_pts counts on, its points beyond the room were
// not stored, see rt2d_cast_write) can never make a later kernel read outside the tile -- the host finds out from
// rt2d_fan_summary and casts again, possibly after such kernels were already queued.
__device__ __forceinline__ int rt2d_stored_pts(const int32_t* n_pts, const int64_t* tile_off, int64_t r) {
    const int64_t t = r >> 6;
    const int64_t room = (tile_off[t + 1] - tile_off[t]) >> 6;
    const int np = n_pts[r];
    return (int)((int64_t)np < room ? (int64_t)np : room);
}

// ------------------------------------------------------------------------------------------------------
// stepping
struct CastArgs {
    DevEnv env;
    const double* blob;
    const double* angles;
    const double* init_slope;
    const double* init_mult;
    int64_t n;
    double src_x, src_z;
    Rt2dParams p;
    Rt2dRayOut out;
    const int64_t* tile_off;
    double* X; double* Z; double* C;
    double g_cut;                 // power cut-off in dB, 10*log10(power_cutoff); only read by the CUTOFF kernels
    double inv_dx_max;            // RN(1 / dx_max), for rt2d_div_by_const
    double ceil_const;            // the surface depth when the surface table is constant (CEILC kernels)
    int32_t lanes;                // rays per wave (1..64, power of two): small fans are spread over more waves
    int32_t tight_off;            // experiments (RT2D_CAST_TIGHT=0): 1 = general loop only; 2 = no tight loop with the whole post-pass
    unsigned long long* wave_log; // development aid (rt2d_debug_wave_log): per wave [start, end] of the 100 MHz clock and its hardware id
    int64_t wave_log_cap;
    // fused cast -> gain -> heat map (FUSE kernels)
    const uint8_t* ray_mask;      // n bytes, 0 = leave this ray out
    BinTarget bin;
    double heat_freq;
    uint8_t* irregular;           // n bytes (may be NULL): 1 = the ray broke a precondition of the fused binning
    unsigned long long* n_violations;
    int32_t fuse_step_doubles;    // blob prefix staged in LDS: sound speed, floor, surface (slopes are read from global)
    int32_t fuse_abs_doubles;     // length of the absorption table staged behind it
    const double* fuse_loss;      // FUSE == 2: rebound losses of a previous cast of the same rays, [k * n + ray]
    int32_t fuse_loss_cap;        // ... k < fuse_loss_cap
};                                // bin.dbg: experiments (RT2D_FUSE_DBG): 1 = no cross-lane aggregation, 2 = no deposit, 4 = no exp2

// RN(a / b) for a divisor b > 0 whose correctly rounded reciprocal y = RN(1/b) is known (Markstein's correction
// steps, each one FMA): q1 is a faithful quotient, and q1 + (a - b*q1)*y then rounds to the IEEE quotient, so the
// result is the one `a / b` gives, bit for bit, at 5 instructions instead of the 13 of the general fp64 division.
// Valid away from overflow / underflow of the residuals (callers check |a|); tests/test_oracle.py::test_div_by_const
// checks the identity on the host, the cast parity tests check it on the device.
__device__ __forceinline__ double rt2d_div_by_const(double a, double b, double y) {
    double q0 = a * y;
    double r0 = fma(-q0, b, a);
    double q1 = fma(r0, y, q0);
    double r1 = fma(-q1, b, a);
    return fma(r1, y, q1);
}


// a / 1000 of the gain integrand (ray.py:346): five FMAs instead of the 13-instruction division (with its quarter-rate
// reciprocal) in every (point, frequency) of the gain loops; outside the range the identity is tested on, divide.
__device__ __forceinline__ double rt2d_div_1000(double a) {
    const double m = fabs(a);
    if (__builtin_expect(m > 0x1p-200 && m < 0x1p200, 1)) return rt2d_div_by_const(a, 1000.0, 0.001);
    return a / 1000;
}

__device__ __forceinline__ double rt2d_sign(double v) { return (v > 0.0) ? 1.0 : ((v < 0.0) ? -1.0 : v); }

// fit_to_bounds closure (ray.py:84-107)
__device__ __forceinline__ int rt2d_fit(const DevEnv& e, double x_prev, double& x_new, double z_prev, double& z_new,
                                        double dx_z, double x_dir) {
    int out = 0;
    if (x_new < e.range_min[0]) {
        x_new = e.range_min[0];
        z_new = -1 * dx_z * (x_new - x_prev) + z_prev;
        out = 1;
    } else if (x_new > e.range_max[0]) {
        x_new = e.range_max[0];
        z_new = dx_z * (x_new - x_prev) + z_prev;
        out = 2;
    }
    if (z_new < e.range_min[1]) {
        z_new = e.range_min[1];
        x_new = x_dir * (z_new - z_prev) / dx_z + x_prev;
        out = 3;
    } else if (z_new > e.range_max[1]) {
        z_new = e.range_max[1];
        x_new = x_dir * (z_new - z_prev) / dx_z + x_prev;
        out = 4;
    }
    return out;
}

__device__ __forceinline__ double rt2d_bound_margin(const DevEnv& e, double x, double z) {
    return fmin(fmin(fabs(x - e.range_min[0]), fabs(x - e.range_max[0])),
                fmin(fabs(z - e.range_min[1]), fabs(z - e.range_max[1])));
}

// Online form of interp1d(X, Z, 'linear') evaluated at the target abscissa (ray.py:282-287): the bracketing
// points of the x-sorted (stable) trajectory are tracked while the points are produced.
struct TargetTrack {
    double lo_x, lo_z, hi_x, hi_z;      // largest x < tx (latest wins ties), smallest x >= tx (earliest wins ties)
    double m1x, m1z, m2x, m2z;          // the two first points of the stable x-sort (for tx == min x)
    double best_d, best_T; int best_step; int n_lt;
    __device__ __forceinline__ void init() {
        lo_x = -INFINITY; hi_x = INFINITY; lo_z = hi_z = 0.0; m1x = m2x = INFINITY; m1z = m2z = 0.0;
        best_d = INFINITY; best_T = 0.0; best_step = 0; n_lt = 0;
    }
    __device__ __forceinline__ void add(double x, double z, double T, int s, double tx) {
        if (x < tx) { ++n_lt; if (x >= lo_x) { lo_x = x; lo_z = z; } }
        else if (x < hi_x) { hi_x = x; hi_z = z; }
        if (x < m1x) { m2x = m1x; m2z = m1z; m1x = x; m1z = z; }
        else if (x < m2x) { m2x = x; m2z = z; }
        double d = fabs(x - tx);
        if (d < best_d) { best_d = d; best_T = T; best_step = s; }      // np.argmin: first minimum (ray.py:278)
    }
    __device__ __forceinline__ double dist(double tx, double tz, int n_pts, double min_x, double max_x) const {
        if (n_pts < 2 || !(tx >= min_x && tx <= max_x)) return __builtin_nan("");
        double x_lo, z_lo, x_hi, z_hi;
        if (n_lt == 0) { x_lo = m1x; z_lo = m1z; x_hi = m2x; z_hi = m2z; }
        else { x_lo = lo_x; z_lo = lo_z; x_hi = hi_x; z_hi = hi_z; }
        double slope = (z_hi - z_lo) / (x_hi - x_lo);
        double zt = slope * (tx - x_lo) + z_lo;
        return fabs(tz - zt);
    }
};

// ------------------------------------------------------------------------------------------------------
// physics closed forms shared by the gain kernel and the in-cast running gain (ray.py:339-369)
static __constant__ double k_A1 = 0.106, k_A3 = 0.00049;                  // model_absorption.py:34-35
#define RT2D_WAVE_HEIGHT_RMS 2.0                                          // preferences.py:11
#define RT2D_DZ_SEDIMENT 2.0                                              // preferences.py:12
#define RT2D_ZP1 2688000.0                                                // mean of 5 sediments, profile_impedance.py:17-18
#define RT2D_ZP2 17000000.0                                               // rock, profile_impedance.py:19

__device__ __forceinline__ double rt2d_refcoef_surface(double wavelength, double angle) {   // model_reflection.py:11-18
    double v = 2 * M_PI / wavelength * RT2D_WAVE_HEIGHT_RMS * sin(angle);
    return exp(-2 * (v * v));
}

__device__ __forceinline__ double rt2d_refcoef_sediment(double wavelength, double Zp0) {     // model_reflection.py:21-46
    double k1 = 2 * M_PI / wavelength;
    double r01 = (RT2D_ZP1 - Zp0) / (RT2D_ZP1 + Zp0);
    double r12 = (RT2D_ZP2 - RT2D_ZP1) / (RT2D_ZP2 + RT2D_ZP1);
    double th = -2 * k1 * RT2D_DZ_SEDIMENT;
    double sn, cs;
    sincos(th, &sn, &cs);
    double er = r12 * cs, ei = r12 * sn;
    double nr = r01 + er, ni = ei;
    double dr = 1 + r01 * er, di = r01 * ei;
    return hypot(nr, ni) / hypot(dr, di);
}

// alpha(f, z) in dB/km from the six depth factors of one fused absorption record (model_absorption.py:37-44)
// n / d, the IEEE quotient bit for bit for operands and quotients far from the ends of the exponent range: the compiler's
// own sequence (v_rcp_f64, two Newton steps, the product, one residual correction) WITHOUT v_div_scale_f64 /
// v_div_fmas_f64 / v_div_fixup_f64 around it, which leave such operands alone -- they rescale operands whose exponents
// would make an intermediate overflow or go denormal, and patch up zeros, infinities and NaNs.  Three instructions and the
// VCC dependence between them fewer per division; the callers guarantee the range (table values, squared frequencies,
// sound speeds; the stepping kernel's tight loop votes on |k1| >= 1e-250 before it divides).  rt2d_debug_div_check counts
// the operand pairs on which the two differ (tests/test_gpu_cast.py: none in 2^28 pairs over up to 200 decades, both signs,
// numerators +0 included).  The one difference there is: a numerator of MINUS zero over a positive divisor gives +0 here
// (the residual correction adds +0 to it), -0 by IEEE -- the sign of a zero that the callers either cannot produce
// (positive numerators, voted |k1|) or add to a slope, where it changes nothing but the sign of a zero slope.
__device__ __forceinline__ double rt2d_div_fast(double n, double d) {
    double y = __builtin_amdgcn_rcp(d);
    double e = fma(-d, y, 1.0); y = fma(y, e, y);
    e = fma(-d, y, 1.0); y = fma(y, e, y);
    const double q = n * y;
    return fma(fma(-d, q, n), y, q);
}

// sqrt(a) likewise: the compiler's sequence (v_rsq_f64, one coupled Newton step for root and half reciprocal root, two
// residual corrections) without the rescaling of tiny arguments and the class test around it; a = 0 handled by a select.
// Same bits for normal a (rt2d_debug_div_check, mode 1).
__device__ __forceinline__ double rt2d_sqrt_fast(double a) {           // a >= 0
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    g = fma(fma(-g, g, a), h, g);
    g = fma(fma(-g, g, a), h, g);
    return a > 0.0 ? g : 0.0;
}

__device__ __forceinline__ double rt2d_alpha(const double* rec, double d, double ff) {
    double f1 = rt2d_quad_of(rec, d, 0), f2 = rt2d_quad_of(rec, d, 1);
    double A2 = rt2d_quad_of(rec, d, 2), E1 = rt2d_quad_of(rec, d, 3);
    double E2 = rt2d_quad_of(rec, d, 4), E3 = rt2d_quad_of(rec, d, 5);
    double alpha_1 = rt2d_div_fast(k_A1 * (f1 * ff), f1 * f1 + ff);     // model_absorption.py:37-44: the same quotients (rt2d_div_fast)
    double alpha_2 = rt2d_div_fast(A2 * (f2 * ff), f2 * f2 + ff);
    double alpha_3 = k_A3 * ff;
    return alpha_1 * E1 + alpha_2 * E2 + alpha_3 * E3;
}

// The same quantities for the one-pass heat map, whose gain is never stored and only has to keep the map within its
// 1e-12: FMA Horner, quotients and the square root to one ulp without the IEEE sequence's scaling and fix-up (the operands
// are squares of table values plus (f/1000)^2 and squared distances in metres: no denormals, no overflow).  About 30
// vector instructions fewer per point than the exact forms above, which every kernel that STORES a gain keeps.
__device__ __forceinline__ double rt2d_quad_fma(const double* r, double d, int k) {
    return fma(fma(r[1 + 3 * k], d, r[2 + 3 * k]), d, r[3 + 3 * k]);
}
__device__ __forceinline__ double rt2d_alpha_fast(const double* rec, double d, double ff) {
    const double f1 = rt2d_quad_fma(rec, d, 0), f2 = rt2d_quad_fma(rec, d, 1);
    const double A2 = rt2d_quad_fma(rec, d, 2), E1 = rt2d_quad_fma(rec, d, 3);
    const double E2 = rt2d_quad_fma(rec, d, 4), E3 = rt2d_quad_fma(rec, d, 5);
    const double alpha_1 = rt2d_div_fast(k_A1 * (f1 * ff), fma(f1, f1, ff));
    const double alpha_2 = rt2d_div_fast(A2 * (f2 * ff), fma(f2, f2, ff));
    return fma(alpha_1, E1, fma(alpha_2, E2, (k_A3 * ff) * E3));
}

// Rarely used launch data (rebound branch, final stores) is parked in LDS instead of scalar registers: the hot
// loop alone needs ~90 SGPRs of table descriptors and bounds, and everything beyond the budget would be spilled
// to v_writelane/v_readlane pairs inside the loop.
struct CastCold {
    DevTable dx_floor, dx_ceil;
    Rt2dRayOut out;
    int64_t n;
    int32_t rebound_cap, n_rebounds_max;
};

// which side of the bounds a (possibly clipped) point leaves through: the value fit_to_bounds returns (ray.py:84-107)
__device__ __forceinline__ int rt2d_fit_code(const DevEnv& e, double x_prev, double x_new, double z_prev, double z_new, double dx_z) {
    int out = 0;
    if (x_new < e.range_min[0]) { x_new = e.range_min[0]; z_new = -1 * dx_z * (x_new - x_prev) + z_prev; out = 1; }
    else if (x_new > e.range_max[0]) { x_new = e.range_max[0]; z_new = dx_z * (x_new - x_prev) + z_prev; out = 2; }
    if (z_new < e.range_min[1]) out = 3; else if (z_new > e.range_max[1]) out = 4;
    return out;
}

// Development aid: when a log buffer is set (rt2d_debug_wave_log), every wave records when it started and ended (100 MHz
// clock) and where it ran; the destructor runs wherever the lane of the wave's first thread leaves the kernel.
#ifndef RT2D_WAVE_LOG
struct WaveLog { __device__ __forceinline__ WaveLog(const CastArgs&) {} };     // compiled out (costs four registers): -DRT2D_WAVE_LOG
#define RT2D_WLOG(...)
#else
struct WaveLog {
    unsigned long long* slot; unsigned long long t0;
    // steps the wave took in its tight loop / in the general loop, and the 100 MHz ticks spent in each
    unsigned long long n_tight, n_general, ticks_tight, ticks_general;
    unsigned long long clk_head, clk_tail;     // shader clocks of the tight loop: top -> decision, decision -> end of the step
    unsigned long long n_gen_flag = 0, n_gen_steep = 0;    // general iterations in which some ray left the common step / was steep
    __device__ __forceinline__ WaveLog(const CastArgs& a) : slot(nullptr), t0(0), n_tight(0), n_general(0), ticks_tight(0), ticks_general(0), clk_head(0), clk_tail(0) {
        if (a.wave_log == nullptr || (threadIdx.x & 63) != 0) return;
        const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        if (w >= a.wave_log_cap) return;
        slot = a.wave_log + 12 * w;
        t0 = __builtin_amdgcn_s_memrealtime();
    }
    __device__ __forceinline__ ~WaveLog() {
        if (!slot) return;
        slot[0] = t0; slot[1] = __builtin_amdgcn_s_memrealtime();
        slot[2] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11));          // HW_ID
        slot[3] = (unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11));         // XCC_ID
        slot[4] = n_tight; slot[5] = n_general; slot[6] = ticks_tight; slot[7] = ticks_general;
        slot[8] = clk_head; slot[9] = clk_tail; slot[10] = n_gen_flag; slot[11] = n_gen_steep;
    }
};
#define RT2D_WLOG(...) __VA_ARGS__
#endif

// WRITE: store the trajectory.  TAB: where the step tables are read from (see the kernel).
// POST: 0 = counts only (the counting pass of a stored cast); 1 = + bounding box of the ray (what the heat-map
// pipeline needs); 2 = the whole scalar post-pass (L, T, target products; the decision margin when out.margin is set).
// CUTOFF: opt-in termination on ray power (not in the reference, SURVEY trap 2): the gain of populate_NEW at
// p.cutoff_freq is carried along (S = absorption cumsum, A = rebound losses, exactly the gain kernel's expressions)
// and the ray ends with the first point whose G = S + A is below the cut-off; that point is kept.  The gain tables
// are read from global memory here (they sit behind the step section of the blob).
// CEILC: the surface table is one constant (the sea surface z = 0 of every configuration of the reference): its lookup,
// the verification of its guess and its interpolation drop out of the step; the value is the same bit for bit
// (slope 0: 0 * d + c == c).
// FUSE: no trajectory leaves the kernel: the gain at a.heat_freq is carried along (as under CUTOFF, but with the
// absorption table in LDS) and every point is binned into the heat map H on the fly.  Only for rays whose columns never
// decrease, whose points all fall inside the map and that visit at most 8 rows per column (the caller's ray_mask,
// RayFan.fused_mask): then "last write wins" (functions.py:16) is local to the column a ray is in -- a register entry for
// the row it is in, 8 LDS slots per lane for rows it has left but may come back to -- and the column is deposited when
// the ray leaves it.  Slope tables and the impedance table are read from global memory (rebounds only) to make room.
#define RT2D_FUSE_MAX_WAVES 8      // launch bound of the FUSE kernels: they want ~250 vector registers; at 12 waves (168
                                   // registers) the spills cost more than the lost occupancy (27 vs 22 ms per 1M rays)
#ifndef RT2D_FUSE2_MAX_WAVES
#define RT2D_FUSE2_MAX_WAVES 8     // with the rebound losses precomputed (FUSE == 2): 216 registers; at 12 waves (168, ~30 words
                                   // spilled) the step itself is faster (10.7 vs 12.0 ms per 1M rays) but the deposits twice as slow
#endif
#define RT2D_TAB1_MAX_WAVES 4      // launch bound of the instances with all step tables in LDS (small fans, one wave per workgroup):
                                   // 256 vector registers instead of 128, so that the straight-line rebound does not spill
template <bool WRITE, int TAB, int POST, bool CUTOFF, bool CEILC, int FUSE = 0>
__global__ void __launch_bounds__(FUSE == 2 ? RT2D_FUSE2_MAX_WAVES * RT2D_WAVE : FUSE ? RT2D_FUSE_MAX_WAVES * RT2D_WAVE : (TAB == 1 || TAB == 3) ? RT2D_TAB1_MAX_WAVES * RT2D_WAVE : 1024) rt2d_cast_kernel(const CastArgs a) {
    extern __shared__ __attribute__((aligned(16))) double s_mem[];
    constexpr int COLD_DOUBLES = (sizeof(CastCold) + 7) / 8;
    CastCold* cold = reinterpret_cast<CastCold*>(s_mem);
    double* s_tab = s_mem + ((COLD_DOUBLES + 1) & ~1);
    if (threadIdx.x == 0) {
        cold->dx_floor = a.env.dx_floor; cold->dx_ceil = a.env.dx_ceil; cold->out = a.out; cold->n = a.n;
        cold->rebound_cap = a.p.rebound_cap; cold->n_rebounds_max = a.p.n_rebounds_max;
    }
    // TAB = 3: the tables of 2 (sound speed in LDS, the rest through the caches) with the launch bound and the rebound forms of 1:
    // a small fan that must leave the LDS to a kernel running beside it (RT2D_LAUNCH_SMALL_LDS).
    // TAB: where the step tables are read from.  1: the whole step section staged in LDS (small fans: one wave per CU, the
    // LDS round trip is on the critical path of every step).  2: only the sound-speed table (the blob's first table, a few
    // KB) in LDS, floor and surface through L1 / L2 (large fans: workgroups of a few waves, many per CU, so that the
    // hardware refills every wave slot the moment a wave ends -- with the whole section in LDS a CU holds ONE workgroup
    // of 16 waves, whose slots stay empty from the first wave that ends to the last).  0: everything through L1 / L2.
    const double* tab_cz;       // sound speed
    const double* tab_fl;       // floor and surface
    const int n_stage = FUSE ? a.fuse_step_doubles : ((TAB == 2 || TAB == 3) ? a.env.cz_doubles : a.env.step_doubles);
    if (TAB != 0) {
        for (int i = threadIdx.x; i < n_stage; i += blockDim.x) s_tab[i] = a.blob[i];
        tab_cz = s_tab;
        if ((TAB == 2 || TAB == 3) && !FUSE) tab_fl = a.blob; else tab_fl = s_tab;
    } else {
        tab_cz = a.blob; tab_fl = a.blob;
    }
    double* s_abs = s_tab + n_stage;                       // FUSE: absorption table, then 8 x 64 slots per wave
    if (FUSE) {
        const double* src_abs = a.blob + a.env.gain_off + a.env.absorb.off;
        for (int i = threadIdx.x; i < a.fuse_abs_doubles; i += blockDim.x) s_abs[i] = src_abs[i];
    }
    const double* tab_slope = (FUSE || TAB == 2 || TAB == 3) ? a.blob : tab_fl;         // where the slope tables are read from
    __syncthreads();
    WaveLog wave_log(a);
    // Ray of this lane.  A wave executes the rebound branch whenever ANY of its lanes rebounds, which with 64 rays is
    // most steps; a small fan therefore uses only `lanes` lanes of each wave and spreads over up to one wave per CU
    // (launch_cast2), so that a wave pays for its own few rebounds only.
    const int lane = threadIdx.x & (RT2D_WAVE - 1);
    if (lane >= a.lanes) return;
    const int64_t r = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * a.lanes + lane;
    if (r >= a.n) return;
    if (FUSE && a.ray_mask && !a.ray_mask[r]) return;
    const DevEnv& e = a.env;
    const Rt2dParams& p = a.p;

    // trajectory slot of this lane: its tile's start (one tile per wave: `lanes` is a power of two, so the start sits
    // in scalar registers and one 32-bit byte offset per lane addresses X, Z and C alike) and how many points the tile has
    // room for
    double* tx_base = nullptr; double* tz_base = nullptr; double* tc_base = nullptr;
    unsigned int p_off = 0;                    // byte offset of this lane's CURRENT point inside its tile
    int room = 0;
    if (WRITE) {
        const int64_t t0 = a.tile_off[r >> 6], t1 = a.tile_off[(r >> 6) + 1];
        room = (int)((t1 - t0) / RT2D_WAVE);
        const int64_t t0u = (int64_t)(((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(t0 >> 32)) << 32)
                                      | (unsigned int)__builtin_amdgcn_readfirstlane((int)t0));
        tx_base = a.X + t0u; tz_base = a.Z + t0u;
        if (a.C) tc_base = a.C + t0u;
        p_off = (unsigned int)(r & 63) * 8u;
    }
    auto put = [](double* base, unsigned int off, double v) { *(double*)((char*)base + off) = v; };
    constexpr unsigned int P_STEP = RT2D_WAVE * 8u;

    int status = RT2D_RAY_OK, stop = RT2D_STOP_NONE, n_reb = 0, n_pts = 1;
    double x = a.src_x, z = a.src_z;

    const double dx_z_max = fabs(p.dz_max / p.dx_max);                    // ray.py:132
    double c = e.c_surface;                                               // ray.py:145
    double dx_z, mult;
    if (a.init_slope) {
        dx_z = a.init_slope[r];
        mult = a.init_mult[r];
    } else {
        double angle = -1 * a.angles[r] + (M_PI / 2);                     // ray.py:138
        dx_z = 1 / tan(angle);                                            // ray.py:139
        double q = c / sin(angle);
        mult = -1 * (q * q);                                              // ray.py:146
    }
    double x_dir = 1.0, kx = x_dir, kz = dx_z;

    // post-pass state (ray.py:263-279)
    constexpr bool FULL = POST >= 2;
    double c_pt = 0.0, L = 0.0, T = 0.0;
    double min_x = x, max_x = x, min_z = z, max_z = z;
    double margin = INFINITY;
    bool mono = true;
    double slope_max = 0.0;
    const bool want_margin = FULL && a.out.margin != nullptr;
    const bool has_target = FULL && p.has_target != 0;
    const double tx = p.target[0], tz = p.target[1];
    TargetTrack tt;
    if (FULL || (WRITE && tc_base)) c_pt = rt2d_eval_c(tab_cz, e.cz, z);          // C[0] = calc_c(z0), ray.py:273
    if (FULL) { tt.init(); if (has_target) tt.add(x, z, 0.0, 0, tx); }
    if (WRITE && room > 0) { put(tx_base, p_off, x); put(tz_base, p_off, z); if (tc_base) put(tc_base, p_off, c_pt); }
    // running gain at the cut-off frequency
    double g_S = 0.0, g_A = 0.0, g_pend = 0.0, g_ff = 0.0;
    const double* gtab = a.blob + e.gain_off;
    const double g_freq = FUSE ? a.heat_freq : p.cutoff_freq;
    if (CUTOFF || FUSE) { double f = g_freq / 1000; g_ff = f * f; }
    DevTable ab_lds = e.absorb; ab_lds.off = 0;            // the LDS copy of the absorption table starts at s_abs

    // ---- fused binning (FUSE): the source is the ray's first point, G[0] = 0 (ray.py:347) ----
    RayBinner binner;
    binner.start(a.bin, FUSE, s_abs + a.fuse_abs_doubles + (threadIdx.x >> 6) * (8 * RT2D_WAVE) + lane, x, z, 0.0);

    // fmin / fmax through the bare instructions: in IEEE mode the compiler quiets every operand it cannot prove canonical
    // with an extra v_max_f64 x, x in front of each fmin / fmax, and the proof does not reach loop-invariant values across
    // the loop boundary -- five extra instructions per step.  v_max_f64 / v_min_f64 themselves return the non-NaN
    // operand and quiet a NaN result, which is what fmax / fmin do.
    auto vmax = [](double u, double v) -> double { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(u), "v"(v)); return r; };
    auto vmin = [](double u, double v) -> double { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(u), "v"(v)); return r; };
    auto vmax_abs = [](double u, double v) -> double { double r; asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(u), "v"(v)); return r; };   // max(u, |v|)
    const int fl_jmax = e.floor.n_pieces - 1, ce_jmax = e.ceil.n_pieces - 1;
    const int cz_jmax = e.cz.n_pieces - 1;
    // the bounds those instructions take, parked in vector registers for the whole loop (the empty asm keeps the compiler
    // from re-creating them from scalar registers in front of every use)
    double fl_lo = e.floor.lo, fl_hi = e.floor.hi, ce_lo = e.ceil.lo, ce_hi = e.ceil.hi;
    double z_lo_bound = e.range_min[1], z_hi_bound = e.range_max[1];
    if (CEILC) z_hi_bound = vmin(z_hi_bound, a.ceil_const);
    asm volatile("" : "+v"(fl_lo), "+v"(fl_hi), "+v"(z_lo_bound), "+v"(z_hi_bound));
    if (!CEILC) asm volatile("" : "+v"(ce_lo), "+v"(ce_hi));
    // rt2d_guess with the clamp on the integer side: v_cvt_i32_f64 saturates and turns a NaN into 0, so the piece is the
    // same for every input (fewer instructions than the max / min pair in double and the idle cycles behind each of them)
    auto guess = [&](const DevTable& t, double v, int jmax) -> int {
        const double g = (v - t.x0) * t.inv_h;
        int j;
        asm("v_cvt_i32_f64 %0, %1" : "=v"(j) : "v"(g));
        return min(max(j, 0), jmax);
    };
    // the same with the clamp to [j_lo, j_hi] in one instruction (j_lo <= j_hi; j_hi in a vector register)
    auto guess_in = [&](const DevTable& t, double v, int j_lo, int j_hi) -> int {
        const double g = (v - t.x0) * t.inv_h;
        int j;
        asm("v_cvt_i32_f64 %0, %1\n\tv_med3_i32 %0, %0, %2, %3" : "=&v"(j) : "v"(g), "s"(j_lo), "v"(j_hi));
        return j;
    };
    // Lane masks of single comparisons as 64-bit scalars (a bit is set for an enabled lane whose comparison holds, as the
    // comparison of the source language would be: an unordered pair fails ge / lt / le and passes ngt): combined with
    // scalar AND and tested against the mask of enabled lanes in one instruction, where the compiler's own lowering of
    // "all lanes" goes back through a vector register.
    auto m_ge = [](double u, double v) -> unsigned long long { unsigned long long m; asm("v_cmp_ge_f64 %0, %1, %2" : "=s"(m) : "v"(u), "v"(v)); return m; };
    auto m_lt = [](double u, double v) -> unsigned long long { unsigned long long m; asm("v_cmp_lt_f64 %0, %1, %2" : "=s"(m) : "v"(u), "v"(v)); return m; };
    auto m_le = [](double u, double v) -> unsigned long long { unsigned long long m; asm("v_cmp_le_f64 %0, %1, %2" : "=s"(m) : "v"(u), "v"(v)); return m; };
    auto m_abs_ngt = [](double u, double v) -> unsigned long long { unsigned long long m; asm("v_cmp_ngt_f64 %0, |%1|, %2" : "=s"(m) : "v"(u), "v"(v)); return m; };
    // (second operand wave-uniform, taken from a scalar register pair)
    auto m_abs_ge = [](double u, double v) -> unsigned long long { unsigned long long m; asm("v_cmp_ge_f64 %0, |%1|, %2" : "=s"(m) : "v"(u), "s"(v)); return m; };

    // Record fetches.  Where the table is staged in LDS the record's byte address is ONE 24-bit multiply-add on a scalar
    // base (a piece index is far below 2^24; the generic pointers above cost a move and a 64-bit multiply-add each).
    typedef const __attribute__((address_space(3))) double lds_cdouble;
    constexpr bool FL_LDS = (TAB == 1) || (TAB == 2 && FUSE != 0);
    constexpr bool CZ_LDS = (TAB != 0);
    const unsigned int l_base = (unsigned int)(size_t)(lds_cdouble*)s_tab;
    auto load_fl = [&](const DevTable& t, int j, double (&rr)[4]) {
        if constexpr (FL_LDS) {
            unsigned int addr;
            asm("v_mad_u32_u24 %0, %1, 24, %2" : "=v"(addr) : "v"(j), "s"(l_base + (unsigned int)t.off * 8u));
            lds_cdouble* q = (lds_cdouble*)(size_t)addr;
#pragma unroll
            for (int k = 0; k < 4; ++k) rr[k] = q[k];
        } else {
            auto q = tab_fl + t.off + j * 3;
#pragma unroll
            for (int k = 0; k < 4; ++k) rr[k] = q[k];
        }
    };
    auto load_cz = [&](const DevTable& t, int j, double (&rr)[8]) {
        if constexpr (CZ_LDS) {
            unsigned int addr;
            asm("v_mad_u32_u24 %0, %1, 56, %2" : "=v"(addr) : "v"(j), "s"(l_base + (unsigned int)t.off * 8u));
            lds_cdouble* q = (lds_cdouble*)(size_t)addr;
#pragma unroll
            for (int k = 0; k < 8; ++k) rr[k] = q[k];
        } else {
            auto q = tab_cz + t.off + j * 7;
#pragma unroll
            for (int k = 0; k < 8; ++k) rr[k] = q[k];
        }
    };

    // Plain steps: a step of dx_max (ray not steep) whose point is inside the rectangle, between floor and surface, with
    // all table guesses right and room left in its tile -- nothing but the curvature update happens.  While EVERY ray of
    // the wave takes such a step the wave stays in the tight loop below: the same statements as the common path of the
    // general step further down (same order, same roundings: the trajectories are bit for bit the same), but with ONE
    // wave-uniform exit, so that none of the per-lane exit bookkeeping of the general loop (ended rays, rebounds, clipped
    // points, the choice of the division) is carried through it.  The step that the wave cannot take together is left
    // untouched for the general loop, which then tries the tight loop again.  (The waves of a small fan hold four shallow
    // rays that rebound twenty times in 1500 steps: they spend nine steps in ten here.)
    constexpr bool SMALLF = (TAB == 1 || TAB == 3);      // the instances of few waves per CU: 256 registers, rebounds in straight-line form
    constexpr bool TIGHT = !CUTOFF && (FUSE == 0 || SMALLF) && (POST <= 1 || SMALLF);     // (with the whole post-pass only where 256
                                                                                    // registers are to be had: it needs 175)
    bool tight_on = TIGHT && !(WRITE && a.C != nullptr) && !want_margin && a.tight_off != 1 && !(FULL && a.tight_off == 2);      // (the decision margin is a test output: general loop)
    // Wave-uniform facts the tight loop leans on, established once:
    //  - every guessed piece is verified against its two breaks, which needs the uniform-grid hint of all three tables;
    //  - the guesses are clamped to the pieces that lie WHOLLY inside the rectangle (floor pieces in x, sound-speed
    //    pieces at the lower end in z): a point verified to be in its guessed piece is then inside the rectangle without
    //    a comparison against the rectangle, and a point in a piece that straddles the border fails the verification and
    //    is left to the general step;
    //  - the lanes of a wave share their tile's room, and a ray has at most i + 1 points before step i: step i fits
    //    while i + 1 < room, one scalar comparison instead of a per-lane one.
    tight_on = tight_on && e.floor.uniform != 0 && e.cz.uniform != 0 && (CEILC || e.ceil.uniform != 0);
    int fl_j0 = 0, fl_j1 = fl_jmax, cz_j0 = 0, cz_j1 = cz_jmax;
    if (TIGHT) {
        auto pieces_inside = [](auto tab, const DevTable& t, double lo, double hi, int& j0, int& j1) {
            int j = rt2d_fix_piece(tab, t, lo, rt2d_guess(t, lo));                 // last piece that starts at or below lo
            j0 = (tab[t.off + j * t.stride] >= lo) ? j : j + 1;
            j = rt2d_fix_piece(tab, t, hi, rt2d_guess(t, hi));
            j1 = (tab[t.off + (j + 1) * t.stride] <= hi) ? j : j - 1;
            j0 = __builtin_amdgcn_readfirstlane(j0); j1 = __builtin_amdgcn_readfirstlane(j1);
        };
        if (!((e.floor.lo >= e.range_min[0]) && (e.floor.hi <= e.range_max[0])))
            pieces_inside(tab_fl, e.floor, e.range_min[0], e.range_max[0], fl_j0, fl_j1);
        // (in z only the lower side: the sound-speed table usually reaches above the surface, and the rays of a surface
        // duct spend their time in its top piece -- the upper bound keeps its comparison)
        if (!(e.cz.lo >= e.range_min[1])) pieces_inside(tab_cz, e.cz, e.range_min[1], e.cz.hi, cz_j0, cz_j1);
        tight_on = tight_on && fl_j0 <= fl_j1 && cz_j0 <= cz_j1;
    }
    int tight_end = p.n_steps_max - 1;
    if (WRITE) {
        const int room_u = __builtin_amdgcn_readfirstlane(room);
        tight_on = tight_on && rt2d_all(room == room_u);
        tight_end = min(tight_end, room_u - 1);
    }
    // The ordinary rebound (ray.py:176-220) in straight-line form: pieces guessed right, a floor or (constant) surface hit that
    // is neither the last allowed one nor beyond the log, the reflected point inside the rectangle -- the same expressions
    // in the same order as the general case of the general step, but every table record and the parked launch data
    // fetched in two batches of LDS reads, every decision a select, and nothing written before all of them are made.
    // rebound_examine works on the step from (x, z) with the scaled direction (kxs, kzs) and the surfaces (z_fl, z_ce) under
    // the point that was hit; it returns false for anything unusual (the caller then leaves the ray to the general case).
    struct RebOut { double xr, z_hit, rx, rzz, rh[8]; bool log_it; };
    auto rebound_examine = [&](bool hitf, double z_fl, double z_ce, double kxs, double kzs, RebOut& o) -> bool {
        bool fast = true;
        const double z_hit = hitf ? z_fl : z_ce;
        // batch 1: floor record under the previous point, sound-speed record at the surface that was hit, the slope
        // table's description, the limits
        const double xq = vmin(vmax(x, fl_lo), fl_hi);                                    // rt2d_eval_lin: both tables clamp
        const int jq = guess(e.floor, xq, fl_jmax);
        const int jh = guess(e.cz, z_hit, cz_jmax);
        double rq[4];
        load_fl(e.floor, jq, rq);
        load_cz(e.cz, jh, o.rh);
        const DevTable* st = hitf ? &cold->dx_floor : &cold->dx_ceil;
        const double st_x0 = st->x0, st_inv_h = st->inv_h, st_lo = st->lo, st_hi = st->hi;
        const int st_off = st->off, st_n = st->n_pieces, st_oob = st->oob, st_uniform = st->uniform;
        const int nrm = cold->n_rebounds_max, reb_cap = cold->rebound_cap;
        const bool has_log = cold->out.reb_step != nullptr;
        asm volatile("" :: "v"(nrm), "v"(reb_cap), "v"(has_log), "v"(st_x0), "v"(st_inv_h), "v"(st_lo), "v"(st_hi),
                     "v"(st_off), "v"(st_n), "v"(st_oob), "v"(st_uniform), "v"(rq[1]), "v"(rq[2]), "v"(o.rh[1]), "v"(o.rh[6]));
        fast &= ((xq >= rq[0]) | (jq == 0)) & ((xq < rq[3]) | (jq == e.floor.n_pieces - 1));                  // rt2d_lookup<3>
        fast &= ((z_hit >= o.rh[0]) | (jh == 0)) & ((z_hit < o.rh[7]) | (jh == e.cz.n_pieces - 1)) & (e.cz.uniform != 0);
        // (a rebound beyond the log's room is an error only when there is a log: a cast without one just goes on)
        o.log_it = n_reb < reb_cap;
        fast &= (o.log_it | !has_log) & !((nrm > -1) & (n_reb + 1 > nrm)) & (st_uniform != 0);
        const double fx = hitf ? rq[1] * (xq - rq[0]) + rq[2] : a.ceil_const;
        const double xr = x + (fx - z) / dx_z;                                            // ray.py:177,200
        // rt2d_eval_quad on the slope table of the surface that was hit
        const double xs = (st_oob == 1) ? vmin(vmax(xr, st_lo), st_hi) : xr;
        fast &= (st_oob != 2) | ((xr >= st_lo) & (xr <= st_hi));
        int js;
        {
            const double g = (xs - st_x0) * st_inv_h;
            asm("v_cvt_i32_f64 %0, %1" : "=v"(js) : "v"(g));
            js = min(max(js, 0), st_n - 1);
        }
        double rs[5];
        {
            auto q = tab_slope + st_off + js * 4;
#pragma unroll
            for (int k = 0; k < 5; ++k) rs[k] = q[k];
        }
        fast &= ((xs >= rs[0]) | (js == 0)) & ((xs < rs[4]) | (js == st_n - 1));                                // rt2d_lookup<4>
        const double sl = rt2d_quad_of(rs, xs - rs[0], 0);
        // reflection (ray.py:178-186, 201-209)
        const double u0 = 1.0, u1 = sl;
        const double n0 = hitf ? -1 * u1 : u1;
        const double n1 = hitf ? u0 : -1 * u0;
        const double ku = kxs * u0 + kzs * u1, kn = kxs * n0 + kzs * n1;
        o.rx = ku * u0 - kn * n0; o.rzz = ku * u1 - kn * n1;
        o.xr = xr; o.z_hit = z_hit;
        // all bounds (ray.py:225-230)
        fast &= !((xr < e.range_min[0]) | (xr > e.range_max[0]) | (z_hit < e.range_min[1]) | (z_hit > e.range_max[1]));
        return fast;
    };
    // ... and its entry in the rebound log (x, z, c: still the point before the rebound; step: i + 1, ray.py:211-218)
    auto rebound_log = [&](bool hitf, const RebOut& o, int step_no) {
        // batch 2: the log's addresses (device memory: as the generic pointers they are parked as, the stores would be
        // FLAT instructions, which the next LDS read would have to wait for)
        const Rt2dRayOut& co = cold->out;
        typedef __attribute__((address_space(1))) int32_t* g_i32p;
        typedef __attribute__((address_space(1))) uint8_t* g_u8p;
        typedef __attribute__((address_space(1))) double* g_f64p;
        const g_i32p o_step = (g_i32p)co.reb_step; const g_u8p o_surface = (g_u8p)co.reb_surface;
        const g_f64p o_velocity = (g_f64p)co.reb_velocity, o_angle = (g_f64p)co.reb_angle;
        const g_f64p o_pos_x = (g_f64p)co.reb_pos_x, o_pos_z = (g_f64p)co.reb_pos_z;
        const int64_t n_rays = cold->n;
        asm volatile("" :: "v"(o_step), "v"(o_surface), "v"(o_velocity), "v"(o_angle), "v"(o_pos_x), "v"(o_pos_z), "v"(n_rays));
        if (o.log_it) {
            const int64_t k = (int64_t)n_reb * n_rays + r;
            if (o_step) o_step[k] = step_no;
            if (o_surface) o_surface[k] = hitf ? RT2D_SURFACE_SEDIMENT : RT2D_SURFACE_WATER;
            if (o_velocity) o_velocity[k] = c;
            if (o_angle) {
                const double q = o.rzz / o.rx;
                o_angle[k] = 1 / sqrt(1 + q * q);                                         // ray.py:184,207 (cos of grazing angle)
            }
            if (o_pos_x) o_pos_x[k] = x;
            if (o_pos_z) o_pos_z[k] = z;
        }
        n_reb += 1;
    };
    double dx_z_max_v = dx_z_max;
    int fl_j1_v = fl_j1, cz_j1_v = cz_j1, ce_j1_v = ce_jmax;
    asm volatile("" : "+v"(dx_z_max_v), "+v"(fl_j1_v), "+v"(cz_j1_v));
    if (!CEILC) asm volatile("" : "+v"(ce_j1_v));
    int i = 0;
    int tight_wait = 0, tight_penalty = 1;      // wave-uniform: steps until the tight loop is tried again, doubled by every
                                                // attempt that takes no step (a wave of steep rays rebounds somewhere in
                                                // almost every step and would pay for the attempt each time)
    for (; i < p.n_steps_max; ++i) {
        RT2D_WLOG(const unsigned long long w_a = __builtin_amdgcn_s_memrealtime(); const int w_i = i;)
        if (tight_on && tight_wait > 0) --tight_wait;
        else if (tight_on) {
            const int i_in = i;
            // Through such steps kx stays x_dir = +-1 (ray.py:242-246 give it back unchanged): x moves one way, so the ends of
            // the stretch give the bounding box in x and whether the ray turned back; kz is dx_z at the top of every step.
            // Two forms of the loop.  PLAIN: no ray is steep -- the step in x is +-dx_max for the whole stretch, k1/k0 is a
            // division by that constant.  STEEP (some ray is steep, ray.py:153): the step length is chosen per ray
            // (dz_max / |dx_z| for a steep one) and k1/k0 is the IEEE division; it ends when no ray is steep any more.
            const double x_in = x;
            const unsigned long long lanes_on = __builtin_amdgcn_read_exec();
            auto stretch = [&](auto steep_form) {
                constexpr bool ST = decltype(steep_form)::value;
                const double kx_p = kx * p.dx_max;                                // ray.py:154 (not steep)
                const unsigned int flip = (kx_p < 0.0) ? 0x80000000u : 0u;       // k1/k0 changes sign with k0
                // The loop is written rotated -- examine a step, then { commit it; examine the next } while it can be taken
                // -- so that an iteration ends in its one conditional branch.
                double kx_s = kx_p, kz_s, xn, zn, ta[4], tb[4], tz8[8], zf_t, zc_t = 0.0;
                unsigned long long steep_t = 0ull;
                auto examine = [&]() -> bool {
                    RT2D_WLOG(const unsigned long long c_a = __builtin_amdgcn_s_memtime();)
                    if (ST) {
                        const double ss = rt2d_div_fast(p.dz_max, fabs(dx_z));     // ray.py:153 (the IEEE quotient: see rt2d_div_fast)
                        const bool st = fabs(dx_z) > dx_z_max;
                        steep_t = __builtin_amdgcn_ballot_w64(st);
                        const double sl = st ? ss : p.dx_max;
                        kx_s = kx * sl; kz_s = dx_z * sl;
                    } else {
                        kz_s = dx_z * p.dx_max;
                    }
                    xn = x + kx_s; zn = z + kz_s;                                  // ray.py:157
                    // (no clamp of xn into the tables: a point outside them is in no piece and fails below)
                    const int ja_t = guess_in(e.floor, xn, fl_j0, fl_j1_v), jb_t = CEILC ? 0 : guess_in(e.ceil, xn, 0, ce_j1_v);
                    const int jz_t = guess_in(e.cz, zn, cz_j0, cz_j1_v);
                    load_fl(e.floor, ja_t, ta);
                    if (!CEILC) load_fl(e.ceil, jb_t, tb);
                    load_cz(e.cz, jz_t, tz8);
                    // (every word of the three records is wanted HERE: without this the compiler sinks the loads of the
                    // coefficients behind the exit test below, a second LDS round trip on the critical path of every step)
                    asm volatile("" :: "v"(ta[1]), "v"(ta[2]), "v"(tz8[1]), "v"(tz8[2]), "v"(tz8[3]), "v"(tz8[4]), "v"(tz8[5]), "v"(tz8[6]));
                    // the guessed pieces must be the right ones WITHOUT the end-piece exceptions of the general step (a point
                    // in an end piece that fails here is simply left to the general step)
                    zf_t = ta[1] * (xn - ta[0]) + ta[2];
                    unsigned long long ok_t = m_ge(xn, ta[0]) & m_lt(xn, ta[3]) & m_ge(zn, tz8[0]) & m_lt(zn, tz8[7]) & m_ge(zn, zf_t);
                    if (!ST) ok_t &= m_abs_ngt(dx_z, dx_z_max_v) & m_abs_ge(kz_s, 1e-250);
                    else ok_t &= m_abs_ge(kz_s, 1e-250);              // (k1/k0 below without the IEEE sequence's scaling: normal operands only)
                    if (CEILC) {
                        ok_t &= m_le(zn, z_hi_bound);
                    } else {
                        zc_t = tb[1] * (xn - tb[0]) + tb[2];
                        ok_t &= m_ge(xn, tb[0]) & m_lt(xn, tb[3]) & m_le(zn, vmin(z_hi_bound, zc_t));
                    }
                    RT2D_WLOG(wave_log.clk_head += __builtin_amdgcn_s_memtime() - c_a;)
                    // every ray can take the step as it stands, and the step count / the tile's room are not at their end
                    const unsigned long long go_t = ((i < tight_end) & (!ST | (steep_t != 0ull))) ? ok_t : 0ull;
                    return (lanes_on & ~go_t) == 0ull;
                };
                // A step that cannot be taken as it stands because some rays REBOUND in it -- ordinarily (rebound_examine),
                // without turning back in x -- while every other ray's step is plain is taken here as well (small-fan
                // instances): the exit into the general loop and back costs more than the rebound itself.  The expressions are
                // those of the general step for such a wave (the IEEE k1/k0 for every ray, as there when one ray rebounds).
                auto mixed_step = [&]() -> bool {
                    if (!SMALLF || FUSE || !(i < tight_end) || (ST && steep_t == 0ull)) return false;      // (FUSE: a rebound's loss is the general step's business)
                    const bool in_x = (xn >= ta[0]) & (xn < ta[3]) & (CEILC | ((xn >= tb[0]) & (xn < tb[3])));
                    const bool form = ST | (!(fabs(dx_z) > dx_z_max) & (fabs(kz_s) >= 1e-250));
                    const double z_ce = CEILC ? a.ceil_const : zc_t;
                    const bool hitf = zn < zf_t;
                    const bool hitc = !hitf & (zn > z_ce);
                    const bool plain_l = in_x & form & (zn >= tz8[0]) & (zn < tz8[7]) & !hitf
                                       & (zn <= (CEILC ? z_hi_bound : vmin(z_hi_bound, zc_t)));
                    const bool cand = in_x & form & (hitf | (CEILC & hitc));
                    if (!rt2d_all(plain_l | cand)) return false;
                    RebOut o;
                    bool fast = false;
                    if (cand) {
                        fast = rebound_examine(hitf, zf_t, z_ce, kx_s, kz_s, o);
                        fast &= (kx > 0.0) ? (o.rx > 0.0) : (o.rx < 0.0);                // x_dir (ray.py:242) stays what it is
                    }
                    if (rt2d_any(cand & !fast)) return false;
                    // commit (ray.py:211-218, 235-247)
                    if (POST >= 1) slope_max = vmax_abs(slope_max, dx_z);
                    const double xa_t = cand ? o.xr : xn, za_t = cand ? o.z_hit : zn;
                    if (FULL) {
                        const double ddx = xa_t - x, ddz = za_t - z;
                        const double dL = rt2d_sqrt_fast(ddx * ddx + ddz * ddz);          // (the IEEE root and quotient: rt2d_sqrt_fast, rt2d_div_fast)
                        L += dL; T += rt2d_div_fast(dL, c_pt);
                    }
                    if (cand) rebound_log(hitf, o, i + 1);
                    if (POST >= 1) mono &= !(xa_t < x);
                    x = xa_t; z = za_t;
                    double rec[8];
#pragma unroll
                    for (int k8 = 0; k8 < 8; ++k8) rec[k8] = cand ? o.rh[k8] : tz8[k8];
                    const double dzr_t = z - rec[0];
                    c = rt2d_quad_of(rec, dzr_t, 0);
                    const double dz_c_t = rt2d_quad_of(rec, dzr_t, 1);
                    if (FULL) {
                        c_pt = c;
                        if (has_target) tt.add(x, z, T, n_pts + (i - i_in), tx);
                    }
                    if (POST >= 1) {
                        min_x = vmin(min_x, x); max_x = vmax(max_x, x);
                        min_z = vmin(min_z, z); max_z = vmax(max_z, z);
                    }
                    if (WRITE) { p_off += P_STEP; put(tx_base, p_off, x); put(tz_base, p_off, z); }
                    const double k0 = cand ? o.rx : kx_s, k1 = cand ? o.rzz : kz_s;
                    dx_z = rt2d_div_fast(k1, k0);                                         // ray.py:243
                    const double dxdx_t = rt2d_div_fast(mult * dz_c_t, c * c * c);
                    dx_z += dxdx_t * k0;
                    ++i;
                    return true;
                };
                bool go = examine();
                for (;;) {
                    while (go) {
                        RT2D_WLOG(const unsigned long long c_b = __builtin_amdgcn_s_memtime();)
                        // commit (ray.py:235-247)
                        if (POST >= 1) slope_max = vmax_abs(slope_max, dx_z);
                        if (FULL) {                                                       // ray.py:271-275
                            const double ddx = xn - x, ddz = zn - z;
                            const double dL = rt2d_sqrt_fast(ddx * ddx + ddz * ddz);
                            L += dL; T += rt2d_div_fast(dL, c_pt);
                        }
                        double g_pt_t = 0.0;
                        if (FUSE) {                                                       // the general step's gain (ray.py:346-347)
                            const double ddx = xn - x, ddz = zn - z;
                            const double dL = rt2d_sqrt_fast(ddx * ddx + ddz * ddz);
                            double rec[20];
                            rt2d_lookup<19>((const double*)s_abs, ab_lds, z, rec);
                            const double dG = rt2d_div_1000(-1 * rt2d_alpha(rec, z - rec[0], g_ff)) * dL;
                            g_S += dG;
                            g_pt_t = g_S + g_A;
                            g_A += g_pend; g_pend = 0.0;
                        }
                        x = xn; z = zn;
                        const double dzr_t = z - tz8[0];
                        c = rt2d_quad_of(tz8, dzr_t, 0);
                        const double dz_c_t = rt2d_quad_of(tz8, dzr_t, 1);
                        if (FULL) {
                            c_pt = c;
                            if (has_target) tt.add(x, z, T, n_pts + (i - i_in), tx);
                        }
                        if (POST >= 1) { min_z = vmin(min_z, z); max_z = vmax(max_z, z); }
                        if (WRITE) { p_off += P_STEP; put(tx_base, p_off, x); put(tz_base, p_off, z); }
                        if (FUSE) binner.point(a.bin, x, z, g_pt_t, lane);
                        if (ST) {
                            dx_z = rt2d_div_fast(kz_s, kx_s);                                 // k1/k0 (ray.py:243)
                        } else {
                            const double q_t = rt2d_div_by_const(kz_s, p.dx_max, a.inv_dx_max);
                            dx_z = __longlong_as_double((long long)((unsigned long long)__double_as_longlong(q_t) ^ ((unsigned long long)flip << 32)));
                        }
                        const double dxdx_t = rt2d_div_fast(mult * dz_c_t, c * c * c);
                        dx_z += dxdx_t * kx_s;
                        ++i;
                        RT2D_WLOG(asm volatile("" :: "v"(dx_z)); wave_log.clk_tail += __builtin_amdgcn_s_memtime() - c_b;)
                        go = examine();
                    }
                    if (!mixed_step()) break;
                    go = examine();
                }
            };
            if (rt2d_all(fabs(kx) == 1.0)) {
                stretch(std::false_type{});
                if (rt2d_any(fabs(dx_z) > dx_z_max)) stretch(std::true_type{});
            }
            kz = dx_z;
            n_pts += i - i_in;
            if (i == i_in) { tight_wait = tight_penalty - 1; tight_penalty = min(2 * tight_penalty, 64); }
            else {
                tight_penalty = 1;
                if (POST >= 1) { mono &= !(x < x_in); min_x = vmin(min_x, x); max_x = vmax(max_x, x); }
            }
        }
        RT2D_WLOG(const unsigned long long w_b = __builtin_amdgcn_s_memrealtime();
                  wave_log.ticks_tight += w_b - w_a; wave_log.n_tight += (unsigned long long)(i - w_i);)
        // step sizing (ray.py:153-154)
        const bool steep = fabs(dx_z) > dx_z_max;
        if (POST >= 1) slope_max = vmax(slope_max, fabs(dx_z));
        {
            double s = p.dx_max;
            if (__builtin_expect(rt2d_any(steep), 0)) {     // wave-uniform and out of line: the long, shallow rays that set a
                                                         // small fan's time fall through without the division or a taken branch
                const double ss = p.dz_max / fabs(dx_z);
                if (steep) s = ss;
            }
            kx *= s; kz *= s;
        }
        double x_new = x + kx, z_new = z + kz;                            // ray.py:157

        // The three table records the common step needs -- floor and ceiling under x_new (ray.py:172-173; both tables
        // clamp, checked by rt2d_env_create) and c, dc/dz at z_new (ray.py:237-238) -- are fetched in ONE batch of
        // independent LDS reads from guessed pieces.  The guesses are verified below; the sound-speed record is only
        // used if the point is accepted as it stands (a rebound moves it, and then the lookup is redone).
        const double xa = vmin(vmax(x_new, fl_lo), fl_hi);
        const double xb = CEILC ? 0.0 : vmin(vmax(x_new, ce_lo), ce_hi);
        int ja = guess(e.floor, xa, fl_jmax), jb = CEILC ? 0 : guess(e.ceil, xb, ce_jmax);
        const int jz = guess(e.cz, z_new, cz_jmax);
        double ra[4], rb[4], rz[8];
        load_fl(e.floor, ja, ra);
        if (!CEILC) load_fl(e.ceil, jb, rb);
        load_cz(e.cz, jz, rz);
        // The common step -- inside the rectangle, between floor and surface, all three guessed pieces right -- is
        // recognised by ONE predicate and skips every branch of the general case below; when the predicate fails the
        // general case runs in the reference's order (ray.py:162-230) and none of its branches would fire otherwise.
        const bool oka = ((xa >= ra[0]) | (ja == 0)) & ((xa < ra[3]) | (ja == e.floor.n_pieces - 1)) & (e.floor.uniform != 0);
        const bool okb = CEILC ? true : (((xb >= rb[0]) | (jb == 0)) & ((xb < rb[3]) | (jb == e.ceil.n_pieces - 1)) & (e.ceil.uniform != 0));
        const bool okz = ((z_new >= rz[0]) | (jz == 0)) & ((z_new < rz[7]) | (jz == e.cz.n_pieces - 1)) & (e.cz.uniform != 0);
        double z_floor = ra[1] * (xa - ra[0]) + ra[2];
        double z_ceil = CEILC ? a.ceil_const : rb[1] * (xb - rb[0]) + rb[2];
        const bool general = !(oka & okb & okz) | (x_new < e.range_min[0]) | (x_new > e.range_max[0])
                           | (z_new < vmax(z_lo_bound, z_floor)) | (z_new > (CEILC ? z_hi_bound : vmin(z_hi_bound, z_ceil)));
        if (want_margin && !general) {
            margin = fmin(margin, rt2d_bound_margin(e, x_new, z_new));
            margin = fmin(margin, fmin(fabs(z_new - z_floor), fabs(z_new - z_ceil)));
        }
        bool final_pt = false;                 // the point about to be accepted is the ray's last (max-rebounds)
        // (branch hints: a wave alone on its CU pays an instruction-fetch stall at every taken branch, so the common
        // step should be laid out as straight-line code)
        RT2D_WLOG(if (rt2d_any(general)) wave_log.n_gen_flag += 1ull; if (rt2d_any(steep)) wave_log.n_gen_steep += 1ull;)
        // The ordinary rebound in its straight-line form (rebound_examine above); a lane that fails any test is left,
        // untouched, to the general case.
        bool reb_done = false;
        if (TIGHT && SMALLF && !FUSE && !want_margin && __builtin_expect(rt2d_any(general), 0)) {     // (small fans: their waves wait on every rebound)
            const bool hitf = z_new < z_floor;
            const bool hitc = !hitf & (z_new > z_ceil);
            const bool cand = general & oka & okb & (x_new >= e.range_min[0]) & (x_new <= e.range_max[0]) & (hitf | (CEILC & hitc));
            if (cand) {
                RebOut o;
                if (rebound_examine(hitf, z_floor, z_ceil, kx, kz, o)) {
                    rebound_log(hitf, o, i + 1);
                    x_new = o.xr; z_new = o.z_hit;
                    kx = o.rx; kz = o.rzz;
#pragma unroll
                    for (int k8 = 0; k8 < 8; ++k8) rz[k8] = o.rh[k8];
                    reb_done = true;
                }
            }
        }
        if (__builtin_expect(rt2d_any(general & !reb_done), 0)) if (general & !reb_done) {
            if (want_margin) margin = fmin(margin, rt2d_bound_margin(e, x_new, z_new));
            // horizontal bounds first (ray.py:162-167): stop only if fit_to_bounds ends on code 1 or 2, i.e. the point
            // clipped in x is still inside in z; the clipped point is NOT appended
            if (x_new < e.range_min[0] || x_new > e.range_max[0]) {
                int out = rt2d_fit_code(e, x, x_new, z, z_new, dx_z);
                if (out == 1 || out == 2) { stop = out; break; }
            }
            if (!(oka & okb)) {
                ja = rt2d_fix_piece(tab_fl, e.floor, xa, ja);
                auto pa = tab_fl + e.floor.off + ja * 3;
#pragma unroll
                for (int k = 0; k < 3; ++k) ra[k] = pa[k];
                z_floor = ra[1] * (xa - ra[0]) + ra[2];
                if (!CEILC) {
                    jb = rt2d_fix_piece(tab_fl, e.ceil, xb, jb);
                    auto pb = tab_fl + e.ceil.off + jb * 3;
#pragma unroll
                    for (int k = 0; k < 3; ++k) rb[k] = pb[k];
                    z_ceil = rb[1] * (xb - rb[0]) + rb[2];
                }
            }
            if (want_margin) margin = fmin(margin, fmin(fabs(z_new - z_floor), fabs(z_new - z_ceil)));

            const bool hit_floor = z_new < z_floor;
            const bool hit_ceil = !hit_floor && (z_new > z_ceil);
            const bool rebound = hit_floor || hit_ceil;
            if (rebound) {                                                    // ray.py:176-220
                const DevTable slope_t = hit_floor ? cold->dx_floor : cold->dx_ceil;
                double fx = hit_floor ? rt2d_eval_lin(tab_fl, e.floor, x) : (CEILC ? a.ceil_const : rt2d_eval_lin(tab_fl, e.ceil, x));
                x_new = x + (fx - z) / dx_z;          // closed-form root of surf(x) - dx_z*(x1 - x) - z (ray.py:177,200)
                z_new = hit_floor ? z_floor : z_ceil;
                bool ok;
                double s = rt2d_eval_quad(tab_slope, slope_t, x_new, &ok);
                if (!ok) { status = RT2D_RAY_ERR_INTERP; break; }
                double u0 = 1.0, u1 = s;
                double n0 = hit_floor ? -1 * u1 : u1;
                double n1 = hit_floor ? u0 : -1 * u0;
                double ku = kx * u0 + kz * u1, kn = kx * n0 + kz * n1;
                double rx = ku * u0 - kn * n0, rzz = ku * u1 - kn * n1;
                kx = rx; kz = rzz;
                double q = kz / kx;
                double ang = 1 / sqrt(1 + q * q);                              // ray.py:184,207 (cos of grazing angle)
                // What the rest of the branch needs from the parked launch data, fetched from LDS in ONE batch (read where
                // they are used, each of the eight costs its own LDS round trip).  The log's addresses are device memory:
                // as the generic pointers they are parked as, the stores would be FLAT instructions.
                const Rt2dRayOut& co = cold->out;
                typedef __attribute__((address_space(1))) int32_t* g_i32p;
                typedef __attribute__((address_space(1))) uint8_t* g_u8p;
                typedef __attribute__((address_space(1))) double* g_f64p;
                const g_i32p o_step = (g_i32p)co.reb_step; const g_u8p o_surface = (g_u8p)co.reb_surface;
                const g_f64p o_velocity = (g_f64p)co.reb_velocity, o_angle = (g_f64p)co.reb_angle;
                const g_f64p o_pos_x = (g_f64p)co.reb_pos_x, o_pos_z = (g_f64p)co.reb_pos_z;
                const int nrm = cold->n_rebounds_max, reb_cap = cold->rebound_cap;
                const int64_t n_rays = cold->n;
                asm volatile("" :: "v"(o_step), "v"(o_surface), "v"(o_velocity), "v"(o_angle), "v"(o_pos_x), "v"(o_pos_z),
                             "v"(nrm), "v"(reb_cap), "v"(n_rays));
                if (n_reb < reb_cap) {
                    const int64_t k = (int64_t)n_reb * n_rays + r;
                    if (o_step) o_step[k] = i + 1;
                    if (o_surface) o_surface[k] = hit_floor ? RT2D_SURFACE_SEDIMENT : RT2D_SURFACE_WATER;
                    if (o_velocity) o_velocity[k] = c;
                    if (o_angle) o_angle[k] = ang;
                    if (o_pos_x) o_pos_x[k] = x;
                    if (o_pos_z) o_pos_z[k] = z;
                } else if (o_step) {
                    status = RT2D_RAY_ERR_REBOUND_LOG;
                }
                n_reb += 1;
                if (FUSE == 2) {                                                // the k-th rebound's loss, computed beforehand
                    g_pend = (n_reb <= a.fuse_loss_cap) ? a.fuse_loss[(int64_t)(n_reb - 1) * cold->n + r] : 0.0;
                } else if (CUTOFF || FUSE) {                                    // ray.py:350-362; applies from the NEXT point on
                    double wavelength = c / g_freq;
                    double rc = hit_floor ? rt2d_refcoef_sediment(wavelength, rt2d_eval_quad(gtab, e.imp, z, nullptr))
                                          : rt2d_refcoef_surface(wavelength, ang);
                    g_pend = 10 * log10(rc);
                }
                if (want_margin) margin = fmin(margin, rt2d_bound_margin(e, x_new, z_new));
                if (nrm > -1 && n_reb > nrm) {                                  // ray.py:189-195: the point IS appended,
                    rt2d_fit(e, x, x_new, z, z_new, dx_z, x_dir);               // clipped into the bounds, and the ray ends
                    stop = RT2D_STOP_MAX_REBOUNDS;
                    final_pt = true;
                }
            }

            // all bounds (ray.py:225-230): any non-zero code stops, the point is not appended
            if (!final_pt && (x_new < e.range_min[0] || x_new > e.range_max[0] || z_new < e.range_min[1] || z_new > e.range_max[1])) {
                stop = rt2d_fit_code(e, x, x_new, z, z_new, dx_z);
                if (stop) break;
            }
            // the sound-speed record prefetched for z_new stands unless a rebound moved the point or the guess was off
            if (rebound | !okz) rt2d_lookup<7>(tab_cz, e.cz, z_new, rz);
        }

        // accept the point (ray.py:235-247)
        bool cut = false;
        double g_pt = 0.0;
        if (FULL || CUTOFF || FUSE) {
            double ddx = x_new - x, ddz = z_new - z;
            double dL = sqrt(ddx * ddx + ddz * ddz);                       // ray.py:271
            if (FULL) { L += dL; T += dL / c_pt; }                         // ray.py:272-275
            if (CUTOFF || FUSE) {
                double rec[20];
                if (FUSE) rt2d_lookup<19>((const double*)s_abs, ab_lds, z, rec);
                else rt2d_lookup<19>(gtab, e.absorb, z, rec);              // factors at the START of the segment, ray.py:346
                double dG = rt2d_div_1000(-1 * rt2d_alpha(rec, z - rec[0], g_ff)) * dL;
                g_S += dG;
                g_pt = g_S + g_A;
                cut = CUTOFF && !final_pt && g_pt < a.g_cut;
                g_A += g_pend; g_pend = 0.0;
            }
        }
        if (POST >= 1) mono &= !(x_new < x);
        x = x_new; z = z_new;
        const double dzr = z - rz[0];
        c = rt2d_quad_of(rz, dzr, 0);
        const double dz_c = rt2d_quad_of(rz, dzr, 1);
        if (FULL || (WRITE && a.C != nullptr)) c_pt = c;
        if (POST >= 1) {
            min_x = vmin(min_x, x); max_x = vmax(max_x, x);        // bare instructions: see vmax / vmin above
            min_z = vmin(min_z, z); max_z = vmax(max_z, z);
        }
        if (FULL && has_target) tt.add(x, z, T, n_pts, tx);
        if (WRITE && __builtin_expect(n_pts < room, 1)) {
            p_off += P_STEP;
            put(tx_base, p_off, x); put(tz_base, p_off, z);
            if (__builtin_expect(a.C != nullptr, 0)) put(tc_base, p_off, c);           // uniform test: no per-lane branch to skip
        }
        n_pts += 1;
        if (FUSE) binner.point(a.bin, x, z, g_pt, lane);
        if (__builtin_expect(final_pt, 0)) break;
        if (CUTOFF && cut) { stop = RT2D_STOP_POWER_CUTOFF; break; }
        x_dir = rt2d_sign(kx);
        // k1/k0 (ray.py:243).  In a wave whose rays all took a plain dx_max step, k0 is +-dx_max exactly and the quotient
        // comes from rt2d_div_by_const (same bits); any other wave divides.
        const bool plain = !steep & !general & (fabs(kx) == p.dx_max) & (fabs(kz) >= 1e-250);
        if (__builtin_expect(rt2d_all(plain), 1)) {
            const double q = rt2d_div_by_const(kz, p.dx_max, a.inv_dx_max);
            dx_z = (kx < 0.0) ? -1 * q : q;
        } else {
            dx_z = kz / kx;
        }
        double dxdx_z = mult * dz_c / (c * c * c);
        dx_z += dxdx_z * kx;
        kx = x_dir; kz = dx_z;

        if (__builtin_expect(!p.backprop && x_dir < 0, 0)) { stop = RT2D_STOP_BACKPROP; break; }   // ray.py:252-255
        if (i == p.n_steps_max - 1) stop = RT2D_STOP_MAX_ITER;                 // ray.py:257-259
        RT2D_WLOG(wave_log.ticks_general += __builtin_amdgcn_s_memrealtime() - w_b; wave_log.n_general += 1ull;)
    }

    if (FUSE) {
        binner.finish(a.bin, lane);                        // the column the ray ends in
        if (a.irregular) a.irregular[r] = binner.violated ? 1 : 0;
        if (binner.violated && a.n_violations) atomicAdd(a.n_violations, 1ull);
    }
    const Rt2dRayOut& o = cold->out;
    const int64_t n = cold->n;
    if (o.n_pts) o.n_pts[r] = n_pts;
    if (o.stop_code) o.stop_code[r] = stop;
    if (o.n_rebounds) o.n_rebounds[r] = n_reb;
    if (o.status) o.status[r] = status;
    if (POST >= 1) {
        if (o.range) { o.range[r] = min_x; o.range[n + r] = min_z; o.range[2 * n + r] = max_x; o.range[3 * n + r] = max_z; }
        if (o.monotone) o.monotone[r] = mono ? 1 : 0;
        if (o.slope_max) o.slope_max[r] = slope_max;
    }
    if (FULL) {
        if (o.L_end) o.L_end[r] = L;
        if (o.T_end) o.T_end[r] = T;
        if (o.margin) o.margin[r] = margin;
        if (has_target) {
            if (o.dist_to_target) o.dist_to_target[r] = tt.dist(tx, tz, n_pts, min_x, max_x);
            if (o.step_target) o.step_target[r] = tt.best_step;
            if (o.T_target) o.T_target[r] = tt.best_T;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// trajectory layout: tile_off[t+1] = 64 * max over the tile's lanes of n_pts, then an in-place inclusive scan
__global__ void rt2d_tile_len_kernel(const int32_t* n_pts, int64_t n, int64_t n_tiles, int64_t* tile_off) {
    int64_t t = (int64_t)blockIdx.x * (blockDim.x / RT2D_WAVE) + (threadIdx.x / RT2D_WAVE);
    int lane = threadIdx.x & 63;
    if (t >= n_tiles) return;
    int64_t r = t * RT2D_WAVE + lane;
    int v = (r < n) ? n_pts[r] : 0;
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
    if (lane == 0) tile_off[t + 1] = (int64_t)v * RT2D_WAVE;
    if (t == 0 && lane == 0) tile_off[0] = 0;
}

__global__ void rt2d_tile_scan_kernel(int64_t n_tiles, int64_t* tile_off) {     // one block (64 ... 1024 threads: a small fan
    __shared__ int64_t s_part[1024];                                            // pays for one wave's barriers only)
    __shared__ int64_t s_carry;
    const int B = blockDim.x;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int64_t start = 0; start < n_tiles; start += B) {
        int64_t i = start + threadIdx.x;
        int64_t v = (i < n_tiles) ? tile_off[i + 1] : 0;
        s_part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < B; off <<= 1) {
            int64_t add = (threadIdx.x >= off) ? s_part[threadIdx.x - off] : 0;
            __syncthreads();
            s_part[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < n_tiles) tile_off[i + 1] = s_part[threadIdx.x] + s_carry;
        __syncthreads();
        if (threadIdx.x == B - 1) s_carry += s_part[B - 1];
        __syncthreads();
    }
}

// Everything that follows a small fan's cast in ONE launch (one block of 1024 threads, at most 1024 tiles): the exact
// tile layout (rt2d_tile_len_kernel + rt2d_tile_scan_kernel), the fan's summary (rt2d_fan_summary_kernel, no zeroing
// before it) and the mask of the error-free rays.  Four dependent launches of a few microseconds each otherwise, behind
// a cast of 0.9 ms.
__global__ void __launch_bounds__(1024) rt2d_small_summary_kernel(const int32_t* n_pts, const int32_t* status, const double* range,
                                                                  int64_t n, int64_t n_tiles, int64_t* tile_off, double* summary,
                                                                  int32_t n_summary, uint8_t* ok_mask) {
    __shared__ int64_t s_len[1024];
    __shared__ double s_part[16][7];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x >= n_tiles) s_len[threadIdx.x] = 0;        // (entries past the last tile: one store each, not a turn of the loop below)
    for (int64_t t = wave; t < n_tiles; t += 16) {
        const int64_t r = t * RT2D_WAVE + lane;
        int v = (r < n) ? n_pts[r] : 0;
        for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
        if (lane == 0) s_len[t] = (int64_t)v * RT2D_WAVE;
    }
    __syncthreads();
    double longest = (double)s_len[threadIdx.x];
    for (int off = 1; off < n_tiles; off <<= 1) {              // inclusive scan (entries past the last tile are zero)
        const int64_t add = (threadIdx.x >= off) ? s_len[threadIdx.x - off] : 0;
        __syncthreads();
        s_len[threadIdx.x] += add;
        __syncthreads();
    }
    if (threadIdx.x < n_tiles) tile_off[threadIdx.x + 1] = s_len[threadIdx.x];
    if (threadIdx.x == 0) tile_off[0] = 0;
    double ext[4] = {0.0, 0.0, 0.0, 0.0}, full = 0.0, n_ok = 0.0;
    for (int64_t r = threadIdx.x; r < n; r += 1024) {
        const int st = status[r];
        if (ok_mask) ok_mask[r] = st == RT2D_RAY_OK ? 1 : 0;
        if (st == RT2D_RAY_ERR_REBOUND_LOG) full = 1.0;
        if (st != RT2D_RAY_OK) continue;
        n_ok += 1.0;
        const double lo_x = -range[r], lo_z = -range[n + r], hi_x = range[2 * n + r], hi_z = range[3 * n + r];
        ext[0] = (lo_x > ext[0]) ? lo_x : ext[0];
        ext[1] = (lo_z > ext[1]) ? lo_z : ext[1];
        ext[2] = (hi_x > ext[2]) ? hi_x : ext[2];
        ext[3] = (hi_z > ext[3]) ? hi_z : ext[3];
    }
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const double o = __shfl_xor(ext[k], off); ext[k] = (o > ext[k]) ? o : ext[k]; }
        const double ol = __shfl_xor(longest, off); longest = (ol > longest) ? ol : longest;
        const double of = __shfl_xor(full, off); full = (of > full) ? of : full;
        n_ok += __shfl_xor(n_ok, off);
    }
    if (lane == 0) {
        s_part[wave][0] = longest; s_part[wave][1] = full; s_part[wave][2] = n_ok;
#pragma unroll
        for (int k = 0; k < 4; ++k) s_part[wave][3 + k] = ext[k];
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        const int k = threadIdx.x;
        double v = s_part[0][k];
        for (int w = 1; w < 16; ++w) v = (k == 2) ? v + s_part[w][k] : ((s_part[w][k] > v) ? s_part[w][k] : v);
        summary[1 + k] = v;                                  // longest -> [1], full -> [2], n_ok -> [3], ext -> [4..7]
    }
    if (threadIdx.x == 0) summary[0] = (double)s_len[n_tiles - 1];
    if (threadIdx.x >= 8 && threadIdx.x < n_summary) summary[threadIdx.x] = 0.0;     // the caller's spare words start at zero
}

// copy every tile from a provisional (padded) layout into the exact one: within a tile both are [step][lane]
// Everything the host needs to know about a finished cast in 8 doubles (RayPack2D.add's range fold ray.py:466-472 and
// the checks behind the single-pass layouts): [0] points of the exact layout, [1] elements of its longest tile, [2] 1 if
// a rebound log overflowed, [3] rays without error, [4..7] = (-min x, -min z, max x, max z) over those rays, folded
// from the origin.  All maxima are over non-negative doubles, whose bit patterns order like the numbers.
__global__ void __launch_bounds__(256) rt2d_fan_summary_kernel(const int32_t* status, const double* range, const uint8_t* ray_mask,
                                                               int64_t n, const int64_t* tile_off, int64_t n_tiles, double* summary) {
    double ext[4] = {0.0, 0.0, 0.0, 0.0};
    double longest = 0.0, full = 0.0, n_ok = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) {
        const int st = status[r];
        if (st == RT2D_RAY_ERR_REBOUND_LOG) full = 1.0;
        if (st != RT2D_RAY_OK || (ray_mask && !ray_mask[r])) continue;
        n_ok += 1.0;
        if (range) {
            const double lo_x = -range[r], lo_z = -range[n + r], hi_x = range[2 * n + r], hi_z = range[3 * n + r];
            ext[0] = (lo_x > ext[0]) ? lo_x : ext[0];
            ext[1] = (lo_z > ext[1]) ? lo_z : ext[1];
            ext[2] = (hi_x > ext[2]) ? hi_x : ext[2];
            ext[3] = (hi_z > ext[3]) ? hi_z : ext[3];
        }
    }
    if (tile_off)
        for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_tiles; t += stride) {
            const double len = (double)(tile_off[t + 1] - tile_off[t]);
            longest = (len > longest) ? len : longest;
        }
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const double o = __shfl_xor(ext[k], off); ext[k] = (o > ext[k]) ? o : ext[k]; }
        const double ol = __shfl_xor(longest, off); longest = (ol > longest) ? ol : longest;
        const double of = __shfl_xor(full, off); full = (of > full) ? of : full;
        n_ok += __shfl_xor(n_ok, off);
    }
    // the four waves of the block combine in LDS, then one set of atomics per block
    __shared__ double s_part[4][7];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_part[wave][0] = longest; s_part[wave][1] = full; s_part[wave][2] = n_ok;
#pragma unroll
        for (int k = 0; k < 4; ++k) s_part[wave][3 + k] = ext[k];
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        const int k = threadIdx.x;
        double v = s_part[0][k];
        for (int w = 1; w < 4; ++w) v = (k == 2) ? v + s_part[w][k] : ((s_part[w][k] > v) ? s_part[w][k] : v);
        unsigned long long* bits = reinterpret_cast<unsigned long long*>(summary);
        const int slot = 1 + k;                              // longest -> [1], full -> [2], n_ok -> [3], ext -> [4..7]
        if (v > 0.0) {
            if (k == 2) atomicAdd(&summary[slot], v);
            else atomicMax(&bits[slot], (unsigned long long)__double_as_longlong(v));
        }
    }
    if (tile_off && blockIdx.x == 0 && threadIdx.x == 0) summary[0] = (double)tile_off[n_tiles];
}

__global__ void __launch_bounds__(256) rt2d_compact_kernel(const int64_t* off_src, const int64_t* off_dst, int64_t n_tiles,
                                                           const double* X0, const double* Z0, const double* C0,
                                                           double* X1, double* Z1, double* C1) {
    const int64_t t = blockIdx.x;
    if (t >= n_tiles) return;
    const int64_t s0 = off_src[t], d0 = off_dst[t], len = off_dst[t + 1] - d0;
    for (int64_t i = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; i < len; i += (int64_t)gridDim.y * blockDim.x) {
        X1[d0 + i] = X0[s0 + i];
        Z1[d0 + i] = Z0[s0 + i];
        if (C0) C1[d0 + i] = C0[s0 + i];
    }
}

// ------------------------------------------------------------------------------------------------------
// post-pass arrays (ray.py:271-275), one lane per ray, coalesced over the tile layout
struct PostArgs {
    DevEnv env; const double* blob;
    const double* X; const double* Z; const int64_t* tile_off; const int32_t* n_pts; int64_t n;
    double* dL; double* L; double* C; double* dT; double* T;
};

__global__ void __launch_bounds__(256) rt2d_post_kernel(const PostArgs a) {
    extern __shared__ __attribute__((aligned(16))) double s_tab[];
    const int cz_doubles = a.env.cz_doubles;           // the fused c/dz_c table is the blob's first table
    for (int i = threadIdx.x; i < cz_doubles; i += blockDim.x) s_tab[i] = a.blob[i];
    __syncthreads();
    const double* tab = s_tab;
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n) return;
    const DevTable& cz = a.env.cz;
    const int np = rt2d_stored_pts(a.n_pts, a.tile_off, r);
    const int64_t base = a.tile_off[r >> 6] + (r & 63);
    double xp = a.X[base], zp = a.Z[base];
    double cp = rt2d_eval_c(tab, cz, zp);
    double L = 0.0, T = 0.0;
    if (a.dL) a.dL[base] = 0.0;
    if (a.L) a.L[base] = 0.0;
    if (a.C) a.C[base] = cp;
    if (a.dT) a.dT[base] = 0.0;
    if (a.T) a.T[base] = 0.0;
    for (int s = 1; s < np; ++s) {
        int64_t o = base + (int64_t)s * RT2D_WAVE;
        double x = a.X[o], z = a.Z[o];
        double ddx = x - xp, ddz = z - zp;
        double dL = sqrt(ddx * ddx + ddz * ddz);
        double dT = dL / cp;
        L += dL; T += dT;
        cp = rt2d_eval_c(tab, cz, z);
        if (a.dL) a.dL[o] = dL;
        if (a.L) a.L[o] = L;
        if (a.C) a.C[o] = cp;
        if (a.dT) a.dT[o] = dT;
        if (a.T) a.T[o] = T;
        xp = x; zp = z;
    }
}

// ------------------------------------------------------------------------------------------------------
// gain (ray.py:339-369)
struct GainArgs {
    DevEnv env; const double* blob;
    const double* X; const double* Z; const int64_t* tile_off; const int32_t* n_pts;
    const int32_t* n_rebounds; const int32_t* reb_step; const uint8_t* reb_surface;
    const double* reb_velocity; const double* reb_angle; const double* reb_pos_z;
    int32_t rebound_cap; int64_t n;
    const double* freqs; int32_t n_freqs; int64_t total;
    double* G; double* G_end;
    const double* loss;            // optional: rebound losses from rt2d_reb_loss_kernel, [(f * rebound_cap + k) * n + ray]
    const int32_t* g_slot;         // optional (rt2d_gain_heatmap_rest): G holds only some rays, ray r at G[g_slot[r] * g_room + step]
    int64_t g_room;
};

// Rebound losses 10*log10(refcoef) (ray.py:350-362, model_reflection.py:11-46) of every logged rebound at every
// frequency, one thread per (ray, rebound).  The sediment coefficient carries a sincos with large-argument reduction, a
// hypot pair and a log10: evaluated inline in the gain loop they cost that kernel half of its registers (162 against 86)
// and with them two of five waves per SIMD -- for a few dozen evaluations per ray.  Same expressions, same bits.
__global__ void __launch_bounds__(256) rt2d_reb_loss_kernel(const GainArgs a, double* loss, double one_freq) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.n * a.rebound_cap) return;
    const int64_t r = idx % a.n;
    const int k = (int)(idx / a.n);
    if (k >= min(a.n_rebounds[r], a.rebound_cap)) return;
    const double fr = a.freqs ? a.freqs[blockIdx.y] : one_freq;       // a single frequency may come by value
    const double* gtab = a.blob + a.env.gain_off;
    const int64_t kk = (int64_t)k * a.n + r;
    const double wavelength = a.reb_velocity[kk] / fr;
    const double rc = (a.reb_surface[kk] == RT2D_SURFACE_SEDIMENT)
        ? rt2d_refcoef_sediment(wavelength, rt2d_eval_quad(gtab, a.env.imp, a.reb_pos_z[kk], nullptr))
        : rt2d_refcoef_surface(wavelength, a.reb_angle[kk]);
    loss[((int64_t)blockIdx.y * a.rebound_cap + k) * a.n + r] = 10 * log10(rc);
}

// Rays with hundreds or thousands of rebounds (no rebound limit at the reference's default step sizes: a ray that hugs the
// surface rebounds every few steps) have a rebound log too long for the LDS of the scan kernels.  Their losses are ADDED
// INTO the per-point dG array instead -- the loss of rebound k at point reb_step + 1, the first point that carries it
// (ray.py:360-362), one thread per (ray, rebound, frequency); two rebounds never share a step -- and the scan then runs
// without a log.  Same sum, another order of additions (the scan kernels' own order is not the reference's either).
__global__ void __launch_bounds__(256) rt2d_reb_scatter_kernel(const GainArgs a, const double* loss) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.n * a.rebound_cap) return;
    const int64_t r = idx % a.n;
    const int k = (int)(idx / a.n);
    if (k >= min(a.n_rebounds[r], a.rebound_cap)) return;
    const int at = a.reb_step[(int64_t)k * a.n + r] + 1;
    if (at >= rt2d_stored_pts(a.n_pts, a.tile_off, r)) return;
    const int64_t o = a.tile_off[r >> 6] + (r & 63) + (int64_t)at * RT2D_WAVE;
    a.G[(int64_t)blockIdx.y * a.total + o] += loss[((int64_t)blockIdx.y * a.rebound_cap + k) * a.n + r];
}

template <int FCH, bool LOSS, bool FULL>       // FULL: G is written and every group of the launch has FCH frequencies
__global__ void __launch_bounds__(256, (FCH == 8 && FULL) ? 3 : 1) rt2d_gain_kernel(const GainArgs a) {
    extern __shared__ __attribute__((aligned(16))) double s_tab[];
    for (int i = threadIdx.x; i < a.env.gain_doubles; i += blockDim.x) s_tab[i] = a.blob[a.env.gain_off + i];
    __syncthreads();
    const double* tab = s_tab;                 // gain-section offsets are relative to gain_off
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n) return;
    const int f0 = blockIdx.y * FCH;
    const int nf = FULL ? FCH : min(FCH, a.n_freqs - f0);
    const DevTable& ab = a.env.absorb;
    const int np = rt2d_stored_pts(a.n_pts, a.tile_off, r);
    const int64_t base = a.tile_off[r >> 6] + (r & 63);

    double fr[FCH], ff[FCH], S[FCH], A[FCH];
#pragma unroll
    for (int k = 0; k < FCH; ++k) {
        fr[k] = (k < nf) ? a.freqs[f0 + k] : 1.0;
        double f = fr[k] / 1000;                                           // model_absorption.py:32
        ff[k] = f * f;
        S[k] = 0.0; A[k] = 0.0;
    }
    const int nreb = min(a.n_rebounds[r], a.rebound_cap);
    int next_reb = 0;
    int next_reb_at = (nreb > 0) ? a.reb_step[r] + 1 : 0x7fffffff;         // first point index that carries the loss

    double xp = a.X[base], zp = a.Z[base];
    if (FULL || a.G) {
#pragma unroll
        for (int k = 0; k < FCH; ++k) if (k < nf) a.G[(int64_t)(f0 + k) * a.total + base] = 0.0;
    }
    // the next GAIN_AHEAD points of the ray are in flight while one is worked on (a slot is refilled as soon as it is read).
    // Measured on 1M rays: 1 ahead 6.9 ms, 4 ahead 8.4 ms -- the kernel is bound by its ~130 instructions per step, not by
    // HBM latency, and the extra registers cost occupancy.
    // The next point is loaded one iteration ahead, at the TOP of the iteration (the empty asm keeps the loads above the
    // stores): on this architecture stores count in the same counter as loads and complete in order, so a load issued
    // behind the n_st stores of an iteration is only back when those have reached memory.  Nothing is outstanding when the
    // loop is entered, so that the wait the compiler places at the first use is counted from the loop's own operations.
    auto load_point = [&](int row, double& xo, double& zo) {
        const int64_t on = base + (int64_t)min(row, max(np, 1) - 1) * RT2D_WAVE;      // (clamped into the ray: unconditional)
        xo = a.X[on]; zo = a.Z[on];
    };
    double xq, zq;
    load_point(1, xq, zq);
    __builtin_amdgcn_s_waitcnt(0x0F70);                                   // vmcnt(0)
    constexpr int GAIN_AHEAD = 1;
    for (int s0 = 1; s0 < np; s0 += GAIN_AHEAD) {
      {
        const int s = s0;
        int64_t o = base + (int64_t)s * RT2D_WAVE;
        double x = xq, z = zq;
        load_point(s + 1, xq, zq);
        asm volatile("" ::: "memory");
        double ddx = x - xp, ddz = z - zp;
        double dL = sqrt(ddx * ddx + ddz * ddz);
        // six depth factors at the START point of the segment (ray.py:346 uses XZ[:-1, 1])
        double rec[20];
        rt2d_lookup<19>(tab, ab, zp, rec);
        double d = zp - rec[0];
        double f1 = rt2d_quad_of(rec, d, 0), f2 = rt2d_quad_of(rec, d, 1);
        double A2 = rt2d_quad_of(rec, d, 2), E1 = rt2d_quad_of(rec, d, 3);
        double E2 = rt2d_quad_of(rec, d, 4), E3 = rt2d_quad_of(rec, d, 5);
        double f1s = f1 * f1, f2s = f2 * f2;
        while (s >= next_reb_at) {                                          // ray.py:350-362
            int64_t kk = (int64_t)next_reb * a.n + r;
            if (LOSS) {
#pragma unroll
                for (int k = 0; k < FCH; ++k)
                    if (k < nf) A[k] += a.loss[((int64_t)(f0 + k) * a.rebound_cap + next_reb) * a.n + r];
            } else {
                double vel = a.reb_velocity[kk];
                bool sediment = a.reb_surface[kk] == RT2D_SURFACE_SEDIMENT;
                double Zp0 = 0.0, ang = 0.0;
                if (sediment) Zp0 = rt2d_eval_quad(tab, a.env.imp, a.reb_pos_z[kk], nullptr);
                else ang = a.reb_angle[kk];
#pragma unroll
                for (int k = 0; k < FCH; ++k) {
                    double wavelength = vel / fr[k];
                    double rc = sediment ? rt2d_refcoef_sediment(wavelength, Zp0) : rt2d_refcoef_surface(wavelength, ang);
                    A[k] += 10 * log10(rc);
                }
            }
            ++next_reb;
            next_reb_at = (next_reb < nreb) ? a.reb_step[(int64_t)next_reb * a.n + r] + 1 : 0x7fffffff;
        }
#pragma unroll
        for (int k = 0; k < FCH; ++k) {
            double alpha_1 = rt2d_div_fast(k_A1 * (f1 * ff[k]), f1s + ff[k]);          // model_absorption.py:37-44 (the IEEE quotients: rt2d_div_fast)
            double alpha_2 = rt2d_div_fast(A2 * (f2 * ff[k]), f2s + ff[k]);
            double alpha_3 = k_A3 * ff[k];
            double alpha = alpha_1 * E1 + alpha_2 * E2 + alpha_3 * E3;
            double dG = rt2d_div_1000(-1 * alpha) * dL;                            // ray.py:346
            S[k] += dG;                                                    // ray.py:347
            if (FULL || (a.G && k < nf)) a.G[(int64_t)(f0 + k) * a.total + o] = S[k] + A[k];
        }
        xp = x; zp = z;
      }
    }
    if (a.G_end) {
#pragma unroll
        for (int k = 0; k < FCH; ++k) if (k < nf) a.G_end[(int64_t)(f0 + k) * a.n + r] = S[k] + A[k];
    }
}

// Small fans: with one lane per ray the loop above is a serial chain of n_pts table evaluations on a handful of
// waves.  Two phases instead, same expressions and same summation order (so the results are bit-identical):
//   (1) rt2d_gain_dg_kernel    dG of EVERY segment, one lane per point, all CUs busy; written into G
//   (2) rt2d_gain_scan_kernel  per (ray, frequency) the running sum over G in place + rebound losses: the serial chain
//                              is one add per point, the loads are issued eight points ahead.
template <int FCH>
__global__ void __launch_bounds__(256) rt2d_gain_dg_kernel(const GainArgs a) {
    extern __shared__ __attribute__((aligned(16))) double s_tab[];
    for (int i = threadIdx.x; i < a.env.gain_doubles; i += blockDim.x) s_tab[i] = a.blob[a.env.gain_off + i];
    __syncthreads();
    const double* tab = s_tab;
    const int64_t tile = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int64_t r = tile * RT2D_WAVE + lane;
    const int np = (r < a.n) ? rt2d_stored_pts(a.n_pts, a.tile_off, r) : 0;
    int tmax = np;                                     // the tile's longest ray: a padded layout has more room than points
    for (int off = 32; off > 0; off >>= 1) tmax = max(tmax, __shfl_xor(tmax, off));
    const int64_t t0 = a.tile_off[tile], len = min(a.tile_off[tile + 1] - t0, (int64_t)tmax * RT2D_WAVE);
    const int f0 = blockIdx.z * FCH;
    const int nf = min(FCH, a.n_freqs - f0);
    double ff[FCH];
#pragma unroll
    for (int k = 0; k < FCH; ++k) { double f = ((k < nf) ? a.freqs[f0 + k] : 1.0) / 1000; ff[k] = f * f; }
    const DevTable& ab = a.env.absorb;
    for (int64_t i = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; i < len; i += (int64_t)gridDim.y * blockDim.x) {
        const int s = (int)(i >> 6);
        if (s >= np) continue;
        const int64_t o = t0 + i;
        if (s == 0) {
#pragma unroll
            for (int k = 0; k < FCH; ++k) if (k < nf) a.G[(int64_t)(f0 + k) * a.total + o] = 0.0;
            continue;
        }
        double x = a.X[o], z = a.Z[o], xp = a.X[o - RT2D_WAVE], zp = a.Z[o - RT2D_WAVE];
        double ddx = x - xp, ddz = z - zp;
        double dL = sqrt(ddx * ddx + ddz * ddz);
        double rec[20];
        rt2d_lookup<19>(tab, ab, zp, rec);
        double d = zp - rec[0];
        double f1 = rt2d_quad_of(rec, d, 0), f2 = rt2d_quad_of(rec, d, 1);
        double A2 = rt2d_quad_of(rec, d, 2), E1 = rt2d_quad_of(rec, d, 3);
        double E2 = rt2d_quad_of(rec, d, 4), E3 = rt2d_quad_of(rec, d, 5);
        double f1s = f1 * f1, f2s = f2 * f2;
#pragma unroll
        for (int k = 0; k < FCH; ++k) {
            double alpha_1 = rt2d_div_fast(k_A1 * (f1 * ff[k]), f1s + ff[k]);          // model_absorption.py:37-44 (the IEEE quotients: rt2d_div_fast)
            double alpha_2 = rt2d_div_fast(A2 * (f2 * ff[k]), f2s + ff[k]);
            double alpha_3 = k_A3 * ff[k];
            double alpha = alpha_1 * E1 + alpha_2 * E2 + alpha_3 * E3;
            double dG = rt2d_div_1000(-1 * alpha) * dL;                            // ray.py:346
            if (k < nf) a.G[(int64_t)(f0 + k) * a.total + o] = dG;
        }
    }
}

// One workgroup of 16 waves per (tile, frequency).  All waves move the tile through LDS in chunks of
// GAIN_SCAN_ROWS steps with coalesced loads and stores (the next chunk is already in flight in registers); wave 0
// alone runs the 64 running sums over the chunk in LDS, so the serial chain never waits on global memory.
#define GAIN_SCAN_ROWS 64
#define GAIN_SCAN_THREADS 1024
#define GAIN_SCAN_PER_THREAD (GAIN_SCAN_ROWS * RT2D_WAVE / GAIN_SCAN_THREADS)
__global__ void __launch_bounds__(GAIN_SCAN_THREADS) rt2d_gain_scan_kernel(const GainArgs a) {
    extern __shared__ __attribute__((aligned(16))) double s_buf[];          // [ROWS][64] chunk, then the loss table
    double* s_loss = s_buf + 3 * GAIN_SCAN_ROWS * RT2D_WAVE;               // [cap][64] cumulative rebound loss (behind the three chunk buffers)
    int* s_at = reinterpret_cast<int*>(s_loss + (size_t)a.rebound_cap * RT2D_WAVE);   // [cap][64] first point carrying it
    const int lane = threadIdx.x & 63;
    const bool scanner = threadIdx.x < RT2D_WAVE;
    const int64_t tile = blockIdx.x;
    const int64_t r = tile * RT2D_WAVE + lane;
    const int fi = blockIdx.y;
    const int64_t t0 = a.tile_off[tile];
    int64_t len = a.tile_off[tile + 1] - t0;                                // room of the tile ...
    {                                                                       // ... cut to 64 * steps of its longest ray
        int tmax = (r < a.n) ? rt2d_stored_pts(a.n_pts, a.tile_off, r) : 0;
        for (int off = 32; off > 0; off >>= 1) tmax = max(tmax, __shfl_xor(tmax, off));
        len = min(len, (int64_t)max(tmax, 1) * RT2D_WAVE);
    }
    double* G = a.G + (int64_t)fi * a.total + t0;
    // rebound losses of the tile's rays (ray.py:350-362): every (rebound, ray) pair on its own thread -- each costs a
    // handful of dependent loads, a sincos, a log10 -- then wave 0 turns them into running sums per ray
    const int nreb = (r < a.n) ? min(a.n_rebounds[r], a.rebound_cap) : 0;
    {
        const double fr = a.freqs[fi];
        const double* gtab = a.blob + a.env.gain_off;
        for (int k = threadIdx.x >> 6; k < a.rebound_cap; k += GAIN_SCAN_THREADS / RT2D_WAVE) {
            if (k < nreb) {
                int64_t kk = (int64_t)k * a.n + r;
                double wavelength = a.reb_velocity[kk] / fr;
                double rc = (a.reb_surface[kk] == RT2D_SURFACE_SEDIMENT)
                    ? rt2d_refcoef_sediment(wavelength, rt2d_eval_quad(gtab, a.env.imp, a.reb_pos_z[kk], nullptr))
                    : rt2d_refcoef_surface(wavelength, a.reb_angle[kk]);
                s_loss[k * RT2D_WAVE + lane] = 10 * log10(rc);
                s_at[k * RT2D_WAVE + lane] = a.reb_step[kk] + 1;
            }
        }
    }
    __syncthreads();
    const int np = (r < a.n) ? rt2d_stored_pts(a.n_pts, a.tile_off, r) : 0;          // of the ray in this thread's column (lane == column for every element)
    if (scanner) {
        double A = 0.0;
        for (int k = 0; k < nreb; ++k) { A += s_loss[k * RT2D_WAVE + lane]; s_loss[k * RT2D_WAVE + lane] = A; }
    }
    __syncthreads();
    // Division of labour, pipelined over three LDS buffers: the fifteen loader waves fill chunk k (zeroing the entries past
    // a ray's end) and write chunk k - 2 out with the rebound loss A of each point added (G = S + A as in the one-pass
    // kernel), while wave 0 runs the bare running sums S of chunk k - 1 (one dependent add per point: the reference's
    // cumsum, in its order) -- one barrier per chunk, and the sums never wait for memory.  A loader thread meets its
    // column's points in increasing order, so the loss in force is a cursor into the column's table.
    int next_reb = 0;
    int next_reb_at = (nreb > 0) ? s_at[lane] : 0x7fffffff;
    double A = 0.0, S = 0.0;
    constexpr int LOADERS = GAIN_SCAN_THREADS / RT2D_WAVE - 1;              // waves 1 .. 15
    constexpr int ROWS_PER_LOADER = (GAIN_SCAN_ROWS + LOADERS - 1) / LOADERS;
    const int lw = (threadIdx.x >> 6) - 1;                                  // loader index, -1 for the scanner wave
    const int64_t chunk = (int64_t)GAIN_SCAN_ROWS * RT2D_WAVE;
    const int64_t last = len - 1;
    const int n_chunks = (int)((len + chunk - 1) / chunk);
    double regs[ROWS_PER_LOADER];
    if (!scanner) {
#pragma unroll
        for (int k = 0; k < ROWS_PER_LOADER; ++k)
            regs[k] = G[min((int64_t)(lw + LOADERS * k) * RT2D_WAVE + lane, last)];
    }
    for (int it = 0; it < n_chunks + 2; ++it) {
        if (!scanner) {
            if (it < n_chunks) {
                double* buf = s_buf + (it % 3) * chunk;
                const int s_first = it * GAIN_SCAN_ROWS;
#pragma unroll
                for (int k = 0; k < ROWS_PER_LOADER; ++k) {
                    const int row = lw + LOADERS * k;
                    const int sx = s_first + row;
                    if (row < GAIN_SCAN_ROWS) buf[row * RT2D_WAVE + lane] = (sx >= 1 && sx < np) ? regs[k] : 0.0;
                }
#pragma unroll
                for (int k = 0; k < ROWS_PER_LOADER; ++k)                  // the next chunk on its way during the sums
                    regs[k] = G[min((int64_t)(it + 1) * chunk + (int64_t)(lw + LOADERS * k) * RT2D_WAVE + lane, last)];
            }
            if (it >= 2) {
                const double* buf = s_buf + ((it - 2) % 3) * chunk;
                const int s_first = (it - 2) * GAIN_SCAN_ROWS;
#pragma unroll
                for (int k = 0; k < ROWS_PER_LOADER; ++k) {
                    const int row = lw + LOADERS * k;
                    const int sx = s_first + row;
                    if (row < GAIN_SCAN_ROWS && sx < np) {
                        while (sx >= next_reb_at) {
                            A = s_loss[next_reb * RT2D_WAVE + lane];
                            ++next_reb;
                            next_reb_at = (next_reb < nreb) ? s_at[next_reb * RT2D_WAVE + lane] : 0x7fffffff;
                        }
                        G[(int64_t)sx * RT2D_WAVE + lane] = buf[row * RT2D_WAVE + lane] + A;
                    }
                }
            }
        } else if (it >= 1 && it <= n_chunks) {
            double* buf = s_buf + ((it - 1) % 3) * chunk;
            for (int rb = 0; rb < GAIN_SCAN_ROWS; rb += 16) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = buf[(rb + u) * RT2D_WAVE + lane];
#pragma unroll
                for (int u = 0; u < 16; ++u) { S += v[u]; v[u] = S; }       // ray.py:347
#pragma unroll
                for (int u = 0; u < 16; ++u) buf[(rb + u) * RT2D_WAVE + lane] = v[u];
            }
        }
        __syncthreads();
    }
    if (scanner && r < a.n && a.G_end) {
        double A_end = 0.0;                              // loss in force at the last point
        for (int k = 0; k < nreb && s_at[k * RT2D_WAVE + lane] <= np - 1; ++k) A_end = s_loss[k * RT2D_WAVE + lane];
        a.G_end[(int64_t)fi * a.n + r] = S + A_end;
    }
}


// The same running sums for fans with few (tile, frequency) pairs, where rt2d_gain_scan_kernel would leave most of the chip
// idle (999 rays at one frequency: 16 workgroups): a workgroup takes GAIN_PART_RAYS rays of a tile instead of all 64, so
// that a chunk (of 1024 / GAIN_PART_RAYS rows) is ONE element per thread (the fifteen loader waves of the kernel above issue five rows each per
// chunk, and their instructions, not memory, are what a chunk waits for) and 64 / GAIN_PART_RAYS times as many CUs take part.  The first
// GAIN_PART_RAYS threads run the sums besides their share of the loading.  Same additions in the same order.
#define GAIN_PART_RAYS 8        // (999 rays, one frequency: 0.808 / 0.795 / 0.790 / 0.791 / 0.821 ms per heat map with 32 / 16 / 8 / 4 / 2)
__global__ void __launch_bounds__(GAIN_SCAN_THREADS) rt2d_gain_scan_part_kernel(const GainArgs a) {
    constexpr int R = GAIN_PART_RAYS, ROWS = GAIN_SCAN_THREADS / R;
    extern __shared__ __attribute__((aligned(16))) double s_buf[];          // three [ROWS][R] chunks, then the loss table
    double* s_loss = s_buf + 3 * ROWS * R;                                  // [cap][R] cumulative rebound loss
    int* s_at = reinterpret_cast<int*>(s_loss + (size_t)a.rebound_cap * R); // [cap][R] first point carrying it
    const int ray_l = threadIdx.x & (R - 1), row_l = threadIdx.x / R;
    const bool scanner = threadIdx.x < R;
    const int col = (int)blockIdx.z * R + ray_l;                            // the ray's lane in its tile
    const int64_t tile = blockIdx.x;
    const int64_t r = tile * RT2D_WAVE + col;
    const int fi = blockIdx.y;
    const int64_t t0 = a.tile_off[tile];
    const int np = (r < a.n) ? rt2d_stored_pts(a.n_pts, a.tile_off, r) : 0;
    int n_rows = np;                                                        // rows of the longest of these R rays
    for (int off = R / 2; off > 0; off >>= 1) n_rows = max(n_rows, __shfl_xor(n_rows, off));
    n_rows = max(n_rows, 1);
    double* G = a.G + (int64_t)fi * a.total + t0;
    const int nreb = (r < a.n) ? min(a.n_rebounds[r], a.rebound_cap) : 0;
    {
        const double fr = a.freqs[fi];
        const double* gtab = a.blob + a.env.gain_off;
        for (int k = row_l; k < a.rebound_cap; k += ROWS) {
            if (k < nreb) {
                int64_t kk = (int64_t)k * a.n + r;
                double wavelength = a.reb_velocity[kk] / fr;
                double rc = (a.reb_surface[kk] == RT2D_SURFACE_SEDIMENT)
                    ? rt2d_refcoef_sediment(wavelength, rt2d_eval_quad(gtab, a.env.imp, a.reb_pos_z[kk], nullptr))
                    : rt2d_refcoef_surface(wavelength, a.reb_angle[kk]);
                s_loss[k * R + ray_l] = 10 * log10(rc);
                s_at[k * R + ray_l] = a.reb_step[kk] + 1;
            }
        }
    }
    __syncthreads();
    if (scanner) {
        double A = 0.0;
        for (int k = 0; k < nreb; ++k) { A += s_loss[k * R + ray_l]; s_loss[k * R + ray_l] = A; }
    }
    __syncthreads();
    int next_reb = 0;
    int next_reb_at = (nreb > 0) ? s_at[ray_l] : 0x7fffffff;
    double A = 0.0, S = 0.0;
    const int n_chunks = (n_rows + ROWS - 1) / ROWS;
    double reg = G[(int64_t)min(row_l, n_rows - 1) * RT2D_WAVE + col];
    for (int it = 0; it < n_chunks + 2; ++it) {
        if (it < n_chunks) {
            const int sx = it * ROWS + row_l;
            s_buf[(it % 3) * (ROWS * R) + row_l * R + ray_l] = (sx >= 1 && sx < np) ? reg : 0.0;
            reg = G[(int64_t)min((it + 1) * ROWS + row_l, n_rows - 1) * RT2D_WAVE + col];     // the next chunk on its way
        }
        if (it >= 2) {
            const int sx = (it - 2) * ROWS + row_l;
            if (sx < np) {
                while (sx >= next_reb_at) {
                    A = s_loss[next_reb * R + ray_l];
                    ++next_reb;
                    next_reb_at = (next_reb < nreb) ? s_at[next_reb * R + ray_l] : 0x7fffffff;
                }
                G[(int64_t)sx * RT2D_WAVE + col] = s_buf[((it - 2) % 3) * (ROWS * R) + row_l * R + ray_l] + A;
            }
        }
        if (scanner && it >= 1 && it <= n_chunks) {
            double* buf = s_buf + ((it - 1) % 3) * (ROWS * R);
            for (int rb = 0; rb < ROWS; rb += 16) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = buf[(rb + u) * R + ray_l];
#pragma unroll
                for (int u = 0; u < 16; ++u) { S += v[u]; v[u] = S; }       // ray.py:347
#pragma unroll
                for (int u = 0; u < 16; ++u) buf[(rb + u) * R + ray_l] = v[u];
            }
        }
        __syncthreads();
    }
    if (scanner && r < a.n && a.G_end) {
        double A_end = 0.0;                              // loss in force at the last point
        for (int k = 0; k < nreb && s_at[k * R + ray_l] <= np - 1; ++k) A_end = s_loss[k * R + ray_l];
        a.G_end[(int64_t)fi * a.n + r] = S + A_end;
    }
}


// Which kernel bins which ray of a fan for the one-pass heat map (RayFan.fused_mask, evaluated on the device so that the
// host need not wait for the cast): m[r] = 1 for the rays rt2d_gain_heatmap_kernel can bin (no error, x never decreases,
// bounding box inside the map, at most 8 rows per column by the slope bound), 2 for the other error-free rays, 0 for the
// rest.  The rays of value 2 are appended to `list` ([0] = their count, [2 + k] = ray; a wave's rays stay adjacent) and
// get row k of the compact gain array (g_slot[r] = k); beyond `rows` rows there is none (g_slot = -1, *overflow = 1: the
// map will be incomplete and the caller, who reads the flag with the fan's summary, builds it again another way).
__global__ void __launch_bounds__(256) rt2d_classify_kernel(const int32_t* status, const uint8_t* monotone, const double* range,
                                                            const double* slope_max, const uint8_t* ray_mask, int64_t n,
                                                            double res_x, double res_z, double nx, double nz, int64_t rows,
                                                            uint8_t* m, int32_t* g_slot, int32_t* list, double* overflow) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool has = r < n;
    const int64_t q = has ? r : n - 1;
    const bool sel = has && status[q] == 0 && (!ray_mask || ray_mask[q] != 0);
    const bool ok = sel && monotone[q] != 0 && range[q] > -1 && range[3 * n + q] < 1
                 && floor(trunc(range[2 * n + q]) / res_x) < nx && floor(-trunc(range[n + q]) / res_z) < nz
                 && floor(slope_max[q] * (res_x / res_z)) + 2 <= 8;
    const bool rest = sel && !ok;
    const unsigned long long votes = __builtin_amdgcn_ballot_w64(rest);
    int base = 0;
    if (votes != 0ull) {
        const int first = (int)__builtin_ctzll(votes);
        if (lane == first) base = atomicAdd(&list[0], (int)__builtin_popcountll(votes));
        base = __builtin_amdgcn_readlane(base, first);
    }
    if (rest) {
        const int k = base + (int)__builtin_popcountll(votes & ((1ull << lane) - 1ull));
        list[2 + k] = (int32_t)r;
        g_slot[r] = k < rows ? k : -1;
        if (k >= rows) *overflow = 1.0;
    }
    if (has) m[r] = ok ? 1 : (rest ? 2 : 0);
}

// Gain at ONE frequency and heat-map binning in one forward pass over stored trajectories: the gain of a point is used
// the moment it exists and never written (rt2d_gain + rt2d_heatmap_accumulate move 8 B per point out and 24 B back in).
// Same gain expressions as rt2d_gain_kernel, same binning as the fused stepping kernel (RayBinner).  ray_mask: 1 = bin
// here (the ray meets RayBinner's preconditions), 2 = only compute the gain and store it to G (the caller bins such a
// ray with the general kernel), 0 = skip.
struct GainHeatArgs {
    GainArgs g; BinTarget bin; double freq;
    const uint8_t* ray_mask; uint8_t* irregular; unsigned long long* n_violations;
    const int32_t* ray_list;       // optional: the rays of this launch ([0] = count, [2 + k] = ray), all of them gain-only (value 2)
    int32_t exact;                 // 1: never the fast forms of the gain (RT2D_GH_EXACT=1, tests)
};

template <bool LOSS, bool FAST = false>
__global__ void __launch_bounds__(512, FAST ? 3 : LOSS ? 4 : 2) rt2d_gain_heatmap_kernel(const GainHeatArgs b) {
    const GainArgs& a = b.g;
    // FAST: this launch only bins (no gain is stored, none asked for as a per-ray result): the fast forms of the gain
    // (rt2d_alpha_fast).  A gain that leaves the kernel is the gain rt2d_gain gives (exact forms).
    extern __shared__ __attribute__((aligned(16))) double s_tab[];
    for (int i = threadIdx.x; i < a.env.gain_doubles; i += blockDim.x) s_tab[i] = a.blob[a.env.gain_off + i];
    __syncthreads();
    const double* tab = s_tab;
    const int lane = threadIdx.x & 63;
    // All 64 lanes of a wave stay in the point loop until the wave's longest ray has ended -- a lane without a ray, with
    // a masked ray or past its ray's end carries live = false and computes on stale values -- so that the deposits run
    // with every lane enabled (rt2d_deposit_full) and the loop has one wave-uniform exit.
    int64_t r_raw = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b.ray_list) r_raw = r_raw < (int64_t)b.ray_list[0] ? (int64_t)b.ray_list[2 + r_raw] : a.n;
    const bool has = r_raw < a.n;
    const int64_t r = has ? r_raw : a.n - 1;
    const uint8_t flags = !has ? (uint8_t)0 : b.ray_list ? (uint8_t)2 : (b.ray_mask ? b.ray_mask[r] : (uint8_t)1);
    const bool binme = (flags & 1) != 0;
    const DevTable& ab = a.env.absorb;
    // (a ray that is neither binned here nor has anywhere to put its gain has nothing to do in this launch)
    const int np = (binme || (flags && (a.G != nullptr || a.G_end != nullptr))) ? rt2d_stored_pts(a.n_pts, a.tile_off, r) : 0;
    if (!rt2d_any(np > 0)) return;
    const int64_t base = a.tile_off[r >> 6] + (r & 63);
    const double fr = b.freq;
    const double f = fr / 1000;                                            // model_absorption.py:32
    const double ff = f * f;
    double S = 0.0, A = 0.0;
    const int nreb = flags ? min(a.n_rebounds[r], a.rebound_cap) : 0;
    int next_reb = 0;
    int next_reb_at = (nreb > 0) ? a.reb_step[r] + 1 : 0x7fffffff;         // first point index that carries the loss
    double xp = 0.0, zp = 0.0;
    if (np > 0) { xp = a.X[base]; zp = a.Z[base]; }
    RayBinner binner;
    binner.start(b.bin, binme && np > 0, s_tab + a.env.gain_doubles + (threadIdx.x >> 6) * (8 * RT2D_WAVE) + lane, xp, zp, 0.0);
    // where this ray's gain goes when it is only computed here: its points of the tile layout, or its own row of a compact G
    const int32_t g_row = a.g_slot ? a.g_slot[r] : 0;
    const int64_t g_base = a.g_slot ? (int64_t)g_row * a.g_room : base;
    const int64_t g_step = a.g_slot ? 1 : RT2D_WAVE;
    const bool g_store = !binme && a.G != nullptr && g_row >= 0;
    if (np > 0 && g_store) a.G[g_base] = 0.0;
    double xn = xp, zn = zp;
    if (np > 1) { xn = a.X[base + RT2D_WAVE]; zn = a.Z[base + RT2D_WAVE]; }
    for (int s = 1; rt2d_any(s < np); ++s) {
        const bool live = s < np;
        const int64_t o = base + (int64_t)s * RT2D_WAVE;
        const double x = xn, z = zn;
        if (s + 1 < np) { xn = a.X[o + RT2D_WAVE]; zn = a.Z[o + RT2D_WAVE]; }   // next point in flight during this step
        double ddx = x - xp, ddz = z - zp;
        const double dL = FAST ? rt2d_sqrt_fast(fma(ddx, ddx, ddz * ddz)) : sqrt(ddx * ddx + ddz * ddz);
        double rec[20];
        rt2d_lookup<19>(tab, ab, zp, rec);                                  // factors at the START of the segment, ray.py:346
        while (live && s >= next_reb_at) {                                  // ray.py:350-362
            int64_t kk = (int64_t)next_reb * a.n + r;
            if (LOSS) {
                A += a.loss[kk];
            } else {
                double wavelength = a.reb_velocity[kk] / fr;
                double rc = (a.reb_surface[kk] == RT2D_SURFACE_SEDIMENT)
                    ? rt2d_refcoef_sediment(wavelength, rt2d_eval_quad(tab, a.env.imp, a.reb_pos_z[kk], nullptr))
                    : rt2d_refcoef_surface(wavelength, a.reb_angle[kk]);
                A += 10 * log10(rc);
            }
            ++next_reb;
            next_reb_at = (next_reb < nreb) ? a.reb_step[(int64_t)next_reb * a.n + r] + 1 : 0x7fffffff;
        }
        // ray.py:346; FAST: alpha / 1000 as a product with one correction step
        double dG;
        if (FAST) {
            const double al = -rt2d_alpha_fast(rec, zp - rec[0], ff);
            const double q = al * 0.001;
            dG = fma(fma(-q, 1000.0, al), 0.001, q) * dL;
        } else {
            dG = rt2d_div_1000(-1 * rt2d_alpha(rec, zp - rec[0], ff)) * dL;
        }
        if (live) S += dG;                                                 // ray.py:347
        const double g = S + A;
        binner.point<true>(b.bin, x, z, g, lane, live);                    // (a ray that is not binned here has ok = false)
        if (live && g_store) a.G[g_base + s * g_step] = g;
        xp = x; zp = z;
    }
    binner.finish<true>(b.bin, lane);
    if (!flags || np == 0) return;
    if (b.irregular) b.irregular[r] = binner.violated ? 1 : 0;
    if (binner.violated && b.n_violations) atomicAdd(b.n_violations, 1ull);
    if (a.G_end) a.G_end[r] = S + A;
}

// ------------------------------------------------------------------------------------------------------
// The one pass again, for the launch that ONLY bins (the big pass of rt2d_gain_heatmap_rest: no gain leaves the kernel), by
// instruction count (238 vector instructions per wave-point for ~40 flop of algorithm).
//  * -alpha(f, z) / 1000 of this launch's frequency comes from ONE polynomial of degree 7 per depth piece (AlphaDesc:
//    record = [lo, hi, c0 .. c7] about d = z - lo, same pieces and same O(1) guess as the absorption table), fitted on the
//    host from the exact form at 8 Chebyshev nodes per piece and VERIFIED ON THE DEVICE against the exact form
//    (rt2d_alpha_verify_kernel: 256 depths per piece, worst relative difference <= 1e-14, else the exact kernel runs):
//    5 LDS reads and 7 FMAs instead of a 19-double record, 6 quadratics, two quotients and the /1000.
//  * the common step -- every live ray of the wave has its depth in the guessed piece, and every binned ray stays in its
//    cell or leaves it (new column or new row: half of the C2 fan is steep rays that change ROW every step) for a cell
//    inside the map without a remembered deposit to take the difference to -- is a straight line under ONE wave vote;
//    a step that is not common runs the general code of the existing kernel (RayBinner::point, the exact alpha) in ONE
//    block outside the unrolled steps.  Rebound losses are added where they fall due without leaving the common step.
//  * trajectory addresses are a scalar tile base + lane: no per-point vector address arithmetic; points are loaded
//    ONEPASS_AHEAD steps ahead, unconditionally, from rows clamped into the wave's longest ray.
// Lanes past their ray's end compute on whatever the tile holds there (inside the allocation: a tile has room for its
// longest ray in every column); nothing of it reaches S of a live ray, the binner or the vote.
#define ALPHA_DEG 7
#define ALPHA_REC (ALPHA_DEG + 3)
struct AlphaDesc { const double* tab; int32_t n_pieces; int32_t pad; double x0, inv_h; };

__global__ void __launch_bounds__(64) rt2d_alpha_verify_kernel(DevEnv env, const double* blob, AlphaDesc ad, double freq,
                                                               unsigned long long* worst) {
    const int j = blockIdx.x, lane = threadIdx.x;
    const double* rec = ad.tab + (size_t)j * ALPHA_REC;
    const double lo = rec[0], hi = rec[1];
    const double f = freq / 1000, ff = f * f;
    double err = 0.0;
    for (int q = 0; q < 4; ++q) {
        const int k = q * 64 + lane;                                      // 0 .. 255
        double z = lo + (hi - lo) * ((double)k / 255.0);
        if (k == 255 || !(z < hi)) z = nextafter(hi, -INFINITY);             // just inside
        if (!(z >= lo)) z = lo;
        // the piece the lean kernel would use for this depth must be this one
        double g = (z - ad.x0) * ad.inv_h;
        int jj = (int)fmin(fmax(g, 0.0), (double)(ad.n_pieces - 1));
        if (jj != j) {
            const double* r2 = ad.tab + (size_t)jj * ALPHA_REC;
            if (z >= r2[0] && z < r2[1]) err = INFINITY;                  // two pieces claim the depth: never
            continue;                                                     // (a wrong guess: the kernel takes its general step)
        }
        double r19[20];
        rt2d_lookup<19>(blob + env.gain_off, env.absorb, z, r19);
        const double want = rt2d_div_1000(-1 * rt2d_alpha(r19, z - r19[0], ff));
        const double d = z - lo;
        double pz = rec[2 + ALPHA_DEG];
#pragma unroll
        for (int m = ALPHA_DEG - 1; m >= 0; --m) pz = fma(pz, d, rec[2 + m]);
        const double e = fabs(pz - want) / fabs(want);
        err = (e > err || e != e) ? (e != e ? INFINITY : e) : err;
    }
    for (int off = 32; off > 0; off >>= 1) err = fmax(err, __shfl_xor(err, off));
    if (lane == 0) atomicMax(worst, (unsigned long long)__double_as_longlong(err));     // non-negative doubles order as integers
}

// Points in flight per lane and waves per SIMD (1 M-ray C2 fan, kernel + 0.25 ms of rebound losses): at 139 registers
// (3 waves per SIMD) 6.24 / 6.12 / 6.22 / 6.28 ms with 3 / 4 / 6 / 8 points in flight; bounded to 128 registers (4 waves
// per SIMD) 5.70 / 5.65 / 6.51 ms with 2 / 3 / 4 (the fourth spills).  Before the general block became ONE copy outside
// the unrolled steps the same sweep read 8.3 / 7.0 / 6.6 / 8.0 ms with 1 / 2 / 4 / 6: first HBM latency, then code size.
// With the three-address exp2 (12 copies fewer per step, 174 vector instructions per wave-point by SQ_INSTS_VALU against
// 238 for rt2d_gain_heatmap_kernel<true, true>): 5.57 ms with 2 points in flight, 5.9 with 3 (two registers spilled).
#ifndef ONEPASS_AHEAD
#define ONEPASS_AHEAD 2
#endif
#ifndef ONEPASS_WPE
#define ONEPASS_WPE 4
#endif
// what a step computes before the wave votes (one copy per unrolled common step, one in the general block)
#define ONEPASS_HEAD(X_, Z_)                                                                                                        \
        const double x = (X_), z = (Z_);                                                                                           \
        const bool live = s < np;                                                                                                  \
        /* -alpha / 1000 at the START of the segment (ray.py:346) from the guessed piece */                                        \
        int j = (int)((zp - ad.x0) * ad.inv_h);                            /* saturating conversion, NaN -> 0 */                  \
        j = min(max(j, 0), ad.n_pieces - 1);                                                                                       \
        const double* rec = s_tab + j * ALPHA_REC;                                                                                 \
        const double lo = rec[0], hi = rec[1];                                                                                     \
        const double c0 = rec[2], c1 = rec[3], c2 = rec[4], c3 = rec[5], c4 = rec[6], c5 = rec[7], c6 = rec[8], c7 = rec[9];       \
        const bool piece_ok = (zp >= lo) & (zp < hi);                                                                              \
        const double d = zp - lo;                                                                                                  \
        double pa = fma(c7, d, c6);                                                                                                \
        pa = fma(pa, d, c5); pa = fma(pa, d, c4); pa = fma(pa, d, c3); pa = fma(pa, d, c2); pa = fma(pa, d, c1); pa = fma(pa, d, c0); \
        const double ddx = x - xp, ddz = z - zp;                                                                                   \
        const double dL = rt2d_sqrt_fast(fma(ddx, ddx, ddz * ddz));                                                                \
        /* rebound losses that start with this point (ray.py:350-362): rare, and no reason to leave the common step */             \
        if (rt2d_any(live & (s >= next_reb_at))) {                                                                                 \
            const GainHeatArgs* cold = (const GainHeatArgs*)__builtin_amdgcn_kernarg_segment_ptr();                                \
            asm volatile("" : "+s"(cold));                                                                                         \
            if (live) {                                                                                                            \
                const GainArgs& ca = cold->g;                                                                                      \
                while (s >= next_reb_at) {                                                                                         \
                    A += ca.loss[(int64_t)next_reb * ca.n + r];                                                                    \
                    ++next_reb;                                                                                                    \
                    next_reb_at = (next_reb < nreb) ? ca.reb_step[(int64_t)next_reb * ca.n + r] + 1 : 0x7fffffff;                  \
                }                                                                                                                  \
            }                                                                                                                      \
        }                                                                                                                          \
        const double Sn = fma(pa, dL, S);                                  /* ray.py:346-347 */                                   \
        /* the cell of the new point, and what the binner (RayBinner::point) would do with it */                                   \
        int ix, iz;                                                                                                                \
        const bool in_map = rt2d_cell_i32(geo, x, z, ix, iz) & ((unsigned)ix < (unsigned)nx) & ((unsigned)iz < (unsigned)nz);      \
        const bool newcol = ix != binner.cur_ix;                                                                                   \
        const bool newrow = !newcol & (iz != binner.cur_iz);                                                                       \
        const int k_old = binner.cur_iz & 7;                                                                                       \
        const int lo_n = min(binner.col_lo, iz), hi_n = max(binner.col_hi, iz);                                                    \
        /* common for a binned ray: inside the map, column not decreasing, the cell it leaves (if it leaves one) was not           \
           deposited before in this column (its slot bit is clear: nothing to take the difference to), the plain form of the       \
           value applies, and the rows of the column still fit the eight slots */                                                  \
        const bool easy = in_map & (ix >= binner.cur_ix) & !((newcol | newrow) & (((binner.occ >> k_old) & 1u) != 0u))             \
                          & (fabs(binner.cur_g) < 3000.0) & (newcol | (hi_n - lo_n < 8));                                          \
        const bool common = piece_ok & (!binner.ok | easy);                                                                        \
        const bool all_common = __builtin_amdgcn_ballot_w64(live & !common) == 0ull;

__attribute__((amdgpu_waves_per_eu(ONEPASS_WPE, ONEPASS_WPE)))
__global__ void __launch_bounds__(256) rt2d_onepass_kernel(const GainHeatArgs b, const AlphaDesc ad) {
    const GainArgs& a = b.g;
    extern __shared__ __attribute__((aligned(16))) double s_tab[];         // [n_pieces][ALPHA_REC], then 8 x 64 slots per wave
    const int tab_doubles = ad.n_pieces * ALPHA_REC;
    for (int i = threadIdx.x; i < tab_doubles; i += blockDim.x) s_tab[i] = ad.tab[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t r_raw = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool has = r_raw < a.n;
    const int64_t r = has ? r_raw : a.n - 1;
    const bool binme = has && ((b.ray_mask ? b.ray_mask[r] : (uint8_t)1) & 1) != 0;
    const int np = binme ? rt2d_stored_pts(a.n_pts, a.tile_off, r) : 0;
    int np_max = np;
    for (int off = 32; off > 0; off >>= 1) np_max = max(np_max, __shfl_xor(np_max, off));
    np_max = __builtin_amdgcn_readfirstlane(np_max);
    if (np_max == 0) return;
    // the wave's tile: scalar base, lane offset
    const int64_t tile = (int64_t)__builtin_amdgcn_readfirstlane((int)(r_raw >> 6));
    const int64_t t0 = a.tile_off[tile];
    const double* __restrict__ Xs = a.X + t0;
    const double* __restrict__ Zs = a.Z + t0;
    const unsigned ulane = (unsigned)lane;
    double* const H = b.bin.H;
    double S = 0.0, A = 0.0;
    const int nreb = binme ? min(a.n_rebounds[r], a.rebound_cap) : 0;
    int next_reb = 0;
    int next_reb_at = (nreb > 0) ? a.reb_step[r] + 1 : 0x7fffffff;         // first point index that carries the loss
    double xp = Xs[ulane], zp = Zs[ulane];
    RayBinner binner;
    binner.start(b.bin, binme && np > 0, s_tab + tab_doubles + (threadIdx.x >> 6) * (8 * RT2D_WAVE) + lane, xp, zp, 0.0);
    const HeatGeom& geo = b.bin.geo;
    const int nx = b.bin.nx, nz = b.bin.nz;
    const double sign = b.bin.sign;
    // the queue of points in flight: at the top of the outer loop xq[k] is row s + k (rows clamped into the longest ray)
    double xq[ONEPASS_AHEAD], zq[ONEPASS_AHEAD];
#pragma unroll
    for (int u = 0; u < ONEPASS_AHEAD; ++u) {
        const int sn = min(1 + u, np_max - 1);
        xq[u] = (Xs + (size_t)sn * RT2D_WAVE)[ulane]; zq[u] = (Zs + (size_t)sn * RT2D_WAVE)[ulane];
    }
    int s = 1;                                                             // wave-uniform
    while (s < np_max) {
        bool general = false;
#pragma unroll
        for (int u = 0; u < ONEPASS_AHEAD; ++u) {
            if (s >= np_max) break;
            ONEPASS_HEAD(xq[u], zq[u])
            if (!all_common) {
                // not a common step: bring the queue back into order (slots 0 .. u-1 already hold rows s + AHEAD - u ...) and
                // let the one general block below take this point
                double tx[ONEPASS_AHEAD], tz[ONEPASS_AHEAD];
#pragma unroll
                for (int k = 0; k < ONEPASS_AHEAD; ++k) { tx[k] = xq[(k + u) % ONEPASS_AHEAD]; tz[k] = zq[(k + u) % ONEPASS_AHEAD]; }
#pragma unroll
                for (int k = 0; k < ONEPASS_AHEAD; ++k) { xq[k] = tx[k]; zq[k] = tz[k]; }
                general = true;
                break;
            }
            {
                const int sn = min(s + ONEPASS_AHEAD, np_max - 1);
                xq[u] = (Xs + (size_t)sn * RT2D_WAVE)[ulane]; zq[u] = (Zs + (size_t)sn * RT2D_WAVE)[ulane];
            }
            S = Sn;
            const bool binning = live & binner.ok;
            const bool leaving = binning & (newcol | newrow);
            const double v = sign * rt2d_exp2_3addr(binner.cur_g * -0.033219280948873623);    // ray.py:317, simulation.py:129; -log2(10)/100
            rt2d_deposit_full(H, leaving, binner.cur_ix * nz + binner.cur_iz, v, lane);
            if (binning & newrow) { binner.slot[k_old * RT2D_WAVE] = v; binner.occ |= 1u << k_old; binner.col_lo = lo_n; binner.col_hi = hi_n; }
            if (binning & newcol) { binner.occ = 0u; binner.col_lo = iz; binner.col_hi = iz; }
            if (binning) { binner.cur_ix = ix; binner.cur_iz = iz; binner.cur_g = Sn + A; }
            xp = x; zp = z;
            ++s;
        }
        if (general) {
            // general step (ONE copy): whatever the existing kernel does for this point -- the exact alpha where the guess was
            // wrong, RayBinner's remembered rows, violations.  Everything it needs of the launch arguments is read HERE, through
            // the kernel-argument pointer laundered by an empty asm: loop-invariant loads of a cold path would otherwise be
            // hoisted in front of the loop and hold scalar registers that the common step's constants need.
            ONEPASS_HEAD(xq[0], zq[0])
            (void)Sn; (void)all_common; (void)lo_n; (void)hi_n; (void)k_old; (void)newrow;
#pragma unroll
            for (int k = 0; k + 1 < ONEPASS_AHEAD; ++k) { xq[k] = xq[k + 1]; zq[k] = zq[k + 1]; }
            {
                const int sn = min(s + ONEPASS_AHEAD, np_max - 1);
                xq[ONEPASS_AHEAD - 1] = (Xs + (size_t)sn * RT2D_WAVE)[ulane]; zq[ONEPASS_AHEAD - 1] = (Zs + (size_t)sn * RT2D_WAVE)[ulane];
            }
            const GainHeatArgs* cold = (const GainHeatArgs*)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(cold));
            if (live) {
                const GainArgs& ca = cold->g;
                double pe = pa;
                if (!piece_ok) {
                    double r19[20];
                    rt2d_lookup<19>(ca.blob + ca.env.gain_off, ca.env.absorb, zp, r19);
                    const double fc = cold->freq / 1000;
                    const double al = -rt2d_alpha_fast(r19, zp - r19[0], fc * fc);
                    const double q = al * 0.001;
                    pe = fma(fma(-q, 1000.0, al), 0.001, q);
                }
                S = fma(pe, dL, S);
            }
            binner.point<true>(cold->bin, x, z, S + A, lane, live);
            xp = x; zp = z;
            ++s;
        }
    }
    {
        const GainHeatArgs* cold = (const GainHeatArgs*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(cold));
        binner.finish<true>(cold->bin, lane);
    }
    if (!binme || np == 0) return;
    if (b.irregular) b.irregular[r] = binner.violated ? 1 : 0;
    if (binner.violated && b.n_violations) atomicAdd(b.n_violations, 1ull);
}

// ------------------------------------------------------------------------------------------------------
// distance to target from stored trajectories (ray.py:282-287)
__global__ void __launch_bounds__(256) rt2d_dist_kernel(const double* X, const double* Z, const int64_t* tile_off,
                                                        const int32_t* n_pts, int64_t n, double tx, double tz, double* dist) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int np = rt2d_stored_pts(n_pts, tile_off, r);
    const int64_t base = tile_off[r >> 6] + (r & 63);
    TargetTrack tt; tt.init();
    double mn = INFINITY, mx = -INFINITY;
    for (int s = 0; s < np; ++s) {
        int64_t o = base + (int64_t)s * RT2D_WAVE;
        double x = X[o], z = Z[o];
        mn = fmin(mn, x); mx = fmax(mx, x);
        tt.add(x, z, 0.0, s, tx);
    }
    dist[r] = tt.dist(tx, tz, np, mn, mx);
}

// Eigenray products (Ray2D.gen_filter, ray.py:382-405): per ray the stored point closest to the target (first of the
// minimal distances, as np.argmin), the straight-line residual to the target, and from them the travel time to the
// target and the transmittance 10^(G/10) at every frequency.  One wave per ray: lanes share the points for the
// arg-min, then take one frequency each.
struct TargetArgs {
    DevEnv env; const double* blob;
    const double* X; const double* Z; const double* T; const double* const* G;     // G: F device pointers, tile layout
    const int64_t* tile_off; const int32_t* n_pts; const int32_t* ray_ids; int64_t n_sel;
    double tx, tz; const double* freqs; int32_t n_freqs;
    int32_t* step_out; double* T_target; double* Tmult;                             // Tmult[f * n_sel + i]
};

__global__ void __launch_bounds__(64) rt2d_target_products_kernel(const TargetArgs a) {
    const int64_t i = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t r = a.ray_ids ? (int64_t)a.ray_ids[i] : i;
    const int np = rt2d_stored_pts(a.n_pts, a.tile_off, r);
    const int64_t base = a.tile_off[r >> 6] + (r & 63);
    double best = INFINITY; int best_s = 0x7fffffff;
    for (int s = lane; s < np; s += RT2D_WAVE) {
        const int64_t o = base + (int64_t)s * RT2D_WAVE;
        const double dx = a.tx - a.X[o], dz = a.tz - a.Z[o];
        const double d = sqrt(dx * dx + dz * dz);                                    // ray.py:372
        if (d < best) { best = d; best_s = s; }                                          // strict: the first minimum stays
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double od = __shfl_xor(best, off); const int os = __shfl_xor(best_s, off);
        if (os != 0x7fffffff && (best_s == 0x7fffffff || od < best || (od == best && os < best_s))) { best = od; best_s = os; }
    }
    if (best_s == 0x7fffffff) best_s = 0;                                            // no finite distance at all
    const int64_t o = base + (int64_t)best_s * RT2D_WAVE;
    const double px = a.X[o], pz = a.Z[o];
    const double ddx = a.tx - px, ddz = a.tz - pz;
    const double d_target = sqrt(ddx * ddx + ddz * ddz);                             // np.linalg.norm(target - p)
    const double z_avg = (a.tz + pz) / 2;
    if (lane == 0) {
        a.step_out[i] = best_s;
        a.T_target[i] = a.T[o] + d_target / rt2d_eval_c(a.blob, a.env.cz, z_avg);    // ray.py:403-404
    }
    const double* gtab = a.blob + a.env.gain_off;
    double rec[20];
    rt2d_lookup<19>(gtab, a.env.absorb, z_avg, rec);
    const double dz = z_avg - rec[0];
    for (int f = lane; f < a.n_freqs; f += RT2D_WAVE) {
        const double fk = a.freqs[f] / 1000;
        const double G_add = -1 * rt2d_alpha(rec, dz, fk * fk) / 1000 * d_target;    // ray.py:398
        const double G_target = a.G[f][o] + G_add;
        a.Tmult[(int64_t)f * a.n_sel + i] = pow(10.0, G_target / 10);                 // ray.py:400
    }
}

// ------------------------------------------------------------------------------------------------------
// heat map accumulation: one wave per ray; LDS hash "cell -> last step in that cell"
#define HM_EMPTY 0xffffffffffffffffull

__device__ __forceinline__ uint32_t rt2d_hash32(uint32_t k) { k ^= k >> 16; k *= 0x7feb352du; k ^= k >> 15; k *= 0x846ca68bu; k ^= k >> 16; return k; }

struct HeatArgs {
    const double* X; const double* Z; const double* G; const int64_t* tile_off; const int32_t* n_pts; int64_t n;
    const uint8_t* ray_mask;                        // n bytes, nonzero = bin this ray (NULL = all)
    int32_t* work;                                  // [0] = number of rays left to the general kernel, [2 + k] = their ids
    HeatGeom geo; int64_t nx, nz;
    double* H; int64_t* n_dropped; int32_t cap;     // cap: hash slots per ray, power of two
    int32_t cell_cap;                               // points per ray whose cells are cached in LDS
    int32_t split;                                  // tile kernel: waves per tile, each walking a range of rows
    // several frequencies at once (rt2d_heatmap_accumulate_multi): map f is H + f * map_stride, its gains Gs[f]
    const double* const* Gs; int32_t n_freqs; int64_t map_stride;
    const int32_t* g_slot; int64_t g_room;          // optional: G in the compact form of GainArgs::g_slot (general kernel only)
    int32_t group;                                  // general kernel: consecutive rays one workgroup bins, one after the other
    int32_t acc_cap;                                // ... slots of the LDS table their deposits are summed in (0 = none), power of two
    int32_t minor_fb, pad_minor;                    // maps per group of the scratch below (4 or 8)
    double* H_minor;                                // several maps per pass, map-minor scratch: group g of FB maps at H_minor + g * map_stride * FB,
                                                    // cell c of map k of the group at [c * FB + k] (see MINOR in rt2d_heatmap_fast_kernel)
};

// Fast path: one wave = one 64-ray tile, coalesced over the tile layout.  A ray whose cell column index never
// decreases along the path and whose points all fall inside the map can only revisit a cell while it is still in
// the same column, so "last write wins" reduces to: walk the ray BACKWARDS and deposit the first time a (column,
// row) pair is seen -- a per-lane bit mask of the rows seen in the current column, no hash.  Rays that fail the
// forward check (backward-moving rays, points outside the map, which would alias flat indices) are queued for the
// general kernel below.
#ifndef HM_UNROLL
#define HM_UNROLL 4      // measured: 8 rows in flight cost registers and 1.7 ms per 1M rays
#endif
#ifndef HM_STAGE
#define HM_STAGE 256         // slots of a wave's staging table (several maps per pass)
#endif
#ifndef HM_STAGE_ROWS
#define HM_STAGE_ROWS 16     // rows of the walk between two flushes of it (a multiple of HM_UNROLL)
#endif
// MINOR (FB > 1): the deposits go to a map-minor scratch H_minor[group][cell][FB] instead of the maps.  The L2s take one atomic
// REQUEST at a time whatever it carries (tools/probes/atomic_coalesce: 23.6 G requests/s with 1, 4 or 8 consecutive doubles
// per request), and the FB sums of a run are FB consecutive doubles there: each wave queues its run sums in LDS (cell, FB
// values: no hash, nothing is dropped or sent directly) and empties the queue with lanes laid out as (entry, map), so that
// one request carries a run's deposit for all FB maps.  rt2d_hm_minor_add_kernel adds the scratch into the maps afterwards.
// (Tried: the LDS summing the runs -- every lane of a run adds into its run's queue entry with ds_add_f64 instead of the DPP
// scan per map: 64 maps x 65 536 rays 15.7 -> 17.3 ms for the maps.  Not kept.)
template <int MW, int FB, bool MINOR = false>     // MW: 64-bit mask words per lane (rows nz <= 64 * MW); FB: maps (frequencies) binned per pass
__global__ void __launch_bounds__(256) rt2d_heatmap_fast_kernel(const HeatArgs a) {
    // FB > 1: blockIdx.y picks a group of FB frequencies.  The cells of a point, "first time this (column, row) is seen" and
    // the runs of equal cells across the wave are the same for every frequency: they are worked out once per point and
    // only the value (2^x of that frequency's gain) and the run sums are per frequency.
    const int f0 = (FB > 1) ? (int)blockIdx.y * FB : 0;
    // (the per-map gain pointers are read from device memory, and a pointer that was loaded is a generic one to the
    // compiler: the gains would be FLAT loads, which count in the vector-memory AND the LDS counter and turn every wait of
    // the walk -- the LDS queue's included -- into a wait for both to drain.  They are global memory: said here.)
    typedef const __attribute__((address_space(1))) double* g_cf64p;
    g_cf64p Gk[FB];
    double* Hk[FB];
#pragma unroll
    for (int k = 0; k < FB; ++k) {
        const int f = (FB > 1) ? min(f0 + k, a.n_freqs - 1) : 0;      // a group past the end repeats the last map's gains ...
        Gk[k] = (g_cf64p)((FB > 1) ? a.Gs[f] : a.G);
        Hk[k] = a.H + (int64_t)f * a.map_stride;
    }
    const int nf = (FB > 1) ? min(FB, a.n_freqs - f0) : 1;            // ... but only deposits the first nf
    // FB > 1: the maps of several frequencies are bound by the f64 atomics of the L2s (one per run of equal cells per map).
    // Each wave stages its deposits in a table of its own in LDS (HM_STAGE slots: cell -> FB sums, direct-mapped,
    // ds_add_f64) and sends the table to the maps every HM_STAGE_ROWS rows of its walk.  The table is private to the wave
    // and the wave runs in lockstep: no barrier anywhere.  A deposit whose slot holds another cell goes to the map directly.
    // What it buys is modest and measured (64 maps x 65 536 rays: 375 M -> 338 M atomics, 0.635 -> 0.607 ms per map): inside
    // a tile a cell hardly ever comes up again at another step -- every ray advances by exactly dx_max in x or dz_max in z
    // per step, so the rays of a tile share the column or the row of a step and leave it together -- and where a tile's rays
    // sit in many cells at one step they have been scattered by a dozen rebounds and share cells with nobody.  Tried beyond
    // it: ONE table for four adjacent tiles per workgroup, flushed per window of 16 absolute rows between barriers: 252 M
    // atomics (-25 %) but 18.4 ms against 17.6 for the launch (the barriers), 0.645 ms per map: not kept.
    constexpr int STAGE = (FB >= 8) ? HM_STAGE / 2 : (FB > 1) ? HM_STAGE : 1;      // (8 maps: half the entries, the same LDS)
    constexpr int UNR = (FB >= 8) ? 2 : HM_UNROLL;                                   // rows in flight per lane (8 maps: 2 x 10 loads)
    __shared__ int32_t s_stage_key[(FB > 1) ? 4 : 1][STAGE];
    __shared__ double s_stage_val[(FB > 1) ? 4 : 1][STAGE * FB];
    int32_t* const st_key = s_stage_key[(FB > 1) ? (threadIdx.x / RT2D_WAVE) : 0];
    double* const st_val = s_stage_val[(FB > 1) ? (threadIdx.x / RT2D_WAVE) : 0];
    double* const Hmn = MINOR ? a.H_minor + (int64_t)blockIdx.y * a.map_stride * FB : nullptr;
    int q_count = 0;                                                  // MINOR: entries in the wave's queue (wave-uniform)
    if (FB > 1 && !MINOR) {
        for (int q = threadIdx.x & 63; q < STAGE; q += RT2D_WAVE) {
            st_key[q] = -1;
#pragma unroll
            for (int kk = 0; kk < FB; ++kk) st_val[q * FB + kk] = 0.0;
        }
    }
    auto stage_flush = [&]() {
        if (FB == 1) return;
        if (MINOR) {
            constexpr int SPW = RT2D_WAVE / FB;                        // queue entries per atomic instruction
            const int ln = threadIdx.x & 63, kk = ln % FB, dq = ln / FB;
            for (int q0 = 0; q0 < q_count; q0 += SPW) {
                const int q = q0 + dq;
                if (q < q_count && kk < nf) atomicAdd(&Hmn[(int64_t)st_key[q] * FB + kk], st_val[q * FB + kk]);
            }
            q_count = 0;
            return;
        }
        for (int q = threadIdx.x & 63; q < STAGE; q += RT2D_WAVE) {
            const int32_t key = st_key[q];
            if (key < 0) continue;
#pragma unroll
            for (int kk = 0; kk < FB; ++kk) {
                if (kk < nf) atomicAdd(&Hk[kk][key], st_val[q * FB + kk]);
                st_val[q * FB + kk] = 0.0;
            }
            st_key[q] = -1;
        }
    };
    // Fans of 10^4..10^5 rays have too few tiles to give a SIMD more than one wave, and then the kernel's time is the
    // length of one wave's walk.  Such a tile is walked by `split` waves, each taking a contiguous range of ROWS (see
    // "rows of this wave" below); tiles with unchecked rays are walked whole by their first wave.
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x / RT2D_WAVE) + (threadIdx.x / RT2D_WAVE);
    const int lane = threadIdx.x & 63;
    const int64_t n_tiles = (a.n + RT2D_WAVE - 1) / RT2D_WAVE;
    const int64_t tile = wave_id / a.split;
    const int seg = (int)(wave_id % a.split);
    if (tile >= n_tiles) return;
    const int64_t r = tile * RT2D_WAVE + lane;
    const uint8_t flags = (r < a.n) ? (a.ray_mask ? a.ray_mask[r] : (uint8_t)1) : (uint8_t)0;
    const bool mine = flags != 0;
    const bool hinted = (flags & 2) != 0;               // caller vouches for monotone columns and in-map points
    int np = mine ? rt2d_stored_pts(a.n_pts, a.tile_off, r) : 0;
    const int64_t base = a.tile_off[tile] + lane;
    const int nx = (int)a.nx, nz = (int)a.nz;
    int tmax = np;
    for (int off = 32; off > 0; off >>= 1) tmax = max(tmax, __shfl_xor(tmax, off));

    // Loads are unconditional, from a row index clamped into the ray (row 0 of the tile exists for every lane): with
    // a load per live lane behind a branch the compiler sinks the first use into the branch and waits there, which
    // turns HM_UNROLL loads in flight into HM_UNROLL round trips.
    int last = max(np, 1) - 1;

    // forward check, HM_UNROLL rows in flight per lane.  The launcher only uses this kernel when every in-map point
    // has |x|, |z| < 2^31 (shape * resolution below that), so the 32-bit cell arithmetic decides "inside the map" exactly.
    bool ok = true;
    int prev_ix = INT32_MIN;
    // A shared tile cannot run the point-by-point check (its verdict would have to cross waves): if some of its rays are
    // vouched for, or the caller marked the others as known not to qualify (bit 4 of ray_mask), the others go straight
    // to the general kernel, which takes any ray; a tile without either (the caller gave no hints) is checked and walked
    // whole by its first wave -- the slowest wave of the launch when the fan has few tiles.
    const bool send = a.split > 1 && mine && !hinted && (__any(hinted) || (flags & 4) != 0);     // bit 4: known not to qualify
    const bool all_hinted = __all(hinted || !mine || send);
    if (!all_hinted && seg > 0) return;
    const bool whole = a.split == 1 || !all_hinted;
    const int t_check = all_hinted ? 0 : tmax;
    for (int s0 = 0; s0 < t_check; s0 += HM_UNROLL) {
        double xs[HM_UNROLL], zs[HM_UNROLL];
#pragma unroll
        for (int u = 0; u < HM_UNROLL; ++u) {
            const int64_t o = base + (int64_t)min(s0 + u, last) * RT2D_WAVE;
            xs[u] = a.X[o];
            zs[u] = a.Z[o];
        }
#pragma unroll
        for (int u = 0; u < HM_UNROLL; ++u) {
            if (s0 + u < np) {
                int ix, iz;
                const bool in_range = rt2d_cell_i32(a.geo, xs[u], zs[u], ix, iz);
                ok = ok && in_range && ix >= prev_ix && (unsigned)ix < (unsigned)nx && (unsigned)iz < (unsigned)nz;
                prev_ix = ix;
            }
        }
    }
    // rays for the general kernel: one counter update per wave, and the rays of a tile stay together in the list (that
    // kernel's lines are shared by neighbouring rays, see its item mapping)
    const bool queued = send ? seg == 0 : (!ok && !hinted);
    const unsigned long long qmask = __ballot(queued);
    if (qmask && (FB == 1 || blockIdx.y == 0)) {
        const int leader = __ffsll((long long)qmask) - 1;
        int k0 = 0;
        if (lane == leader) k0 = atomicAdd(&a.work[0], __popcll(qmask));
        k0 = __shfl(k0, leader);
        if (queued) a.work[2 + k0 + __popcll(qmask & ((1ull << lane) - 1ull))] = (int32_t)r;
    }
    if (queued | send) { np = 0; last = 0; }
    tmax = np;
    for (int off = 32; off > 0; off >>= 1) tmax = max(tmax, __shfl_xor(tmax, off));

    // rows of this wave: [row_lo, row_hi)
    int row_lo = 0, row_hi = tmax;
    if (!whole) {
        const int seg_len = (tmax + a.split - 1) / a.split;
        row_lo = seg * seg_len;
        row_hi = min(tmax, row_lo + seg_len);
        if (row_lo >= row_hi) return;
    }

    // backward walk
    unsigned long long seen[MW];
#pragma unroll
    for (int w = 0; w < MW; ++w) seen[w] = 0ull;
    int cur_ix = -1;
    if (row_hi < tmax) {
        // The walk starts inside the ray: the rows already seen in the column of its first row are the ones the ray
        // visits right AFTER that row while it stays in the column (columns never decrease), so they are collected by
        // walking forward from row_hi until every lane has left its column.
        bool more = np > row_hi;                             // this lane has rows above the range
        int ix_top = -1;
        {
            const int64_t o = base + (int64_t)min(max(row_hi - 1, 0), last) * RT2D_WAVE;
            int iz;
            rt2d_cell_i32(a.geo, a.X[o], a.Z[o], ix_top, iz);
        }
        if (more) cur_ix = ix_top;
        for (int s = row_hi; __any(more && s < np); ++s) {
            const int64_t o = base + (int64_t)min(s, last) * RT2D_WAVE;
            int ix, iz;
            rt2d_cell_i32(a.geo, a.X[o], a.Z[o], ix, iz);
            more = more && s < np && ix == ix_top;
            if (more) {
#pragma unroll
                for (int w = 0; w < MW; ++w) if (w == (iz >> 6)) seen[w] |= 1ull << (iz & 63);
            }
        }
    }
    for (int s0 = row_hi - 1; s0 >= row_lo; s0 -= UNR) {
        double xs[UNR], zs[UNR], gs[UNR][FB];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t o = base + (int64_t)min(max(s0 - u, 0), last) * RT2D_WAVE;
            xs[u] = a.X[o];
            zs[u] = a.Z[o];
#pragma unroll
            for (int k = 0; k < FB; ++k) gs[u][k] = Gk[k][o];
        }
        unsigned late = 0u;                                 // (row, map) pairs of this batch whose value needs the two pows
        int late_cell[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int s = s0 - u;
            bool fresh = false;
            int cell = -1 - lane;                           // unique per lane: never merges
            late_cell[u] = 0;
            if (s >= row_lo && s < np) {
                int ix, iz;
                rt2d_cell_i32(a.geo, xs[u], zs[u], ix, iz);
                if (ix != cur_ix) {
                    cur_ix = ix;
#pragma unroll
                    for (int w = 0; w < MW; ++w) seen[w] = 0ull;
                }
                const unsigned long long bit = 1ull << (iz & 63);
                const int word = iz >> 6;
#pragma unroll
                for (int w = 0; w < MW; ++w) if (w == word) { fresh = (seen[w] & bit) == 0; seen[w] |= bit; }
                if (fresh) { cell = ix * nz + iz; late_cell[u] = cell; }
            }
            // Neighbouring lanes are neighbouring angles: at one step they mostly sit in the same cell.  Runs of
            // equal cells across the wave are summed and the last lane of each run issues ONE atomic per map.
            DepositRuns runs;
            if (!runs.init(fresh, cell, lane)) continue;
            // (FB > 1) the slot of the wave's staging table for this run's cell: claimed if free, shared if it is this cell's
            bool staged = false;
            int st_slot = 0;
            if (MINOR) {
                const unsigned long long tmask = __ballot(runs.tail);
                st_slot = q_count + __popcll(tmask & ((1ull << lane) - 1ull));
                if (runs.tail) st_key[st_slot] = cell;
                q_count += __popcll(tmask);
            } else if (FB > 1 && runs.tail) {
                st_slot = (int)(rt2d_hash32((uint32_t)cell) & (uint32_t)(STAGE - 1));
                const int32_t old = atomicCAS(&st_key[st_slot], -1, cell);
                staged = old == -1 || old == cell;
            }
#pragma unroll
            for (int k = 0; k < FB; ++k) {
                if (k >= nf) break;
                double val = 0.0;
                if (fresh) {
                    const double g = gs[u][k];
                    if (__builtin_expect(rt2d_heat_val_is_plain(g), 1)) val = rt2d_heat_val_plain(g);   // ray.py:317, simulation.py:129
                    else if (rt2d_heat_val_is_inf(g)) val = rt2d_heat_val_of_inf(g);
                    else late |= 1u << (u * FB + k);        // adds 0 to its run here, the real value below
                }
                const double v = runs.sum(val);
                if (runs.tail) {
                    if (MINOR) st_val[st_slot * FB + k] = v;
                    else if (FB > 1 && staged) unsafeAtomicAdd(&st_val[st_slot * FB + k], v);
                    else atomicAdd(&Hk[k][cell], v);
                }
            }
            if (MINOR && q_count > STAGE - RT2D_WAVE) stage_flush();              // (room for the next row's run sums)
        }
        if (!MINOR && FB > 1 && (((row_hi - 1 - s0) / UNR) % (HM_STAGE_ROWS / UNR)) == (HM_STAGE_ROWS / UNR) - 1) stage_flush();
        // gains beyond +-3000 dB (a surface coefficient that underflowed to 0 at high frequency): the pow pair of the
        // reference, kept out of the loop above and present once in the program
        while (__builtin_expect(late != 0u, 0)) {
            const int b = __ffs(late) - 1;
            late &= late - 1u;
            const int u = b / FB, k = b % FB;
            double g = gs[0][0]; int cell = late_cell[0]; double* Hm = Hk[0];
#pragma unroll
            for (int uu = 0; uu < UNR; ++uu)
#pragma unroll
                for (int kk = 0; kk < FB; ++kk) if (u == uu && k == kk) { g = gs[uu][kk]; cell = late_cell[uu]; Hm = Hk[kk]; }
            if (MINOR) atomicAdd(&Hmn[(int64_t)cell * FB + k], rt2d_heat_val_pows(g));
            else atomicAdd(&Hm[cell], rt2d_heat_val_pows(g));
        }
    }
    stage_flush();
}

// H[f][cell] += H_minor[f / fb][cell][f % fb] (see MINOR above); one thread per (map, cell), stream-ordered behind the tile
// kernel: plain read-modify-write
__global__ void __launch_bounds__(256) rt2d_hm_minor_add_kernel(const double* Hm, int64_t cells, int32_t n_freqs, int32_t fb, double* H) {
    const int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (cell >= cells || f >= n_freqs) return;
    const double v = Hm[((int64_t)(f / fb) * cells + cell) * fb + (f % fb)];
    if (v != 0.0) H[(int64_t)f * cells + cell] += v;
}

// General path: one workgroup of HM_THREADS per ray, any trajectory.  The cell of every point is computed once into LDS
// (rays longer than the cell cache recompute it from global memory); "last write wins" is resolved by an LDS hash
// cell -> last step, in partitions when the ray has more candidate cells than half the hash.  The LDS of a ray (48 KB
// at the default sizes) allows three rays per CU whatever the workgroup size, and a ray is a chain of dependent phases:
// four waves per ray shorten every phase of an ~800-point ray to one or two points per lane.
#ifndef HM_THREADS
#define HM_THREADS 256
#endif
__global__ void __launch_bounds__(HM_THREADS) rt2d_heatmap_kernel(const HeatArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_hash[];
    __shared__ int s_red[2];
    __shared__ int2 s_own[HM_THREADS / RT2D_WAVE][RT2D_WAVE];      // (several maps through the map-minor scratch: a wave's owning points)
    int32_t* s_cell = reinterpret_cast<int32_t*>(s_hash + a.cap);
    // Neighbouring rays of the list visit mostly the same cells: a workgroup bins `group` consecutive rays one after the
    // other and sums their deposits in an LDS table first (acc_cap slots: cell -> sum, two probes, a deposit that finds
    // both taken goes to the map directly); one atomic per occupied slot reaches L2 when the group is done.
    double* s_acc_val = reinterpret_cast<double*>(s_cell + a.cell_cap);
    int32_t* s_acc_key = reinterpret_cast<int32_t*>(s_acc_val + a.acc_cap);
    const bool use_acc = a.acc_cap > 0 && a.n_freqs == 0;
    const uint32_t acc_mask = (uint32_t)a.acc_cap - 1;
    const int lane = threadIdx.x;
    const int64_t size = a.nx * a.nz;
    const uint32_t mask = (uint32_t)a.cap - 1;
    // rays come either from the queue the fast kernel filled (work != NULL) or straight from the mask
    const int64_t n_items = a.work ? (int64_t)a.work[0] : a.n;
    // One lane reads one point of the ray, i.e. a wave touches 64 different 128-byte lines of the tile layout, each of
    // which also holds the same point of the ray's 15 neighbours.  Workgroups go to the 8 XCDs round-robin, so items
    // are dealt in groups of 16 consecutive list entries per XCD: the neighbours' workgroups then run on the same L2 at
    // about the same time and a line is fetched once for the group instead of once per ray (grid: a multiple of 128).
    const int64_t b_xcd = blockIdx.x & 7, b_in = blockIdx.x >> 3;
    const int64_t first_item = ((b_in >> 4) * 8 + b_xcd) * 16 + (b_in & 15);
    const int64_t group = a.group > 1 ? a.group : 1;
    for (int64_t item0 = first_item * group; item0 < n_items; item0 += (int64_t)gridDim.x * group) {
      if (use_acc) {
          for (int i = lane; i < a.acc_cap; i += HM_THREADS) { s_acc_key[i] = -1; s_acc_val[i] = 0.0; }
          __syncthreads();
      }
      for (int64_t item = item0; item < item0 + group && item < n_items; ++item) {
        const int64_t r = a.work ? (int64_t)a.work[2 + item] : item;
        if (!a.work && a.ray_mask && !a.ray_mask[r]) continue;
        if (a.g_slot && a.g_slot[r] < 0) continue;            // (no row of the compact gain array: rt2d_classify_kernel flagged it)
        const int np = rt2d_stored_pts(a.n_pts, a.tile_off, r);
        const int64_t base = a.tile_off[r >> 6] + (r & 63);
        const bool cached = np <= a.cell_cap;
        auto cell_global = [&](int s) -> int32_t {
            const int64_t o = base + (int64_t)s * RT2D_WAVE;
            return (int32_t)rt2d_cell(a.geo, a.X[o], a.Z[o], a.nz, size);
        };
        auto cell_at = [&](int s) -> int32_t { return cached ? s_cell[s] : cell_global(s); };

        // pass 0: cells (four independent loads per lane in flight), dropped points
        int n_drop = 0;
        for (int s0 = 0; s0 < np; s0 += 4 * HM_THREADS) {
            double xs[4], zs[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int sc = min(s0 + u * HM_THREADS + lane, np - 1);
                const int64_t o = base + (int64_t)sc * RT2D_WAVE;
                xs[u] = a.X[o]; zs[u] = a.Z[o];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int sx = s0 + u * HM_THREADS + lane;
                if (sx < np) {
                    const int32_t cell = (int32_t)rt2d_cell(a.geo, xs[u], zs[u], a.nz, size);
                    if (cached) s_cell[sx] = cell;
                    n_drop += cell < 0;
                }
            }
        }
        if (lane < 2) s_red[lane] = 0;
        __syncthreads();
        int n_cand = 0;
        for (int sx = lane; sx < np; sx += HM_THREADS) {
            const int32_t cell = cell_at(sx);
            const int32_t nxt = (sx + 1 < np) ? cell_at(sx + 1) : -2;
            n_cand += (cell >= 0) & (nxt != cell);
        }
        for (int off = 32; off > 0; off >>= 1) { n_cand += __shfl_xor(n_cand, off); n_drop += __shfl_xor(n_drop, off); }
        if ((lane & 63) == 0) { atomicAdd(&s_red[0], n_cand); atomicAdd(&s_red[1], n_drop); }
        __syncthreads();
        n_cand = s_red[0]; n_drop = s_red[1];
        if (lane == 0 && n_drop && a.n_dropped) atomicAdd((unsigned long long*)a.n_dropped, (unsigned long long)n_drop);
        const int parts = max(1, (2 * n_cand + a.cap - 1) / a.cap);        // load factor <= 1/2 per partition

        for (int part = 0; part < parts; ++part) {
            for (int i = lane; i < a.cap; i += HM_THREADS) s_hash[i] = HM_EMPTY;
            __syncthreads();
            // pass A: cell -> max step
            for (int sx = lane; sx < np; sx += HM_THREADS) {
                const int32_t cell = cell_at(sx);
                if (cell < 0) continue;
                if (sx + 1 < np && cell_at(sx + 1) == cell) continue;
                const uint32_t h = rt2d_hash32((uint32_t)cell);
                if ((int)((h >> 8) % (uint32_t)parts) != part) continue;
                const unsigned long long packed = ((unsigned long long)(uint32_t)cell << 32) | (uint32_t)sx;
                uint32_t slot = h & mask;
                for (int probe = 0; probe < a.cap; ++probe) {
                    unsigned long long old = atomicCAS(&s_hash[slot], HM_EMPTY, packed);
                    if (old == HM_EMPTY) break;
                    if ((old >> 32) == (unsigned long long)(uint32_t)cell) { atomicMax(&s_hash[slot], packed); break; }
                    slot = (slot + 1) & mask;
                }
            }
            __syncthreads();
            // pass B: the step that owns its cell deposits (simulation.py:129-131); up to eight points per lane at a time so
            // that their gain loads travel together
            for (int s0 = 0; s0 < np; s0 += 8 * HM_THREADS) {
                int32_t own[8]; double gv[8]; int own_s[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int sx = s0 + u * HM_THREADS + lane;
                    own[u] = -1;
                    own_s[u] = sx;
                    gv[u] = 0.0;
                    if (sx >= np) continue;
                    const int32_t cell = cell_at(sx);
                    if (cell < 0) continue;
                    if (sx + 1 < np && cell_at(sx + 1) == cell) continue;
                    const uint32_t h = rt2d_hash32((uint32_t)cell);
                    if ((int)((h >> 8) % (uint32_t)parts) != part) continue;
                    uint32_t slot = h & mask;
                    for (int probe = 0; probe < a.cap; ++probe) {
                        const unsigned long long cur = s_hash[slot];
                        if (cur == HM_EMPTY) break;
                        if ((cur >> 32) == (unsigned long long)(uint32_t)cell) {
                            if ((uint32_t)cur == (uint32_t)sx) {
                                own[u] = cell;
                                if (a.n_freqs == 0) gv[u] = a.g_slot ? a.G[(int64_t)a.g_slot[r] * a.g_room + sx] : a.G[base + (int64_t)sx * RT2D_WAVE];
                            }
                            break;
                        }
                        slot = (slot + 1) & mask;
                    }
                }
                if (a.n_freqs == 0) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (own[u] < 0) continue;
                        const double v = rt2d_heat_val(gv[u]);                               // ray.py:317, simulation.py:129
                        bool staged = false;
                        if (use_acc) {
                            const uint32_t h = rt2d_hash32((uint32_t)own[u] * 0x9e3779b1u);
#pragma unroll
                            for (int probe = 0; probe < 2 && !staged; ++probe) {
                                const uint32_t slot = (h + probe) & acc_mask;
                                const int32_t old = atomicCAS(&s_acc_key[slot], -1, own[u]);
                                if (old == -1 || old == own[u]) { unsafeAtomicAdd(&s_acc_val[slot], v); staged = true; }
                            }
                        }
                        if (!staged) atomicAdd(&a.H[own[u]], v);
                    }
                } else if (a.H_minor) {
                    // several maps, map-minor scratch (see MINOR in rt2d_heatmap_fast_kernel): the owning points of a wave are
                    // compacted into LDS, then lanes are laid out as (owning point, map of a group of 4): one atomic request
                    // carries a point's deposit for four maps, and every lane of the instruction has one to make
                    const int FBG = a.minor_fb, SPW = RT2D_WAVE / FBG;      // 4 or 8 maps per group, as the scratch is laid out
                    int2* const own_q = s_own[lane >> 6];
                    const int wl = lane & 63, kq = wl % FBG, dq = wl / FBG;
                    const int n_groups = (a.n_freqs + FBG - 1) / FBG;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const unsigned long long om = __ballot(own[u] >= 0);
                        if (om == 0ull) continue;
                        const int c = __popcll(om);
                        if (own[u] >= 0) own_q[__popcll(om & ((1ull << wl) - 1ull))] = make_int2(own[u], own_s[u]);
                        for (int g = 0; g < n_groups; ++g) {
                            const int f = g * FBG + kq;
                            const auto Gf = (const __attribute__((address_space(1))) double*)a.Gs[min(f, a.n_freqs - 1)];   // (global, not FLAT)
                            double* Hg = a.H_minor + (int64_t)g * a.map_stride * FBG;
                            for (int e0 = 0; e0 < c; e0 += SPW) {
                                const int e = e0 + dq;
                                if (e < c && f < a.n_freqs) {
                                    const int2 q = own_q[e];
                                    atomicAdd(&Hg[(int64_t)q.x * FBG + kq], rt2d_heat_val(Gf[base + (int64_t)q.y * RT2D_WAVE]));
                                }
                            }
                        }
                    }
                } else {
                    // several maps: the owning step of a cell is the same for every frequency, only the value differs
                    for (int f = 0; f < a.n_freqs; ++f) {
                        const auto Gf = (const __attribute__((address_space(1))) double*)a.Gs[f];
                        double* Hf = a.H + (int64_t)f * a.map_stride;
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (own[u] >= 0) atomicAdd(&Hf[own[u]], rt2d_heat_val(Gf[base + (int64_t)own_s[u] * RT2D_WAVE]));
                    }
                }
            }
            __syncthreads();
        }
      }
      if (use_acc) {                                  // (every ray of the group ended on a barrier)
          for (int i = lane; i < a.acc_cap; i += HM_THREADS)
              if (s_acc_key[i] >= 0) atomicAdd(&a.H[s_acc_key[i]], s_acc_val[i]);
          __syncthreads();
      }
    }
}

// ------------------------------------------------------------------------------------------------------
// Small fans (one workgroup per ray is all there is to fill the chip with): gain at one frequency AND binning of a ray in
// ONE launch, nothing written but the map.  Replaces, for Simulation2D.heatmap() of a freshly cast small fan, the three
// launches rt2d_gain_dg_kernel -> rt2d_gain_scan_part_kernel -> rt2d_heatmap_kernel and the gain array between them
// (ten launches behind a 0.59 ms cast).  A thread owns a CONTIGUOUS chunk of the ray's points (at most
// TAIL_MAXC): their dG (ray.py:346, alpha from the verified table of rt2d_onepass_kernel or the exact form), a block scan
// for the running sum (ray.py:347; summation order differs from np.cumsum's, the map's tolerance is 1e-12 and no gain
// leaves the kernel), the rebound losses in force at each point (ray.py:350-362), then "last write wins" through the LDS
// hash cell -> last step exactly as rt2d_heatmap_kernel does it, the owning point depositing its value with one atomic.
#define TAIL_THREADS 256
#define TAIL_MAXC 8
#define TAIL_CAP (TAIL_THREADS * TAIL_MAXC)      // 2048 points per ray (at 16 per thread the kernel needs 179 registers: two
                                                 // workgroups per CU, 999 rays in two rounds)
#define TAIL_HASH 4096                           // slots of the hash: load factor <= 1/2 without partitions
struct TailArgs {
    GainArgs g; const int32_t* status; const uint8_t* ray_mask;
    HeatGeom geo; int64_t nx, nz; double* H; int64_t* n_dropped;
    double freq; AlphaDesc ad; int32_t alpha_ok; int32_t pad;
    unsigned long long* scratch;                 // optional: the two words rt2d_hm_max_kernel accumulates into, cleared here
    double* overflow;                            // optional: set to 1 when a ray has more stored points than a workgroup holds
    const int32_t* ray_list;                     // optional: [0] = number of rays (read on the device), [2 + k] = ray: only these, status and mask not consulted
    int32_t* long_list;                          // optional, same form: the rays with more stored points than a workgroup holds are appended instead of binned
    unsigned long long* dbg; int64_t dbg_cap;    // (RT2D_TAIL_LOG builds)
};

#ifdef RT2D_TAIL_LOG      // development aid: 100 MHz clock at the phase boundaries of every workgroup, into rt2d_debug_wave_log's buffer
#define TAIL_STAMP(K) if (tid == 0 && t.dbg && blockIdx.x < (unsigned)t.dbg_cap) t.dbg[(size_t)blockIdx.x * 8 + (K)] = __builtin_amdgcn_s_memrealtime();
#else
#define TAIL_STAMP(K)
#endif
__attribute__((amdgpu_waves_per_eu(4, 4)))
__global__ void __launch_bounds__(TAIL_THREADS) rt2d_small_tail_kernel(const TailArgs t) {
    __shared__ int32_t s_hash[TAIL_HASH];       // cell -> its last step: a slot holds the STEP, the cell it stands for is s_cell[step]
    __shared__ int32_t s_cell[TAIL_CAP + 1];
    __shared__ int16_t s_slot[TAIL_CAP];        // the slot a candidate point ended up in (pass A), read back by pass B    // (28 KB of LDS and 128 registers per ray: four workgroups per CU, 999 rays resident at once)
    __shared__ double s_part[TAIL_THREADS / RT2D_WAVE];
    __shared__ int s_red[2];
    extern __shared__ __attribute__((aligned(16))) double s_loss[];        // [cap] cumulative rebound loss, then int [cap] first point carrying it
    const GainArgs& a = t.g;
    int* s_at = reinterpret_cast<int*>(s_loss + a.rebound_cap);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // A thread reads one point of the ray, i.e. one 8-byte word of a 128-byte line that also holds the same point of the
    // ray's 15 neighbours: workgroups go to the 8 XCDs round-robin, so rays are dealt in groups of 16 consecutive rays per
    // XCD and a line is fetched into ONE L2 (grid: a multiple of 128, as in rt2d_heatmap_kernel)
    const int64_t b_xcd = blockIdx.x & 7, b_in = blockIdx.x >> 3;
    int64_t r = ((b_in >> 4) * 8 + b_xcd) * 16 + (b_in & 15);
    if (blockIdx.x == 0 && tid == 0 && t.scratch) { t.scratch[0] = 0ull; t.scratch[1] = 0ull; }
    if (t.ray_list) {                                                       // (the side rays of rt2d_gain_heatmap_rest)
        if (r >= (int64_t)t.ray_list[0]) return;
        r = t.ray_list[2 + r];
    }
    if (r >= a.n) return;
    // everything that depends on r alone is asked for at once (one global round trip instead of a chain of them): status,
    // mask, point count, the tile's bounds, the rebound count and -- whatever that count turns out to be -- this thread's
    // entry of the rebound log (an entry beyond the count is allocated, unspecified, and not used)
    const int st = t.ray_list ? 0 : t.status[r];
    const int mk = (t.ray_mask && !t.ray_list) ? (int)t.ray_mask[r] : 1;
    const int np_raw = a.n_pts[r];
    const int64_t toff0 = a.tile_off[r >> 6], toff1 = a.tile_off[(r >> 6) + 1];
    const int nreb_raw = a.n_rebounds[r];
    double rb_vel = 1.0, rb_ang = 0.0, rb_z = 0.0; int rb_step = 0; int rb_surf = 0;
    if (tid < a.rebound_cap) {
        const int64_t kk = (int64_t)tid * a.n + r;
        rb_vel = a.reb_velocity[kk]; rb_ang = a.reb_angle[kk]; rb_z = a.reb_pos_z[kk]; rb_step = a.reb_step[kk]; rb_surf = a.reb_surface[kk];
    }
    if (st != 0 || mk == 0) return;                                         // (the same for the whole workgroup)
    const int np = (int)min((int64_t)np_raw, (toff1 - toff0) >> 6);         // rt2d_stored_pts
    if (np <= 0) return;
    if (np > TAIL_CAP) {
        if (tid == 0) {
            if (t.long_list) {                                              // list mode: the ray is handed on (rt2d_gain_heatmap_rest bins these the other way)
                const int k = atomicAdd(&t.long_list[0], 1);
                t.long_list[2 + k] = (int32_t)r;
            } else if (t.overflow) *t.overflow = 1.0;                       // the map will be incomplete: the caller is told and bins the fan another way
        }
        return;
    }
    TAIL_STAMP(0)
    const int64_t base = toff0 + (r & 63);
    const int64_t size = t.nx * t.nz;
    const int c = (np + TAIL_THREADS - 1) / TAIL_THREADS;                   // points per thread, <= TAIL_MAXC
    const int i0 = tid * c;
    // the chunk's points and the one before it, all loads in flight together
    double xs[TAIL_MAXC + 1], zs[TAIL_MAXC + 1];
#pragma unroll
    for (int u = 0; u <= TAIL_MAXC; ++u) {
        if (u <= c) {
            const int i = min(max(i0 - 1 + u, 0), np - 1);
            const int64_t o = base + (int64_t)i * RT2D_WAVE;
            xs[u] = a.X[o]; zs[u] = a.Z[o];
        } else { xs[u] = 0.0; zs[u] = 0.0; }
    }
    for (int i = tid; i < TAIL_HASH; i += TAIL_THREADS) s_hash[i] = -1;
    if (tid < 2) s_red[tid] = 0;
    if (tid == 0) s_cell[np] = -2;
    // rebound losses of this ray (one thread per rebound; rebounds beyond TAIL_THREADS in a strided loop), then their running sum
    const int nreb = min(nreb_raw, a.rebound_cap);
    {
        const double* gtab = a.blob + a.env.gain_off;
        for (int k = tid; k < nreb; k += TAIL_THREADS) {
            if (k >= TAIL_THREADS) {
                const int64_t kk = (int64_t)k * a.n + r;
                rb_vel = a.reb_velocity[kk]; rb_ang = a.reb_angle[kk]; rb_z = a.reb_pos_z[kk]; rb_step = a.reb_step[kk]; rb_surf = a.reb_surface[kk];
            }
            const double wavelength = rb_vel / t.freq;
            const double rc = (rb_surf == RT2D_SURFACE_SEDIMENT)
                ? rt2d_refcoef_sediment(wavelength, rt2d_eval_quad(gtab, a.env.imp, rb_z, nullptr))
                : rt2d_refcoef_surface(wavelength, rb_ang);
            s_loss[k] = 10 * log10(rc);
            s_at[k] = rb_step + 1;
        }
    }
    TAIL_STAMP(1)
    // cells and dG of the chunk; gl[u] = sum of the chunk's dG up to and including point i0 + u
    const double f = t.freq / 1000, ff = f * f;                            // model_absorption.py:32
    double gl[TAIL_MAXC];
    double run = 0.0;
    int n_drop = 0;
#pragma unroll
    for (int u = 0; u < TAIL_MAXC; ++u) {
        gl[u] = run;
        const int i = i0 + u;
        if (u < c && i < np) {
            const double x = xs[u + 1], z = zs[u + 1], xp = xs[u], zp = zs[u];
            const int32_t cell = (int32_t)rt2d_cell(t.geo, x, z, t.nz, size);
            s_cell[i] = cell;
            n_drop += cell < 0;
            if (i >= 1) {
                const double ddx = x - xp, ddz = z - zp;
                const double dL = rt2d_sqrt_fast(fma(ddx, ddx, ddz * ddz));
                double pa;
                bool have = false;
                if (t.alpha_ok) {
                    int j = (int)((zp - t.ad.x0) * t.ad.inv_h);
                    j = min(max(j, 0), t.ad.n_pieces - 1);
                    const double* rec = t.ad.tab + (size_t)j * ALPHA_REC;
                    const double lo = rec[0], hi = rec[1];
                    const double d = zp - lo;
                    pa = fma(rec[9], d, rec[8]);
                    pa = fma(pa, d, rec[7]); pa = fma(pa, d, rec[6]); pa = fma(pa, d, rec[5]); pa = fma(pa, d, rec[4]);
                    pa = fma(pa, d, rec[3]); pa = fma(pa, d, rec[2]);
                    have = (zp >= lo) & (zp < hi);
                }
                if (!have) {
                    double r19[20];
                    rt2d_lookup<19>(a.blob + a.env.gain_off, a.env.absorb, zp, r19);
                    const double al = -rt2d_alpha_fast(r19, zp - r19[0], ff);
                    const double q = al * 0.001;
                    pa = fma(fma(-q, 1000.0, al), 0.001, q);
                }
                run = fma(pa, dL, run);                                     // ray.py:346-347
            }
            gl[u] = run;
        }
    }
    TAIL_STAMP(2)
    // exclusive scan of the chunk sums over the workgroup
    double incl = run;
#pragma unroll
    for (int off = 1; off < RT2D_WAVE; off <<= 1) {
        const double up = __shfl_up(incl, off);
        if (lane >= off) incl += up;
    }
    if (lane == RT2D_WAVE - 1) s_part[wv] = incl;
    __syncthreads();                                                        // s_part, s_cell, s_loss / s_at, s_hash, s_red
    double offset = incl - run;
    for (int k = 0; k < wv; ++k) offset += s_part[k];
    if (tid == 0) { double A = 0.0; for (int k = 0; k < nreb; ++k) { A += s_loss[k]; s_loss[k] = A; } }
    TAIL_STAMP(3)
    // candidates: last point of a run of equal cells
    unsigned cand = 0u;                                                     // bit u: point i0 + u is the last of its run of equal cells
#pragma unroll
    for (int u = 0; u < TAIL_MAXC; ++u) {
        const int i = i0 + u;
        if (u < c && i < np) { const int32_t cell = s_cell[i]; cand |= (unsigned)((cell >= 0) & (s_cell[i + 1] != cell)) << u; }
    }
    for (int off = 32; off > 0; off >>= 1) n_drop += __shfl_xor(n_drop, off);
    if (lane == 0 && n_drop) atomicAdd(&s_red[1], n_drop);
    __syncthreads();                                                        // s_red, the summed s_loss
    n_drop = s_red[1];
    if (tid == 0 && n_drop && t.n_dropped) atomicAdd((unsigned long long*)t.n_dropped, (unsigned long long)n_drop);
    // gain of every point of the chunk: running sum + the loss in force (ray.py:360-362)
    {
        int k = 0; double A = 0.0;
#pragma unroll
        for (int u = 0; u < TAIL_MAXC; ++u) {
            const int i = i0 + u;
            if (u < c && i < np) {
                while (k < nreb && s_at[k] <= i) { A = s_loss[k]; ++k; }
                gl[u] = (offset + gl[u]) + A;
            }
        }
    }
    TAIL_STAMP(4)
    // At most TAIL_CAP candidates in TAIL_HASH = 2 TAIL_CAP slots: load factor <= 1/2, no partitions.
    // pass A: cell -> last step.  One loop with every lane busy: a lane makes ONE probe of its current candidate per
    // iteration and moves on to its next candidate when that one is placed -- the wave is done after the max over lanes of
    // their probes, not after the sum over points of the longest probe sequence.
    const uint32_t mask = TAIL_HASH - 1;
    {
        unsigned todo = cand;
        bool probing = false;
        int i = 0; int32_t cell = 0; uint32_t slot = 0;
        while (todo != 0u || probing) {
            if (!probing) {
                const int u = __ffs(todo) - 1;
                todo &= todo - 1u;
                i = i0 + u; cell = s_cell[i];
                slot = rt2d_hash32((uint32_t)cell) & mask; probing = true;
            }
            const int32_t old = atomicCAS(&s_hash[slot], -1, i);
            // (a slot's step only ever changes to a later step of the same cell: s_cell[old] names the slot's cell for good)
            if (old == -1) { s_slot[i] = (int16_t)slot; probing = false; }
            else if (s_cell[old] == cell) { atomicMax(&s_hash[slot], i); s_slot[i] = (int16_t)slot; probing = false; }
            else slot = (slot + 1) & mask;
        }
    }
    __syncthreads();
    TAIL_STAMP(5)
    // pass B: the step that owns its cell deposits (simulation.py:129-131)
#pragma unroll
    for (int u = 0; u < TAIL_MAXC; ++u) {
        if (!((cand >> u) & 1u)) continue;
        const int i = i0 + u;
        if (s_hash[(uint16_t)s_slot[i]] == i) atomicAdd(&t.H[s_cell[i]], rt2d_heat_val(gl[u]));      // ray.py:317, simulation.py:129
    }
    TAIL_STAMP(6)
}

// presentation (simulation.py:133-138)
// np.max over the map: doubles of either sign are ordered through a key whose unsigned order is the numbers' order
// (sign bit flipped for values >= 0, all bits flipped for negative ones), so one 64-bit atomicMax per wave does it and a
// map accumulated with a negative sign (rt2d_cast_heatmap's `sign`) normalises by its true maximum.
__device__ __forceinline__ unsigned long long rt2d_order_key(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double rt2d_order_value(unsigned long long k) {
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
}

__global__ void rt2d_hm_max_kernel(const double* H, int64_t size, unsigned long long* scratch) {
    // scratch[0]: order key of the maximum (0 = nothing seen yet); scratch[1]: NaN flag; blockIdx.y: map of a batch
    H += (int64_t)blockIdx.y * size; scratch += 2 * blockIdx.y;
    double m = -INFINITY; bool nan = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < size; i += (int64_t)gridDim.x * blockDim.x) {
        double v = H[i];
        if (v != v) nan = true; else m = fmax(m, v);
    }
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    const bool any_nan = __builtin_amdgcn_ballot_w64(nan) != 0ull;      // wave vote (inactive lanes count as false)
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&scratch[0], rt2d_order_key(m));
        if (any_nan) atomicMax(&scratch[1], 1ull);
    }
}

__global__ void rt2d_hm_finalize_kernel(const double* H, int64_t nx, int64_t nz, double cutoff, double power,
                                        const unsigned long long* scratch, double* out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nx * nz) return;
    H += (int64_t)blockIdx.y * nx * nz; out += (int64_t)blockIdx.y * nx * nz; scratch += 2 * blockIdx.y;
    double mx = scratch[1] ? __builtin_nan("") : rt2d_order_value(scratch[0]);
    int64_t ix = i / nz, iz = i % nz;
    double v = H[i] / mx;
    if (v > cutoff) v = cutoff;
    if (power != 1.0) v = pow(v, power);
    out[iz * nx + ix] = v;
}

// ======================================================================================================
// C ABI
// ======================================================================================================
extern "C" {

const char* rt2d_last_error(void) { return g_err.c_str(); }
int32_t rt2d_version(void) { return 100; }

static double host_eval_quad0(const std::vector<double>& blob, const DevTable& t, double x) {
    int j = 0;
    for (int k = 0; k < t.n_pieces; ++k) if (blob[t.off + (size_t)k * t.stride] <= x) j = k;
    const double* r = &blob[t.off + (size_t)j * t.stride];
    double d = x - r[0];
    return (r[1] * d + r[2]) * d + r[3];
}

static bool same_breaks(const Rt2dTable& a, const Rt2dTable& b) {
    if (a.n_pieces != b.n_pieces) return false;
    return memcmp(a.breaks, b.breaks, sizeof(double) * (a.n_pieces + 1)) == 0;
}

// appends `count` host tables that share their breaks as one fused device table of records
// [break_j, coefficients of table 0, of table 1, ...] plus a final record holding break_n
static int32_t pack_table(std::vector<double>& blob, const Rt2dTable* tabs, int count, int order, int rel_to,
                          DevTable* out, const char* name) {
    const Rt2dTable& t0 = tabs[0];
    if (t0.n_pieces < 1 || !t0.breaks || !t0.coef) return fail("table %s is empty", name);
    for (int k = 0; k < count; ++k) {
        if (tabs[k].order != order) return fail("table %s: order %d expected, got %d", name, order, tabs[k].order);
        if (!same_breaks(t0, tabs[k])) return fail("table %s: fused tables must share their breaks", name);
    }
    const int n = t0.n_pieces;
    for (int j = 0; j < n; ++j) if (!(t0.breaks[j + 1] > t0.breaks[j])) return fail("table %s: breaks not increasing", name);
    const int stride = 1 + order * count;
    out->off = (int32_t)blob.size() - rel_to;
    for (int j = 0; j < n; ++j) {
        blob.push_back(t0.breaks[j]);
        for (int k = 0; k < count; ++k)
            for (int q = 0; q < order; ++q) blob.push_back(tabs[k].coef[(size_t)j * order + q]);
    }
    blob.push_back(t0.breaks[n]);
    for (int q = 1; q < stride; ++q) blob.push_back(0.0);      // the last record is only read at [0]
    out->n_pieces = n;
    out->stride = stride;
    out->oob = t0.oob;
    out->lo = t0.breaks[0];
    out->hi = t0.breaks[n];
    // uniform-grid hint: interior breaks b_j ~ x0 + j*h (the two end pieces of a quadratic table are 1.5 h long)
    double h, x0;
    if (n >= 3) { h = (t0.breaks[n - 1] - t0.breaks[1]) / (n - 2); x0 = t0.breaks[1] - h; }
    else { h = (t0.breaks[n] - t0.breaks[0]) / n; x0 = t0.breaks[0]; }
    bool uniform = h > 0;
    for (int j = 1; j < n && uniform; ++j) if (fabs(t0.breaks[j] - (x0 + j * h)) > 0.25 * h) uniform = false;
    out->uniform = uniform ? 1 : 0;
    out->x0 = x0;
    out->inv_h = uniform ? 1.0 / h : 0.0;
    return 0;
}

extern "C++" { static int32_t noalloc_scratch(int32_t** out); }

int32_t rt2d_env_create(const Rt2dEnvDesc* d, rt2d_env_t* env_out) {
    if (!d || !env_out) return fail("rt2d_env_create: null argument");
    rt2d_env_s* env = new rt2d_env_s();
    DevEnv& e = env->dev;
    memset(&e, 0, sizeof e);
    for (int k = 0; k < 2; ++k) { e.range_min[k] = d->range_min[k]; e.range_max[k] = d->range_max[k]; }
    std::vector<double>& blob = env->host_blob;
    int32_t rc = 0;
    Rt2dTable cz[2] = {d->c, d->dz_c};
    if ((rc = pack_table(blob, cz, 2, 3, 0, &e.cz, "c/dz_c"))) { delete env; return rc; }
    e.cz_doubles = (int32_t)blob.size();
    if (d->floor.oob != RT2D_OOB_CLAMP || d->ceil.oob != RT2D_OOB_CLAMP) {
        delete env;
        return fail("rt2d_env_create: the floor and ceil tables must use RT2D_OOB_CLAMP (the stepping kernel clamps them unconditionally)");
    }
    if ((rc = pack_table(blob, &d->floor, 1, 2, 0, &e.floor, "floor"))) { delete env; return rc; }
    if ((rc = pack_table(blob, &d->ceil, 1, 2, 0, &e.ceil, "ceil"))) { delete env; return rc; }
    if ((rc = pack_table(blob, &d->dx_floor, 1, 3, 0, &e.dx_floor, "dx_floor"))) { delete env; return rc; }
    if ((rc = pack_table(blob, &d->dx_ceil, 1, 3, 0, &e.dx_ceil, "dx_ceil"))) { delete env; return rc; }
    if (blob.size() & 1) blob.push_back(0.0);
    e.step_doubles = (int32_t)blob.size();
    e.gain_off = (int32_t)blob.size();
    if ((rc = pack_table(blob, &d->impedance, 1, 3, e.gain_off, &e.imp, "impedance"))) { delete env; return rc; }
    if ((rc = pack_table(blob, d->absorb, 6, 3, e.gain_off, &e.absorb, "absorb"))) { delete env; return rc; }
    if (blob.size() & 1) blob.push_back(0.0);
    e.gain_doubles = (int32_t)blob.size() - e.gain_off;
    e.total_doubles = (int32_t)blob.size();
    e.c_surface = host_eval_quad0(blob, e.cz, 0.0);
    env->ceil_is_const = true;
    env->ceil_const = d->ceil.coef[1] + 0.0;                 // -0 -> +0
    for (int j = 0; j < d->ceil.n_pieces; ++j)
        if (d->ceil.coef[2 * j] != 0.0 || d->ceil.coef[2 * j + 1] != env->ceil_const) env->ceil_is_const = false;
    if (getenv("RT2D_NO_CEILC")) env->ceil_is_const = false;   // tests compare both kernels
    if (hipGetDevice(&env->device) != hipSuccess) { delete env; return fail("no HIP device available"); }
    if (hipDeviceGetAttribute(&env->n_cu, hipDeviceAttributeMultiprocessorCount, env->device) != hipSuccess || env->n_cu <= 0) env->n_cu = 256;
    if (hipMalloc(&env->blob, blob.size() * sizeof(double)) != hipSuccess) { delete env; return fail("hipMalloc of the table blob failed"); }
    if (hipMemcpy(env->blob, blob.data(), blob.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(env->blob); delete env; return fail("hipMemcpy of the table blob failed");
    }
    // rt2d_heatmap_accumulate_noalloc: the per-device address of its scratch array is taken HERE, on the device the
    // environment lives on (hipGetSymbolAddress may load the code object, which is not legal under stream capture), so that a
    // first rt2d_heatmap_accumulate_noalloc on this device makes no set-up call of its own
    { int32_t* unused = nullptr; if ((rc = noalloc_scratch(&unused))) { (void)hipFree(env->blob); delete env; return rc; } }
    *env_out = env;
    return 0;
}

int32_t rt2d_env_destroy(rt2d_env_t env) {
    if (!env) return 0;
    if (env->blob) (void)hipFree(env->blob);
    for (auto& kv : env->alpha) if (kv.second.dev) (void)hipFree(kv.second.dev);
    delete env;
    return 0;
}

int32_t rt2d_env_surface_speed(rt2d_env_t env, double* c0_out) {
    if (!env || !c0_out) return fail("rt2d_env_surface_speed: null argument");
    *c0_out = env->dev.c_surface;
    return 0;
}

#define RT2D_LDS_BUDGET (152 * 1024)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel, size growth) instead of once per launch: a
// 999-ray step is ten launches inside 1.3 ms, and the attribute call costs a few microseconds of host time each.
extern "C++" {
static int32_t rt2d_allow_lds(const void* kernel, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> granted;
    int dev = 0;
    HIP_OK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = granted[std::make_pair(dev, kernel)];
    if (bytes <= have) return 0;
    HIP_OK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    have = bytes;
    return 0;
}
}
#define ALLOW_LDS(kernel, bytes) do { int32_t rc_ = rt2d_allow_lds((const void*)(kernel), (size_t)(bytes)); if (rc_) return rc_; } while (0)

extern "C++" {
static int post_level(const Rt2dRayOut& o) {
    if (o.L_end || o.T_end || o.dist_to_target || o.step_target || o.T_target || o.margin) return 2;
    return (o.range || o.monotone || o.slope_max) ? 1 : 0;
}

template <bool WRITE, int POST, bool CUTOFF>
static int32_t launch_cast2(rt2d_env_t env, const CastArgs& a_in, hipStream_t stream) {
    CastArgs a = a_in;
    const int64_t n = a.n;
    const size_t cold = ((sizeof(CastCold) + 15) / 16) * 16;
    const size_t tab_bytes = (size_t)env->dev.step_doubles * sizeof(double);
    const size_t cz_bytes = (size_t)env->dev.cz_doubles * sizeof(double);
    static const int tab_env = getenv("RT2D_CAST_TAB") ? atoi(getenv("RT2D_CAST_TAB")) : -1;      // experiments: force a table mode
    const bool in_lds = cold + tab_bytes <= RT2D_LDS_BUDGET && tab_env != 0;
    static const int64_t n_cu_env = getenv("RT2D_CAST_CUS") ? atoll(getenv("RT2D_CAST_CUS")) : 0;
    const int64_t n_cu = n_cu_env > 0 ? n_cu_env : env->n_cu;
    // large fans (more than eight full waves per CU): sound speed in LDS, floor and surface through the caches (TAB = 2).
    // Below that there are too few waves to hide the cache latency of a floor record whose neighbours in the wave are
    // kilometres apart (count pass of 65 536 rays at the default step sizes 108 -> 78 ms with everything in LDS, 131 072
    // rays 112 -> 82 ms; at dx_max 50 / dz_max 25: 32 768 rays stored 1.94 -> 1.42 ms, 65 536 1.88 -> 1.44, 131 072
    // 1.74 -> 1.51, 262 144 2.15 -> 2.28).  RT2D_LAUNCH_SMALL_LDS asks for it whatever the size.
    // ... and so does a small fan whose step tables do not fit in LDS at all (the 100 km floor of tests_3.ipynb: 166 KB):
    // the sound speed from LDS and the small-fan forms of the loop instead of every record through the caches
    const bool small_lds = a.p.launch_hint == RT2D_LAUNCH_SMALL_LDS
                        || (!in_lds && tab_env < 0 && n <= n_cu * RT2D_WAVE && !getenv("RT2D_NO_SMALL_HYBRID"));
    const bool hybrid = (n > n_cu * RT2D_WAVE * 8 || small_lds) && cold + cz_bytes <= 32 * 1024 && tab_env != 0 && tab_env != 1;
    const size_t lds = cold + (hybrid ? cz_bytes : in_lds ? tab_bytes : 0);
    // Launch shape.  Measured on MI355X (tools/microbench_lanes.py): a wave steps fastest when it is alone on its CU
    // (co-resident waves cost each other 0.4-0.7 us per step even on different SIMDs) and when few of its lanes are in
    // use (a wave runs the rebound branch whenever ANY lane rebounds).  So: up to one wave per CU with as few rays per
    // wave as that allows; beyond 64 rays per wave, as many waves per workgroup as keeps every CU busy, at most 16
    // (1024 threads share one LDS copy of the tables).
    int block = RT2D_WAVE;
    a.lanes = RT2D_WAVE;
    { static const int t_env = getenv("RT2D_CAST_TIGHT") ? atoi(getenv("RT2D_CAST_TIGHT")) : -1; a.tight_off = t_env == 0 ? 1 : t_env == 1 ? 2 : 0; }
    if (n <= n_cu * RT2D_WAVE) {
        int lanes = 1;
        while ((int64_t)lanes * n_cu < n) lanes *= 2;
        a.lanes = lanes;
        // (experiment: the same rays per CU spread over more waves of fewer lanes, RT2D_CAST_LANES = lanes per wave)
        static const int lanes_env = getenv("RT2D_CAST_LANES") ? atoi(getenv("RT2D_CAST_LANES")) : 0;
        // (measured on the refine cones, 4590 near-vertical rays of up to 100 000 steps: 669 / 800 / 891 ms per three
        // iterations with 32 / 16 / 8 rays per wave -- co-resident waves cost more than shared rebound branches)
        // More than 16 rays per CU: 16 rays per wave and up to four waves per CU (one per SIMD) -- a wave runs the rebound
        // branch whenever any of its rays rebounds, and with a bounded rebound count the shorter waves pay for the
        // company: stored casts at dx_max 50 / dz_max 25, 20 rebounds, 6000 / 12 288 / 16 384 rays 1.03 -> 0.94, 1.27 -> 0.88,
        // 1.24 -> 0.89 ms (8192: 0.99 either way; 8 rays per wave: slower again).  Not without a rebound limit: rays that
        // rebound every few steps keep the LDS busy, and the refine cones (4590 rays) took 669 / 800 ms at 32 / 16 per wave.
        if (lanes > 16 && lanes_env == 0 && a.p.n_rebounds_max >= 0 && !hybrid && in_lds) {
            block = (lanes / 16) * RT2D_WAVE;
            a.lanes = lanes = 16;
        }
        if (lanes_env >= 1 && lanes_env < lanes && (lanes_env & (lanes_env - 1)) == 0 &&
            (lanes / lanes_env) <= (hybrid || !in_lds ? 16 : RT2D_TAB1_MAX_WAVES)) {          // within the instance's launch bound
            block = (lanes / lanes_env) * RT2D_WAVE;
            a.lanes = lanes_env;
        }
    } else {
        // smallest workgroup with which all workgroups are resident at once (LDS decides how many fit on a CU); with the
        // small LDS footprint of the hybrid mode: two waves per workgroup, ten workgroups per CU
        // (fewer rays per wave here, i.e. more waves: no gain.  Count pass of 65 536 rays at the default step sizes with 64 /
        // 32 / 16 rays per wave: 78 / 82 / 136 ms -- the longest wave is ONE ray that rebounds on almost every step)
        // ... except that a fan which still fits one round of four-wave workgroups with fewer rays per wave takes them (as
        // above: bounded rebounds, all tables in LDS): stored casts of 24 576 / 32 768 rays 1.35 -> 1.12, 1.37 -> 1.11 ms
        if (!hybrid && in_lds && a.p.n_rebounds_max >= 0) {
            int l = 16;
            while (l < RT2D_WAVE && (int64_t)l * n_cu * RT2D_TAB1_MAX_WAVES < n) l *= 2;
            a.lanes = l;
        }
        const int64_t waves = (n + a.lanes - 1) / a.lanes;
        const int64_t per_cu = lds > 0 ? (int64_t)((160 * 1024) / lds > 0 ? (160 * 1024) / lds : 1) : 16;
        int wpb = 1;
        while (wpb < 16 && (waves + wpb - 1) / wpb > n_cu * per_cu) wpb *= 2;
        if (hybrid && !small_lds) wpb = 2;
        if (const char* w = getenv("RT2D_CAST_WPB")) { const int v = atoi(w); if (v >= 1 && v <= 16) wpb = v; }
        if (!hybrid && in_lds && wpb > RT2D_TAB1_MAX_WAVES) wpb = RT2D_TAB1_MAX_WAVES;     // that instance's launch bound
        block = wpb * RT2D_WAVE;
    }
    const int64_t rays_per_block = (int64_t)(block / RT2D_WAVE) * a.lanes;
    if (getenv("RT2D_DEBUG_LDS")) fprintf(stderr, "[rt2d] cast: n %lld cold %zu tab %zu cz %zu hybrid %d in_lds %d lds %zu block %d lanes %d\n", (long long)n, cold, tab_bytes, cz_bytes, (int)hybrid, (int)in_lds, lds, block, a.lanes);
    const int64_t grid = (n + rays_per_block - 1) / rays_per_block;
    if (grid > 0x7fffffff) return fail("too many rays for one launch");
    if (hybrid && small_lds && WRITE && POST == 1 && !CUTOFF && env->ceil_is_const && block <= RT2D_TAB1_MAX_WAVES * RT2D_WAVE) {
        // (the one combination the side rays of rt2d_cast_heatmap are cast with; every other one takes the instance below)
        auto k = rt2d_cast_kernel<WRITE, 3, (POST == 1 ? 1 : 0), false, true>;
        a.ceil_const = env->ceil_const;
        ALLOW_LDS(k, lds);
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(block), lds, stream, a);
    } else if (hybrid) {
        auto k = env->ceil_is_const ? rt2d_cast_kernel<WRITE, 2, POST, CUTOFF, true> : rt2d_cast_kernel<WRITE, 2, POST, CUTOFF, false>;
        a.ceil_const = env->ceil_const;
        ALLOW_LDS(k, lds);
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(block), lds, stream, a);
    } else if (in_lds) {
        auto k = env->ceil_is_const ? rt2d_cast_kernel<WRITE, 1, POST, CUTOFF, true> : rt2d_cast_kernel<WRITE, 1, POST, CUTOFF, false>;
        a.ceil_const = env->ceil_const;
        ALLOW_LDS(k, lds);
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(block), lds, stream, a);
    } else {
        hipLaunchKernelGGL((rt2d_cast_kernel<WRITE, 0, POST, CUTOFF, false>), dim3((unsigned)grid), dim3(block), lds, stream, a);
    }
    HIP_OK(hipGetLastError());
    return 0;
}

template <bool WRITE>
static int32_t launch_cast(rt2d_env_t env, const CastArgs& a, hipStream_t stream) {
    if (a.p.power_cutoff > 0) return launch_cast2<WRITE, 2, true>(env, a, stream);
    switch (post_level(a.out)) {
        case 2: return launch_cast2<WRITE, 2, false>(env, a, stream);
        case 1: return launch_cast2<WRITE, 1, false>(env, a, stream);
        default: return launch_cast2<WRITE, 0, false>(env, a, stream);
    }
}
}  // extern "C++"

static BinTarget make_bin_target(double* H, const int32_t* res, const int32_t* shape, double sign) {
    BinTarget t; memset(&t, 0, sizeof t);
    t.H = H; t.nx = shape[0]; t.nz = shape[1]; t.sign = sign;
    t.geo.res_x = res[0]; t.geo.res_z = res[1]; t.geo.rx = res[0]; t.geo.rz = res[1];
    t.geo.inv_rx = 1.0 / res[0]; t.geo.inv_rz = 1.0 / res[1]; t.geo.lim = 2147483648.0;
    t.dbg = getenv("RT2D_FUSE_DBG") ? atoi(getenv("RT2D_FUSE_DBG")) : 0;
    return t;
}

static int32_t check_cast(rt2d_env_t env, const double* angles, int64_t n, const double* src, const Rt2dParams* p,
                          const double* init_slope, const double* init_mult, const Rt2dRayOut* out) {
    if (!env || !src || !p || !out) return fail("rt2d_cast: null argument");
    if (n < 0) return fail("rt2d_cast: negative ray count");
    if (n == 0) return 0;
    if (!angles && !(init_slope && init_mult)) return fail("rt2d_cast: angles is null");
    if ((init_slope == nullptr) != (init_mult == nullptr)) return fail("rt2d_cast: init_slope and init_mult go together");
    if (!(p->dx_max > 0) || !(p->dz_max > 0)) return fail("rt2d_cast: dx_max and dz_max must be positive");
    if (p->n_steps_max < 0) return fail("rt2d_cast: negative n_steps_max");
    if (p->rebound_cap < 0) return fail("rt2d_cast: negative rebound_cap");
    if (p->power_cutoff != p->power_cutoff) return fail("rt2d_cast: power_cutoff is NaN");
    if (p->power_cutoff > 0 && !(p->cutoff_freq > 0)) return fail("rt2d_cast: power_cutoff needs cutoff_freq > 0");
    return 0;
}

// development aid, not part of include/rt2d.h: on how many of n pseudo-random operand pairs (magnitudes log-uniform over
// [10^lo10, 10^hi10], random signs, every 16th numerator +0) rt2d_div_fast and the compiler's IEEE division differ
__global__ void __launch_bounds__(256) rt2d_div_check_kernel(int64_t n, uint64_t seed, double lo10, double hi10, unsigned long long* bad, int mode) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t s = seed + 0x9e3779b97f4a7c15ull * (uint64_t)(i + 1);
    auto next = [&]() { s ^= s >> 30; s *= 0xbf58476d1ce4e5b9ull; s ^= s >> 27; s *= 0x94d049bb133111ebull; s ^= s >> 31; return s; };
    auto draw = [&]() {
        const uint64_t u = next();
        const double t = (double)(u >> 11) * (1.0 / 9007199254740992.0);
        const double m = exp10(lo10 + (hi10 - lo10) * t) * (1.0 + (double)(next() >> 12) * (1.0 / 4503599627370496.0));
        return (u & 1ull) ? -m : m;
    };
    double a = draw();
    const double b = draw();
    if ((i & 15) == 0) a = 0.0;
    if (mode == 1) {                                                      // square roots of |b| (every 16th: of +0)
        const double v = ((i & 15) == 0) ? 0.0 : fabs(b);
        if (__double_as_longlong(sqrt(v)) != __double_as_longlong(rt2d_sqrt_fast(v))) atomicAdd(bad, 1ull);
        return;
    }
    const double q_ieee = a / b, q_fast = rt2d_div_fast(a, b);
    if (__double_as_longlong(q_ieee) != __double_as_longlong(q_fast)) atomicAdd(bad, 1ull);
}
int64_t rt2d_debug_div_check(int64_t n, uint64_t seed, double lo10, double hi10, unsigned long long* bad_dev, int32_t mode, void* stream) {
    if (n <= 0 || !bad_dev) return -1;
    hipLaunchKernelGGL(rt2d_div_check_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, seed, lo10, hi10, bad_dev, (int)mode);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// development aid, not part of include/rt2d.h: waves of the following cast launches log [start, end, hw id, xcc id]
// (4 words per wave) into `buf` (device, room for `cap` waves); NULL switches it off
static unsigned long long* g_wave_log = nullptr;
static int64_t g_wave_log_cap = 0;
void rt2d_debug_wave_log(unsigned long long* buf, int64_t cap) { g_wave_log = buf; g_wave_log_cap = cap; }

static CastArgs make_cast_args(rt2d_env_t env, const double* angles, int64_t n, const double* src, const Rt2dParams* p,
                               const double* init_slope, const double* init_mult, const Rt2dRayOut* out) {
    CastArgs a;
    memset(&a, 0, sizeof a);
    a.env = env->dev; a.blob = env->blob; a.angles = angles; a.init_slope = init_slope; a.init_mult = init_mult;
    a.n = n; a.src_x = src[0]; a.src_z = src[1]; a.p = *p; a.out = *out;
    a.g_cut = p->power_cutoff > 0 ? 10 * log10(p->power_cutoff) : -INFINITY;
    a.inv_dx_max = 1.0 / p->dx_max;
    a.wave_log = g_wave_log; a.wave_log_cap = g_wave_log_cap;
    return a;
}

int32_t rt2d_cast_count(rt2d_env_t env, const double* angles, int64_t n, const double* src, const Rt2dParams* params,
                        const double* init_slope, const double* init_mult, const Rt2dRayOut* out, void* stream) {
    int32_t rc = check_cast(env, angles, n, src, params, init_slope, init_mult, out);
    if (rc) return rc;
    if (n == 0) return 0;
    CastArgs a = make_cast_args(env, angles, n, src, params, init_slope, init_mult, out);
    return launch_cast<false>(env, a, (hipStream_t)stream);
}

int32_t rt2d_cast_write(rt2d_env_t env, const double* angles, int64_t n, const double* src, const Rt2dParams* params,
                        const double* init_slope, const double* init_mult, const Rt2dRayOut* out,
                        const int64_t* tile_off, double* X, double* Z, double* C, void* stream) {
    int32_t rc = check_cast(env, angles, n, src, params, init_slope, init_mult, out);
    if (rc) return rc;
    if (!tile_off || !X || !Z) return fail("rt2d_cast_write: tile_off, X and Z are required");
    // (a lane addresses its tile with a 32-bit byte offset: 512 B per stored point of a ray)
    if (params->n_steps_max > (1 << 23) - 2) return fail("rt2d_cast_write: at most 8 388 606 steps per ray when trajectories are stored");
    if (n == 0) return 0;
    CastArgs a = make_cast_args(env, angles, n, src, params, init_slope, init_mult, out);
    a.tile_off = tile_off; a.X = X; a.Z = Z; a.C = C;
    return launch_cast<true>(env, a, (hipStream_t)stream);
}

int32_t rt2d_cast_heatmap(rt2d_env_t env, const double* angles, int64_t n, const double* src, const Rt2dParams* params,
                          const double* init_slope, const double* init_mult, const uint8_t* ray_mask, double freq,
                          const int32_t* res, const int32_t* shape, double sign, double* H, const Rt2dRayOut* out,
                          uint8_t* irregular, int64_t* n_violations, const double* loss, int32_t loss_cap, void* stream) {
    Rt2dRayOut none; memset(&none, 0, sizeof none);
    if (!out) out = &none;
    int32_t rc = check_cast(env, angles, n, src, params, init_slope, init_mult, out);
    if (rc) return rc;
    if (!res || !shape || !H) return fail("rt2d_cast_heatmap: null argument");
    if (params->power_cutoff > 0) return fail("rt2d_cast_heatmap: power_cutoff is not supported by the fused kernel");
    if (out->L_end || out->T_end || out->dist_to_target || out->step_target || out->T_target || out->margin || out->reb_step)
        return fail("rt2d_cast_heatmap: only n_pts, stop_code, n_rebounds, status, range, monotone and slope_max are produced");
    if (!(freq > 0)) return fail("rt2d_cast_heatmap: freq must be positive");
    if (res[0] <= 0 || res[1] <= 0) return fail("rt2d_cast_heatmap: resolution must be positive");
    if (res[0] >= (1 << 20) || res[1] >= (1 << 20)) return fail("rt2d_cast_heatmap: resolution too coarse for the fused kernel");
    if (shape[0] <= 0 || shape[1] <= 0 || (int64_t)shape[0] * shape[1] >= (1ll << 31)) return fail("rt2d_cast_heatmap: bad shape");
    if (n == 0) return 0;
    CastArgs a = make_cast_args(env, angles, n, src, params, init_slope, init_mult, out);
    a.p.rebound_cap = 0;
    a.ray_mask = ray_mask; a.heat_freq = freq;
    a.bin = make_bin_target(H, res, shape, sign);
    a.irregular = irregular; a.n_violations = (unsigned long long*)n_violations;
    a.fuse_step_doubles = env->dev.dx_floor.off;                           // sound speed, floor, surface
    a.fuse_abs_doubles = (env->dev.absorb.n_pieces + 1) * env->dev.absorb.stride;
    a.fuse_abs_doubles += a.fuse_abs_doubles & 1;
    a.lanes = RT2D_WAVE;
    a.ceil_const = env->ceil_const;
    const size_t cold = ((sizeof(CastCold) + 15) / 16) * 16;
    const size_t tab_bytes = ((size_t)a.fuse_step_doubles + a.fuse_abs_doubles) * sizeof(double);
    const size_t per_wave = 8 * RT2D_WAVE * sizeof(double);
    if (cold + tab_bytes + per_wave > RT2D_LDS_BUDGET)
        return fail("rt2d_cast_heatmap: the sound-speed, floor, surface and absorption tables (%zu bytes) leave no room in LDS; "
                    "use rt2d_cast_write + rt2d_gain + rt2d_heatmap_accumulate", tab_bytes);
    int waves = (int)((RT2D_LDS_BUDGET - cold - tab_bytes) / per_wave);
    if (waves > (loss ? RT2D_FUSE2_MAX_WAVES : RT2D_FUSE_MAX_WAVES)) waves = loss ? RT2D_FUSE2_MAX_WAVES : RT2D_FUSE_MAX_WAVES;
    static const int waves_env = getenv("RT2D_FUSE_WAVES") ? atoi(getenv("RT2D_FUSE_WAVES")) : 0;
    if (waves_env > 0 && waves_env < waves) waves = waves_env;
    // fans of few waves: smaller workgroups, so that every CU gets one and every wave a SIMD of its own (65 536 rays of
    // 100 000 steps: 194 -> 155 ms).  The workgroups then hold most of the LDS of every CU: a stored cast that is to run
    // beside this kernel (the side rays) must be launched with RT2D_LAUNCH_SMALL_LDS, or it waits for this kernel to end.
    while (waves > 1 && (n + (int64_t)waves * RT2D_WAVE - 1) / ((int64_t)waves * RT2D_WAVE) < env->n_cu) waves /= 2;
    const int block = waves * RT2D_WAVE;
    const size_t lds = cold + tab_bytes + (size_t)waves * per_wave;
    const int64_t grid = (n + block - 1) / block;
    if (getenv("RT2D_DEBUG_LDS")) fprintf(stderr, "[rt2d] fused: cold %zu tab %zu waves %d lds %zu grid %lld; pieces: cz %d floor %d surface %d floor slope %d surface slope %d absorption %d\n", cold, tab_bytes, waves, lds, (long long)grid,
                                          env->dev.cz.n_pieces, env->dev.floor.n_pieces, env->dev.ceil.n_pieces, env->dev.dx_floor.n_pieces, env->dev.dx_ceil.n_pieces, env->dev.absorb.n_pieces);
    if (grid > 0x7fffffff) return fail("too many rays for one launch");
    a.fuse_loss = loss; a.fuse_loss_cap = loss ? loss_cap : 0;
    auto k = loss ? (env->ceil_is_const ? rt2d_cast_kernel<false, 1, 1, false, true, 2> : rt2d_cast_kernel<false, 1, 1, false, false, 2>)
                  : (env->ceil_is_const ? rt2d_cast_kernel<false, 1, 1, false, true, 1> : rt2d_cast_kernel<false, 1, 1, false, false, 1>);
    ALLOW_LDS(k, lds);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(block), lds, (hipStream_t)stream, a);
    HIP_OK(hipGetLastError());
    return 0;
}

int32_t rt2d_traj_layout(const int32_t* n_pts, int64_t n, int64_t* tile_off, int64_t* total_out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!n_pts || !tile_off) return fail("rt2d_traj_layout: null argument");
    if (n < 0) return fail("rt2d_traj_layout: negative ray count");
    const int64_t n_tiles = (n + RT2D_TILE - 1) / RT2D_TILE;
    if (n_tiles == 0) {
        HIP_OK(hipMemsetAsync(tile_off, 0, sizeof(int64_t), stream));
    } else {
        const int wpb = 4;
        hipLaunchKernelGGL(rt2d_tile_len_kernel, dim3((unsigned)((n_tiles + wpb - 1) / wpb)), dim3(wpb * RT2D_WAVE), 0, stream,
                           n_pts, n, n_tiles, tile_off);
        hipLaunchKernelGGL(rt2d_tile_scan_kernel, dim3(1), dim3(n_tiles <= 64 ? 64 : n_tiles <= 256 ? 256 : 1024), 0, stream, n_tiles, tile_off);
        HIP_OK(hipGetLastError());
    }
    if (total_out) {
        HIP_OK(hipMemcpyAsync(total_out, tile_off + n_tiles, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
    }
    return 0;
}

int32_t rt2d_fan_summary(const int32_t* status, const double* range, const uint8_t* ray_mask, int64_t n,
                         const int64_t* tile_off, double* summary, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!status || !summary) return fail("rt2d_fan_summary: null argument");
    if (n < 0) return fail("rt2d_fan_summary: negative ray count");
    HIP_OK(hipMemsetAsync(summary, 0, 8 * sizeof(double), stream));
    if (n == 0) return 0;
    const int64_t n_tiles = (n + RT2D_TILE - 1) / RT2D_TILE;
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(rt2d_fan_summary_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, stream,
                       status, range, ray_mask, n, tile_off, n_tiles, summary);
    HIP_OK(hipGetLastError());
    return 0;
}

int32_t rt2d_layout_summary(const int32_t* n_pts, const int32_t* status, const double* range, int64_t n,
                            int64_t* tile_off, double* summary, int32_t n_summary, uint8_t* ok_mask, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!n_pts || !status || !range || !tile_off || !summary) return fail("rt2d_layout_summary: null argument");
    if (n < 0) return fail("rt2d_layout_summary: negative ray count");
    const int64_t n_tiles = (n + RT2D_TILE - 1) / RT2D_TILE;
    if (n_summary < 8 || n_summary > 1024) return fail("rt2d_layout_summary: the summary has 8 doubles (plus the caller's spare words)");
    if (n_tiles == 0 || n_tiles > 1024) {
        if (ok_mask) return fail("rt2d_layout_summary: the mask is only produced for fans of at most 65 536 rays");
        if (n_summary > 8) HIP_OK(hipMemsetAsync(summary + 8, 0, (size_t)(n_summary - 8) * sizeof(double), stream));
        int32_t rc = rt2d_traj_layout(n_pts, n, tile_off, nullptr, stream_);
        return rc ? rc : rt2d_fan_summary(status, range, nullptr, n, tile_off, summary, stream_);
    }
    hipLaunchKernelGGL(rt2d_small_summary_kernel, dim3(1), dim3(1024), 0, stream, n_pts, status, range, n, n_tiles, tile_off, summary, n_summary, ok_mask);
    HIP_OK(hipGetLastError());
    return 0;
}

int32_t rt2d_traj_compact(const int64_t* tile_off_src, const int64_t* tile_off_dst, int64_t n,
                          const double* X_src, const double* Z_src, const double* C_src,
                          double* X_dst, double* Z_dst, double* C_dst, void* stream) {
    if (!tile_off_src || !tile_off_dst || !X_src || !Z_src || !X_dst || !Z_dst) return fail("rt2d_traj_compact: null argument");
    if ((C_src == nullptr) != (C_dst == nullptr)) return fail("rt2d_traj_compact: C_src and C_dst go together");
    if (n <= 0) return n == 0 ? 0 : fail("rt2d_traj_compact: negative ray count");
    const int64_t n_tiles = (n + RT2D_TILE - 1) / RT2D_TILE;
    if (n_tiles > 0x7fffffff) return fail("rt2d_traj_compact: too many tiles");
    const unsigned split = n_tiles < 512 ? 16 : 1;       // few tiles: several workgroups per tile
    hipLaunchKernelGGL(rt2d_compact_kernel, dim3((unsigned)n_tiles, split), dim3(256), 0, (hipStream_t)stream,
                       tile_off_src, tile_off_dst, n_tiles, X_src, Z_src, C_src, X_dst, Z_dst, C_dst);
    HIP_OK(hipGetLastError());
    return 0;
}

int32_t rt2d_post(rt2d_env_t env, const double* X, const double* Z, const int64_t* tile_off, const int32_t* n_pts,
                  int64_t n, double* dL, double* L, double* C, double* dT, double* T, void* stream) {
    if (!env || !X || !Z || !tile_off || !n_pts) return fail("rt2d_post: null argument");
    if (n <= 0) return n == 0 ? 0 : fail("rt2d_post: negative ray count");
    PostArgs a; memset(&a, 0, sizeof a);
    a.env = env->dev; a.blob = env->blob; a.X = X; a.Z = Z; a.tile_off = tile_off; a.n_pts = n_pts; a.n = n;
    a.dL = dL; a.L = L; a.C = C; a.dT = dT; a.T = T;
    const size_t lds = (size_t)env->dev.cz_doubles * sizeof(double);
    if (lds > RT2D_LDS_BUDGET) return fail("rt2d_post: sound-speed table does not fit in LDS");
    ALLOW_LDS(rt2d_post_kernel, lds);
    const int block = (n <= 64 * 512) ? 64 : 256;
    hipLaunchKernelGGL(rt2d_post_kernel, dim3((unsigned)((n + block - 1) / block)), dim3(block), lds, (hipStream_t)stream, a);
    HIP_OK(hipGetLastError());
    return 0;
}

int32_t rt2d_rebound_loss(rt2d_env_t env, const Rt2dRayOut* reb, int32_t rebound_cap, int64_t n, const double* freqs,
                          int32_t n_freqs, double* loss, void* stream) {
    if (!env || !reb || !freqs || !loss) return fail("rt2d_rebound_loss: null argument");
    if (!reb->n_rebounds || !reb->reb_surface || !reb->reb_velocity || !reb->reb_angle || !reb->reb_pos_z)
        return fail("rt2d_rebound_loss: the rebound log (n_rebounds, reb_surface, reb_velocity, reb_angle, reb_pos_z) is required");
    if (n < 0 || n_freqs < 0 || rebound_cap < 0) return fail("rt2d_rebound_loss: negative count");
    if (n == 0 || n_freqs == 0 || rebound_cap == 0) return 0;
    GainArgs a; memset(&a, 0, sizeof a);
    a.env = env->dev; a.blob = env->blob;
    a.n_rebounds = reb->n_rebounds; a.reb_surface = reb->reb_surface;
    a.reb_velocity = reb->reb_velocity; a.reb_angle = reb->reb_angle; a.reb_pos_z = reb->reb_pos_z;
    a.rebound_cap = rebound_cap; a.n = n; a.freqs = freqs; a.n_freqs = n_freqs;
    const int64_t pairs = n * rebound_cap;
    hipLaunchKernelGGL(rt2d_reb_loss_kernel, dim3((unsigned)((pairs + 255) / 256), (unsigned)n_freqs), dim3(256), 0,
                       (hipStream_t)stream, a, loss, 0.0);
    HIP_OK(hipGetLastError());
    return 0;
}

int32_t rt2d_gain(rt2d_env_t env, const double* X, const double* Z, const int64_t* tile_off, const int32_t* n_pts,
                  const Rt2dRayOut* reb, int32_t rebound_cap, int64_t n, const double* freqs, int32_t n_freqs,
                  int64_t total, double* G, double* G_end, double* loss_ws, void* stream) {
    if (!env || !X || !Z || !tile_off || !n_pts || !reb || !freqs) return fail("rt2d_gain: null argument");
    if (!reb->n_rebounds || !reb->reb_step || !reb->reb_surface || !reb->reb_velocity || !reb->reb_angle || !reb->reb_pos_z)
        return fail("rt2d_gain: the rebound log (n_rebounds, reb_step, reb_surface, reb_velocity, reb_angle, reb_pos_z) is required");
    if (n < 0 || n_freqs < 0) return fail("rt2d_gain: negative count");
    if (n == 0 || n_freqs == 0) return 0;
    GainArgs a; memset(&a, 0, sizeof a);
    a.env = env->dev; a.blob = env->blob; a.X = X; a.Z = Z; a.tile_off = tile_off; a.n_pts = n_pts;
    a.n_rebounds = reb->n_rebounds; a.reb_step = reb->reb_step; a.reb_surface = reb->reb_surface;
    a.reb_velocity = reb->reb_velocity; a.reb_angle = reb->reb_angle; a.reb_pos_z = reb->reb_pos_z;
    a.rebound_cap = rebound_cap; a.n = n; a.freqs = freqs; a.n_freqs = n_freqs; a.total = total; a.G = G; a.G_end = G_end;
    const size_t lds = (size_t)env->dev.gain_doubles * sizeof(double);
    if (lds > RT2D_LDS_BUDGET) return fail("rt2d_gain: absorption tables do not fit in LDS");
    const int fch = n_freqs >= 8 ? 8 : (n_freqs >= 3 ? 4 : n_freqs);      // frequencies per lane: amortises the 6 table evals
    const char* tp = getenv("RT2D_GAIN_TWO_PHASE_MAX_RAYS");              // tests force either path with this
    const size_t scan_lds = 3 * (size_t)GAIN_SCAN_ROWS * RT2D_WAVE * 8 + (size_t)(rebound_cap > 0 ? rebound_cap : 1) * RT2D_WAVE * 12;
    // (measured crossovers: one frequency 65536 rays 1.40 -> 1.06 ms, 131072 rays 1.59 -> 1.96 ms; 65536 rays x 8
    // frequencies 4.8 -> 3.7 ms, x 64 frequencies slower again: the scan launches a workgroup per (tile, frequency))
    const bool two_phase = tp ? n <= atoll(tp) : (n <= 64 * 512 || (n <= 90000 && n * (int64_t)n_freqs <= 600000));
    const int64_t n_tiles = (n + RT2D_TILE - 1) / RT2D_TILE;
    static const int part_env = getenv("RT2D_GAIN_SCAN_PART") ? atoi(getenv("RT2D_GAIN_SCAN_PART")) : -1;    // experiments: 0 / 1
    // few (tile, frequency) pairs: a part of a tile's rays per workgroup, all of them resident at once (rt2d_gain_scan_part_kernel);
    // its LDS holds the rebound log of 8 rays instead of 64, so it still fits when rays have hundreds of rebounds (no rebound
    // limit at the reference's default step sizes) -- the ray-sequential loop below needs 1.1 us per point on such fans
    const bool use_part = part_env == 1 || (part_env != 0 && n_tiles * (int64_t)n_freqs * (RT2D_WAVE / GAIN_PART_RAYS) <= 512);
    const size_t part_lds = 3 * (size_t)GAIN_SCAN_THREADS * 8 + (size_t)(rebound_cap > 0 ? rebound_cap : 1) * GAIN_PART_RAYS * 12;
    // (the log does not fit: losses scattered into dG by their own kernel, the scan runs without a log)
    const bool scatter = G && two_phase && use_part && part_lds > RT2D_LDS_BUDGET && loss_ws && rebound_cap > 0;
    if (scatter) {
        const unsigned split = (unsigned)(n_tiles >= 512 ? 4 : (2048 + n_tiles - 1) / n_tiles);
        dim3 grid1((unsigned)n_tiles, split, (unsigned)((n_freqs + fch - 1) / fch));
        void (*k1)(const GainArgs) = fch == 8 ? rt2d_gain_dg_kernel<8> : fch == 4 ? rt2d_gain_dg_kernel<4>
                                    : fch == 2 ? rt2d_gain_dg_kernel<2> : rt2d_gain_dg_kernel<1>;
        ALLOW_LDS(k1, lds);
        hipLaunchKernelGGL(k1, grid1, dim3(256), lds, (hipStream_t)stream, a);
        const int64_t pairs = n * rebound_cap;
        hipLaunchKernelGGL(rt2d_reb_loss_kernel, dim3((unsigned)((pairs + 255) / 256), (unsigned)n_freqs), dim3(256), 0,
                           (hipStream_t)stream, a, loss_ws, 0.0);
        hipLaunchKernelGGL(rt2d_reb_scatter_kernel, dim3((unsigned)((pairs + 255) / 256), (unsigned)n_freqs), dim3(256), 0,
                           (hipStream_t)stream, a, (const double*)loss_ws);
        GainArgs b = a;
        b.rebound_cap = 0;
        const size_t lds0 = 3 * (size_t)GAIN_SCAN_THREADS * 8 + GAIN_PART_RAYS * 12;
        ALLOW_LDS(rt2d_gain_scan_part_kernel, lds0);
        hipLaunchKernelGGL(rt2d_gain_scan_part_kernel, dim3((unsigned)n_tiles, (unsigned)n_freqs, RT2D_WAVE / GAIN_PART_RAYS),
                           dim3(GAIN_SCAN_THREADS), lds0, (hipStream_t)stream, b);
        HIP_OK(hipGetLastError());
        return 0;
    }
    if (G && two_phase && (use_part ? part_lds : scan_lds) <= RT2D_LDS_BUDGET) {
        // small fan: per-point dG on the whole chip, then the running sums (see rt2d_gain_dg_kernel)
        const unsigned split = (unsigned)(n_tiles >= 512 ? 4 : (2048 + n_tiles - 1) / n_tiles);
        dim3 grid1((unsigned)n_tiles, split, (unsigned)((n_freqs + fch - 1) / fch));
        void (*k1)(const GainArgs) = fch == 8 ? rt2d_gain_dg_kernel<8> : fch == 4 ? rt2d_gain_dg_kernel<4>
                                    : fch == 2 ? rt2d_gain_dg_kernel<2> : rt2d_gain_dg_kernel<1>;
        ALLOW_LDS(k1, lds);
        hipLaunchKernelGGL(k1, grid1, dim3(256), lds, (hipStream_t)stream, a);
        if (use_part) {
            ALLOW_LDS(rt2d_gain_scan_part_kernel, part_lds);
            hipLaunchKernelGGL(rt2d_gain_scan_part_kernel, dim3((unsigned)n_tiles, (unsigned)n_freqs, RT2D_WAVE / GAIN_PART_RAYS),
                               dim3(GAIN_SCAN_THREADS), part_lds, (hipStream_t)stream, a);
        } else {
            ALLOW_LDS(rt2d_gain_scan_kernel, scan_lds);
            hipLaunchKernelGGL(rt2d_gain_scan_kernel, dim3((unsigned)n_tiles, (unsigned)n_freqs), dim3(GAIN_SCAN_THREADS), scan_lds, (hipStream_t)stream, a);
        }
        HIP_OK(hipGetLastError());
        return 0;
    }
    const int block = (n <= 64 * 512) ? 64 : 256;
    dim3 grid((unsigned)((n + block - 1) / block), (unsigned)((n_freqs + fch - 1) / fch));
    void (*kern)(const GainArgs);
    // every group full and G written: the stores of an iteration are a fixed count, which lets the wait for the next point
    // be a counted one (see the kernel)
    const bool full = a.G && n_freqs % fch == 0 && !getenv("RT2D_GAIN_NOT_FULL");
    if (loss_ws && rebound_cap > 0) {
        // rebound losses first, one thread per (ray, rebound, frequency); the gain loop then only reads them
        const int64_t pairs = n * rebound_cap;
        hipLaunchKernelGGL(rt2d_reb_loss_kernel, dim3((unsigned)((pairs + 255) / 256), (unsigned)n_freqs), dim3(256), 0,
                           (hipStream_t)stream, a, loss_ws, 0.0);
        a.loss = loss_ws;
        kern = full ? (fch == 8 ? rt2d_gain_kernel<8, true, true> : fch == 4 ? rt2d_gain_kernel<4, true, true>
                     : fch == 2 ? rt2d_gain_kernel<2, true, true> : rt2d_gain_kernel<1, true, true>)
                    : (fch == 8 ? rt2d_gain_kernel<8, true, false> : fch == 4 ? rt2d_gain_kernel<4, true, false>
                     : fch == 2 ? rt2d_gain_kernel<2, true, false> : rt2d_gain_kernel<1, true, false>);
    } else {
        kern = fch == 8 ? rt2d_gain_kernel<8, false, false> : fch == 4 ? rt2d_gain_kernel<4, false, false>
             : fch == 2 ? rt2d_gain_kernel<2, false, false> : rt2d_gain_kernel<1, false, false>;
    }
    ALLOW_LDS(kern, lds);
    hipLaunchKernelGGL(kern, grid, dim3(block), lds, (hipStream_t)stream, a);
    HIP_OK(hipGetLastError());
    return 0;
}


extern "C++" {
// -alpha(f, z) / 1000 of ONE frequency as a polynomial of degree ALPHA_DEG per depth piece of the absorption table (see
// rt2d_onepass_kernel).  Host: the exact form (model_absorption.py:37-44 on the six quadratics of the piece, the very
// arithmetic of rt2d_alpha: only + * /) at the Chebyshev nodes of the piece, interpolated in long double.  The end pieces
// reach to the environment's rectangle (every stored point but a source outside it lies inside).  Device: the finished
// table is checked against the exact form (rt2d_alpha_verify_kernel); a table that misses 1e-14 relative anywhere is kept
// as "not ok" and the caller uses the exact kernel.  One table per (environment, frequency), made once (one small
// allocation, one copy, one launch, one 8-byte read back) and kept until the environment is destroyed.
static int32_t alpha_table_for(rt2d_env_t env, double freq, AlphaDesc* out, bool* ok) {
    *ok = false;
    const DevTable& ab = env->dev.absorb;
    if (!ab.uniform || ab.stride != 19 || ab.n_pieces < 1) return 0;
    uint64_t key; memcpy(&key, &freq, sizeof key);
    std::lock_guard<std::mutex> lock(env->alpha_mu);
    auto it = env->alpha.find(key);
    if (it == env->alpha.end()) {
        if (env->alpha.size() >= 4096) return 0;                          // (a sweep over thousands of frequencies: exact kernel)
        const int n = ab.n_pieces, D = ALPHA_DEG, N = D + 1;
        const double* base = env->host_blob.data() + env->dev.gain_off + ab.off;
        const double f = freq / 1000, ff = f * f;
        std::vector<double> tab((size_t)n * ALPHA_REC);
        for (int j = 0; j < n; ++j) {
            const double* rec = base + (size_t)j * 19;
            double lo = rec[0], hi = rec[19];
            if (j == 0) lo = fmin(lo, env->dev.range_min[1]) - 1.0;
            if (j == n - 1) hi = fmax(hi, env->dev.range_max[1]) + 1.0;
            long double V[N][N + 1];
            for (int k = 0; k < N; ++k) {
                const long double u = (1.0L + cosl(3.14159265358979323846264338327950288L * (k + 0.5L) / N)) / 2;
                const double z = (double)((long double)lo + ((long double)hi - lo) * u);
                const double d = z - rec[0];
                auto quad = [&](int q) { return (rec[1 + 3 * q] * d + rec[2 + 3 * q]) * d + rec[3 + 3 * q]; };
                const double f1 = quad(0), f2 = quad(1), A2 = quad(2), E1 = quad(3), E2 = quad(4), E3 = quad(5);
                const double alpha_1 = 0.106 * (f1 * ff) / (f1 * f1 + ff);
                const double alpha_2 = A2 * (f2 * ff) / (f2 * f2 + ff);
                const double alpha_3 = 0.00049 * ff;
                const double alpha = alpha_1 * E1 + alpha_2 * E2 + alpha_3 * E3;
                const long double uu = ((long double)z - lo) / ((long double)hi - lo);
                long double pw = 1;
                for (int m = 0; m < N; ++m) { V[k][m] = pw; pw *= uu; }
                V[k][N] = -(long double)alpha / 1000;
            }
            for (int c = 0; c < N; ++c) {                                  // Gauss-Jordan with partial pivoting
                int piv = c;
                for (int k = c + 1; k < N; ++k) if (fabsl(V[k][c]) > fabsl(V[piv][c])) piv = k;
                if (piv != c) for (int m = 0; m <= N; ++m) std::swap(V[c][m], V[piv][m]);
                if (V[c][c] == 0) return 0;
                for (int k = 0; k < N; ++k) {
                    if (k == c) continue;
                    const long double g = V[k][c] / V[c][c];
                    for (int m = c; m <= N; ++m) V[k][m] -= g * V[c][m];
                }
            }
            double* o = &tab[(size_t)j * ALPHA_REC];
            o[0] = lo; o[1] = hi;
            long double scale = 1;
            for (int m = 0; m < N; ++m) { o[2 + m] = (double)(V[m][N] / V[m][m] / scale); scale *= ((long double)hi - lo); }
        }
        rt2d_env_s::AlphaTab t;
        t.n_pieces = n;
        unsigned long long* worst = nullptr;
        HIP_OK(hipMalloc(&t.dev, tab.size() * sizeof(double) + 8));
        worst = (unsigned long long*)(t.dev + tab.size());
        unsigned long long zero = 0, got = ~0ull;
        hipError_t e1 = hipMemcpy(t.dev, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice);
        hipError_t e2 = hipMemcpy(worst, &zero, 8, hipMemcpyHostToDevice);
        AlphaDesc ad; ad.tab = t.dev; ad.n_pieces = n; ad.pad = 0; ad.x0 = ab.x0; ad.inv_h = ab.inv_h;
        if (e1 == hipSuccess && e2 == hipSuccess) {
            hipLaunchKernelGGL(rt2d_alpha_verify_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)0, env->dev, env->blob, ad, freq, worst);
            if (hipGetLastError() == hipSuccess && hipMemcpy(&got, worst, 8, hipMemcpyDeviceToHost) == hipSuccess) {
                memcpy(&t.max_rel_err, &got, 8);
                t.ok = t.max_rel_err <= 1e-14;
            }
        }
        it = env->alpha.emplace(key, t).first;
    }
    const rt2d_env_s::AlphaTab& t = it->second;
    out->tab = t.dev; out->n_pieces = t.n_pieces; out->pad = 0; out->x0 = ab.x0; out->inv_h = ab.inv_h;
    *ok = t.ok;
    return 0;
}

static int32_t gain_heatmap_impl(rt2d_env_t env, const double* X, const double* Z, const int64_t* tile_off, const int32_t* n_pts,
                          const Rt2dRayOut* reb, int32_t rebound_cap, int64_t n, double freq, const uint8_t* ray_mask,
                          const int32_t* res, const int32_t* shape, double* H, double* G, double* G_end,
                          uint8_t* irregular, int64_t* n_violations, double* loss_ws, void* stream,
                          const int32_t* g_slot, int64_t g_room,
                          const int32_t* ray_list = nullptr, int64_t n_list = 0, bool loss_ready = false) {
    if (!env || !X || !Z || !tile_off || !n_pts || !reb || !res || !shape || !H) return fail("rt2d_gain_heatmap: null argument");
    if (!reb->n_rebounds || !reb->reb_step || !reb->reb_surface || !reb->reb_velocity || !reb->reb_angle || !reb->reb_pos_z)
        return fail("rt2d_gain_heatmap: the rebound log (n_rebounds, reb_step, reb_surface, reb_velocity, reb_angle, reb_pos_z) is required");
    if (n < 0) return fail("rt2d_gain_heatmap: negative count");
    if (!(freq > 0)) return fail("rt2d_gain_heatmap: freq must be positive");
    if (res[0] <= 0 || res[1] <= 0 || res[0] >= (1 << 20) || res[1] >= (1 << 20)) return fail("rt2d_gain_heatmap: bad resolution");
    if (shape[0] <= 0 || shape[1] <= 0 || (int64_t)shape[0] * shape[1] >= (1ll << 31)) return fail("rt2d_gain_heatmap: bad shape");
    if (n == 0) return 0;
    GainHeatArgs b; memset(&b, 0, sizeof b);
    GainArgs& a = b.g;
    a.env = env->dev; a.blob = env->blob; a.X = X; a.Z = Z; a.tile_off = tile_off; a.n_pts = n_pts;
    a.n_rebounds = reb->n_rebounds; a.reb_step = reb->reb_step; a.reb_surface = reb->reb_surface;
    a.reb_velocity = reb->reb_velocity; a.reb_angle = reb->reb_angle; a.reb_pos_z = reb->reb_pos_z;
    a.rebound_cap = rebound_cap; a.n = n; a.n_freqs = 1; a.G = G; a.G_end = G_end;
    b.bin = make_bin_target(H, res, shape, 1.0);
    b.freq = freq; b.ray_mask = ray_mask; b.irregular = irregular; b.n_violations = (unsigned long long*)n_violations;
    a.g_slot = g_slot; a.g_room = g_room;
    b.ray_list = ray_list;
    static const int exact_env = getenv("RT2D_GH_EXACT") ? atoi(getenv("RT2D_GH_EXACT")) : 0;
    b.exact = exact_env;
    const int64_t n_launch = ray_list ? n_list : n;          // threads with a ray (list mode: an upper bound, the count is in the list)
    static const int block_env = getenv("RT2D_GH_BLOCK") ? atoi(getenv("RT2D_GH_BLOCK")) : 0;       // development knob
    // big fans: 3 workgroups of 4 waves per CU (with 8 waves a CU holds one workgroup: 2 waves per SIMD)
    const int block = (block_env >= 64 && block_env <= 512) ? block_env / 64 * 64 : (n_launch <= 64 * 512) ? 64 : 256;
    const size_t lds = ((size_t)env->dev.gain_doubles + (size_t)(block / RT2D_WAVE) * 8 * RT2D_WAVE) * sizeof(double);
    if (lds > RT2D_LDS_BUDGET) return fail("rt2d_gain_heatmap: absorption tables do not fit in LDS");
    void (*kern)(const GainHeatArgs) = rt2d_gain_heatmap_kernel<false>;
    if (loss_ws && rebound_cap > 0) {
        const int64_t pairs = n * rebound_cap;
        if (!loss_ready)
            hipLaunchKernelGGL(rt2d_reb_loss_kernel, dim3((unsigned)((pairs + 255) / 256), 1u), dim3(256), 0, (hipStream_t)stream, a, loss_ws, freq);
        a.loss = loss_ws;
        kern = rt2d_gain_heatmap_kernel<true>;
        if (!G && !G_end && !b.exact) kern = rt2d_gain_heatmap_kernel<true, true>;
        // ... and when nothing but the map is asked of every ray of a whole fan: the lean form (rt2d_onepass_kernel), if the
        // alpha table of this frequency passes its check on the device
        static const int lean_env = getenv("RT2D_GH_LEAN") ? atoi(getenv("RT2D_GH_LEAN")) : 1;
        if (lean_env && !G && !G_end && !b.exact && !ray_list && !irregular && b.bin.dbg == 0) {
            AlphaDesc ad; bool ok = false;
            int32_t rc = alpha_table_for(env, freq, &ad, &ok);
            if (rc) return rc;
            if (ok) {
                const int lean_block = 256;
                const size_t lean_lds = ((size_t)ad.n_pieces * ALPHA_REC + (size_t)(lean_block / RT2D_WAVE) * 8 * RT2D_WAVE) * sizeof(double);
                if (lean_lds <= RT2D_LDS_BUDGET) {
                    ALLOW_LDS(rt2d_onepass_kernel, lean_lds);
                    hipLaunchKernelGGL(rt2d_onepass_kernel, dim3((unsigned)((n + lean_block - 1) / lean_block)), dim3(lean_block), lean_lds,
                                       (hipStream_t)stream, b, ad);
                    HIP_OK(hipGetLastError());
                    return 0;
                }
            }
        }
    }
    ALLOW_LDS(kern, lds);
    if (n_launch > 0)
        hipLaunchKernelGGL(kern, dim3((unsigned)((n_launch + block - 1) / block)), dim3(block), lds, (hipStream_t)stream, b);
    HIP_OK(hipGetLastError());
    return 0;
}
}  // extern "C++"

int32_t rt2d_alpha_table_error(rt2d_env_t env, double freq, double* max_rel_err, int32_t* in_use) {
    if (!env || !max_rel_err || !in_use) return fail("rt2d_alpha_table_error: null argument");
    if (!(freq > 0)) return fail("rt2d_alpha_table_error: freq must be positive");
    AlphaDesc ad; bool ok = false;
    int32_t rc = alpha_table_for(env, freq, &ad, &ok);
    if (rc) return rc;
    uint64_t key; memcpy(&key, &freq, sizeof key);
    std::lock_guard<std::mutex> lock(env->alpha_mu);
    auto it = env->alpha.find(key);
    *max_rel_err = it == env->alpha.end() ? INFINITY : it->second.max_rel_err;
    *in_use = ok ? 1 : 0;
    return 0;
}

int32_t rt2d_gain_heatmap(rt2d_env_t env, const double* X, const double* Z, const int64_t* tile_off, const int32_t* n_pts,
                          const Rt2dRayOut* reb, int32_t rebound_cap, int64_t n, double freq, const uint8_t* ray_mask,
                          const int32_t* res, const int32_t* shape, double* H, double* G, double* G_end,
                          uint8_t* irregular, int64_t* n_violations, double* loss_ws, void* stream) {
    return gain_heatmap_impl(env, X, Z, tile_off, n_pts, reb, rebound_cap, n, freq, ray_mask, res, shape, H, G, G_end,
                             irregular, n_violations, loss_ws, stream, nullptr, 0);
}

int32_t rt2d_dist(const double* X, const double* Z, const int64_t* tile_off, const int32_t* n_pts, int64_t n,
                  const double* target, double* dist, void* stream) {
    if (!X || !Z || !tile_off || !n_pts || !target || !dist) return fail("rt2d_dist: null argument");
    if (n <= 0) return n == 0 ? 0 : fail("rt2d_dist: negative ray count");
    const int block = (n <= 64 * 512) ? 64 : 256;
    hipLaunchKernelGGL(rt2d_dist_kernel, dim3((unsigned)((n + block - 1) / block)), dim3(block), 0, (hipStream_t)stream,
                       X, Z, tile_off, n_pts, n, target[0], target[1], dist);
    HIP_OK(hipGetLastError());
    return 0;
}

int32_t rt2d_target_products(rt2d_env_t env, const double* X, const double* Z, const double* T, const double* const* G,
                             const int64_t* tile_off, const int32_t* n_pts, const int32_t* ray_ids, int64_t n_sel,
                             const double* target, const double* freqs, int32_t n_freqs,
                             int32_t* step_out, double* T_target, double* Tmult, void* stream) {
    if (!env || !X || !Z || !T || !tile_off || !n_pts || !target || !step_out || !T_target)
        return fail("rt2d_target_products: null argument");
    if (n_freqs < 0 || (n_freqs > 0 && (!G || !freqs || !Tmult))) return fail("rt2d_target_products: frequencies need G, freqs and Tmult");
    if (n_sel <= 0) return n_sel == 0 ? 0 : fail("rt2d_target_products: negative ray count");
    if (n_sel > 0x7fffffff) return fail("rt2d_target_products: too many rays for one launch");
    TargetArgs a; memset(&a, 0, sizeof a);
    a.env = env->dev; a.blob = env->blob; a.X = X; a.Z = Z; a.T = T; a.G = G; a.tile_off = tile_off; a.n_pts = n_pts;
    a.ray_ids = ray_ids; a.n_sel = n_sel; a.tx = target[0]; a.tz = target[1]; a.freqs = freqs; a.n_freqs = n_freqs;
    a.step_out = step_out; a.T_target = T_target; a.Tmult = Tmult;
    hipLaunchKernelGGL(rt2d_target_products_kernel, dim3((unsigned)n_sel), dim3(RT2D_WAVE), 0, (hipStream_t)stream, a);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C++" {
static int knob_pow2(const char* name, int dflt, int lo, int hi) {
    const char* v = getenv(name);
    if (!v) return dflt;
    const int k = atoi(v);
    return (k >= lo && k <= hi && (k & (k - 1)) == 0) ? k : dflt;
}
static int knob_even(const char* name, int dflt, int lo, int hi) {
    const char* v = getenv(name);
    if (!v) return dflt;
    const int k = atoi(v);
    return (k >= lo && k <= hi && (k & 1) == 0) ? k : dflt;
}
static int32_t heatmap_accumulate_impl(const double* X, const double* Z, const double* G, const int64_t* tile_off,
                                       const int32_t* n_pts, int64_t n, const uint8_t* ray_mask,
                                       const int32_t* res, const int32_t* shape,
                                       double* H, int64_t* n_dropped, int32_t* work, hipStream_t stream,
                                       const double* const* Gs = nullptr, int32_t n_freqs = 0,
                                       const int32_t* rest_list = nullptr, int64_t n_rest = 0,
                                       const int32_t* g_slot = nullptr, int64_t g_room = 0, double* H_minor = nullptr,
                                       int64_t max_pts = 0);
}

int32_t rt2d_heatmap_accumulate(const double* X, const double* Z, const double* G, const int64_t* tile_off,
                                const int32_t* n_pts, int64_t n, const uint8_t* ray_mask,
                                const int32_t* res, const int32_t* shape,
                                double* H, int64_t* n_dropped, int32_t* work, void* stream_) {
    return heatmap_accumulate_impl(X, Z, G, tile_off, n_pts, n, ray_mask, res, shape, H, n_dropped, work, (hipStream_t)stream_);
}

int32_t rt2d_heatmap_accumulate_hinted(const double* X, const double* Z, const double* G, const int64_t* tile_off,
                                       const int32_t* n_pts, int64_t n, const uint8_t* ray_mask,
                                       const int32_t* res, const int32_t* shape,
                                       double* H, int64_t* n_dropped, int32_t* work, int64_t max_pts, void* stream_) {
    if (max_pts < 0) return fail("rt2d_heatmap_accumulate_hinted: negative max_pts");
    return heatmap_accumulate_impl(X, Z, G, tile_off, n_pts, n, ray_mask, res, shape, H, n_dropped, work, (hipStream_t)stream_,
                                   nullptr, 0, nullptr, 0, nullptr, 0, nullptr, max_pts);
}

extern "C++" {
static int32_t heatmap_accumulate_impl(const double* X, const double* Z, const double* G, const int64_t* tile_off,
                                       const int32_t* n_pts, int64_t n, const uint8_t* ray_mask,
                                       const int32_t* res, const int32_t* shape,
                                       double* H, int64_t* n_dropped, int32_t* work, hipStream_t stream,
                                       const double* const* Gs, int32_t n_freqs,
                                       const int32_t* rest_list, int64_t n_rest, const int32_t* g_slot, int64_t g_room, double* H_minor,
                                       int64_t max_pts) {
    if (!X || !Z || (!G && !Gs) || !tile_off || !n_pts || !res || !shape || !H) return fail("rt2d_heatmap_accumulate: null argument");
    if (n <= 0) return n == 0 ? 0 : fail("rt2d_heatmap_accumulate: negative ray count");
    if (res[0] <= 0 || res[1] <= 0) return fail("rt2d_heatmap_accumulate: resolution must be positive");
    if (shape[0] <= 0 || shape[1] <= 0 || (int64_t)shape[0] * shape[1] >= (1ll << 31)) return fail("rt2d_heatmap_accumulate: bad shape");
    if (n > 0x7fffffff) return fail("rt2d_heatmap_accumulate: too many rays for one launch");
    HeatArgs a; memset(&a, 0, sizeof a);
    a.X = X; a.Z = Z; a.G = G; a.tile_off = tile_off; a.n_pts = n_pts; a.n = n;
    a.ray_mask = ray_mask;
    a.Gs = Gs; a.n_freqs = Gs ? n_freqs : 0; a.map_stride = (int64_t)shape[0] * shape[1];
    n_freqs = a.n_freqs;
    a.geo.res_x = res[0]; a.geo.res_z = res[1]; a.geo.rx = res[0]; a.geo.rz = res[1];
    a.geo.inv_rx = 1.0 / res[0]; a.geo.inv_rz = 1.0 / res[1];
    a.geo.lim = (res[0] < (1 << 20) && res[1] < (1 << 20)) ? 2147483648.0 : 0.0;
    a.nx = shape[0]; a.nz = shape[1]; a.H = H; a.n_dropped = n_dropped;
    // LDS per one-ray workgroup: cap * 8 B of hash + cell_cap * 4 B of cached cells; it decides how many rays a CU works on
    // (development knobs: the hash and the LDS table are addressed with masks, the doubles behind the cached cells must stay
    // 8-byte aligned -- anything but a power of two / an even count falls back to the default)
    static const int hm_cap = knob_pow2("RT2D_HM_CAP", 4096, 64, 8192);
    static const int hm_cells = knob_even("RT2D_HM_CELLS", 4096, 64, 16384);
    a.cap = hm_cap; a.cell_cap = hm_cells;
    // Rays of tens of thousands of points (the reference's default step sizes: 100 001) have tens of thousands of candidate
    // cells, and the general kernel walks the ray once per partition of its hash: twice the slots, half the walks (the side
    // rays of the 65 536-ray default-step fan: 33 -> 16 ms).  Only when the caller says so (max_pts, or the room of a
    // compact gain array): the LDS it costs takes a workgroup per CU away from ordinary rays.
    if ((max_pts > 16384 || g_room > 16384) && !getenv("RT2D_HM_CAP")) a.cap = 8192;
    // the tile kernel walks a ray sequentially (latency ~ steps), the general kernel spreads one ray over a wave:
    // below ~6000 rays the latter finishes first (measured with the rows of a tile shared by up to 16 waves)
    static const int64_t fast_min_rays = getenv("RT2D_HEATMAP_FAST_MIN_RAYS") ? atoll(getenv("RT2D_HEATMAP_FAST_MIN_RAYS")) : 6000;
    // (the tile kernel works on 32-bit cells: every point inside the map must have |x|, |z| < 2^31)
    const bool small_coords = res[0] < (1 << 20) && res[1] < (1 << 20) &&
                              (int64_t)shape[0] * res[0] < (1ll << 31) && (int64_t)shape[1] * res[1] < (1ll << 31);
    a.g_slot = g_slot; a.g_room = g_room;
    // general kernel: a workgroup takes `group` consecutive rays of its list when there are rays enough to fill the chip
    // that way, and sums their deposits in LDS before they go to the map (then 2048 cached cells per ray leave room for
    // three workgroups per CU as before)
    static const int group_env = getenv("RT2D_HM_GROUP") ? atoi(getenv("RT2D_HM_GROUP")) : 0;
    auto launch_general = [&](int64_t items_est, int64_t items_max) -> int32_t {
        int64_t group = group_env > 0 ? group_env : items_est / 4096;
        group = group < 1 ? 1 : (group > 4 ? 4 : group);     // measured on the 37 000 turned-back rays of the 1M-ray C2 fan:
        // 1 ray 1.64 ms / 29.6 M atomics, 4 rays 1.26 ms / 18.1 M, 8 rays 1.46 ms / 17.0 M (the table is full: with 4096 slots
        // 11.6 M atomics but one workgroup less per CU and 2.7 ms) -- these rays are the ones that scatter, their
        // neighbours share fewer cells than the rays of the tile kernel do
        a.group = (int32_t)group;
        static const int acc_env = knob_pow2("RT2D_HM_ACC", 1024, 64, 4096);
        a.acc_cap = (group > 1 && n_freqs == 0) ? acc_env : 0;
        if (a.acc_cap && a.cell_cap > 2048) a.cell_cap = 2048;
        const int64_t groups = (items_max + group - 1) / group;
        const int64_t blocks = groups < 16384 ? (groups + 127) / 128 * 128 : 16384;
        const size_t lds = (size_t)a.cap * 8 + (size_t)a.cell_cap * 4 + (size_t)a.acc_cap * 12;
        if (blocks > 0) hipLaunchKernelGGL(rt2d_heatmap_kernel, dim3((unsigned)blocks), dim3(HM_THREADS), lds, stream, a);
        HIP_OK(hipGetLastError());
        return 0;
    };
    if (rest_list) {
        // the caller's own list of rays ([0] = their number, [2 + k] = ray ids), all of them for the general kernel
        a.work = const_cast<int32_t*>(rest_list);
        return launch_general(n_rest, n_rest);
    }
    const bool fast = work != nullptr && shape[1] <= 256 && n > fast_min_rays && small_coords;
    if (n_freqs > 0 && H_minor && !rest_list) {
        // map-minor scratch (see MINOR in the tile kernel): zeroed here, filled by both kernels, added into the maps at the end
        // groups of 8 maps when that needs no more scratch than groups of 4 (the size the header asks for)
        a.minor_fb = ((n_freqs + 7) / 8 * 8 == (n_freqs + 3) / 4 * 4) ? 8 : 4;
        static const int fb_env = getenv("RT2D_MINOR_FB") ? atoi(getenv("RT2D_MINOR_FB")) : 0;      // experiments: 4
        if (fb_env == 4) a.minor_fb = 4;
        a.H_minor = H_minor;
        HIP_OK(hipMemsetAsync(H_minor, 0, (size_t)((n_freqs + 3) / 4) * 4 * (size_t)a.map_stride * sizeof(double), stream));
    }
    if (work && !fast) HIP_OK(hipMemsetAsync(work, 0xff, sizeof(int32_t), stream));     // work[0] = -1: every ray went general
    if (fast) {
        a.work = work;
        HIP_OK(hipMemsetAsync(work, 0, 2 * sizeof(int32_t), stream));
        const int64_t n_tiles = (n + RT2D_TILE - 1) / RT2D_TILE;
        static const int max_split = getenv("RT2D_HM_SPLIT") ? atoi(getenv("RT2D_HM_SPLIT")) : 16;
        int split = 1;
        while (split < max_split && n_tiles * split < 4096) split *= 2;    // enough waves to give every SIMD several
        a.split = split;
        const int64_t n_waves = n_tiles * split;
        static const int wpb_env = getenv("RT2D_HM_WPB") ? atoi(getenv("RT2D_HM_WPB")) : 0;
        const int wpb = wpb_env > 0 ? wpb_env : ((n_waves <= 1024) ? 1 : 4);
        const unsigned blocks = (unsigned)((n_waves + wpb - 1) / wpb);
        if (n_freqs > 0 && H_minor && a.minor_fb == 8) {
            const unsigned groups = (unsigned)((n_freqs + 7) / 8);
            if (shape[1] <= 64) hipLaunchKernelGGL((rt2d_heatmap_fast_kernel<1, 8, true>), dim3(blocks, groups), dim3(wpb * RT2D_WAVE), 0, stream, a);
            else hipLaunchKernelGGL((rt2d_heatmap_fast_kernel<4, 8, true>), dim3(blocks, groups), dim3(wpb * RT2D_WAVE), 0, stream, a);
        } else if (n_freqs > 0 && H_minor) {
            const unsigned groups = (unsigned)((n_freqs + 3) / 4);
            if (shape[1] <= 64) hipLaunchKernelGGL((rt2d_heatmap_fast_kernel<1, 4, true>), dim3(blocks, groups), dim3(wpb * RT2D_WAVE), 0, stream, a);
            else hipLaunchKernelGGL((rt2d_heatmap_fast_kernel<4, 4, true>), dim3(blocks, groups), dim3(wpb * RT2D_WAVE), 0, stream, a);
        } else if (n_freqs > 0) {
            const unsigned groups = (unsigned)((n_freqs + 3) / 4);
            if (shape[1] <= 64) hipLaunchKernelGGL((rt2d_heatmap_fast_kernel<1, 4>), dim3(blocks, groups), dim3(wpb * RT2D_WAVE), 0, stream, a);
            else hipLaunchKernelGGL((rt2d_heatmap_fast_kernel<4, 4>), dim3(blocks, groups), dim3(wpb * RT2D_WAVE), 0, stream, a);
        } else if (shape[1] <= 64) hipLaunchKernelGGL((rt2d_heatmap_fast_kernel<1, 1>), dim3(blocks), dim3(wpb * RT2D_WAVE), 0, stream, a);
        else hipLaunchKernelGGL((rt2d_heatmap_fast_kernel<4, 1>), dim3(blocks), dim3(wpb * RT2D_WAVE), 0, stream, a);
        HIP_OK(hipGetLastError());
    }
    // general kernel: everything (no workspace) or what the fast kernel queued (how many is only known on the device: a
    // few per cent of the fan as a rule); blocks stride over the items
    int32_t rc = launch_general(fast ? n / 32 : n, n);
    if (!rc && a.H_minor) {
        hipLaunchKernelGGL(rt2d_hm_minor_add_kernel, dim3((unsigned)((a.map_stride + 255) / 256), (unsigned)n_freqs), dim3(256), 0, stream,
                           (const double*)H_minor, a.map_stride, n_freqs, a.minor_fb, H);
        HIP_OK(hipGetLastError());
    }
    return rc;
}
}  // extern "C++"

// ------------------------------------------------------------------------------------------------------
// rt2d_heatmap_accumulate for callers that cannot allocate: the work list of the tile kernel (a counter, a spare word and
// one ray id per queued ray) lives in this file-scope device array instead of a caller-provided workspace.  Every GPU
// the library runs on holds its own copy (4 MiB) until the process ends; host code only ever uses the per-device
// address obtained from hipGetSymbolAddress, never the array's name.
#define RT2D_NOALLOC_MAX_RAYS (1 << 20)
static __device__ __align__(256) int32_t scratch[RT2D_NOALLOC_MAX_RAYS + 2];

#define RT2D_MAX_DEVICES 64
extern "C++" {
static int32_t noalloc_scratch(int32_t** out) {
    static std::mutex mu;
    static int32_t* ptr[RT2D_MAX_DEVICES];          // per device: each GPU has its own copy of the array at its own address
    int dev = 0;
    HIP_OK(hipGetDevice(&dev));
    if (dev < 0 || dev >= RT2D_MAX_DEVICES) return fail("rt2d_heatmap_accumulate_noalloc: device %d out of range", dev);
    std::lock_guard<std::mutex> lock(mu);
    if (!ptr[dev]) {
        void* p = nullptr;
        HIP_OK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail("rt2d_heatmap_accumulate_noalloc: no device address for the scratch array");
        ptr[dev] = (int32_t*)p;
    }
    *out = ptr[dev];
    return 0;
}
}

int32_t rt2d_heatmap_accumulate_noalloc(const double* X, const double* Z, const double* G, const int64_t* tile_off,
                                        const int32_t* n_pts, int64_t n, const uint8_t* ray_mask,
                                        const int32_t* res, const int32_t* shape,
                                        double* H, int64_t* n_dropped, void* stream) {
    if (n > RT2D_NOALLOC_MAX_RAYS)
        return fail("rt2d_heatmap_accumulate_noalloc: %lld rays exceed the built-in work list of %d rays; "
                    "use rt2d_heatmap_accumulate with a caller-provided workspace", (long long)n, RT2D_NOALLOC_MAX_RAYS);
    int32_t* work = nullptr;
    int32_t rc = noalloc_scratch(&work);
    if (rc) return rc;
    // the implementation zeroes the two header words of the list (hipMemsetAsync on `stream`) before the tile kernel fills
    // it: nothing of the previous call's contents is read
    return heatmap_accumulate_impl(X, Z, G, tile_off, n_pts, n, ray_mask, res, shape, H, n_dropped, work, (hipStream_t)stream);
}

int32_t rt2d_heatmap_accumulate_multi(const double* X, const double* Z, const double* const* Gs, int32_t n_freqs,
                                      const int64_t* tile_off, const int32_t* n_pts, int64_t n, const uint8_t* ray_mask,
                                      const int32_t* res, const int32_t* shape,
                                      double* H, int64_t* n_dropped, int32_t* work, double* H_minor, void* stream) {
    if (!Gs) return fail("rt2d_heatmap_accumulate_multi: null argument");
    if (n_freqs <= 0) return n_freqs == 0 ? 0 : fail("rt2d_heatmap_accumulate_multi: negative frequency count");
    return heatmap_accumulate_impl(X, Z, nullptr, tile_off, n_pts, n, ray_mask, res, shape, H, n_dropped, work, (hipStream_t)stream, Gs, n_freqs,
                                   nullptr, 0, nullptr, 0, H_minor);
}

// the library's own second stream per device (rt2d_gain_heatmap_rest), made on first use and kept
struct RestSide { hipStream_t stream; hipEvent_t ready, losses, done; };
extern "C++" {
static int32_t rest_side(RestSide** out) {
    static std::mutex mu;
    static RestSide side[RT2D_MAX_DEVICES];
    static bool made[RT2D_MAX_DEVICES];
    int dev = 0;
    HIP_OK(hipGetDevice(&dev));
    if (dev < 0 || dev >= RT2D_MAX_DEVICES) return fail("rt2d_gain_heatmap_rest: device %d out of range", dev);
    std::lock_guard<std::mutex> lock(mu);
    if (!made[dev]) {
        HIP_OK(hipStreamCreateWithFlags(&side[dev].stream, hipStreamNonBlocking));
        HIP_OK(hipEventCreateWithFlags(&side[dev].ready, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&side[dev].losses, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&side[dev].done, hipEventDisableTiming));
        made[dev] = true;
    }
    *out = &side[dev];
    return 0;
}
}

// The side rays of rt2d_gain_heatmap_rest by rt2d_small_tail_kernel in list mode: gain and binning of a ray by ONE workgroup
// in ONE launch, no gain array (instead of the ray-sequential gain of the listed rays followed by the general binner).
extern "C++" {
static int32_t launch_tail_list(rt2d_env_t env, const double* X, const double* Z, const int64_t* tile_off, const int32_t* n_pts,
                                const Rt2dRayOut* reb, int32_t rebound_cap, int64_t n, double freq, const int32_t* res,
                                const int32_t* shape, double* H, int64_t* n_dropped, const int32_t* rest_list, int64_t n_rest,
                                int32_t* long_list, hipStream_t stream) {
    TailArgs t; memset(&t, 0, sizeof t);
    GainArgs& a = t.g;
    a.env = env->dev; a.blob = env->blob; a.X = X; a.Z = Z; a.tile_off = tile_off; a.n_pts = n_pts;
    a.n_rebounds = reb->n_rebounds; a.reb_step = reb->reb_step; a.reb_surface = reb->reb_surface;
    a.reb_velocity = reb->reb_velocity; a.reb_angle = reb->reb_angle; a.reb_pos_z = reb->reb_pos_z;
    a.rebound_cap = rebound_cap; a.n = n; a.n_freqs = 1;
    t.ray_list = rest_list;
    t.geo.res_x = res[0]; t.geo.res_z = res[1]; t.geo.rx = res[0]; t.geo.rz = res[1];
    t.geo.inv_rx = 1.0 / res[0]; t.geo.inv_rz = 1.0 / res[1];
    t.geo.lim = (res[0] < (1 << 20) && res[1] < (1 << 20)) ? 2147483648.0 : 0.0;
    t.nx = shape[0]; t.nz = shape[1]; t.H = H; t.n_dropped = n_dropped; t.freq = freq;
    t.long_list = long_list;
    if (long_list) HIP_OK(hipMemsetAsync(long_list, 0, 2 * sizeof(int32_t), stream));
    static const int lean_env = getenv("RT2D_GH_LEAN") ? atoi(getenv("RT2D_GH_LEAN")) : 1;
    bool ok = false;
    if (lean_env) { int32_t rc = alpha_table_for(env, freq, &t.ad, &ok); if (rc) return rc; }
    t.alpha_ok = ok ? 1 : 0;
    const size_t lds = (size_t)(rebound_cap > 0 ? rebound_cap : 1) * 12 + 8;
    hipLaunchKernelGGL(rt2d_small_tail_kernel, dim3((unsigned)((n_rest + 127) / 128 * 128)), dim3(TAIL_THREADS), lds, stream, t);
    HIP_OK(hipGetLastError());
    return 0;
}
}

int32_t rt2d_gain_heatmap_rest(rt2d_env_t env, const double* X, const double* Z, const int64_t* tile_off, const int32_t* n_pts,
                               const Rt2dRayOut* reb, int32_t rebound_cap, int64_t n, double freq, const uint8_t* ray_mask,
                               const int32_t* res, const int32_t* shape, double* H,
                               const int32_t* rest_list, int64_t n_rest, const int32_t* g_slot, int64_t g_room, double* G_rest,
                               int64_t* n_dropped, int64_t* n_violations, double* loss_ws, int32_t* rest_long, void* stream) {
    if (n_rest < 0 || (n_rest > 0 && (!rest_list || !g_slot || !G_rest || g_room <= 0)))
        return fail("rt2d_gain_heatmap_rest: rest_list, g_slot, G_rest and g_room describe the rays of ray_mask value 2");
    // the same argument checks as gain_heatmap_impl, BEFORE anything is dereferenced or queued on either stream
    if (!env || !X || !Z || !tile_off || !n_pts || !reb || !res || !shape || !H) return fail("rt2d_gain_heatmap_rest: null argument");
    if (!reb->n_rebounds || !reb->reb_step || !reb->reb_surface || !reb->reb_velocity || !reb->reb_angle || !reb->reb_pos_z)
        return fail("rt2d_gain_heatmap_rest: the rebound log (n_rebounds, reb_step, reb_surface, reb_velocity, reb_angle, reb_pos_z) is required");
    if (n < 0 || rebound_cap < 0) return fail("rt2d_gain_heatmap_rest: negative count");
    if (!(freq > 0)) return fail("rt2d_gain_heatmap_rest: freq must be positive");
    if (res[0] <= 0 || res[1] <= 0 || res[0] >= (1 << 20) || res[1] >= (1 << 20)) return fail("rt2d_gain_heatmap_rest: bad resolution");
    if (shape[0] <= 0 || shape[1] <= 0 || (int64_t)shape[0] * shape[1] >= (1ll << 31)) return fail("rt2d_gain_heatmap_rest: bad shape");
    if (n == 0) return 0;
    static const bool side_on = !(getenv("RT2D_REST_SIDE_STREAM") && atoi(getenv("RT2D_REST_SIDE_STREAM")) == 0);
    if (n_rest == 0 || !side_on || !(loss_ws && rebound_cap > 0)) {
        // one stream: the pass writes the side rays' gain, the general kernel follows it
        int32_t rc = gain_heatmap_impl(env, X, Z, tile_off, n_pts, reb, rebound_cap, n, freq, ray_mask, res, shape, H,
                                       n_rest > 0 ? G_rest : nullptr, nullptr, nullptr, n_violations, loss_ws, stream,
                                       n_rest > 0 ? g_slot : nullptr, g_room);
        if (rc || n_rest == 0) return rc;
        return heatmap_accumulate_impl(X, Z, G_rest, tile_off, n_pts, n, nullptr, res, shape, H, n_dropped, nullptr, (hipStream_t)stream,
                                       nullptr, 0, rest_list, n_rest, g_slot, g_room);
    }
    // Two streams: the side rays are few (a few per cent) and their chain -- gain of the listed rays by the same kernel in
    // list mode, then the general kernel -- is two short launches of long-lived waves; queued BEHIND the big pass they
    // add their whole length to the map's time, beside it they disappear in it.  Both start when the rebound losses are
    // there; the caller's stream waits for the side chain at the end.
    RestSide* side = nullptr;
    int32_t rc = rest_side(&side);
    if (rc) return rc;
    // (the side stream and its two events are shared by every caller on this device: one call queues at a time)
    static std::mutex queue_mu;
    std::lock_guard<std::mutex> queue_lock(queue_mu);
    hipStream_t main = (hipStream_t)stream;
    // (the one-workgroup-per-ray kernel works out the rebound losses of its ray itself: the side stream starts from HERE when
    // it opens with that kernel, and only the chain that reads the loss table waits for the launch below)
    HIP_OK(hipEventRecord(side->ready, main));
    {   // rebound losses, once, on the caller's stream (the same launch gain_heatmap_impl would make)
        GainArgs a; memset(&a, 0, sizeof a);
        a.env = env->dev; a.blob = env->blob; a.n_rebounds = reb->n_rebounds; a.reb_step = reb->reb_step; a.reb_surface = reb->reb_surface;
        a.reb_velocity = reb->reb_velocity; a.reb_angle = reb->reb_angle; a.reb_pos_z = reb->reb_pos_z;
        a.rebound_cap = rebound_cap; a.n = n; a.n_freqs = 1;
        const int64_t pairs = n * rebound_cap;
        if (pairs > 0)
            hipLaunchKernelGGL(rt2d_reb_loss_kernel, dim3((unsigned)((pairs + 255) / 256), 1u), dim3(256), 0, main, a, loss_ws, freq);
        HIP_OK(hipGetLastError());
    }
    HIP_OK(hipEventRecord(side->losses, main));
    // The side rays: one workgroup per ray does gain and binning in one launch (rt2d_small_tail_kernel in list mode) when a
    // ray fits a workgroup.  Known to (g_room bounds the stored points of a side ray), or sorted out by the kernel: a longer
    // ray is appended to rest_long, and that list -- as a rule empty -- goes through the chain below.  1 M-ray C2 fan, side
    // chain beside the big pass: 3.9 + 3.0 ms (ray-sequential gain of the listed rays, then the general binner) -> one launch.
    static const int tail_env = getenv("RT2D_REST_TAIL") ? atoi(getenv("RT2D_REST_TAIL")) : 1;
    const bool by_tail = tail_env != 0 && (size_t)rebound_cap * 12 + 8 <= 15 * 1024 && (g_room <= TAIL_CAP || rest_long != nullptr)
                         && n_rest <= 0x7fffff00;
    HIP_OK(hipStreamWaitEvent(side->stream, by_tail ? side->ready : side->losses, 0));
    if (by_tail) {
        int32_t* const longer = g_room <= TAIL_CAP ? nullptr : rest_long;
        rc = launch_tail_list(env, X, Z, tile_off, n_pts, reb, rebound_cap, n, freq, res, shape, H, n_dropped, rest_list, n_rest,
                              longer, side->stream);
        if (!rc && longer) {
            HIP_OK(hipStreamWaitEvent(side->stream, side->losses, 0));
            rc = gain_heatmap_impl(env, X, Z, tile_off, n_pts, reb, rebound_cap, n, freq, ray_mask, res, shape, H,
                                   G_rest, nullptr, nullptr, nullptr, loss_ws, side->stream, g_slot, g_room, longer, n_rest, true);
            if (!rc) rc = heatmap_accumulate_impl(X, Z, G_rest, tile_off, n_pts, n, nullptr, res, shape, H, n_dropped, nullptr, side->stream,
                                                  nullptr, 0, longer, n_rest, g_slot, g_room);
        }
    } else {
        rc = gain_heatmap_impl(env, X, Z, tile_off, n_pts, reb, rebound_cap, n, freq, ray_mask, res, shape, H,
                               G_rest, nullptr, nullptr, nullptr, loss_ws, side->stream, g_slot, g_room, rest_list, n_rest, true);
        if (!rc) rc = heatmap_accumulate_impl(X, Z, G_rest, tile_off, n_pts, n, nullptr, res, shape, H, n_dropped, nullptr, side->stream,
                                              nullptr, 0, rest_list, n_rest, g_slot, g_room);
    }
    if (!rc) rc = gain_heatmap_impl(env, X, Z, tile_off, n_pts, reb, rebound_cap, n, freq, ray_mask, res, shape, H,
                                    nullptr, nullptr, nullptr, n_violations, loss_ws, stream, nullptr, 0, nullptr, 0, true);
    // (whatever happened above, the caller's stream must not run ahead of work already queued on the side stream)
    (void)hipEventRecord(side->done, side->stream);
    (void)hipStreamWaitEvent(main, side->done, 0);
    return rc;
}

// ray_mask of rt2d_heatmap_accumulate with its hint bits, from a cast's per-ray results: 1 = bin this ray, | 2 = the caller
// vouches that its x never decreases and its bounding box lies inside the map, | 4 = known not to qualify
__global__ void __launch_bounds__(256) rt2d_heatmap_hints_kernel(const uint8_t* monotone, const double* range, const uint8_t* ray_mask,
                                                                 int64_t n, double res_x, double res_z, double nx, double nz, uint8_t* out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const bool base = !ray_mask || ray_mask[r] != 0;
    const bool good = base && monotone[r] != 0 && range[r] > -1 && range[3 * n + r] < 1
                   && floor(trunc(range[2 * n + r]) / res_x) < nx && floor(-trunc(range[n + r]) / res_z) < nz;
    out[r] = base ? (good ? 3 : 5) : 0;
}

int32_t rt2d_heatmap_hints(const uint8_t* monotone, const double* range, const uint8_t* ray_mask, int64_t n,
                           const int32_t* res, const int32_t* shape, uint8_t* out, void* stream) {
    if (!monotone || !range || !res || !shape || !out) return fail("rt2d_heatmap_hints: null argument");
    if (n < 0) return fail("rt2d_heatmap_hints: negative ray count");
    if (res[0] <= 0 || res[1] <= 0 || shape[0] <= 0 || shape[1] <= 0) return fail("rt2d_heatmap_hints: bad resolution or shape");
    if (n == 0) return 0;
    hipLaunchKernelGGL(rt2d_heatmap_hints_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, monotone, range,
                       ray_mask, n, (double)res[0], (double)res[1], (double)shape[0], (double)shape[1], out);
    HIP_OK(hipGetLastError());
    return 0;
}

int32_t rt2d_classify_rays(const int32_t* status, const uint8_t* monotone, const double* range, const double* slope_max,
                           const uint8_t* ray_mask, int64_t n, const int32_t* res, const int32_t* shape, int64_t rows,
                           uint8_t* m, int32_t* g_slot, int32_t* rest_list, double* overflow, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!status || !monotone || !range || !slope_max || !res || !shape || !m || !g_slot || !rest_list || !overflow)
        return fail("rt2d_classify_rays: null argument");
    if (n < 0 || rows < 0) return fail("rt2d_classify_rays: negative count");
    if (res[0] <= 0 || res[1] <= 0 || shape[0] <= 0 || shape[1] <= 0) return fail("rt2d_classify_rays: bad resolution or shape");
    if (n > 0x7fffffff - 2) return fail("rt2d_classify_rays: too many rays for one list");
    HIP_OK(hipMemsetAsync(rest_list, 0, 2 * sizeof(int32_t), stream));
    if (n == 0) return 0;
    hipLaunchKernelGGL(rt2d_classify_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, status, monotone, range,
                       slope_max, ray_mask, n, (double)res[0], (double)res[1], (double)shape[0], (double)shape[1], rows,
                       m, g_slot, rest_list, overflow);
    HIP_OK(hipGetLastError());
    return 0;
}

int32_t rt2d_heatmap_finalize_multi(const double* H, const int32_t* shape, int32_t n_maps, double cutoff, double reduction_power,
                                    double* out, double* scratch, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!H || !shape || !out || !scratch) return fail("rt2d_heatmap_finalize_multi: null argument");
    const int64_t nx = shape[0], nz = shape[1], size = nx * nz;
    if (nx <= 0 || nz <= 0) return fail("rt2d_heatmap_finalize_multi: bad shape");
    if (n_maps <= 0) return n_maps == 0 ? 0 : fail("rt2d_heatmap_finalize_multi: negative map count");
    HIP_OK(hipMemsetAsync(scratch, 0, (size_t)n_maps * 2 * sizeof(double), stream));
    const int block = 256;
    const int64_t blocks = (size + block - 1) / block;
    hipLaunchKernelGGL(rt2d_hm_max_kernel, dim3((unsigned)(blocks < 128 ? blocks : 128), (unsigned)n_maps), dim3(block), 0, stream,
                       H, size, (unsigned long long*)scratch);
    hipLaunchKernelGGL(rt2d_hm_finalize_kernel, dim3((unsigned)blocks, (unsigned)n_maps), dim3(block), 0, stream,
                       H, nx, nz, cutoff, reduction_power, (const unsigned long long*)scratch, out);
    HIP_OK(hipGetLastError());
    return 0;
}

int32_t rt2d_heatmap_finalize(const double* H, const int32_t* shape, double cutoff, double reduction_power,
                              double* out, double* scratch, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!H || !shape || !out || !scratch) return fail("rt2d_heatmap_finalize: null argument");
    const int64_t nx = shape[0], nz = shape[1], size = nx * nz;
    if (nx <= 0 || nz <= 0) return fail("rt2d_heatmap_finalize: bad shape");
    HIP_OK(hipMemsetAsync(scratch, 0, 2 * sizeof(double), stream));
    const int block = 256;
    const int64_t blocks = (size + block - 1) / block;
    hipLaunchKernelGGL(rt2d_hm_max_kernel, dim3((unsigned)(blocks < 128 ? blocks : 128)), dim3(block), 0, stream,     // few waves: each ends in an atomic on one word
                       H, size, (unsigned long long*)scratch);
    hipLaunchKernelGGL(rt2d_hm_finalize_kernel, dim3((unsigned)blocks), dim3(block), 0, stream,
                       H, nx, nz, cutoff, reduction_power, (const unsigned long long*)scratch, out);
    HIP_OK(hipGetLastError());
    return 0;
}

int32_t rt2d_small_fan_heatmap(rt2d_env_t env, const double* X, const double* Z, const int64_t* tile_off, const int32_t* n_pts,
                               const Rt2dRayOut* reb, int32_t rebound_cap, int64_t n, int64_t max_pts, double freq,
                               const uint8_t* ray_mask, const int32_t* res, const int32_t* shape, double* H, int64_t* n_dropped,
                               double cutoff, double reduction_power, double* out, double* scratch, double* overflow, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!env || !X || !Z || !tile_off || !n_pts || !reb || !res || !shape || !H) return fail("rt2d_small_fan_heatmap: null argument");
    if (!reb->status || !reb->n_rebounds || !reb->reb_step || !reb->reb_surface || !reb->reb_velocity || !reb->reb_angle || !reb->reb_pos_z)
        return fail("rt2d_small_fan_heatmap: status and the rebound log (n_rebounds, reb_step, reb_surface, reb_velocity, reb_angle, reb_pos_z) are required");
    if (n < 0 || rebound_cap < 0) return fail("rt2d_small_fan_heatmap: negative count");
    if (n > 0x7fffffff) return fail("rt2d_small_fan_heatmap: too many rays for one launch");
    if (max_pts > TAIL_CAP && !overflow)
        return fail("rt2d_small_fan_heatmap: rays of up to %lld stored points may exceed the %d points a workgroup holds: pass `overflow` "
                    "(and bin the fan with rt2d_gain + rt2d_heatmap_accumulate when it comes back set), or use those directly",
                    (long long)max_pts, TAIL_CAP);
    if (!(freq > 0)) return fail("rt2d_small_fan_heatmap: freq must be positive");
    if (res[0] <= 0 || res[1] <= 0) return fail("rt2d_small_fan_heatmap: resolution must be positive");
    if (shape[0] <= 0 || shape[1] <= 0 || (int64_t)shape[0] * shape[1] >= (1ll << 31)) return fail("rt2d_small_fan_heatmap: bad shape");
    if (out && !scratch) return fail("rt2d_small_fan_heatmap: the normalised map needs the two-double scratch");
    const size_t lds = (size_t)(rebound_cap > 0 ? rebound_cap : 1) * 12 + 8;
    if (lds > 15 * 1024) return fail("rt2d_small_fan_heatmap: rebound log of %d entries per ray does not fit beside the hash; "
                                     "use rt2d_gain + rt2d_heatmap_accumulate", rebound_cap);
    TailArgs t; memset(&t, 0, sizeof t);
    GainArgs& a = t.g;
    a.env = env->dev; a.blob = env->blob; a.X = X; a.Z = Z; a.tile_off = tile_off; a.n_pts = n_pts;
    a.n_rebounds = reb->n_rebounds; a.reb_step = reb->reb_step; a.reb_surface = reb->reb_surface;
    a.reb_velocity = reb->reb_velocity; a.reb_angle = reb->reb_angle; a.reb_pos_z = reb->reb_pos_z;
    a.rebound_cap = rebound_cap; a.n = n; a.n_freqs = 1;
    t.status = reb->status; t.ray_mask = ray_mask;
    t.geo.res_x = res[0]; t.geo.res_z = res[1]; t.geo.rx = res[0]; t.geo.rz = res[1];
    t.geo.inv_rx = 1.0 / res[0]; t.geo.inv_rz = 1.0 / res[1];
    t.geo.lim = (res[0] < (1 << 20) && res[1] < (1 << 20)) ? 2147483648.0 : 0.0;
    t.nx = shape[0]; t.nz = shape[1]; t.H = H; t.n_dropped = n_dropped; t.freq = freq;
    t.scratch = out ? (unsigned long long*)scratch : nullptr;
    t.overflow = overflow;
    t.dbg = g_wave_log; t.dbg_cap = g_wave_log_cap / 8;
    static const int lean_env = getenv("RT2D_GH_LEAN") ? atoi(getenv("RT2D_GH_LEAN")) : 1;
    bool ok = false;
    if (lean_env) { int32_t rc = alpha_table_for(env, freq, &t.ad, &ok); if (rc) return rc; }
    t.alpha_ok = ok ? 1 : 0;
    if (n > 0) {
        hipLaunchKernelGGL(rt2d_small_tail_kernel, dim3((unsigned)((n + 127) / 128 * 128)), dim3(TAIL_THREADS), lds, stream, t);
        HIP_OK(hipGetLastError());
    } else if (out) {
        HIP_OK(hipMemsetAsync(scratch, 0, 2 * sizeof(double), stream));
    }
    if (!out) return 0;
    const int64_t nx = shape[0], nz = shape[1], size = nx * nz;
    const int block = 256;
    const int64_t blocks = (size + block - 1) / block;
    hipLaunchKernelGGL(rt2d_hm_max_kernel, dim3((unsigned)(blocks < 128 ? blocks : 128)), dim3(block), 0, stream, H, size, (unsigned long long*)scratch);
    hipLaunchKernelGGL(rt2d_hm_finalize_kernel, dim3((unsigned)blocks), dim3(block), 0, stream,
                       H, nx, nz, cutoff, reduction_power, (const unsigned long long*)scratch, out);
    HIP_OK(hipGetLastError());
    return 0;
}

}  // extern "C"

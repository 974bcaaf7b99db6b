/*
 * rt2d.h -- C ABI of librt2d_hip.so: 2D acoustic ray propagation + heatmap binning on MI355X (gfx950).
 *
 * The reference (ahernot/ray-tracer) has no FFI: its boundary for this path is the Python class API
 * (Environment2D / Ray2D / RayPack2D / Simulation2D).  Each entry point below names the reference
 * routine (file:line under the reference tree) whose work it takes over; the Python mirror in
 * ray-tracer_amd/ binds them with ctypes (see INTEGRATION.md for the stub a maintainer would add).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - Every call returns 0 on success, <0 on error; rt2d_last_error() gives a thread-local message.
 *   - Pointers documented "device" are device pointers valid on the CURRENT HIP device (the one the
 *     env handle was created on).  Pointers documented "host" are read during the call only.
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).  Launches are
 *     asynchronous unless stated otherwise.
 *   - Nothing here allocates per call except rt2d_env_create and the documented scratch of
 *     rt2d_heatmap_accumulate (none); callers own every buffer.
 *
 * Trajectory layout ("tile layout")
 *   Rays are grouped in tiles of 64 consecutive rays (one wavefront).  Tile t stores its points
 *   step-major:  X[tile_off[t] + s*64 + lane]  is point s of ray 64*t + lane, for
 *   s < n_pts[ray]; slots with s >= n_pts[ray] are unspecified.  tile_off has n_tiles+1 entries
 *   (element counts, not bytes), tile_off[t+1]-tile_off[t] = 64 * max_lane(n_pts).  A wavefront's
 *   store of one step is one contiguous 512-byte line per coordinate.
 */
#ifndef RT2D_H
#define RT2D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT2D_TILE 64

/* stop codes (reference: ray.py:83,165,194,218,228,253,258) */
#define RT2D_STOP_NONE          0
#define RT2D_STOP_EXIT_XMIN     1
#define RT2D_STOP_EXIT_XMAX     2
#define RT2D_STOP_EXIT_ZMIN     3
#define RT2D_STOP_EXIT_ZMAX     4
#define RT2D_STOP_MAX_REBOUNDS  5
#define RT2D_STOP_BACKPROP      6
#define RT2D_STOP_MAX_ITER      7
#define RT2D_STOP_POWER_CUTOFF  8   /* opt-in extension, see Rt2dParams.power_cutoff */

/* Rt2dParams.launch_hint */
#define RT2D_LAUNCH_SMALL_LDS     1

/* per-ray status */
#define RT2D_RAY_OK               0
#define RT2D_RAY_ERR_INTERP      -1  /* slope table queried outside its range: the reference raises
                                        ValueError from env.dx_floor / env.dx_ceil (environment.py:50-51, ray.py:179,202) */
#define RT2D_RAY_ERR_REBOUND_LOG -2  /* more rebounds than log slots (only possible with n_rebounds_max = -1) */

/* surfaces in the rebound log (reference: ray.py:185,208) */
#define RT2D_SURFACE_SEDIMENT 0
#define RT2D_SURFACE_WATER    1

/* out-of-range policy of a table */
#define RT2D_OOB_EXTRAPOLATE 0
#define RT2D_OOB_CLAMP       1
#define RT2D_OOB_ERROR       2

/* Piecewise polynomial on host memory: piece j covers breaks[j] <= x < breaks[j+1];
 * order 3: ((c0*d + c1)*d + c2), order 2: (c0*d + c1), d = x - breaks[j], coef[j*order + k]. */
typedef struct {
    int32_t order;
    int32_t n_pieces;
    int32_t oob;
    int32_t reserved;
    const double* breaks;   /* host, n_pieces + 1 */
    const double* coef;     /* host, n_pieces * order */
} Rt2dTable;

/* Environment description.  Replaces the interpolant objects built by
 * PhysicsEnvironment2D.__generate (physics_environment.py:61-86) and Environment2D.__init__
 * (environment.py:43-53). c and dz_c must share their breaks; absorb[0..5] must share theirs. */
typedef struct {
    double range_min[2];
    double range_max[2];
    Rt2dTable c;          /* sound speed c(z), quadratic, extrapolating */
    Rt2dTable dz_c;       /* dc/dz, quadratic, extrapolating */
    Rt2dTable floor;      /* sea floor z(x), linear, clamped */
    Rt2dTable ceil;       /* surface z(x), linear, clamped */
    Rt2dTable dx_floor;   /* floor slope, quadratic, out of range = error */
    Rt2dTable dx_ceil;    /* surface slope, quadratic, out of range = error */
    Rt2dTable impedance;  /* Z(z) = rho*c, quadratic, extrapolating */
    Rt2dTable absorb[6];  /* f1, f2, A2, E1, E2, E3 of the absorption model, quadratic, extrapolating */
} Rt2dEnvDesc;

typedef struct rt2d_env_s* rt2d_env_t;

/* Propagation parameters (reference: ray.py:123-132 kwargs and preferences.py:6-9 defaults). */
typedef struct {
    double  dx_max;          /* default 10.0 */
    double  dz_max;          /* default 0.1 */
    int32_t n_steps_max;     /* default 100000 */
    int32_t n_rebounds_max;  /* default -1 = unbounded */
    int32_t backprop;        /* default 1 */
    int32_t has_target;      /* 0/1: when 1, dist_to_target / step_target / T_target are produced */
    double  target[2];
    int32_t rebound_cap;     /* slots per ray in the rebound log arrays; a ray with more rebounds gets
                                RT2D_RAY_ERR_REBOUND_LOG (n_rebounds_max+1 slots always suffice when bounded) */
    int32_t launch_hint;     /* 0: the library chooses the launch shape.  RT2D_LAUNCH_SMALL_LDS (1): this cast runs BESIDE a
                                kernel whose workgroups hold most of every CU's LDS (the side rays of rt2d_cast_heatmap):
                                only the sound-speed table is staged in LDS, floor and surface records come through the
                                caches, so that its workgroups find room on the same CUs.  Results do not depend on it */
    double  power_cutoff;    /* <= 0: off (reference behaviour).  > 0: stop the ray at the first point whose
                                transmittance 10^(G/10) at cutoff_freq falls below it (opt-in, SURVEY trap 2) */
    double  cutoff_freq;
} Rt2dParams;

/* Per-ray results of a cast; every pointer is a device array of n rays, any of them may be NULL.
 * Rebound-log arrays hold rebound_cap * n entries, entry k of ray r at [k*n + r]. */
typedef struct {
    int32_t* n_pts;        /* trajectory points incl. the source = Ray2D.steps (ray.py:264) */
    int32_t* stop_code;    /* RT2D_STOP_* = Ray2D.stop_reason */
    int32_t* n_rebounds;   /* Ray2D.n_rebounds */
    int32_t* status;       /* RT2D_RAY_* */
    double*  range;        /* 4*n: [0*n+r]=min x, [1*n+r]=min z, [2*n+r]=max x, [3*n+r]=max z (ray.py:267-268) */
    double*  L_end;        /* path length at the last point (ray.py:272) */
    double*  T_end;        /* travel time at the last point (ray.py:275) */
    double*  dist_to_target; /* |target.z - z(target.x)| by linear interpolation, NaN if out of range (ray.py:282-287) */
    int32_t* step_target;  /* first step minimising |x - target.x| (ray.py:278) */
    double*  T_target;     /* travel time at step_target (ray.py:279) */
    int32_t* reb_step;     /* rebound log: point index of the rebound, = 'step' (ray.py:185) */
    uint8_t* reb_surface;  /* RT2D_SURFACE_* */
    double*  reb_velocity; /* c at the point before the rebound */
    double*  reb_angle;    /* the reference's 'angle' = cos(grazing), reproduced as is (ray.py:184) */
    double*  reb_pos_x;    /* point before the rebound */
    double*  reb_pos_z;
    double*  margin;       /* smallest distance [m] by which any bound / floor / surface comparison of the loop was
                              decided; a ray with margin ~ 0 is "knife-edge": its step count depends on the last bits
                              of the arithmetic (no counterpart in the reference; used to report parity honestly) */
    uint8_t* monotone;     /* 1 when no appended point lies left of its predecessor (x never decreases): the hint bit 2
                              of rt2d_heatmap_accumulate's ray_mask may be set for such a ray */
    double*  slope_max;    /* largest |dz/dx| of any step of the ray: bounds the rows a ray can visit inside one heat-map
                              column (rt2d_cast_heatmap's precondition) */
} Rt2dRayOut;

const char* rt2d_last_error(void);
int32_t rt2d_version(void);

/* Copies the tables to the current device (hipMalloc + hipMemcpy, synchronous). */
int32_t rt2d_env_create(const Rt2dEnvDesc* desc, rt2d_env_t* env_out);
int32_t rt2d_env_destroy(rt2d_env_t env);
/* c(0), the surface sound speed every ray starts with (reference: ray.py:145). Host query. */
int32_t rt2d_env_surface_speed(rt2d_env_t env, double* c0_out);

/* Ray2D.propagate main loop for n rays, one lane per ray, no trajectory stored (reference:
 * ray.py:70-259 plus the scalar parts of the post-pass ray.py:263-287).
 * angles: device, n casting angles in radians.  src: host, 2 doubles.
 * init_slope / init_mult: optional device arrays (n) overriding tan(angle) and -(c0/cos(angle))^2
 * (ray.py:138-146) so a caller can inject host-computed values; NULL = computed on device. */
int32_t rt2d_cast_count(rt2d_env_t env, const double* angles, int64_t n, const double* src,
                        const Rt2dParams* params, const double* init_slope, const double* init_mult,
                        const Rt2dRayOut* out, void* stream);

/* Tile offsets for the trajectory layout from n_pts (device, n) -> tile_off (device, n_tiles+1).
 * total_out (host, may be NULL) receives tile_off[n_tiles]; when non-NULL the call synchronises the stream. */
int32_t rt2d_traj_layout(const int32_t* n_pts, int64_t n, int64_t* tile_off, int64_t* total_out, void* stream);

/* What the host needs to know about a finished cast, in one kernel and one 64-byte read (reference: the range fold
 * of RayPack2D.add, ray.py:466-472; the rest has no counterpart -- it validates the single-pass layouts).
 * status: device, n.  range: device, 4*n as in Rt2dRayOut (may be NULL).  ray_mask: device, n bytes, nonzero = count
 * this ray (may be NULL = all).  tile_off: device, n_tiles+1 offsets of the EXACT layout from rt2d_traj_layout (may be
 * NULL).  summary: device, 8 doubles: [0] tile_off[n_tiles], [1] elements of the longest tile, [2] 1 if any ray has
 * RT2D_RAY_ERR_REBOUND_LOG, [3] rays with status 0 (and mask), [4..7] = (-min x, -min z, max x, max z) over those
 * rays folded from the origin (all four >= 0), i.e. the plot range; one max-all-reduce of [4..7] merges ranks. */
int32_t rt2d_fan_summary(const int32_t* status, const double* range, const uint8_t* ray_mask, int64_t n,
                         const int64_t* tile_off, double* summary, void* stream);

/* rt2d_traj_layout followed by rt2d_fan_summary of every ray, as one call -- and, for fans of at most 65 536 rays, as ONE
 * launch that also writes ok_mask[r] = (status[r] == 0) when a mask is asked for (it must be NULL for bigger fans): a small
 * fan's cast is followed by a dozen launches of a few microseconds each, and these four are one.  No zeroing of `summary`
 * is needed: summary has n_summary >= 8 doubles, the 8 of rt2d_fan_summary and the caller's spare words behind them, which
 * are set to zero.  Same outputs as the two calls (ray.py:466-472 for the range fold). */
int32_t rt2d_layout_summary(const int32_t* n_pts, const int32_t* status, const double* range, int64_t n,
                            int64_t* tile_off, double* summary, int32_t n_summary, uint8_t* ok_mask, void* stream);

/* Same propagation, storing every point (reference: the np.insert at ray.py:192,216,236) in tile layout.
 * C (device, may be NULL) additionally receives the sound speed at every point (ray.py:273).
 * A tile never writes outside [tile_off[t], tile_off[t+1]): points beyond the tile's room are counted in n_pts but
 * not stored.  This allows a single speculative pass over a provisional layout (e.g. the same generous room for
 * every tile) followed by rt2d_traj_layout + rt2d_traj_compact when max(n_pts) fits, instead of count + store.
 * params->n_steps_max is limited to 8 388 606 here (a lane addresses its tile with a 32-bit byte offset). */
int32_t rt2d_cast_write(rt2d_env_t env, const double* angles, int64_t n, const double* src,
                        const Rt2dParams* params, const double* init_slope, const double* init_mult,
                        const Rt2dRayOut* out, const int64_t* tile_off, double* X, double* Z, double* C,
                        void* stream);

/* Fused cast -> gain at one frequency -> heat-map accumulation: the rays are propagated, their gain (rt2d_gain's
 * expressions, same bits) is carried along and every point is binned into H as it is produced; no trajectory is
 * written to HBM (16 + 8 B per ray-step saved twice over).  Takes over, for the rays it is given, the sequence
 * Ray2D.propagate (ray.py:70-289) -> populate_NEW (ray.py:339-369) -> heat-map assignment and sum
 * (simulation.py:120-131, functions.py:12-19); the reference itself has no such fused form.
 * Precondition per ray (the caller selects with ray_mask, n bytes, 0 = skip the ray, NULL = all): no ray error, x never
 * decreases (Rt2dRayOut.monotone), every point inside the map (Rt2dRayOut.range against shape), and at most 8 rows
 * visited inside one column: floor(res_x * slope_max / res_z) + 2 <= 8 -- all known from a prior rt2d_cast_count.  "Last
 * write wins" is then local to the column a ray is in.  The other rays go through rt2d_cast_write + rt2d_gain +
 * rt2d_heatmap_accumulate into the same H.
 * sign multiplies every deposited value (1.0 normally).  out: only n_pts, stop_code, n_rebounds, status, range,
 * monotone, slope_max (may be NULL).  irregular (device, n bytes, may be NULL) / n_violations (device int64, may be
 * NULL, caller zeroes it): a ray that breaks the precondition stops being binned, is flagged and counted.  A count
 * other than 0 means H is NOT the reference's map and the caller must bin again with the flagged rays left to the stored
 * path: ALWAYS read it.  The slope bound above is exact for plain steps only -- a rebound ends on the surface under the
 * hit abscissa (ray.py:177-178), which on a rough floor can lie far off the line of the step -- so a mask built as above
 * makes the count 0 on every fixture, not on every environment (the host side starts over: Simulation2D._violated).
 * loss / loss_cap: optional (NULL, 0): the rebound losses at `freq` of these same rays from rt2d_rebound_loss over the
 * rebound log of the prior cast, entry k of ray r at [k*n + r], k < loss_cap.  The propagation is deterministic, so the
 * k-th rebound met here is the k-th logged one; the kernel then carries no reflection-coefficient code (sincos, hypot,
 * log10) and needs a third fewer registers.  Same map.
 * Fails (<0) when the sound-speed, floor, surface and absorption tables do not fit in LDS together. */
int32_t rt2d_cast_heatmap(rt2d_env_t env, const double* angles, int64_t n, const double* src,
                          const Rt2dParams* params, const double* init_slope, const double* init_mult,
                          const uint8_t* ray_mask, double freq, const int32_t* res, const int32_t* shape,
                          double sign, double* H, const Rt2dRayOut* out, uint8_t* irregular,
                          int64_t* n_violations, const double* loss, int32_t loss_cap, void* stream);

/* Rebound losses 10*log10(reflection coefficient) of every logged rebound at F frequencies (reference: ray.py:350-362,
 * model_reflection.py:11-46), one thread per (ray, rebound, frequency).  loss: device, F * rebound_cap * n doubles,
 * entry k of ray r at frequency f at [(f*rebound_cap + k)*n + r]; entries k >= n_rebounds[r] are left untouched. */
int32_t rt2d_rebound_loss(rt2d_env_t env, const Rt2dRayOut* rebounds, int32_t rebound_cap, int64_t n,
                          const double* freqs, int32_t n_freqs, double* loss, void* stream);

/* Copies trajectories from one tile layout to another (typically provisional -> exact); the first
 * tile_off_dst[t+1]-tile_off_dst[t] elements of every tile are copied.  C_src/C_dst may both be NULL.
 * No counterpart in the reference (it grows XZ point by point with np.insert). */
int32_t rt2d_traj_compact(const int64_t* tile_off_src, const int64_t* tile_off_dst, int64_t n,
                          const double* X_src, const double* Z_src, const double* C_src,
                          double* X_dst, double* Z_dst, double* C_dst, void* stream);

/* Post-pass arrays in tile layout (reference: ray.py:271-275): L and T are cumulative (value 0 at point 0);
 * dL[s] / dT[s] belong to the segment ending at point s (slot 0 is written as 0). Any output may be NULL. */
int32_t rt2d_post(rt2d_env_t env, const double* X, const double* Z, const int64_t* tile_off,
                  const int32_t* n_pts, int64_t n, double* dL, double* L, double* C, double* dT, double* T,
                  void* stream);

/* Ray2D.populate_NEW for F frequencies (reference: ray.py:339-369, model_absorption.py:19-46,
 * model_reflection.py:11-46): cumulative gain in dB at every point.
 * freqs: device, F values in Hz.  G: device, F * total elements, frequency f at G + f*total, tile layout
 * (may be NULL).  G_end: device, F*n, gain at the last point of each ray at [f*n + r] (may be NULL).
 * loss_ws: optional device workspace of F * rebound_cap * n doubles (contents irrelevant on entry).  With it the rebound
 * losses are computed first by their own kernel and the gain loop only reads them: same numbers bit for bit, and the
 * loop keeps 86 instead of 162 vector registers (five instead of three waves per SIMD).  NULL = evaluated in the loop. */
int32_t rt2d_gain(rt2d_env_t env, const double* X, const double* Z, const int64_t* tile_off,
                  const int32_t* n_pts, const Rt2dRayOut* rebounds, int32_t rebound_cap, int64_t n,
                  const double* freqs, int32_t n_freqs, int64_t total, double* G, double* G_end,
                  double* loss_ws, void* stream);

/* Gain at ONE frequency and heat-map accumulation in one forward pass over stored trajectories (reference:
 * ray.py:339-369 followed by simulation.py:120-131, functions.py:12-19): the gain of a point is binned the moment it is
 * computed and never stored -- rt2d_gain + rt2d_heatmap_accumulate write 8 B per point and read 24 B back.
 * ray_mask (device, n bytes, NULL = all 1): 1 = bin the ray here; it must meet the preconditions of rt2d_cast_heatmap
 * (x never decreases, every point inside the map, at most 8 rows per column).  2 = only compute the ray's gain and
 * store it to G (device, tile layout, `total` elements as for rt2d_gain; may be NULL when no ray has mask 2): the caller
 * then bins these rays with rt2d_heatmap_accumulate.  0 = skip.  G_end (device, n, may be NULL): gain at the last point.
 * irregular / n_violations: as for rt2d_cast_heatmap.
 * loss_ws: optional device workspace of rebound_cap * n doubles for the rebound losses (as for rt2d_gain). */
int32_t rt2d_gain_heatmap(rt2d_env_t env, const double* X, const double* Z, const int64_t* tile_off,
                          const int32_t* n_pts, const Rt2dRayOut* rebounds, int32_t rebound_cap, int64_t n,
                          double freq, const uint8_t* ray_mask, const int32_t* res, const int32_t* shape,
                          double* H, double* G, double* G_end, uint8_t* irregular, int64_t* n_violations,
                          double* loss_ws, void* stream);

/* rt2d_gain_heatmap for a whole fan in one call: the rays of ray_mask value 1 are binned by the one-pass kernel, those of
 * value 2 (turned back, left the map, too steep: RayFan.fused_mask says which) only get their gain computed by it, into
 * a COMPACT array -- ray r at G_rest[g_slot[r] * g_room + step], g_room >= its points -- and are then binned by the
 * general one-ray-per-workgroup kernel ("last write wins" by an LDS hash, simulation.py:120-131), which reads that array.
 * rest_list: int32 device array, [0] = n_rest, [1] unused, [2 + k] = the k-th such ray; n_rest: that count on the host, or an
 * upper bound of it when the list was made on the device (rt2d_classify_rays); a listed ray with g_slot < 0 is skipped.
 * No gain array of the layout's size exists at any time.  Same map as rt2d_gain + rt2d_heatmap_accumulate.
 * When a listed ray is known to fit a workgroup (g_room <= 2048 stored points, rebound_cap <= 1279) the listed rays take a
 * shorter way: gain AND binning of a ray by one workgroup in one launch (the kernel of rt2d_small_fan_heatmap in list mode),
 * G_rest is not touched.  rest_long (device, n_rest + 2 int32, contents irrelevant, may be NULL): lets them take it when g_room
 * is larger -- the listed rays that turn out longer than a workgroup holds are collected there (same form as rest_list) and
 * only those go through the compact array and the general kernel. */
int32_t rt2d_gain_heatmap_rest(rt2d_env_t env, const double* X, const double* Z, const int64_t* tile_off, const int32_t* n_pts,
                               const Rt2dRayOut* reb, int32_t rebound_cap, int64_t n, double freq, const uint8_t* ray_mask,
                               const int32_t* res, const int32_t* shape, double* H,
                               const int32_t* rest_list, int64_t n_rest, const int32_t* g_slot, int64_t g_room, double* G_rest,
                               int64_t* n_dropped, int64_t* n_violations, double* loss_ws, int32_t* rest_long, void* stream);

/* Diagnostic of the binning-only pass of rt2d_gain_heatmap / rt2d_gain_heatmap_rest (reference: the absorption of
 * ray.py:346 = model_absorption.py:19-46 at ONE frequency; no counterpart of the table itself).  When a launch stores no
 * gain, -alpha(freq, z) / 1000 is read from one polynomial of degree 7 per depth piece of the absorption table, fitted on
 * the host and CHECKED ON THE DEVICE against the exact form at 256 depths per piece; the table is only used when the worst
 * relative difference is <= 1e-14 (else the exact form runs).  This call builds the table of `freq` if it is not there yet
 * (one small device allocation, kept until rt2d_env_destroy) and reports that worst relative difference and whether the
 * table is in use (1) or not (0).  RT2D_GH_LEAN=0 in the environment switches the table and its kernel off altogether. */
int32_t rt2d_alpha_table_error(rt2d_env_t env, double freq, double* max_rel_err, int32_t* in_use);

/* The ray_mask of rt2d_heatmap_accumulate with its hint bits, made on the device from a cast's per-ray results (no reference
 * counterpart; the reference bins every ray the same way, simulation.py:120-131): out[r] = 0 when ray_mask (may be NULL) leaves
 * the ray out, else 1 | 2 when its x never decreases (monotone) and its bounding box (range, 4 x n as in Rt2dRayOut) lies
 * inside the map of `shape` cells of `res` metres -- the tile kernel need not check such a ray point by point --, else 1 | 4. */
int32_t rt2d_heatmap_hints(const uint8_t* monotone, const double* range, const uint8_t* ray_mask, int64_t n,
                           const int32_t* res, const int32_t* shape, uint8_t* out, void* stream);

/* The ray_mask, rest_list and g_slot arguments of rt2d_gain_heatmap_rest, made on the device from a cast's per-ray results
 * (no reference counterpart: RayFan.fused_mask, which it restates, needs the host to wait for the cast): m[r] = 1 / 2 / 0 as
 * described there for the rays with status 0 (and ray_mask != 0 when one is given); the rays of value 2 are appended to
 * rest_list ([0] = count, [2 + k] = ray; n + 2 ints) and get row g_slot[r] = k of a compact gain array of `rows` rows.  A
 * ray beyond the last row gets g_slot = -1, is left out by both kernels, and *overflow (a device double, e.g. a spare
 * word behind the fan's summary) is set to 1: the caller reads it with the summary and bins the fan another way. */
int32_t rt2d_classify_rays(const int32_t* status, const uint8_t* monotone, const double* range, const double* slope_max,
                           const uint8_t* ray_mask, int64_t n, const int32_t* res, const int32_t* shape, int64_t rows,
                           uint8_t* m, int32_t* g_slot, int32_t* rest_list, double* overflow, void* stream);

/* Distance to a target from stored trajectories (reference: ray.py:282-287). dist: device, n. */
int32_t rt2d_dist(const double* X, const double* Z, const int64_t* tile_off, const int32_t* n_pts, int64_t n,
                  const double* target, double* dist, void* stream);

/* Eigenray products (reference: Ray2D.gen_filter ray.py:382-405 with closest_step_to_point ray.py:371-373), for
 * n_sel rays at once: the stored point closest to `target` (first of equal distances, as np.argmin), the travel time
 * T[step] + d / c(z_avg) and, per frequency, the transmittance 10^((G[step] + G_add) / 10) with
 * G_add = -alpha(f, z_avg) / 1000 * d over the straight residual d from that point to the target.
 * T: device, tile layout, from rt2d_post.  G: device array of n_freqs device pointers, each one frequency's gain in
 * tile layout (rt2d_gain).  ray_ids: device, n_sel ray indices (NULL = rays 0..n_sel-1).  target: host, 2 doubles.
 * freqs: device, n_freqs.  step_out (device int32, n_sel), T_target (device, n_sel), Tmult (device, n_freqs * n_sel,
 * [f * n_sel + i]). */
int32_t rt2d_target_products(rt2d_env_t env, const double* X, const double* Z, const double* T, const double* const* G,
                             const int64_t* tile_off, const int32_t* n_pts, const int32_t* ray_ids, int64_t n_sel,
                             const double* target, const double* freqs, int32_t n_freqs,
                             int32_t* step_out, double* T_target, double* Tmult, void* stream);

/* Heat-map accumulation (reference: simulation.py:120-131 + functions.py:12-19): for every ray, the
 * value 10^(G/10)^-0.1 of the LAST point falling in each cell (x//res_x, z//-res_z) is added to H.
 * The cell arithmetic is the reference's integer arithmetic: trunc toward zero, then floor division, then the
 * flat index ix*nz + iz (negative flat indices wrap once, as NumPy indexing does).
 * ray_mask: device, n bytes, nonzero = bin this ray (NULL = all n rays; a RayPack may hold a subset of a fan).
 *   Bit 2 (value 2) is a HINT the caller may add: "x never decreases along this ray (Rt2dRayOut.monotone) and every
 *   point of it falls inside the map"; the tile kernel then skips its own check of exactly that.  A wrong hint gives
 *   a wrong map.  Bit 4 (value 4) says the opposite, "known not to qualify": where the tile kernel shares a tile among
 *   several waves (fans of 10^4..10^5 rays) such a ray goes to the general kernel without being checked.
 * G: device, one frequency, tile layout.  res: host, 2 ints (metres per cell in x and z).
 * shape: host, 2 ints (nx, nz).  H: device, nx*nz doubles row-major [ix*nz + iz], accumulated into
 * (caller zeroes it).  n_dropped (device int64, may be NULL, caller zeroes it) counts points whose flat index
 * falls outside [-nx*nz, nx*nz) -- the reference raises IndexError there.
 * work: device workspace of n + 2 int32 (contents irrelevant on entry), or NULL.  With a workspace, nz <= 256 and
 * more than 6000 rays (env RT2D_HEATMAP_FAST_MIN_RAYS overrides), rays that only move forward in x and stay inside
 * the map are binned by a coalesced one-wave-per-tile kernel and only the remaining rays go through the general
 * one-wave-per-ray hash kernel; otherwise every ray does.  On return work[0] = number of rays the general kernel
 * handled after the tile kernel, or -1 if the tile kernel was not used. */
int32_t rt2d_heatmap_accumulate(const double* X, const double* Z, const double* G, const int64_t* tile_off,
                                const int32_t* n_pts, int64_t n, const uint8_t* ray_mask,
                                const int32_t* res, const int32_t* shape,
                                double* H, int64_t* n_dropped, int32_t* work, void* stream);

/* rt2d_heatmap_accumulate with one more piece of knowledge: max_pts, an upper bound of the stored points of any binned ray
 * (0 = unknown: then exactly rt2d_heatmap_accumulate).  Rays of tens of thousands of points -- the reference's default step
 * sizes, preferences.py:6-9 -- get a larger hash in the one-ray-per-workgroup kernel, which otherwise walks such a ray once
 * per partition of its candidate cells.  Same map. */
int32_t rt2d_heatmap_accumulate_hinted(const double* X, const double* Z, const double* G, const int64_t* tile_off,
                                       const int32_t* n_pts, int64_t n, const uint8_t* ray_mask,
                                       const int32_t* res, const int32_t* shape,
                                       double* H, int64_t* n_dropped, int32_t* work, int64_t max_pts, void* stream);

/* Heat maps of n_freqs frequencies in ONE pass over the trajectories (reference: the frequency loop of populate_NEW,
 * ray.py:345-362, followed by simulation.py:120-131 once per frequency): the cell of a point and which point of a ray owns
 * a cell do not depend on the frequency, so they are worked out once and only the deposited value differs per map.
 * Gs: device array of n_freqs device pointers, each one frequency's gain in tile layout (rt2d_gain).  H: device,
 * n_freqs maps of nx*nz doubles one after the other, accumulated into (caller zeroes them).  Everything else as for
 * rt2d_heatmap_accumulate (ray_mask with its hint bits, work, n_dropped: counted once, not per map).
 * H_minor: optional device scratch of 4 * ceil(n_freqs / 4) * nx*nz doubles (contents irrelevant on entry, garbage on
 * return), or NULL.  With it both binning kernels deposit into a map-minor copy [group of 8 or 4 maps][cell][8 or 4] -- one
 * L2 atomic request then carries a run's deposit for the maps of a group (the L2s are bound by requests, not by the doubles
 * in them; groups of 8 when 8 * ceil(n_freqs / 8) doubles per cell fit this size, i.e. n_freqs mod 8 is 0 or above 4) -- which
 * is added into H before the call's remaining launches; without it every (run, map) pair is a request of its own.  Same maps
 * up to the order of the sums. */
int32_t rt2d_heatmap_accumulate_multi(const double* X, const double* Z, const double* const* Gs, int32_t n_freqs,
                                      const int64_t* tile_off, const int32_t* n_pts, int64_t n, const uint8_t* ray_mask,
                                      const int32_t* res, const int32_t* shape,
                                      double* H, int64_t* n_dropped, int32_t* work, double* H_minor, void* stream);

/* rt2d_heatmap_finalize for n_maps maps stored one after the other (H and out), each normalised by its own maximum
 * (reference: simulation.py:133-138 per map).  scratch: device, at least 2 * n_maps doubles. */
int32_t rt2d_heatmap_finalize_multi(const double* H, const int32_t* shape, int32_t n_maps, double cutoff,
                                    double reduction_power, double* out, double* scratch, void* stream);

/* Heat map of a SMALL fan in one call: gain at `freq` (ray.py:339-362) and binning (simulation.py:120-131, functions.py:12-19)
 * of every ray by one kernel launch -- one workgroup per ray, nothing written but the map -- followed, when `out` is given,
 * by the normalisation of rt2d_heatmap_finalize (simulation.py:133-138).  Same map as rt2d_gain + rt2d_heatmap_accumulate
 * (+ rt2d_heatmap_finalize) of the same stored fan to 1e-12 (the running sum of a ray is a block scan here, not
 * np.cumsum's order, and no gain array exists).  Rays with reb->status != 0 (or ray_mask == 0 when one is given) are left
 * out.  H: nx * nz doubles, zeroed by the caller (the call ACCUMULATES).  A workgroup holds a ray of at most 2048 stored points.
 * max_pts: an upper bound of the stored points of any ray (the room of the layout); when it exceeds 2048 the caller must
 * pass `overflow`, a device double that the kernel sets to 1 if a ray really is longer (the map is then INCOMPLETE and the
 * fan must be binned with rt2d_gain + rt2d_heatmap_accumulate instead; the word is never cleared here), else the call is
 * rejected and nothing runs.  scratch: 2 doubles (with `out`).  For fans of a few thousand rays at most: beyond,
 * rt2d_gain_heatmap_rest and the tile kernels are the faster forms. */
int32_t rt2d_small_fan_heatmap(rt2d_env_t env, const double* X, const double* Z, const int64_t* tile_off, const int32_t* n_pts,
                               const Rt2dRayOut* reb, int32_t rebound_cap, int64_t n, int64_t max_pts, double freq,
                               const uint8_t* ray_mask, const int32_t* res, const int32_t* shape, double* H, int64_t* n_dropped,
                               double cutoff, double reduction_power, double* out, double* scratch, double* overflow, void* stream);

/* rt2d_heatmap_accumulate for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments, same kernels and same map
 * as rt2d_heatmap_accumulate called with a workspace, but the work list lives in a device array built into the library
 * (4 MiB per GPU, held until the process ends), so the call allocates nothing: no hipMalloc, no hipFree, no allocating
 * library routine.  Limit: n <= 1048576 rays (RT2D_NOALLOC_MAX_RAYS); more is rejected with an error, nothing is run.
 * Every call on a GPU shares that GPU's list (its header words are cleared at the start of each call): calls that
 * overlap on one GPU (different streams, different threads) race -- serialise them.  On return nothing is reported
 * about the number of rays the general kernel handled (rt2d_heatmap_accumulate's work[0]).
 * Set-up: the per-device address of the built-in array is taken by rt2d_env_create on the device the environment is
 * created on; on a device where no environment was ever created, the FIRST call takes it itself (hipGetSymbolAddress,
 * which may load the code object) and must therefore be made outside stream capture. */
int32_t rt2d_heatmap_accumulate_noalloc(const double* X, const double* Z, const double* G, const int64_t* tile_off,
                                        const int32_t* n_pts, int64_t n, const uint8_t* ray_mask,
                                        const int32_t* res, const int32_t* shape,
                                        double* H, int64_t* n_dropped, void* stream);

/* Heat-map presentation (reference: simulation.py:133-138): out[iz*nx + ix] = min(H/max(H), cutoff)^power.
 * scratch: device, at least 2 doubles. */
int32_t rt2d_heatmap_finalize(const double* H, const int32_t* shape, double cutoff, double reduction_power,
                              double* out, double* scratch, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RT2D_H */

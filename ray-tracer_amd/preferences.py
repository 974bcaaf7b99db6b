# Default parameters of the 2D acoustic ray path.
# Values mirror the reference's constants (reference: preferences.py:1-12) so that
# a caller that relied on the defaults sees identical behaviour.

# Environment parameters
RES_X_DEFAULT = 100
RES_Z_DEFAULT = 10

# Ray tracing parameters
DX_MAX_DEFAULT = 10.
DZ_MAX_DEFAULT = 0.1
N_STEPS_MAX = 100000
N_REBOUNDS_MAX = -1  # -1 = unbounded

WAVE_HEIGHT_RMS_DEFAULT = 2.
DZ_SEDIMENT_DEFAULT = 2.

# Build-specific knobs (no counterpart in the reference)
FLOOR_OVERSAMPLE_DEFAULT = 4      # floor/ceiling callables are tabulated at res_x / FLOOR_OVERSAMPLE
REBOUND_LOG_CAP_DEFAULT = 64      # rebound-log slots per ray when n_rebounds_max == -1

// TEST-ONLY host emulation harness.
// Compiles the *device* math headers (astroz_amd/csrc/*.h) for the host with stand-ins for the
// gfx950 intrinsics (AZ_HOST_EMUL), so that the algebra of the kernels (rotation-carried angles,
// eps-form Kepler iteration, rcp/rsqrt refinement, no-atan2 short-period step, init formulas)
// can be checked against the oracle in the CPU-only test tier.  It is a single-lane emulation:
// wave votes degenerate to the lane's own predicate.  Never built into or loaded by the product.
#define AZ_HOST_EMUL 1
#define AZ_DEVICE static inline
#define AZ_COLD_STRIDE 1
#include <cstddef>
#include <cstring>
#include "../../astroz_amd/csrc/init_device.h"
#include "../../astroz_amd/csrc/propagate_device.h"
#include "../../astroz_amd/csrc/fast_step.h"
#include "../../astroz_amd/csrc/fast_step_f32.h"

extern "C" {

int emul_num_fields() { return AZ_NUM_FIELDS; }

// raw[8] -> fields[AZ_NUM_FIELDS]; returns flags
unsigned emul_init(const double* raw, const double* grav6, double* fields)
{
    AzGrav g{grav6[0], grav6[1], grav6[2], grav6[3], grav6[4], grav6[5], 0.5 * grav6[1]};
    return az_init_satellite(raw, g, fields, 1, 0);
}

// propagate one satellite (fields as produced by emul_init) over ts[n]; incremental=1 carries the
// (sin,cos) pairs from step to step exactly as the lane=satellite kernel does.
void emul_propagate(const double* fields, unsigned flags, const double* grav6, const double* ts, int n,
                    int incremental, double* out6, int* rc_out)
{
    AzGrav g{grav6[0], grav6[1], grav6[2], grav6[3], grav6[4], grav6[5], 0.5 * grav6[1]};
    if (flags & AZ_FLAG_DEEP) {
        Sdp4Lane e; Sdp4Carry cy; double cold_store[D_NUM];
        const ColdLds cold{cold_store};
        az_load_sdp4(fields, 1, 0, flags, e, cold);
        cy.atime = 0.0; cy.xli = e(H_xlamo); cy.xni = e(H_no_unkozai);
        for (int i = 0; i < n; ++i) {
            if (!incremental) { cy.atime = 0.0; cy.xli = e(H_xlamo); cy.xni = e(H_no_unkozai); }
            double r[3], v[3];
            int rc = az_sdp4_step<true>(e, cold, g, az_rotk(), ts[i], cy, r, v);
            if (rc) { r[0]=r[1]=r[2]=v[0]=v[1]=v[2]=0.0; }
            memcpy(out6 + 6*i, r, 24); memcpy(out6 + 6*i + 3, v, 24);
            rc_out[i] = rc;
        }
    } else {
        Sgp4Lane e; Sgp4Carry st; ColdRegs cold;
        az_load_sgp4(fields, 1, 0, flags, e, cold);
        st.t_prev = 0; st.sW = st.cW = st.sO = st.cO = st.sA = st.cA = 0;
        for (int i = 0; i < n; ++i) {
            double r[3], v[3];
            az_sgp4_step<true>(e, cold, fields, 1, 0, g, az_rotk(), ts[i], (i == 0) || !incremental, st, r, v);
            memcpy(out6 + 6*i, r, 24); memcpy(out6 + 6*i + 3, v, 24);
            rc_out[i] = 0;
        }
    }
}

// the branch-free uniform-grid step (fast_step.h) as one lane of the lane = time kernel runs it: windows of 12
// steps of `dt` minutes (a 768-point segment of k_rows_fast), each seeded one increment before its first step.
// bad_out[i] = the step's validation predicate.
static void emul_fast_run(const double* fields, unsigned flags, const double* grav6, double ts0, double dt, int n, int ecc,
                          double* out6, int* bad_out)
{
    AzGrav g{grav6[0], grav6[1], grav6[2], grav6[3], grav6[4], grav6[5], 0.5 * grav6[1]};
    double inc[2 * AZ_INC_NUM];
    const double rate[2] = {fields[F_mdot], fields[F_argpdot]};
    for (int which = 0; which < 2; ++which)
        for (int a = 0; a < 2; ++a) az_sincos(rate[a] * dt, inc[AZ_INC_NUM * which + 2 * a], inc[AZ_INC_NUM * which + 2 * a + 1]);
    // window length as the host picks it: at most 12 steps (768 grid points) and at most ~3,000 minutes
    int wlen = (int)(3000.0 / (dt < 0 ? -dt : dt));
    wlen = wlen < 1 ? 1 : (wlen > 12 ? 12 : wlen);
    for (int w0 = 0; w0 < n; w0 += wlen) {
        const int w1 = (w0 + wlen < n) ? w0 + wlen : n;
        FastK k;
        az_load_fast(fields, 1, 0, flags, inc, 0, k);
        az_fast_window(fields, 1, 0, ts0 + w0 * dt, ts0 + (w1 - 1) * dt, dt, k);
        const bool win_bad = ecc ? !az_fast_window_ok<true>(k, g, ts0 + w0 * dt, ts0 + (w1 - 1) * dt)
                                 : !az_fast_window_ok<false>(k, g, ts0 + w0 * dt, ts0 + (w1 - 1) * dt);
        FastCarry st;
        az_seed_fast(fields, 1, 0, ts0 + (w0 - 1) * dt, k.tc_, st);
        for (int i = w0; i < w1; ++i) {
            double r[3], v[3];
            // the kernels never run the step in a window the bounds reject: report the whole window as bad
            bad_out[i] = ((ecc ? az_sgp4_fast_step<true, true>(k, g, RotCoefLit(), ts0 + i * dt, st, r, v)
                               : az_sgp4_fast_step<true, false>(k, g, RotCoefLit(), ts0 + i * dt, st, r, v)) || win_bad) ? 1 : 0;
            memcpy(out6 + 6*i, r, 24); memcpy(out6 + 6*i + 3, v, 24);
        }
    }
}
// k_rows_fast as the kernel runs one satellite row: time segments of `tile` grid points (a multiple of 64), one window
// set-up and validation per segment over ALL its grid points, lane l producing points t_lo + l + 64 j with increments of
// 64 grid steps.  out6: n_times rows; bad_out[i] = 1 where the point belongs to a rejected segment (window bounds) or,
// eccentric form, follows a rejected step of its segment (the kernel hands the rest of the segment to the generic path).
// delta (may be null): quasi-uniform grid, point i sits at t_first + i step + delta[i] (fast_step.h, DELTA); dmax >= max |delta|
static void emul_rows_fast_impl(const double* fields, unsigned flags, const double* grav6, double t_first, double step, int n_times, int tile,
                                int ecc, double* out6, int* bad_out, const float* delta, double dmax, const double* delta64 = nullptr)
{
    AzGrav g{grav6[0], grav6[1], grav6[2], grav6[3], grav6[4], grav6[5], 0.5 * grav6[1]};
    double inc[2 * AZ_INC_NUM];
    const double rate[2] = {fields[F_mdot], fields[F_argpdot]};
    for (int which = 0; which < 2; ++which)
        for (int a = 0; a < 2; ++a)
            az_sincos(rate[a] * (which == 0 ? 64.0 * step : step), inc[AZ_INC_NUM * which + 2 * a], inc[AZ_INC_NUM * which + 2 * a + 1]);
    for (int t_lo = 0; t_lo < n_times; t_lo += tile) {
        const int t_hi = t_lo + tile < n_times ? t_lo + tile : n_times;
        FastK k;
        az_load_fast(fields, 1, 0, flags, inc, 0, k);
        const double w_a = fma((double)t_lo, step, t_first), w_b = fma((double)(t_hi - 1), step, t_first);
        az_fast_window(fields, 1, 0, w_a, w_b, 64.0 * step, k);
        const bool ok = ecc ? az_fast_window_ok<true>(k, g, w_a, w_b, dmax) : az_fast_window_ok<false>(k, g, w_a, w_b, dmax);
        int first_bad_base = ok ? t_hi : t_lo; // the wave leaves the loop at the first iteration any live lane rejects
        for (int lane = 0; lane < 64 && ok; ++lane) {
            FastCarry st;
            az_seed_fast(fields, 1, 0, fma((double)(t_lo + lane) - 64.0, step, t_first), k.tc_, st);
            for (int base = t_lo; base < t_hi; base += 64) {
                const int i = base + lane;
                double r[3], v[3];
                double t = fma((double)i, step, t_first);
                bool bad;
                if (delta64) { // the WIDE form: fp64 deviations of seconds
                    const double dl = i < n_times ? delta64[i] : 0.0;
                    t += dl;
                    bad = ecc ? az_sgp4_fast_step<true, true, 2>(k, g, RotCoefLit(), t, st, r, v, dl)
                              : az_sgp4_fast_step<true, false, 2>(k, g, RotCoefLit(), t, st, r, v, dl);
                } else if (delta) {
                    const double dl = i < n_times ? (double)delta[i] : 0.0;
                    t += dl;
                    bad = ecc ? az_sgp4_fast_step<true, true, 1>(k, g, RotCoefLit(), t, st, r, v, dl)
                              : az_sgp4_fast_step<true, false, 1>(k, g, RotCoefLit(), t, st, r, v, dl);
                } else {
                    bad = ecc ? az_sgp4_fast_step<true, true>(k, g, RotCoefLit(), t, st, r, v)
                              : az_sgp4_fast_step<true, false>(k, g, RotCoefLit(), t, st, r, v);
                }
                if (i >= t_hi) break;
                if (bad && base < first_bad_base) first_bad_base = base;
                memcpy(out6 + 6 * (size_t)i, r, 24); memcpy(out6 + 6 * (size_t)i + 3, v, 24);
            }
        }
        for (int i = t_lo; i < t_hi; ++i) bad_out[i] = (t_lo + (i - t_lo) / 64 * 64 >= first_bad_base) ? 1 : 0;
    }
}

void emul_rows_fast(const double* fields, unsigned flags, const double* grav6, double t_first, double step, int n_times, int tile,
                    int ecc, double* out6, int* bad_out)
{
    emul_rows_fast_impl(fields, flags, grav6, t_first, step, n_times, tile, ecc, out6, bad_out, nullptr, 0.0);
}
void emul_rows_fast_delta(const double* fields, unsigned flags, const double* grav6, double t_first, double step, int n_times, int tile,
                          int ecc, const float* delta, double dmax, double* out6, int* bad_out)
{
    emul_rows_fast_impl(fields, flags, grav6, t_first, step, n_times, tile, ecc, out6, bad_out, delta, dmax);
}

void emul_rows_fast_wide(const double* fields, unsigned flags, const double* grav6, double t_first, double step, int n_times, int tile,
                         int ecc, const double* delta64, double dmax, double* out6, int* bad_out)
{
    emul_rows_fast_impl(fields, flags, grav6, t_first, step, n_times, tile, ecc, out6, bad_out, nullptr, dmax, delta64);
}

void emul_propagate_fast(const double* fields, unsigned flags, const double* grav6, double ts0, double dt, int n,
                         int ecc, double* out6, int* bad_out)
{
    emul_fast_run(fields, flags, grav6, ts0, dt, n, ecc, out6, bad_out);
}

// the packed fp32 step (fast_step_f32.h) as one lane of k_rows_fast32 runs it: the lane produces the grid points
// (2j, 2j+1) of a grid with step `step`, j = 0..n-1, one lane step (`lane_steps` grid steps; 128 in the kernel) apart;
// windows of at most 6 lane steps (a 768-point segment) and ~3,000 minutes.  out6: 2n rows in the order
// (even_0, odd_0, even_1, ...), bad_out: n.
static void emul_fast32_run(const double* fields, unsigned flags, const double* grav6, double ts0, double step, int lane_steps,
                            int n, double* out6, int* bad_out, bool precise, const float* delta2 = nullptr)
{
    AzGrav g{grav6[0], grav6[1], grav6[2], grav6[3], grav6[4], grav6[5], 0.5 * grav6[1]};
    const double dt = step * lane_steps;
    double inc[2 * AZ_INC_NUM];
    const double rate[2] = {fields[F_mdot], fields[F_argpdot]};
    // as k_prep_inc lays it out: [0] 64 grid steps (doubled below, as the kernel does), [1] one grid step
    for (int a = 0; a < 2; ++a) {
        az_sincos(rate[a] * dt * 0.5, inc[2 * a], inc[2 * a + 1]);
        az_sincos(rate[a] * step, inc[AZ_INC_NUM + 2 * a], inc[AZ_INC_NUM + 2 * a + 1]);
    }
    int wlen = (int)(3000.0 / (dt < 0 ? -dt : dt));
    wlen = wlen < 1 ? 1 : (wlen > 6 ? 6 : wlen);
    for (int w0 = 0; w0 < n; w0 += wlen) {
        const int w1 = (w0 + wlen < n) ? w0 + wlen : n;
        const double w_a = ts0 + w0 * dt, w_b = ts0 + (w1 - 1) * dt + step;
        FastK k0, k1;
        az_load_fast(fields, 1, 0, flags, inc, 0, k0);
        az_double_increments(k0);
        az_fast_window(fields, 1, 0, w_a, w_b, dt, k0);
        const bool win_bad = !az_fast_window_ok<false>(k0, g, w_a, w_b);
        az_load_fast(fields, 1, 0, flags, inc, 1, k1);
        az_fast_window(fields, 1, 0, w_a, w_b, step, k1);
        FastK32 k32;
        az_load_fast32(k0, k1, step, g, k32);
        FastCarry f0;
        az_seed_fast(fields, 1, 0, ts0 + (w0 - 1) * dt, k0.tc_, f0);
        FastCarry32 st;
        az_seed_fast32(f0, k1, st);
        for (int i = w0; i < w1; ++i) {
            az_f2 r[3], v[3];
            if (delta2) {
                // delta2[2 i], [2 i + 1]: deviations of the lane's two grid points from the ideal grid
                az_f2 dl; dl.x = delta2[2 * i]; dl.y = delta2[2 * i + 1];
                if (precise) az_sgp4_fast_step_f32p<true, true>(k32, g, ts0 + i * dt, st, r, v, dl);
                else az_sgp4_fast_step_f32<true, true>(k32, g, ts0 + i * dt, st, r, v, dl);
            } else if (precise) az_sgp4_fast_step_f32p<true>(k32, g, ts0 + i * dt, st, r, v);
            else az_sgp4_fast_step_f32<true>(k32, g, ts0 + i * dt, st, r, v);
            bad_out[i] = (win_bad || (precise && !az_fast32p_window_ok(k0, w_a, w_b))) ? 1 : 0;
            for (int j = 0; j < 3; ++j) {
                out6[12*i + j] = r[j].x; out6[12*i + 3 + j] = v[j].x;
                out6[12*i + 6 + j] = r[j].y; out6[12*i + 9 + j] = v[j].y;
            }
        }
    }
}

void emul_propagate_fast32(const double* fields, unsigned flags, const double* grav6, double ts0, double step, int lane_steps,
                           int n, double* out6, int* bad_out)
{
    emul_fast32_run(fields, flags, grav6, ts0, step, lane_steps, n, out6, bad_out, false);
}
// ... and the mixed-precision step (az_sgp4_fast_step_f32p)
void emul_propagate_fast32p(const double* fields, unsigned flags, const double* grav6, double ts0, double step, int lane_steps,
                            int n, double* out6, int* bad_out)
{
    emul_fast32_run(fields, flags, grav6, ts0, step, lane_steps, n, out6, bad_out, true);
}

void emul_propagate_fast32p_delta(const double* fields, unsigned flags, const double* grav6, double ts0, double step, int lane_steps,
                                  int n, const float* delta2, double* out6, int* bad_out)
{
    emul_fast32_run(fields, flags, grav6, ts0, step, lane_steps, n, out6, bad_out, true, delta2);
}

// the generic deep-space step as one lane of k_rows_deep runs it: per iteration the lane starts from a chunk seed (the
// integrator state at seed_t[i], what k_deep_seed prepares), brings it to its own time with the cached accelerations
// (az_resonance_cached) and runs az_sdp4_step on the prepared state
void emul_propagate_deep_cached(const double* fields, unsigned flags, const double* grav6, const double* ts, const double* seed_t,
                                int n, double* out6, int* rc_out, int* evals_out)
{
    AzGrav g{grav6[0], grav6[1], grav6[2], grav6[3], grav6[4], grav6[5], 0.5 * grav6[1]};
    Sdp4Lane e; double cold_store[D_NUM];
    const ColdLds cold{cold_store};
    az_load_sdp4(fields, 1, 0, flags, e, cold);
    Sdp4Acc acc{__builtin_nan(""), 0.0, 0.0, 0.0};
    int evals = 0;
    for (int i = 0; i < n; ++i) {
        Sdp4Carry cy{0.0, e(H_xlamo), e(H_no_unkozai)};
        if (e.irez != 0) az_resonance_advance(e, cold, seed_t[i], cy);
        double r[3], v[3];
        const double before = acc.atime;
        if (e.irez != 0) az_resonance_cached(e, cold, ts[i], cy, acc);
        if (!(before == acc.atime)) ++evals;
        int rc = az_sdp4_step_pre<true>(e, cold, g, az_rotk(), ts[i], cy, r, v, acc);
        if (rc) { r[0]=r[1]=r[2]=v[0]=v[1]=v[2]=0.0; }
        memcpy(out6 + 6*i, r, 24); memcpy(out6 + 6*i + 3, v, 24);
        rc_out[i] = rc;
    }
    *evals_out = evals;
}

void emul_sincos(double x, double* s, double* c) { az_sincos(x, *s, *c); }
double emul_rcp(double x) { return az_rcp(x); }
double emul_rsqrt(double x) { return az_rsqrt(x); }
void emul_rotate(double* s, double* c, double d) { az_rotate(*s, *c, d, az_rotk()); }
void emul_geodetic(double* p) { az_ecef_to_geodetic(p); }
double emul_atan2(double y, double x) { return az_atan2(y, x); }
}

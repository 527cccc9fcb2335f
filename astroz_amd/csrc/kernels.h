// kernels.h -- the gfx950 kernels.  Included only by astroz_hip.hip (compiled with hipcc).
//
// Work decomposition (the reverse of the reference's): the reference vectorises 8 satellites per
// AVX-512 register and calls a kernel 1,685 x 1,440 times through a function pointer
// (src/Constellation.zig L387-434, src/dispatch.zig L18-23).  Here the lane mapping follows the
// output layout, because the layout decides which mapping stores coalesced:
//   k_rows / k_rows_deep  one wave64 per satellite, lane = time: per-satellite constants are
//                         wave-uniform (SGPRs + LDS broadcast), control flow is uniform by
//                         construction, a wave's 64 results are contiguous in a satellite-major row;
//   k_propagate           one lane per satellite for a tile of time steps: constants in VGPRs / LDS
//                         columns, 64 satellites' results of one step are contiguous in a time-major row.
// Slowly drifting angles are carried as (sin,cos) pairs between a lane's steps in both mappings; 2-D
// grids (rows x time segments, satellite blocks x time tiles) supply the tens of thousands of waves
// needed to fill 256 CUs, with the workgroup -> XCD assignment chosen so that every XCD's L2 sees a
// contiguous part of the element table.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "init_device.h"
#include "propagate_device.h"
#include "fast_step.h"
#include "fast_step_f32.h"

#ifndef AZ_BLOCK
// 256 satellites per workgroup = four independent waves (no barriers anywhere).  The host orders
// each 256-satellite group by eccentricity class, so the few members that need extra
// Kepler-Newton trips share one wave instead of slowing all four, while the group's output rows
// stay one contiguous 6-KB span per step that a single CU fills within microseconds.
#define AZ_BLOCK 256
#endif
#ifndef AZ_PROP_FAST
#define AZ_PROP_FAST 1 /* k_propagate (near-earth): try the branch-free fast step first (fast_step.h) */
#endif
#ifndef AZ_MIN_WAVES
#define AZ_MIN_WAVES 1 /* __launch_bounds__ 2nd argument: waves per SIMD the register allocator must allow */
#endif
#ifndef AZ_ROWS_WAVES
/* k_rows: waves per SIMD the register allocator must allow.  3 (<= 168 VGPRs): the TEME kernel needs 135 and
 * runs without a single spill; 4 (<= 128 VGPRs, 8 spilled) measures the same or 2% slower. */
#define AZ_ROWS_WAVES 3
#endif
#ifndef AZ_DEEP_WAVES
#define AZ_DEEP_WAVES 2
#endif
#define AZ_RESEED 256 /* re-seed carried (sin,cos) pairs with a full sincos every 256 steps */
#ifndef AZ_SM_CHUNK
#define AZ_SM_CHUNK 4 /* time steps staged in LDS per flush in the satellite-major store path; 0 = direct */
#endif

struct PropArgs {
    const double *el;
    const unsigned *flags;
    size_t n_pad;
    const unsigned *list; // table indices of the satellites this launch handles
    unsigned n_list;
    const double *times; // minutes from the reference instant
    unsigned n_times;
    const double *offsets; // per-satellite minutes (table index), may be null
    double *pos, *vel;
    const double *sin_g, *cos_g; // GMST per time (ECEF / geodetic)
    const unsigned char *mask;   // per satellite, may be null
    unsigned char *err;          // n_sats x n_times, may be null (pre-zeroed)
    size_t stride_sats;          // time-major row length
    unsigned tile;               // time steps per workgroup
    unsigned tile_e, screen_nseg; // fused screen on the fast kernels: segment length of the eccentric form (tile: the
                                  // near-circular one's), partial-minimum rows the fast kernels own
    unsigned tile_forced;        // user override (0 = automatic)
    const double *seeds;         // deep space: resonance state at each tile start, [tile][3][n_list]; may be null
    int mode;
    int f32; // outputs are float arrays (pos/vel point to float): fp64 arithmetic, results rounded once at the store
    int tm_rows; // k_rows (redo pass behind the tile kernel): write this lane's 24 bytes of the time-major layout
    int rows_compact; // k_rows_deep: the output row of a satellite is its LIST SLOT (a compact satellite-major scratch array
                      // that k_deep_transpose turns into time-major runs), not its catalog index
    unsigned ecc_row0; // k_rows_fast with rows_compact (ahead of k_cols_fast): scratch row of the launch's first list slot
    // fused single-target conjunction screen (sink instead of stores): the target's TEME track,
    // [n_times][3], NaN where the target itself failed; partial minima per (segment or tile, list slot)
    const double *screen_target;
    double *part_d2;
    unsigned *part_t;
    // uniform grids: times[i] = times[0] + i * uniform_step (0 = not uniform), and the per-satellite
    // rotation increments k_prep_inc prepared for it (fast_step.h); null = generic path only
    double uniform_step;
    int grid_exact_uniform; // the staged grid is exactly uniform (whether or not the fast path is switched on): k_rows caches a lane's increments
    // quasi-uniform grid (what jd + fr arithmetic produces): times[i] = times[0] + i uniform_step + delta[i], |delta| <= delta_max
    // (fast_step.h, DELTA); null = exactly uniform
    const float *delta;
    double delta_max;
    // ... or the WIDE form (jitter of seconds, |delta| <= AZ_DELTA_WIDE_MAX): deviations as fp64; the ideal grid is a fit, its
    // origin grid_t0 (= times[0] on exact and tight grids) replaces times[0] in the fast kernels and the plan
    const double *delta64; // (also set on tight grids: k_tiles_fast stages fp64 deviations in both forms)
    int delta_wide;        // the wide form applies
    double grid_t0;
    const double *inc;
    const double *fast_rec; // [n_pad][FR_NUM]: per-satellite record of the lane = time fast kernels (k_prep_rec, fast_step.h)
    // row window: only satellites with row_lo <= table index < row_hi are produced by this launch (chunked
    // launches whose results feed a collective while the next chunk is still being computed)
    unsigned row_lo, row_hi;
    // ... and, for the lane = time kernels whose list is in catalog order, the LIST SLOTS that window covers: the launch's grid
    // is cut to them (a window launch over the full grid starts 80,000 workgroups of which three in four return at once:
    // 0.13 ms per window in round 4's chunked pipeline).  slot_hi = 0: the whole list.  win_*: the slot ranges of the two
    // sub-lists [class 0 | other classes] of a fast launch (host: launch_all -> launch_rows2)
    unsigned slot_lo, slot_hi;
    unsigned win_circ_lo, win_circ_hi, win_ecc_lo, win_ecc_hi;
    // k_rows_fast -> k_rows hand-over: (list slot, first grid point, end) of every segment remainder the fast
    // step rejected
    int arith32;         // fp32 outputs: 0 mixed-precision step, 1 packed fp32 step (both k_rows_fast32), 2 fp64 rounded at the store
    unsigned n_circ;     // row kernels on a uniform grid: list = [n_circ members of eccentricity class 0 | the rest]
    unsigned redo_slot0; // added to a k_rows_fast launch's list slots when it files redo items (sub-list launches)
    unsigned *redo_count;
    unsigned *redo_next; // the other launch parity's counter (re-armed by the redo pass for the launch after this one)
    unsigned *redo_items;
    // the window plan of the staged grid (k_plan_windows): per (time segment, list slot) the window constants of the fast
    // step and whether its validation bounds hold; rejected windows are items [0, *redo_static) of the redo list
    // k_tiles_fast: tiles are 16 consecutive CATALOG rows; rowmap[s] = kind << 30 | slot (AZ_ROW_*), and the rows it does not
    // compute itself arrive in a compact satellite-major scratch array (row = slot), see k_rows_deep / rows_compact
    const unsigned *rowmap;
    const double *tmp_pos, *tmp_vel;
    unsigned n_rows;
    const double *plan_win;         // [n_seg][plan_stride][AZ_PLAN_NUM]
    const unsigned char *plan_flag; // [n_seg][plan_stride]: AZ_PLAN_OK | AZ_PLAN_TC
    unsigned plan_stride;
    const unsigned *redo_static;
    AzGrav g;
};
enum { AZ_ROW_NEAR = 0, AZ_ROW_COPY = 1, AZ_ROW_ZERO = 2 }; // rowmap kinds: slot = near-earth list slot | scratch row | -
enum { AZ_PLAN_sOc, AZ_PLAN_cOc, AZ_PLAN_sdU, AZ_PLAN_cdU, AZ_PLAN_s1U, AZ_PLAN_c1U, AZ_PLAN_NUM };
#define AZ_PLAN_OK 1u /* the fast step's validation bounds hold over this window (az_fast_window_ok) */
#define AZ_PLAN_TC 2u /* the drag phase is expanded about the window centre (tc = tmid), else tc = 0 */

// one result vector -> memory, fp64 or fp32
template <class T>
__device__ __forceinline__ void az_put3(T *o, const double r[3])
{
    o[0] = (T)r[0];
    o[1] = (T)r[1];
    o[2] = (T)r[2];
}
// ... as streaming ("nt") stores: one 16-byte + one 8-byte piece (fp64) / one 12-byte piece (fp32).
// Measured on k_rows at sustained clocks: 0.256 vs 0.288 ms -- the write-once output no longer
// competes for L2 / Infinity-Cache space with itself, and the waves spend less time blocked at store
// issue.  (The same hint slows a pure write stream and the lane = satellite kernel's staged stores.)
typedef double az_d2s __attribute__((ext_vector_type(2), aligned(8)));
typedef float az_f3s __attribute__((ext_vector_type(3), aligned(4)));
#ifndef AZ_ROWS_NT
#define AZ_ROWS_NT 1
#endif
__device__ __forceinline__ void az_put3_stream(double *o, const double r[3])
{
#if AZ_ROWS_NT
    const az_d2s a = {r[0], r[1]};
    __builtin_nontemporal_store(a, reinterpret_cast<az_d2s *>(o));
    __builtin_nontemporal_store(r[2], o + 2);
#else
    az_put3(o, r);
#endif
}
__device__ __forceinline__ void az_put3_stream(float *o, const double r[3])
{
#if AZ_ROWS_NT
    const az_f3s a = {(float)r[0], (float)r[1], (float)r[2]};
    __builtin_nontemporal_store(a, reinterpret_cast<az_f3s *>(o));
#else
    az_put3(o, r);
#endif
}

__device__ __forceinline__ void az_epilogue(double r[3], double v[3], int mode, bool vel, const double *sin_g,
                                            const double *cos_g, unsigned i)
{
    if (mode != 0) {
        const double sg = sin_g[i], cg = cos_g[i];
        az_to_ecef(r, sg, cg);
        if (vel) az_to_ecef(v, sg, cg);
        if (mode == 2) az_ecef_to_geodetic(r);
    }
}

// LDS staging.  All staging is wave-private (each wave owns a slice of the array and only ever
// exchanges data between its own lanes), so no s_barrier and -- more importantly -- no
// `s_waitcnt vmcnt(0)` is needed: DS operations of one wave execute in issue order, and the
// outstanding global stores of earlier steps keep draining underneath the arithmetic.
//  time-major (t, s, 3): the wave's 64 x 24 B of one step are contiguous in memory when its
//      satellites are consecutive; they are transposed through LDS so that every lane stores
//      16 aligned bytes (global_store_dwordx4: 1,024 + 512 contiguous bytes per wave) instead of
//      three 8-byte pieces at a 24-byte stride.  Full cache lines leave the CU -> no partial-line
//      read-modify-write at the memory side.
//  sat-major (s, t, 3): rows are n_times*24 B apart; AZ_SM_CHUNK steps are staged and flushed as
//      8-byte words that are contiguous along each satellite's row.
#define AZ_TM_ROW 256 /* doubles per wave and per array in the time-major staging slice: a 1,536-byte row + 512 */
#define AZ_SM_ROW (AZ_SM_CHUNK * 3 + 1) /* +1 double: 26-bank row stride, conflict-free ds_write_b64 */

typedef double az_d2 __attribute__((ext_vector_type(2)));

// output stores.  Plain (L2 write-back) stores are the default: measured on config 2, `nt` stores
// leave the full kernel unchanged (0.431 vs 0.433 ms) but halve the rate of a pure write stream
// (0.56 vs 0.24 ms) because partial-line pieces no longer combine in L2.
#if defined(AZ_STORE_POLICY) && AZ_STORE_POLICY == 1
#define AZ_ST1(ptr, val) __builtin_nontemporal_store((val), (ptr))
#define AZ_ST2(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define AZ_ST1(ptr, val) (*(ptr) = (val))
#define AZ_ST2(ptr, val) (*(ptr) = (val))
#endif

__device__ __forceinline__ double az_readlane_f64(double x, unsigned lane_uniform)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), (int)lane_uniform);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), (int)lane_uniform);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void az_wave_lds_fence()
{
    // compiler-level ordering of LDS accesses within the wave (the hardware already executes one
    // wave's DS instructions in order); emits no instruction
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// time-major staged stores, one call per step and array (see k_propagate): L = the 2,048-byte staging slice, row = this
// step's row in global memory, pitch = doubles between consecutive rows, par = step parity within the tile
__device__ __forceinline__ void az_tm_flush(const double *L, double *row, size_t pitch, unsigned lane, unsigned par, bool last)
{
    const az_d2 *lp = reinterpret_cast<const az_d2 *>(L); // 128 sixteen-byte pieces
    az_d2 *g = reinterpret_cast<az_d2 *>(row);
    if (par == 0u) {
        AZ_ST2(g + lane, lp[32u + lane]);                             // row pieces 0..63 (staged from byte 512)
        if (last && lane < 32u) AZ_ST2(g + 64u + lane, lp[96u + lane]); // no odd step follows: the tail now
    } else {
        az_d2 *gprev = reinterpret_cast<az_d2 *>(row - pitch);
        AZ_ST2(lane < 32u ? gprev + (64u + lane) : g + (lane - 32u), lp[lane < 32u ? 96u + lane : lane - 32u]);
        AZ_ST2(g + (32u + lane), lp[32u + lane]);                     // row pieces 32..95 (staged from byte 0)
    }
}

// FRAME = false: TEME output, no epilogue code at all (keeps its registers and SGPRs out of the
// hot kernel); FRAME = true: ECEF / geodetic chosen at run time by p.mode.
template <int LAYOUT, bool VEL, bool DEEP, bool FRAME, bool SCREEN = false>
// (round 6: the time-major deep-space instantiations -- grids shorter than 32 points only, everything longer takes k_rows_deep --
// get a whole SIMD's registers: at two waves they spilled 12-28 bytes for nothing)
__global__ void __launch_bounds__(AZ_BLOCK, (AZ_MIN_WAVES > 1 ? AZ_MIN_WAVES : (FRAME ? 1 : (DEEP ? (LAYOUT == 1 ? 1 : AZ_DEEP_WAVES) : 3)))) k_propagate(PropArgs p)
{
    constexpr int WAVES = AZ_BLOCK / 64;
    // deep-space lists are never runs of consecutive rows: no time-major staging (LDS is needed for
    // the 37 cold coefficients; 18.9 KB/wave lets two workgroups share a CU)
    constexpr int SLICE = (LAYOUT == 1) ? (DEEP ? 0 : AZ_TM_ROW) : (AZ_SM_CHUNK > 0 ? 64 * AZ_SM_ROW : 0);
    constexpr int COLD = (DEEP ? (int)D_NUM : (int)C_NUM) * 64; // once-per-step constants, one column per lane
    __shared__ __attribute__((aligned(16))) double lds[WAVES * ((VEL ? 2 : 1) * SLICE + COLD)];

    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = threadIdx.x >> 6;
    double *lds_w = lds + wave * ((VEL ? 2 : 1) * SLICE + COLD); // this wave's private slice
    double *lds_p = lds_w;
    double *lds_v = lds_w + SLICE;
    double *cold = lds_w + (VEL ? 2 : 1) * SLICE + lane;
    ColdLds cold4{cold};

    // gridDim.x is padded to a multiple of 8, so workgroup (x, y) lands on XCD x % 8 for every time
    // tile y: a satellite group's element columns are fetched into one XCD's L2 only
    const unsigned li0 = blockIdx.x * AZ_BLOCK + wave * 64; // first list slot of this wave
    if (li0 >= p.n_list) return;                            // padding workgroup / wave beyond the list
    const unsigned li = li0 + lane;
    const bool in_range = li < p.n_list;
    // out-of-range lanes shadow the last satellite (the reference pads its last batch the same way,
    // Constellation.zig L145-147) so the wave stays convergent; they never store
    const unsigned s = p.list[in_range ? li : p.n_list - 1];
    const unsigned fl = p.flags[s];
    const bool wr = in_range && (p.mask == nullptr || p.mask[s] != 0) && s >= p.row_lo && s < p.row_hi;
    if (!SCREEN && !az_any(wr)) return; // nothing to write (masked out / outside the row window)
    const double off = p.offsets ? p.offsets[s] : 0.0;
    const unsigned t0 = blockIdx.y * p.tile;
    const unsigned t1 = min(t0 + p.tile, p.n_times);

    // time-major staged stores: this wave's 64 satellites are consecutive catalog rows IN LANE ORDER (the host
    // orders every 256-slot group of the list by eccentricity class, so a wave that straddles a class boundary
    // holds a permutation of -- or not even -- a run of rows: the vote looks at the real mapping)
    const unsigned s_first = p.list[li0];
    const bool dense = (LAYOUT == 1) && !DEEP && (p.mask == nullptr) && !p.f32 && !az_any(!(wr && s == s_first + lane));

    double tcache = 0.0;
    const RotK rk = az_rotk();
    double best_d2 = __builtin_inf(); // fused screen only (SCREEN)
    unsigned best_t = 0xffffffffu;

    // what happens to one step's result: the fused screen, or the store path of the layout
    auto emit = [&](unsigned i, double (&r)[3], double (&v)[3], int rc) {
        if (FRAME) az_epilogue(r, v, p.mode, VEL, p.sin_g, p.cos_g, i);
        if (SCREEN) {
            // fused single-target screen (lane = satellite): running minimum over this tile, no stores
            const double *q = p.screen_target + (size_t)i * 3; // wave-uniform address
            const double dx = q[0] - r[0], dy = q[1] - r[1], dz = q[2] - r[2];
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (rc == 0 && d2 < best_d2) {
                best_d2 = d2;
                best_t = i;
            }
            return;
        }
        if (DEEP) {
            if (rc != 0) {
                r[0] = r[1] = r[2] = 0.0;
                v[0] = v[1] = v[2] = 0.0;
                if (p.err && wr) p.err[(size_t)s * p.n_times + i] = (unsigned char)rc;
            }
        }
#if defined(AZ_ABLATE) && AZ_ABLATE == 1 /* tuning experiment: arithmetic only */
        if (r[0] != 1.2345e300) return;
#endif
        if (LAYOUT == 1) {
            if (dense) {
                // A row of 64 satellites is 1,536 bytes = one and a HALF 1-KB store instructions, and a half-filled
                // store instruction costs the write path as much as a full one (tools/store_pattern_probe.hip: 4.8 TB/s
                // with the half stores, 6.2 without).  Two consecutive rows make three full instructions:
                //   even step:  row i   pieces  0..63                          (its last 512 B wait in LDS)
                //   odd step :  row i-1 pieces 64..95 | row i pieces 0..31,    then row i pieces 32..95
                // The staging slice is 2,048 bytes per array: an even step stages its row at byte 512, an odd one at
                // byte 0, so the even row's tail [1536, 2048) survives the odd step's staging without a copy.
                const unsigned par = (i - t0) & 1u; // wave-uniform
                double *sp = lds_p + (par ? 0 : 64);
                sp[lane * 3 + 0] = r[0];
                sp[lane * 3 + 1] = r[1];
                sp[lane * 3 + 2] = r[2];
                if (VEL) {
                    double *sv = lds_v + (par ? 0 : 64);
                    sv[lane * 3 + 0] = v[0];
                    sv[lane * 3 + 1] = v[1];
                    sv[lane * 3 + 2] = v[2];
                }
                az_wave_lds_fence();
                const size_t ob = ((size_t)i * p.stride_sats + s_first) * 3;
                az_tm_flush(lds_p, p.pos + ob, (size_t)p.stride_sats * 3, lane, par, i + 1 == t1);
                if (VEL) az_tm_flush(lds_v, p.vel + ob, (size_t)p.stride_sats * 3, lane, par, i + 1 == t1);
                az_wave_lds_fence();
            } else if (wr) {
#if defined(AZ_ABLATE) && AZ_ABLATE == 3
                const size_t ob = (((size_t)i * p.stride_sats + s) * 3) & 0x1ffffu;
#else
                const size_t ob = ((size_t)i * p.stride_sats + s) * 3;
#endif
                if (p.f32) {
                    az_put3(reinterpret_cast<float *>(p.pos) + ob, r);
                    if (VEL) az_put3(reinterpret_cast<float *>(p.vel) + ob, v);
                    return;
                }
                AZ_ST1(p.pos + ob, r[0]);
                AZ_ST1(p.pos + ob + 1, r[1]);
                AZ_ST1(p.pos + ob + 2, r[2]);
                if (VEL) {
                    AZ_ST1(p.vel + ob, v[0]);
                    AZ_ST1(p.vel + ob + 1, v[1]);
                    AZ_ST1(p.vel + ob + 2, v[2]);
                }
            }
        } else if (AZ_SM_CHUNK == 0) {
            // satellite-major, direct: every lane appends 24 B to its own row each step.  One store
            // instruction touches 64 cache lines, but a lane's consecutive steps are adjacent in
            // memory, so the pieces combine in L2 before they reach HBM -- and no LDS is spent on
            // staging, which keeps the occupancy of the time-major kernel.
            if (wr) {
                const size_t ob = ((size_t)s * p.n_times + i) * 3;
                if (p.f32) {
                    az_put3(reinterpret_cast<float *>(p.pos) + ob, r);
                    if (VEL) az_put3(reinterpret_cast<float *>(p.vel) + ob, v);
                    return;
                }
                AZ_ST1(p.pos + ob, r[0]);
                AZ_ST1(p.pos + ob + 1, r[1]);
                AZ_ST1(p.pos + ob + 2, r[2]);
                if (VEL) {
                    AZ_ST1(p.vel + ob, v[0]);
                    AZ_ST1(p.vel + ob + 1, v[1]);
                    AZ_ST1(p.vel + ob + 2, v[2]);
                }
            }
        } else {
            const unsigned k = (i - t0) % (AZ_SM_CHUNK > 0 ? AZ_SM_CHUNK : 1);
            double *row = lds_p + lane * AZ_SM_ROW + k * 3;
            row[0] = r[0];
            row[1] = r[1];
            row[2] = r[2];
            if (VEL) {
                double *vrow = lds_v + lane * AZ_SM_ROW + k * 3;
                vrow[0] = v[0];
                vrow[1] = v[1];
                vrow[2] = v[2];
            }
            const bool flush = (k == AZ_SM_CHUNK - 1) || (i + 1 == t1);
            if (flush) {
                az_wave_lds_fence();
                const unsigned nsteps = k + 1;
                const unsigned tb = i - k; // first time index of this chunk
                const unsigned words = nsteps * 3;
                const unsigned total = 64 * words;
                for (unsigned w = lane; w < total; w += 64) {
                    const unsigned rl = w / words, cw = w - rl * words;
                    const unsigned lj = li0 + rl;
                    if (lj < p.n_list) {
                        const unsigned sj = p.list[lj];
                        if (p.mask == nullptr || p.mask[sj] != 0) {
                            const size_t ob = ((size_t)sj * p.n_times + tb) * 3 + cw;
                            if (p.f32) {
                                reinterpret_cast<float *>(p.pos)[ob] = (float)lds_p[rl * AZ_SM_ROW + cw];
                                if (VEL) reinterpret_cast<float *>(p.vel)[ob] = (float)lds_v[rl * AZ_SM_ROW + cw];
                            } else {
                                p.pos[ob] = lds_p[rl * AZ_SM_ROW + cw];
                                if (VEL) p.vel[ob] = lds_v[rl * AZ_SM_ROW + cw];
                            }
                        }
                    }
                }
                az_wave_lds_fence();
            }
        }
    };
    // Time values: one coalesced 512-B vector load per 64 steps parks 64 of them in a VGPR (one
    // per lane); each step then broadcasts its value with v_readlane.  A per-step load would be
    // followed by s_waitcnt vmcnt(0), which also waits for every outstanding global STORE of the
    // previous steps and serialises the arithmetic against the write stream.
    auto time_at = [&](unsigned i) {
        const unsigned k64 = __builtin_amdgcn_readfirstlane((i - t0) & 63u);
        if (k64 == 0) tcache = p.times[min(i + lane, p.n_times - 1)];
        return az_readlane_f64(tcache, k64) + off;
    };

    unsigned i = t0;
#if AZ_PROP_FAST
    // Optimistic straight-line loop (fast_step.h): uniform grid, all 64 orbits near-circular, every small
    // angle inside its usual tier; one vote per step, the generic loop takes over on a violation.
    if (!DEEP && p.inc != nullptr && p.uniform_step != 0.0 && p.delta == nullptr && p.delta64 == nullptr && !az_any(AZ_FLAG_ECLASS(fl) != 0)) {
        FastKCol k;
        bool window_ok;
        {
            FastK k0;
            az_load_fast(p.el, p.n_pad, s, fl, p.inc, 1, k0);
            az_fast_window(p.el, p.n_pad, s, p.times[t0] + off, p.times[t1 - 1] + off, p.uniform_step, k0);
            window_ok = !az_any(in_range && !az_fast_window_ok<false>(k0, p.g, p.times[t0] + off, p.times[t1 - 1] + off));
            static_assert((int)FC_NUM <= (int)C_NUM, "the fast step's cold constants share the generic step's LDS columns");
#define X(n) cold[FC_##n * AZ_COLD_STRIDE] = k0.n##_;
            AZ_FASTK_COLD(X)
#undef X
#define X(n) k.n##_ = k0.n##_;
            AZ_FASTK_HOT(X)
#undef X
            k.cold = cold;
            az_wave_lds_fence();
        }
        FastCarry fc;
        az_seed_fast(p.el, p.n_pad, s, p.times[t0] + off - p.uniform_step, k.tc_, fc);
#pragma unroll 1
        for (; window_ok && i < t1; ++i) {
            const double t = time_at(i);
            double r[3], v[3];
#if defined(AZ_ABLATE) && AZ_ABLATE == 2 /* tuning experiment: stores only */
            r[0] = t; r[1] = t + 1.0; r[2] = t + 2.0; v[0] = t + 3.0; v[1] = t + 4.0; v[2] = t + 5.0;
#else
            az_sgp4_fast_step<VEL, false>(k, p.g, RotCoefLit(), t, fc, r, v); // (validated for the whole tile above)
#endif
            emit(i, r, v, 0);
        }
    }
#endif
    if (i < t1) {
        const unsigned g0 = i; // first step of the generic loop
        if ((g0 - t0) & 63u) tcache = p.times[min(g0 - ((g0 - t0) & 63u) + lane, p.n_times - 1)];
        Sgp4Lane e4;
        Sgp4Carry c4;
        Sdp4Lane e8;
        Sdp4Carry c8;
        if (DEEP) {
            az_load_sdp4(p.el, p.n_pad, s, fl, e8, ColdLds{cold});
            if (p.seeds) {
                // resonance state at this tile's first time, prepared once by k_deep_seed: a tile never
                // re-integrates from epoch (a 300-day-old resonant element set would cost 600+ integrator
                // steps per tile otherwise)
                const double *sd = p.seeds + (size_t)blockIdx.y * 3 * p.n_list + (in_range ? li : p.n_list - 1);
                c8.atime = sd[0];
                c8.xli = sd[p.n_list];
                c8.xni = sd[2 * (size_t)p.n_list];
            } else {
                c8.atime = 0.0;
                c8.xli = e8(H_xlamo);
                c8.xni = e8(H_no_unkozai);
            }
        } else {
            az_load_sgp4(p.el, p.n_pad, s, fl, e4, cold4);
            c4.t_prev = 0.0;
            c4.sW = c4.sO = c4.sA = 0.0;
            c4.cW = c4.cO = c4.cA = 1.0;
        }
#pragma unroll 1
        for (; i < t1; ++i) {
            const double t = time_at(i);
            double r[3], v[3];
            int rc = 0;
#if defined(AZ_ABLATE) && AZ_ABLATE == 2 /* tuning experiment: stores only */
            r[0] = t; r[1] = t + 1.0; r[2] = t + 2.0; v[0] = t + 3.0; v[1] = t + 4.0; v[2] = t + 5.0;
#else
            if (DEEP) {
                rc = az_sdp4_step<VEL>(e8, ColdLds{cold}, p.g, rk, t, c8, r, v);
            } else {
                const bool first = ((i - g0) % AZ_RESEED) == 0;
                az_sgp4_step<VEL>(e4, cold4, p.el, p.n_pad, s, p.g, rk, t, first, c4, r, v);
            }
#endif
            emit(i, r, v, rc);
        }
    }
    if (SCREEN && in_range) {
        p.part_d2[(size_t)blockIdx.y * p.n_list + li] = best_d2;
        p.part_t[(size_t)blockIdx.y * p.n_list + li] = best_t;
    }
}

// Deep space: walk the resonance integrator ONCE per satellite through the tile start times and
// record (atime, xli, xni) for every tile.  Lane = deep-space list slot; sequential in time like the
// reference's carry (src/Constellation.zig L448-476), but only the integrator -- a few dozen
// instructions per 720 minutes of elapsed time -- so the serial chain is short.
// node_cache (may be null): [3][n_pad], per SATELLITE the integrator state (atime, xli, xni) the previous seeding pass left
// nearest to epoch.  The integrator's nodes -- epoch + k * 720 min -- depend on the satellite alone, not on the grid
// (src/Sdp4.zig L774-820; the reference keeps them as its carry, src/Sdp4Batch.zig L241-249), so a new grid continues from the
// cached node instead of re-integrating from epoch (a 300-day-old resonant element set: 600 steps per new grid otherwise);
// az_resonance_advance's own restart rule rejects a cached node the new grid lies inside of or on the other side of epoch.
__global__ void __launch_bounds__(64) k_deep_seed(const double *el, const unsigned *flags, size_t n_pad,
                                                  const unsigned *list, unsigned n_list, const double *times,
                                                  unsigned n_times, const double *offsets, unsigned tile,
                                                  double *seeds, int nearest, double *node_cache)
{
    __shared__ double cold_lds[D_NUM * 64];
    const unsigned li = blockIdx.x * 64 + threadIdx.x;
    const bool in_range = li < n_list;
    const unsigned s = list[in_range ? li : n_list - 1];
    const unsigned fl = flags[s];
    Sdp4Lane e;
    double *cold = cold_lds + threadIdx.x;
    az_load_sdp4(el, n_pad, s, fl, e, ColdLds{cold});
    Sdp4Carry cy;
    cy.atime = 0.0;
    cy.xli = e(H_xlamo);
    cy.xni = e(H_no_unkozai);
    if (node_cache && node_cache[s] != 0.0) {
        cy.atime = node_cache[s];
        cy.xli = node_cache[n_pad + s];
        cy.xni = node_cache[2 * n_pad + s];
    }
    Sdp4Carry near = cy; // the state nearest to epoch this pass comes by
    bool have_near = false;
    const double off = offsets ? offsets[s] : 0.0;
    const unsigned n_tiles = (n_times + tile - 1) / tile;
    for (unsigned k = 0; k < n_tiles; ++k) {
        double t = times[k * tile] + off;
        if (nearest) {
            // lane = time consumers (k_rows_deep): every lane of the chunk starts from this state and only
            // ever integrates AWAY from epoch, so seed the chunk at its grid point nearest to epoch
            // (t = 0, the epoch state itself, when the chunk straddles it)
            const double t_end = times[min((k + 1) * tile, n_times) - 1] + off;
            if (t * t_end <= 0.0) t = 0.0;
            else if (fabs(t_end) < fabs(t)) t = t_end;
        }
        if (az_any(e.irez != 0)) az_resonance_advance(e, ColdLds{cold}, t, cy);
        if (!have_near || fabs(cy.atime) < fabs(near.atime)) {
            near = cy;
            have_near = true;
        }
        if (in_range) {
            double *sd = seeds + (size_t)k * 3 * n_list + li;
            sd[0] = cy.atime;
            sd[n_list] = cy.xli;
            sd[2 * (size_t)n_list] = cy.xni;
        }
    }
    if (node_cache && in_range && e.irez != 0 && have_near) {
        node_cache[s] = near.atime;
        node_cache[n_pad + s] = near.xli;
        node_cache[2 * n_pad + s] = near.xni;
    }
}

// one satellite x many times: lane = time, the satellite's constants are wave-uniform.  Every
// evaluation is a 'first' step (full sincos seeds); deep-space lanes integrate the resonance from
// epoch themselves, like the reference's sdp4Times8 (src/Sdp4.zig L1105-1128).
#ifndef AZ_ONE_SEG
#define AZ_ONE_SEG 1024 /* points per wave of k_one_fast; k_one_satellite's item lists count in these */
#endif
// The hand-over lists of k_one_fast: AZ_ONE_LISTS of them (a segment goes onto list seg mod AZ_ONE_LISTS), each with its own
// counter on its own 128-byte line -- ten thousand waves pushing onto ONE list head with an atomic each took as long as the
// arithmetic they had skipped (118 us for an all-irregular series of 10^7 points; spread over 64 heads: 16 us).
// items[32 k] = length of list k; items[AZ_ONE_HEAD + k cap + j] = its j-th segment index.
#define AZ_ONE_LISTS 64u
#define AZ_ONE_HEAD (32u * AZ_ONE_LISTS)
// items (ITEMS only): only the points of the listed AZ_ONE_SEG-point segments are produced (what k_one_fast handed over);
// workgroup b works on list b mod AZ_ONE_LISTS, striding over its (segment, 64-point block) pairs; gridDim.x is a multiple
// of AZ_ONE_LISTS
template <bool ITEMS = false> // (a separate instantiation: the item loop costs the plain form registers)
__global__ void __launch_bounds__(64) k_one_satellite(const double *__restrict__ el, const unsigned *__restrict__ flags,
                                                            size_t n_pad, unsigned sat, const double *__restrict__ tsince,
                                                            unsigned n, double *pos, double *vel,
                                                            unsigned char *err, int interleaved, AzGrav g,
                                                            const double *__restrict__ offsets, int nan_on_error,
                                                            const unsigned *__restrict__ items = nullptr, unsigned list_cap = 0)
{
    const unsigned sub = ITEMS ? (blockIdx.x % AZ_ONE_LISTS) : 0u;
    const unsigned n_work = ITEMS ? items[32u * sub] * (AZ_ONE_SEG / 64u) : 1u;
#pragma unroll 1
    for (unsigned w = ITEMS ? blockIdx.x / AZ_ONE_LISTS : 0u; w < n_work; w += ITEMS ? gridDim.x / AZ_ONE_LISTS : 1u) {
    const unsigned i = ITEMS ? items[AZ_ONE_HEAD + sub * list_cap + w / (AZ_ONE_SEG / 64u)] * AZ_ONE_SEG + (w % (AZ_ONE_SEG / 64u)) * 64u + threadIdx.x
                             : blockIdx.x * 64 + threadIdx.x;
    if (ITEMS && i - threadIdx.x >= n) continue; // (the last segment's blocks past the end)
    const double t = tsince[i < n ? i : n - 1] + (offsets ? offsets[sat] : 0.0);
    const unsigned fl = flags[sat];
    double r[3], v[3];
    int rc = AZ_FLAG_ERR(fl);
    if (rc == 0) {
        if (fl & AZ_FLAG_DEEP) {
            Sdp4Lane e;
            Sdp4Carry c;
            __shared__ double cold_deep[D_NUM * 64];
            double *cold = cold_deep + threadIdx.x;
            az_load_sdp4(el, n_pad, sat, fl, e, ColdLds{cold});
            c.atime = 0.0;
            c.xli = e(H_xlamo);
            c.xni = e(H_no_unkozai);
            rc = az_sdp4_step<true>(e, ColdLds{cold}, g, az_rotk(), t, c, r, v);
        } else {
            Sgp4Lane e;
            Sgp4Carry c;
            ColdRegs cold;
            az_load_sgp4(el, n_pad, sat, fl, e, cold);
            c.t_prev = 0.0;
            az_sgp4_step<true>(e, cold, el, n_pad, sat, g, az_rotk(), t, true, c, r, v);
        }
    }
    if (rc != 0) {
        r[0] = r[1] = r[2] = nan_on_error ? __builtin_nan("") : 0.0;
        v[0] = v[1] = v[2] = 0.0;
    }
    if (i < n) {
        if (interleaved) {
            double *o = pos + (size_t)i * 6;
            o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = v[0]; o[4] = v[1]; o[5] = v[2];
        } else {
            pos[(size_t)i * 3] = r[0]; pos[(size_t)i * 3 + 1] = r[1]; pos[(size_t)i * 3 + 2] = r[2];
            if (vel) { vel[(size_t)i * 3] = v[0]; vel[(size_t)i * 3 + 1] = v[1]; vel[(size_t)i * 3 + 2] = v[2]; }
        }
        if (err) err[i] = (unsigned char)rc;
    }
    }
}

// Window plan of a staged uniform grid, once per (grid, launch shape): one lane per (list slot, time segment) evaluates
// what every wave of the row / tile kernels used to redo at its start -- the window-centred constants of the fast step
// (fast_step.h: az_fast_window) and its validation bounds (az_fast_window_ok) -- and files the windows the bounds reject
// as the STATIC part of the redo list.  The fast kernels then read four scalars and a flag per wave (no sincos, no
// divisions, no compares, no vote), never file class-0 items themselves, and the generic pass over the rejected windows
// no longer depends on them: it runs beside the bulk launch instead of after it.
//   slots [0, n_circ): near-circular form, segments of tile_c points; the rest: eccentric form, tile_e; TILES: the class
//   is the satellite's own (k_tiles_fast: catalog-ordered list, one tile length).  dt_mult: grid steps per lane step
//   (64; 128 for the packed fp32 kernel, which also needs the one-grid-step increment of U).
struct PlanArgs {
    const double *el;
    const unsigned *flags;
    size_t n_pad;
    const unsigned *list;
    unsigned n_list, n_circ, n_times, tile_c, tile_e, by_flags;
    unsigned f32_mixed; // near-circular slots go to the mixed-precision fp32 step: its extra bound (az_fast32p_window_ok)
    const double *times, *offsets, *inc;
    double step, dt_mult, delta_max, grid_t0;
    double *win;
    unsigned char *flag;
    unsigned *redo_static, *redo_c0, *redo_c1, *redo_items;
    AzGrav g;
};
__global__ void __launch_bounds__(256) k_plan_windows(PlanArgs a)
{
    const unsigned slot = blockIdx.x * 256 + threadIdx.x;
    const unsigned seg = blockIdx.y;
    if (slot >= a.n_list) return;
    const unsigned s = a.list[slot];
    const unsigned fl = a.flags[s];
    const bool ecc = a.by_flags ? AZ_FLAG_ECLASS(fl) != 0 : slot >= a.n_circ;
    const unsigned tile = ecc ? a.tile_e : a.tile_c;
    const unsigned t_lo = seg * tile;
    if (t_lo >= a.n_times) return;
    const unsigned t_hi = min(t_lo + tile, a.n_times);
    const double t_first = a.grid_t0 + (a.offsets ? a.offsets[s] : 0.0);
    const double w_a = fma((double)t_lo, a.step, t_first), w_b = fma((double)(t_hi - 1), a.step, t_first);
    FastK k0, k1;
    az_load_fast(a.el, a.n_pad, s, fl, a.inc, 0, k0);
    const double dt_mult = ecc ? 64.0 : a.dt_mult; // (eccentric members always take the fp64 row kernel: 64 grid points per lane step)
    if (dt_mult == 128.0) az_double_increments(k0);
    az_fast_window(a.el, a.n_pad, s, w_a, w_b, dt_mult * a.step, k0);
    az_load_fast(a.el, a.n_pad, s, fl, a.inc, 1, k1);
    az_fast_window(a.el, a.n_pad, s, w_a, w_b, a.step, k1);
    // a class the form cannot take (an eccentric member in the near-circular list) is rejected like a failed bound
    bool ok = ecc ? az_fast_window_ok<true>(k0, a.g, w_a, w_b, a.delta_max) : az_fast_window_ok<false>(k0, a.g, w_a, w_b, a.delta_max);
    if (!ecc && AZ_FLAG_ECLASS(fl) != 0) ok = false;
    if (!ecc && a.f32_mixed && !az_fast32p_window_ok(k0, w_a, w_b)) ok = false;
    const size_t at = (size_t)seg * a.n_list + slot;
    double *w = a.win + at * AZ_PLAN_NUM;
    w[AZ_PLAN_sOc] = k0.sOc_; w[AZ_PLAN_cOc] = k0.cOc_; w[AZ_PLAN_sdU] = k0.sdU_; w[AZ_PLAN_cdU] = k0.cdU_;
    w[AZ_PLAN_s1U] = k1.sdU_; w[AZ_PLAN_c1U] = k1.cdU_;
    a.flag[at] = (unsigned char)((ok ? AZ_PLAN_OK : 0u) | (k0.tc_ != 0.0 ? AZ_PLAN_TC : 0u));
    if (!ok) {
        const unsigned k = atomicAdd(a.redo_static, 1u);
        a.redo_items[3 * (size_t)k + 0] = slot;
        a.redo_items[3 * (size_t)k + 1] = t_lo;
        a.redo_items[3 * (size_t)k + 2] = t_hi;
    }
}
// after k_plan_windows: both dynamic counters start behind the static items
__global__ void k_plan_arm(unsigned *redo_static, unsigned *c0, unsigned *c1)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *c0 = *c1 = *redo_static;
}
// window constants of one wave's segment from the plan (wave-uniform scalar loads)
template <class K>
__device__ __forceinline__ bool az_plan_window(const PropArgs &p, unsigned seg, unsigned slot, double w_a, double w_b, K &k0)
{
    const size_t at = (size_t)seg * p.plan_stride + slot;
    const double *w = p.plan_win + at * AZ_PLAN_NUM;
    const unsigned f = p.plan_flag[at];
    k0.tmid_ = 0.5 * (w_a + w_b);
    k0.tc_ = (f & AZ_PLAN_TC) ? k0.tmid_ : 0.0;
    k0.sOc_ = w[AZ_PLAN_sOc]; k0.cOc_ = w[AZ_PLAN_cOc]; k0.sdU_ = w[AZ_PLAN_sdU]; k0.cdU_ = w[AZ_PLAN_cdU];
    return (f & AZ_PLAN_OK) != 0;
}

// Satellite-major output, near-earth: ONE WAVE PER SATELLITE ROW, lane = time.
// The row (s, :, :) is contiguous in memory, so with lane = time a wave's 64 results of one
// iteration are 1,536 contiguous bytes per array -- perfectly coalesced without any staging.  All
// per-satellite constants are wave-uniform (forced into SGPRs with v_readfirstlane): no LDS, ~100
// VGPRs less than the lane = satellite kernel, and the Kepler-Newton trip count and every rotation
// tier are uniform by construction (one orbit per wave).  Each lane carries its own (sin,cos) pairs
// of the slow angles across iterations (64 grid points apart); on a uniform grid the increments of
// the mean anomaly and of the argument of perigee are the same every iteration and are cached.
AZ_DEVICE double az_uniform(double x)
{
#ifdef AZ_HOST_EMUL
    return x;
#else
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)),
                            __builtin_amdgcn_readfirstlane(__double2loint(x)));
#endif
}
AZ_DEVICE float az_uniform32(float x)
{
#ifdef AZ_HOST_EMUL
    return x;
#else
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
#endif
}
// p, as an LDS address held in a VGPR that the compiler cannot see through (loads through it are not hoisted out of a
// loop, and stay ds_read with immediate offsets from this one base)
__device__ __forceinline__ const double *az_opaque_lds(const double *p)
{
    typedef __attribute__((address_space(3))) const double lds_cdouble;
    lds_cdouble *q = (lds_cdouble *)p;
    asm volatile("" : "+v"(q));
    return (const double *)q;
}
struct ColdBroadcast;
// once-per-step constants of a lane = time kernel: one LDS word per constant, read by all 64 lanes
// at once (broadcast ds_read_b64: no VALU slot, no SGPRs -- the 33 uniform doubles of a satellite do
// not fit the SGPR file next to the kernel's pointers, and every SGPR spilled to a VGPR lane costs
// a v_readlane per use)
struct ColdBroadcast {
    double *p;
    const double *m = nullptr; // polynomial coefficients (devmath.h AzMathConst) as LDS words; null: literals
    __device__ __forceinline__ double operator()(int k) const { return p[k]; }
    __device__ __forceinline__ void set(int k, double v) const { p[k] = v; }
    __device__ __forceinline__ double mc(int k) const { return m[k]; }
    // the same table behind an address the compiler cannot see through: one per call site of a polynomial, so that the
    // coefficient reads of one call are not merged with the next call's (merged, they stay in registers across the step)
    __device__ __forceinline__ ColdBroadcast fresh() const;
};
// ... the same with literal coefficients (set-up code, kernels without the table)
struct ColdBroadcastLit {
    double *p;
    __device__ __forceinline__ double operator()(int k) const { return p[k]; }
    __device__ __forceinline__ void set(int k, double v) const { p[k] = v; }
    __device__ __forceinline__ double mc(int k) const { return az_mc_literal(k); }
    __device__ __forceinline__ const ColdBroadcastLit &fresh() const { return *this; }
};
__device__ const double az_mc_table[MC_NUM] = AZ_MC_VALUES;
__device__ const double az_rc_table[RC_NUM] = AZ_RC_VALUES;
// sincos coefficients of a wave's seeds through its LDS table (the first AZ_SC_NUM entries of AzMathConst)
#define AZ_SC_NUM (MC_C5 + 1)
#define AZ_FAST_TABLE ((FCX_NUM + RC_NUM + AZ_SC_NUM + 1) & ~1) /* doubles per wave, 16-byte multiple */
struct McLds {
    const double *p;
    __device__ __forceinline__ double mc(int k) const { return p[k]; }
};
// One vector load fills a wave's constant table in LDS: lane j < FCX_NUM fetches cold field j of the satellite's record, the
// next RC_NUM lanes the rotation coefficients, the next AZ_SC_NUM the sincos coefficients (round 3 wrote the first two
// groups from lane 0, two register moves and a share of an LDS write per constant: 100 VALU slots per segment).
__device__ __forceinline__ void az_fill_fast_table(double *table, const double *__restrict__ rec, unsigned lane)
{
    static_assert(FCX_NUM + RC_NUM + AZ_SC_NUM <= 64, "one lane per table entry");
    const double *src = lane < FCX_NUM ? rec + lane
                        : lane < FCX_NUM + RC_NUM ? az_rc_table + (lane - FCX_NUM) : az_mc_table + (lane - (FCX_NUM + RC_NUM));
    if (lane < FCX_NUM + RC_NUM + AZ_SC_NUM) table[lane] = *src;
}
__device__ __forceinline__ ColdBroadcast ColdBroadcast::fresh() const { return ColdBroadcast{p, az_opaque_lds(m)}; }
#ifndef AZ_ROWSF_WAVES
#define AZ_ROWSF_WAVES 6 /* k_rows_fast, near-circular Kepler form: 74 VGPRs; forced to 7 (70 VGPRs) or 8 (64 + spills) it measures slower */
#endif
#ifndef AZ_ROWSF32_WAVES
#define AZ_ROWSF32_WAVES 4 /* k_rows_fast32: two grid points per lane */
#endif
#ifndef AZ_ROWSF32P_WAVES
#define AZ_ROWSF32P_WAVES 3 /* k_rows_fast32<MIXED>: two grid points per lane, their O(1) chains in fp64 */
#endif
#ifndef AZ_ROWSF_ECC_WAVES
#define AZ_ROWSF_ECC_WAVES 4 /* k_rows_fast, eccentric form: ~100 VGPRs (its own instantiation and launch, so that the
                                 few eccentric members do not cost every wave of the bulk its occupancy) */
#endif
#ifndef AZ_ROWS_TLDS
#define AZ_ROWS_TLDS 1024 /* k_rows: time values staged in LDS per refill (a power of two >= 64) */
#endif

// SINK: what happens to a result -- 0: fp64 rows, 1: fp32 rows, 2: fused single-target conjunction
// screen (nothing is stored; each lane keeps the running minimum of |r - r_target|^2 over its grid
// points, the wave reduces once at the end; src/Constellation.zig L683-756)
enum { AZ_SINK_F64 = 0, AZ_SINK_F32 = 1, AZ_SINK_SCREEN = 2 };

// lexicographic (d2, t) minimum across the wave: smallest distance, earliest grid point among equals
// (the reference scans time ascending with a strict '<', Constellation.zig L744)
__device__ __forceinline__ void az_wave_argmin(double &d2, unsigned &t)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double od = __shfl_xor(d2, off, 64);
        const unsigned ot = __shfl_xor(t, off, 64);
        if (od < d2 || (od == d2 && ot < t)) {
            d2 = od;
            t = ot;
        }
    }
}


// Row stores of the lane = time kernels through a wave-private LDS transpose: a wave's 64 results of one
// array are 64 x 24 (12) contiguous bytes of a satellite-major row; written as they sit in the lanes (a
// 16-byte and an 8-byte piece per lane at a 24-byte stride) every store instruction leaves holes that
// only close when its sibling instruction reaches the same L2 lines -- fine while the arithmetic paces the
// stores, but a store-bound kernel collapses to ~2.5 TB/s that way (measured: stores-only 0.377 ms for
// 932 MB).  Through LDS every lane stores 16 ALIGNED bytes and one instruction covers 1,024 contiguous
// bytes = 8 full 128-byte lines.  ds_write_b64 at a 24-byte lane stride is conflict-free (banks 6k mod 32).
#ifndef AZ_ROWS_LDS_STORE
#define AZ_ROWS_LDS_STORE 1
#endif
typedef float az_f4 __attribute__((ext_vector_type(4)));
template <class T>
__device__ __forceinline__ void az_stage3(T *stage, unsigned lane, const double r[3])
{
    stage[lane * 3 + 0] = (T)r[0];
    stage[lane * 3 + 1] = (T)r[1];
    stage[lane * 3 + 2] = (T)r[2];
}
// Flush of one staged iteration (64 lanes x 3 components of T, positions then velocities) as 16-byte pieces: 96 per array
// in fp64, 48 in fp32.  fp64: all 64 lanes move pieces 0-63 of each array, lanes 0-31 pieces 64-95 -- unmasked stores and
// ONE exec-masked block; fp32: lanes 0-47 move one piece per array.  The stores go through a buffer descriptor per row
// (built once per segment from scalars): address = row base (descriptor) + the iteration's byte offset (one SGPR) + this
// lane's piece offset (one VGPR shared by all stores, + an immediate) -- no vector ALU work at all.  (As independent
// `if (lane < PIECES)` global stores per array the compiler kept a masked block per store -- it does not know lane < 64 --
// and rebuilt an LDS address and a lane offset inside each, or hoisted a 64-bit address per array and re-derived it with
// v_mad_u64_u32: 8-11 VALU and 20 scalar slots per iteration.)
typedef unsigned az_u4 __attribute__((ext_vector_type(4)));
struct AzRowSink {
    __amdgpu_buffer_rsrc_t pos, vel;
};
template <class T>
__device__ __forceinline__ AzRowSink az_row_sink(T *prow, T *vrow, unsigned n_times)
{
    // (flags 0x00020000: raw buffer, no swizzle, as in the guide's T8 recipe; num_records = the row's bytes)
    const unsigned row_bytes = n_times * 3u * (unsigned)sizeof(T);
    return AzRowSink{__builtin_amdgcn_make_buffer_rsrc(prow, 0, row_bytes, 0x00020000),
                     __builtin_amdgcn_make_buffer_rsrc(vrow ? vrow : prow, 0, row_bytes, 0x00020000)};
}
#ifndef AZ_AUX_NT
#define AZ_AUX_NT 2 /* cache-policy bits of the buffer builtins on gfx94x/95x: 1 = sc0, 2 = nt, 16 = sc1 */
#endif
template <bool VEL, class T>
__device__ __forceinline__ void az_flush_stage(const T *stage, const AzRowSink &sink, unsigned base, unsigned lane)
{
    constexpr unsigned PIECES = 64 * 3 * sizeof(T) / 16; // 96 (fp64) / 48 (fp32); the velocity pieces follow at stage + 192
    constexpr int aux = AZ_ROWS_NT ? AZ_AUX_NT : 0;
    const az_u4 *src = reinterpret_cast<const az_u4 *>(stage) + lane;
    const unsigned voff = lane * 16u, soff = base * 3u * (unsigned)sizeof(T);
    if constexpr (PIECES >= 64) {
        const az_u4 a = src[0];
        __builtin_amdgcn_raw_buffer_store_b128(a, sink.pos, voff, soff, aux);
        if (VEL) {
            const az_u4 c = src[PIECES];
            __builtin_amdgcn_raw_buffer_store_b128(c, sink.vel, voff, soff, aux);
        }
        if (lane < PIECES - 64) {
            const az_u4 b = src[64];
            __builtin_amdgcn_raw_buffer_store_b128(b, sink.pos, voff + 1024u, soff, aux);
            if (VEL) {
                const az_u4 d = src[PIECES + 64];
                __builtin_amdgcn_raw_buffer_store_b128(d, sink.vel, voff + 1024u, soff, aux);
            }
        }
    } else if (lane < PIECES) {
        const az_u4 a = src[0];
        __builtin_amdgcn_raw_buffer_store_b128(a, sink.pos, voff, soff, aux);
        if (VEL) {
            const az_u4 c = src[PIECES];
            __builtin_amdgcn_raw_buffer_store_b128(c, sink.vel, voff, soff, aux);
        }
    }
}
// The row a workgroup of the lane = time row kernels takes (grid: x = rows rounded up to a multiple of 8, y = time segments).
// Workgroup b runs on XCD b % 8, and gridDim.x is a multiple of 8: the low three bits of blockIdx.x ARE the XCD.  The rows are
// dealt out in eight contiguous ranges, one per XCD (k_rows).  With a handful of rows and a long time series that alone would
// leave most XCDs idle -- one satellite x 10^7 times ran on ONE of the eight (0.54 ms; 0.30 ms through k_one_satellite) -- so
// there the ranges rotate with the time segment: segment y of range j runs on XCD (j - y) mod 8.
__device__ __forceinline__ unsigned az_xcd_row()
{
    const unsigned per_xcd = gridDim.x >> 3;
    const unsigned range = per_xcd <= 8u ? ((blockIdx.x + blockIdx.y) & 7u) : (blockIdx.x & 7u);
    return range * per_xcd + (blockIdx.x >> 3);
}
// the list slot of a workgroup of the row kernels: the XCD-aware row of a grid cut to the slot window [slot_lo, slot_hi)
__device__ __forceinline__ unsigned az_list_slot(const PropArgs &p) { return p.slot_lo + az_xcd_row(); }
__device__ __forceinline__ unsigned az_list_end(const PropArgs &p) { return p.slot_hi ? p.slot_hi : p.n_list; }
// one wave's work on a row segment, shared by the two row kernels: where a finished iteration goes
template <bool VEL, class out_t>
__device__ __forceinline__ void az_rows_store(bool staged, bool full, bool live, unsigned lane, out_t *stage, out_t *prow,
                                              out_t *vrow, unsigned base, const double r[3], const double v[3], const AzRowSink &sink)
{
    if (staged && full) {
        az_stage3(stage, lane, r);
        if (VEL) az_stage3(stage + 192, lane, v);
        az_wave_lds_fence();
        az_flush_stage<VEL>(stage, sink, base, lane);
        az_wave_lds_fence();
    } else if (live) {
        // partial iteration / unaligned row: direct 24-byte (12-byte) pieces per lane
        az_put3_stream(prow + (size_t)(base + lane) * 3, r);
        if (VEL) az_put3_stream(vrow + (size_t)(base + lane) * 3, v);
    }
}

// ONE SATELLITE x MANY TIMES, the times in device memory (azh_propagate_one_device / _host, sgp4_propagate_batch).
// Nothing on the host has seen the times, so every wave finds out for itself: it takes AZ_ONE_SEG consecutive points, fits the
// ideal grid t0 + i step through its first and last time and measures the deviations; if they stay inside the tight
// quasi-uniform form (fast_step.h, DELTA = 1: what arange, linspace and jd + fr arithmetic produce) and the window passes
// the fast step's validation bounds, the wave sets the branch-free step up by itself -- the increments of a lane step (two
// sincos), the window constants (one or two), its seeds (two or three): the work k_prep_inc / k_prep_rec / k_plan_windows do
// once per staged grid for the constellation kernels, ~600 instructions here against sixteen iterations of ~205 -- and
// runs it; otherwise (irregular times, a short tail, a window the bounds reject, an eccentric member whose Newton
// validation fails, deep space) the segment goes onto a list that k_one_satellite, launched behind, works off point by
// point with the generic step (~600 instructions and full sincos seeds per point: 0.29 ms for 10^7 points).
// The constants stay where the compiler puts them (uniform values in vector registers: this kernel runs at 3 waves/SIMD).
template <bool VEL>
__global__ void __launch_bounds__(64, 3) k_one_fast(const double *el, const unsigned *flags, size_t n_pad, unsigned sat,
                                                    const double *tsince, unsigned n, double *pos, double *vel, unsigned char *err,
                                                    AzGrav g, const double *offsets, unsigned *items, unsigned list_cap)
{
    const unsigned lane = threadIdx.x, seg = blockIdx.x, lo = seg * AZ_ONE_SEG;
    const unsigned cnt = min((unsigned)AZ_ONE_SEG, n - lo);
    const unsigned fl = flags[sat];
    __shared__ float dl_lds[AZ_ONE_SEG];
    __shared__ __attribute__((aligned(16))) double rows_stage[2 * 64 * 3];
    const double off = az_uniform(offsets ? offsets[sat] : 0.0);
    const bool ecc = AZ_FLAG_ECLASS(fl) != 0; // wave-uniform
    bool ok = AZ_FLAG_ERR(fl) == 0 && !(fl & AZ_FLAG_DEEP) && cnt >= 128u;
    double t0 = 0.0, step = 0.0;
    FastK k;
    if (ok) {
        t0 = az_uniform(tsince[lo]) + off;
        step = ((az_uniform(tsince[lo + cnt - 1]) + off) - t0) / (double)(cnt - 1u);
        bool far = false;
#pragma unroll 4
        for (unsigned j = lane; j < (unsigned)AZ_ONE_SEG; j += 64u) {
            const unsigned jj = min(j, cnt - 1u);
            const double d = (tsince[lo + jj] + off) - fma((double)jj, step, t0);
            dl_lds[j] = (float)d;
            far |= !(fabs(d) <= AZ_DELTA_MAX); // (NaN: far)
        }
        // (the window may not outgrow the window-centred constants either: fast_window_cap on the host side)
        ok = !az_any(far) && step != 0.0 && fabs(step) * (double)cnt <= 3000.0;
    }
    if (ok) {
#define L(f) el[(size_t)F_##f * n_pad + sat]
        double sdA, cdA, sdW, cdW;
        az_sincos(L(mdot) * (64.0 * step), sdA, cdA);
        az_sincos(L(argpdot) * (64.0 * step), sdW, cdW);
#undef L
        az_load_fast_with(el, n_pad, sat, fl, sdA, cdA, sdW, cdW, k);
        const double w_a = t0, w_b = fma((double)(cnt - 1u), step, t0);
        az_fast_window(el, n_pad, sat, w_a, w_b, 64.0 * step, k);
        ok = ecc ? az_fast_window_ok<true>(k, g, w_a, w_b, AZ_DELTA_MAX) : az_fast_window_ok<false>(k, g, w_a, w_b, AZ_DELTA_MAX);
    }
    if (ok) {
        FastCarry fc;
        az_seed_fast(el, n_pad, sat, fma((double)lane - 64.0, step, t0), k.tc_, fc); // one lane step before the lane's first point
        double *prow = pos + (size_t)lo * 3, *vrow = VEL ? vel + (size_t)lo * 3 : nullptr;
        const bool staged = ((reinterpret_cast<size_t>(prow) | (VEL ? reinterpret_cast<size_t>(vrow) : 0)) & 15u) == 0;
        const AzRowSink sink = az_row_sink(prow, vrow, cnt);
        az_wave_lds_fence();
#pragma unroll 1
        for (unsigned base = 0; base < cnt; base += 64u) {
            const unsigned i = base + lane;
            const bool live = i < cnt;
            const double dl = (double)dl_lds[i]; // (i < AZ_ONE_SEG always; past cnt the entry is the last point's)
            const double t = fma((double)i, step, t0) + dl;
            double r[3], v[3];
            const bool bad = ecc ? az_sgp4_fast_step<VEL, true, 1>(k, g, RotCoefLit(), t, fc, r, v, dl)
                                 : az_sgp4_fast_step<VEL, false, 1>(k, g, RotCoefLit(), t, fc, r, v, dl);
            if (az_any(bad && live)) { // (eccentric form only) the generic kernel redoes the whole segment
                ok = false;
                break;
            }
            az_rows_store<VEL>(staged, base + 64u <= cnt, live, lane, rows_stage, prow, vrow, base, r, v, sink);
            if (err && live) err[lo + i] = 0;
        }
    }
    if (!ok && lane == 0) {
        const unsigned sub = seg % AZ_ONE_LISTS;
        items[AZ_ONE_HEAD + sub * list_cap + atomicAdd(items + 32u * sub, 1u)] = seg;
    }
}

// Near-earth rows on a UNIFORM grid: the branch-free step of fast_step.h, one wave per (satellite row, time
// segment), lane = time.  74 VGPRs (6 waves/SIMD; the generic k_rows needs 135 = 3).  The satellite's constants come
// from its fast record (k_prep_rec): the 23 hot ones by scalar loads into SGPRs, the 16 once-per-step ones and the
// polynomial coefficients by one vector load into the wave's LDS table (broadcast reads); the window's constants and the
// verdict of the validation bounds from the plan (k_plan_windows); time is t0 + i*step (no staging, no loads in the
// loop).  Two instantiations, launched on the host's two near-earth lists: ECC = false for eccentricity class 0
// (near-circular Kepler form), ECC = true for the other classes (general form).  A window the plan rejected is already an
// item of the redo list and its wave exits at once; a wave of the eccentric form whose Newton iteration needs more than
// its five fixed-tier trips appends the rest of its segment to that list; the generic kernel runs the list beside the bulk.
// FRAME: 0 TEME, 1 ECEF, 2 geodetic (compile-time: the geodetic conversion's registers stay out of the ECEF kernel).  The
// (sin,cos) of the Greenwich angle of AZ_FRAME_SEG points of this wave's segment at a time (refilled inside the loop)
// are staged in LDS before the loop: no table load inside it (round 2's FRAME kernels loaded two doubles per iteration,
// each load waiting for the stores in flight, and carried the run-time choice of ECEF / geodetic: 149-173 VGPRs + scratch).
// The pair is NOT carried by a constant rotation although the angle is linear in time: the reference evaluates GMST from
// jd = reference_jd + t / 1440 (src/Constellation.zig L573-581), whose rounding at 2.46e6 days moves every table entry by up
// to 3e-9 rad -- the table, noise included, is what parity is measured against (a carried pair drifts 1e-8 rad = 85 mm).
#define AZ_FRAME_SEG 256
// DELTA: quasi-uniform grid (fast_step.h): the segment's deviations delta_i, staged in LDS as fp32 before the loop (|delta| <=
// 4e-6 min: an fp32 delta is good to 2.4e-13 min, the rounding of t itself), ride on the time value and on the small rotation of U.
template <bool VEL, int FRAME, int SINK, bool ECC, int DELTA = 0> // DELTA: 0 exact grid, 1 tight (fp32 deviations), 2 wide (fp64)
__global__ void __launch_bounds__(64, FRAME == 2 ? 4 : (ECC ? AZ_ROWSF_ECC_WAVES : (FRAME == 1 || DELTA == 2 ? 5 : AZ_ROWSF_WAVES))) k_rows_fast(PropArgs p)
{
    typedef typename std::conditional<SINK == AZ_SINK_F32, float, double>::type out_t;
    const unsigned lane = threadIdx.x;
    const unsigned row = az_list_slot(p); // XCD-aware row assignment, see k_rows
    if (row >= az_list_end(p)) return;
    const unsigned s = p.list[row]; // wave-uniform
    if (SINK != AZ_SINK_SCREEN && ((p.mask != nullptr && p.mask[s] == 0) || s < p.row_lo || s >= p.row_hi)) return;
    const unsigned t_lo = blockIdx.y * p.tile;
    if (t_lo >= p.n_times) return; // (the screen launches both forms on one grid: the finer tiling decides its height)
    const unsigned t_hi = min(t_lo + p.tile, p.n_times);
    unsigned base = t_lo;
    bool window_ok;
    double best_d2 = __builtin_inf(); // SINK_SCREEN: this lane's running minimum of |r - r_target|^2 and its grid point
    unsigned best_t = 0xffffffffu;
    {
        __shared__ __attribute__((aligned(16))) double cold_lds[AZ_FAST_TABLE];
        __shared__ __attribute__((aligned(16))) out_t rows_stage[AZ_ROWS_LDS_STORE ? 2 * 64 * 3 : 4];
        const double off = az_uniform(p.offsets ? p.offsets[s] : 0.0);
        const double *__restrict__ rec = p.fast_rec + (size_t)s * FR_NUM;
        az_fill_fast_table(cold_lds, rec, lane);
        const size_t out_row = p.rows_compact ? (size_t)p.ecc_row0 + row : (size_t)s; // (compact scratch rows ahead of k_cols_fast)
        out_t *prow = SINK == AZ_SINK_SCREEN ? nullptr : reinterpret_cast<out_t *>(p.pos) + out_row * p.n_times * 3;
        out_t *vrow = VEL ? reinterpret_cast<out_t *>(p.vel) + out_row * p.n_times * 3 : nullptr;
        const bool staged = AZ_ROWS_LDS_STORE && SINK != AZ_SINK_SCREEN &&
                            (((reinterpret_cast<size_t>(prow) | (VEL ? reinterpret_cast<size_t>(vrow) : 0)) & 15u) == 0);
        FastKBcast k;
        {
            const double w_a = fma((double)t_lo, p.uniform_step, p.grid_t0 + off), w_b = fma((double)(t_hi - 1), p.uniform_step, p.grid_t0 + off);
            // window constants and the verdict of the validation bounds: prepared once per staged grid (k_plan_windows); a
            // rejected window is already on the redo list
            window_ok = az_plan_window(p, blockIdx.y, row + p.redo_slot0, w_a, w_b, k);
            if (!window_ok) return;
            az_fast_rec_hot(rec, k); // scalar loads: the hot constants arrive in SGPRs
            az_wave_lds_fence();
            if (DELTA && k.tc_ != 0.0 && lane == 0) cold_lds[FCX_udot] = fma(2.0 * rec[FC_nl2], k.tc_, rec[FCX_udot]); // rate of U about tc
        }
        typedef typename std::conditional<DELTA == 2, double, float>::type dl_t;
        __shared__ dl_t dl_lds[DELTA ? AZ_DELTA_SEG : 1]; // (the host keeps DELTA segments at AZ_DELTA_SEG points)
        if (DELTA) {
            const dl_t *src = DELTA == 2 ? reinterpret_cast<const dl_t *>(p.delta64) : reinterpret_cast<const dl_t *>(p.delta);
#pragma unroll
            for (unsigned j = lane; j < AZ_DELTA_SEG; j += 64) dl_lds[j] = src[t_lo + j]; // (the table is zero-padded by one segment)
        }
        az_wave_lds_fence();
        const double step = p.uniform_step;
        const double t_first = p.grid_t0 + off; // tsince of grid point 0; grid point i is t_first + i*step
        FastCarry fc;
        // seed one increment (64 grid steps) BEFORE this lane's first grid point
        az_seed_fast_rec(rec, rec[FC_nl2], fma((double)(t_lo + lane) - 64.0, step, t_first), k.tc_, fc,
                         McLds{cold_lds + FCX_NUM + RC_NUM});
        __shared__ double gst_lds[FRAME ? 2 * AZ_FRAME_SEG : 2];
        if (FRAME) {
#pragma unroll
            for (unsigned j = lane; j < AZ_FRAME_SEG; j += 64) {
                const unsigned jj = min(t_lo + j, p.n_times - 1);
                gst_lds[2 * j] = p.sin_g[jj];
                gst_lds[2 * j + 1] = p.cos_g[jj];
            }
            az_wave_lds_fence();
        }
        AzRowSink sink{};
        if constexpr (SINK != AZ_SINK_SCREEN) sink = az_row_sink(prow, vrow, p.n_times);
#pragma unroll 1
        for (; window_ok && base < t_hi; base += 64) {
            const unsigned i = base + lane;
            const bool live = i < t_hi;
            double t = fma((double)i, step, t_first), dl = 0.0;
            if (DELTA) {
                dl = (double)dl_lds[i - t_lo]; // (segments are multiples of 64 points, at most AZ_DELTA_SEG)
                t += dl;
            }
            // the cold constants are re-read from LDS where they are used: an address the compiler cannot see through
            // keeps it from hoisting 16 loop-invariant loads into 32 VGPRs.  The opaque value is the LDS byte address of
            // the table itself, in a VGPR: DS addresses are VGPRs, and from one VGPR base every read is an immediate
            // offset (round 2's opaque scalar zero cost an s_add + v_mov per read: twelve VALU slots per iteration)
            const double *cold_now = az_opaque_lds(cold_lds);
            k.cold = cold_now;
            const RotCoefLds rk{cold_now + FCX_NUM};
            double r[3], v[3];
#if defined(AZ_ABLATE) && AZ_ABLATE == 2 /* tuning experiment: stores only */
            r[0] = t; r[1] = t + 1.0; r[2] = t + 2.0; v[0] = t + 3.0; v[1] = t + 4.0; v[2] = t + 5.0;
#else
            const bool bad = az_sgp4_fast_step<VEL, ECC, DELTA>(k, p.g, rk, t, fc, r, v, dl);
            if (az_any(bad && live)) break;
#endif
            if (SINK == AZ_SINK_SCREEN) {
                // fused single-target screen (src/Constellation.zig L683-756): nothing is stored; distances are
                // frame-independent, so the TEME vectors serve.  (The target track is read through scalar-friendly
                // loads: the loop has no stores for them to wait on.)
                const double *q = p.screen_target + (size_t)(live ? i : t_hi - 1) * 3;
                const double dx = q[0] - r[0], dy = q[1] - r[1], dz = q[2] - r[2];
                const double d2 = fma(dx, dx, fma(dy, dy, dz * dz));
                if (live && d2 < best_d2) { // NaN (failed target step) never wins
                    best_d2 = d2;
                    best_t = i;
                }
                continue;
            }
            if (FRAME) {
                // the Greenwich table holds AZ_FRAME_SEG points: refilled every AZ_FRAME_SEG / 64 iterations (round 5: segments of
                // ECEF / geodetic launches are as long as TEME ones -- cut to 256 points each wave paid its set-up, 185 VALU +
                // the table, every four iterations: ECEF satellite-major ran 15 % behind TEME for 5 % more arithmetic)
                if (((base - t_lo) & (AZ_FRAME_SEG - 1u)) == 0u && base != t_lo) {
                    az_wave_lds_fence();
#pragma unroll
                    for (unsigned j = lane; j < AZ_FRAME_SEG; j += 64) {
                        const unsigned jn = min(base + j, p.n_times - 1);
                        gst_lds[2 * j] = p.sin_g[jn];
                        gst_lds[2 * j + 1] = p.cos_g[jn];
                    }
                    az_wave_lds_fence();
                }
                const unsigned jj = (i - t_lo) & (AZ_FRAME_SEG - 1u);
                const double sg = gst_lds[2 * jj], cg = gst_lds[2 * jj + 1];
                az_to_ecef(r, sg, cg);
                if (VEL) az_to_ecef(v, sg, cg);
                if (FRAME == 2) az_ecef_to_geodetic(r);
            }
#if defined(AZ_ABLATE) && AZ_ABLATE == 1 /* tuning experiment: arithmetic only (every component stays live) */
            if (!(live && (r[0] + r[1] + r[2] + (VEL ? v[0] + v[1] + v[2] : 0.0)) == 1.2345e300)) continue;
#endif
            az_rows_store<VEL>(staged, base + 64 <= t_hi, live, lane, rows_stage, prow, vrow, base, r, v, sink);
        }
    }
    if (SINK == AZ_SINK_SCREEN) {
        // partial minimum of [t_lo, base) for (this form's segment, list slot); what a Newton hand-over leaves goes into the
        // redo pass's own partial slots
        az_wave_argmin(best_d2, best_t);
        if (lane == 0) {
            p.part_d2[(size_t)blockIdx.y * p.plan_stride + row + p.redo_slot0] = best_d2;
            p.part_t[(size_t)blockIdx.y * p.plan_stride + row + p.redo_slot0] = best_t;
        }
    }
    if (ECC && base < t_hi && lane == 0) {
        // the eccentric form's Newton iteration left its tiers: rest of the segment -> generic kernel (one item: list slot,
        // first grid point, end), behind the static items of the plan
        const unsigned k = atomicAdd(p.redo_count, 1u);
        p.redo_items[3 * (size_t)k + 0] = row + p.redo_slot0;
        p.redo_items[3 * (size_t)k + 1] = base;
        p.redo_items[3 * (size_t)k + 2] = t_hi;
    }
}

// TIME-MAJOR output from the lane = time arithmetic (uniform grid, TEME, fp64): a workgroup of 16 waves takes 16
// consecutive slots of the catalog-ordered near-earth list, every wave runs one satellite exactly like k_rows_fast
// (scalar operands, branch-free step, near-circular or eccentric Kepler form by the satellite's class), and the
// workgroup transposes each 64-step iteration through LDS: 64 time rows x (16 satellites x 24 bytes), flushed as
// 384-byte runs.  Positions and velocities of four time rows are 192 sixteen-byte pieces = three full store instructions
// per wave.  Two tile buffers alternate, so one barrier per iteration is enough: a wave writes buffer (k+1)&1 only after
// passing the barrier of iteration k, which every wave reaches after it has read buffer (k-1)&1 out.
// The lane = satellite kernel (k_propagate) spends 23 % more instructions per propagation with every operand in a VGPR
// or a per-lane LDS word; this one has k_rows_fast's instruction stream.  Validation failures go to the redo list like
// there (the generic kernel then writes that satellite's 24-byte pieces row by row); until it has run, the tile holds
// stale values for that satellite.
#define AZ_TILE_SATS 16
#define AZ_TILE_SEG_MAX 1024 /* longest time segment of an ECEF launch (its Greenwich-angle table sits in LDS) */
#define AZ_TILE_PITCH 49 /* doubles per staged time row: 48 + 1 (lane stride 98 dwords: ds_write_b64 conflict-free per half-wave) */
template <bool VEL, int FRAME = 0, int DELTA = 0> // FRAME: 0 TEME, 1 ECEF, 2 geodetic positions (+ ECEF velocities); DELTA: see k_rows_fast
__global__ void __launch_bounds__(1024, 1) k_tiles_fast(PropArgs p)
{
    constexpr unsigned NA = VEL ? 2u : 1u;
    constexpr bool ECEF = FRAME != 0;
    __shared__ __attribute__((aligned(16))) double cold_all[AZ_TILE_SATS * AZ_FAST_TABLE];
    __shared__ __attribute__((aligned(16))) double tile[2 * NA * 64 * AZ_TILE_PITCH];
    // ECEF output: (sin,cos) of the Greenwich angle of this segment's time steps, staged once (a load inside the loop would
    // wait for the stores in flight); geodetic output stays with the lane = satellite kernel
    __shared__ double gst[ECEF ? 2 * AZ_TILE_SEG_MAX : 2];
    const unsigned lane = threadIdx.x & 63u, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // (uniform, and known to be)
    double *cold_lds = cold_all + w * AZ_FAST_TABLE;
    // XCD-aware tile assignment (workgroup b runs on XCD b % 8; gridDim.x is a multiple of 8): every XCD takes a contiguous
    // range of tiles, so the cache lines that two neighbouring tiles share at the ends of their 384-byte runs meet in ONE L2
    const unsigned tile_id = az_xcd_row(); // (a handful of tiles: rotated over the XCDs with the time segment, like the rows)
    // A tile is 16 consecutive CATALOG rows, whatever they are: a near-earth member is computed here; a deep-space member was
    // computed by k_rows_deep into the compact scratch array just before and is copied through (a coalesced 1.5-KB read per
    // wave and iteration); a member whose initialisation failed is zeros.  So every tile is a run of consecutive rows and
    // leaves as 384-byte runs even when the catalog mixes the populations in any order (round 2 tiled the near-earth LIST:
    // with one deep-space member in ten scattered through the catalog four tiles in five fell back to 8-byte pieces).
    const unsigned s_first = tile_id * AZ_TILE_SATS;
    if (s_first >= p.n_rows) return; // padding workgroup (uniform over the workgroup)
    const unsigned n_valid = min((unsigned)AZ_TILE_SATS, p.n_rows - s_first);
    const bool have = w < n_valid;
    const unsigned s = s_first + (have ? w : 0u);
    const unsigned fl = p.flags[s];
    const unsigned rm = p.rowmap[s], kind = rm >> 30, slot = rm & 0x3fffffffu;
    // a full tile inside the row window with every member writable leaves as unconditional 16-byte pieces (384-byte runs); the
    // last tile of the catalog, a tile a row window cuts through and a tile with members the satellite mask switches off leave
    // through the same piece mapping with two write-enable bits per piece (a piece is two doubles: both writable = the same
    // 16-byte store, one = an 8-byte store, none = nothing) -- masked launches used to fall back to the lane = satellite kernel
    const unsigned t_lo = blockIdx.y * p.tile, t_hi = min(t_lo + p.tile, p.n_times);
    const bool in_window = have && s >= p.row_lo && s < p.row_hi && (p.mask == nullptr || p.mask[s] != 0);
    // (wave w's verdict for member w, collected over the workgroup below: bit j of `writable` = member j is written)
    __shared__ unsigned writable_lds;
    if (threadIdx.x == 0) writable_lds = 0u;
    __syncthreads();
    if (lane == 0 && in_window) atomicOr(&writable_lds, 1u << w);
    __syncthreads();
    const unsigned writable = writable_lds;
    if (writable == 0u) return; // nothing of this tile is written (uniform over the workgroup)
    const bool contig = writable == 0xffffu;
    bool dead = !(in_window && kind == AZ_ROW_NEAR);   // no arithmetic in this wave
    const bool copy = in_window && kind == AZ_ROW_COPY, zero = in_window && kind == AZ_ROW_ZERO;
    if (ECEF) {
        for (unsigned j = threadIdx.x; j < t_hi - t_lo; j += 1024u) {
            gst[2 * j] = p.sin_g[t_lo + j];
            gst[2 * j + 1] = p.cos_g[t_lo + j];
        }
    }
    const bool ecc = AZ_FLAG_ECLASS(fl) != 0; // wave-uniform
    const double off = az_uniform(p.offsets ? p.offsets[s] : 0.0);
    const double step = p.uniform_step, t_first = p.grid_t0 + off;
    // (fp64 deviations: the host launches the wide instantiation of this kernel for both quasi-uniform forms)
    typedef double dl_t;
    __shared__ dl_t dl_lds[DELTA ? AZ_DELTA_SEG : 1]; // the segment's deviations from the ideal grid, one table per tile
    if (DELTA) {
        const dl_t *src = p.delta64;
        if (threadIdx.x < AZ_DELTA_SEG) dl_lds[threadIdx.x] = src[t_lo + threadIdx.x]; // (zero-padded by one segment)
        if (!ECEF) __syncthreads();
    }
    if (ECEF) __syncthreads(); // (gst above, and dl_lds)
    FastKBcast k;
    FastCarry fc;
    if (!dead) {
        const double *__restrict__ rec = p.fast_rec + (size_t)s * FR_NUM;
        az_fill_fast_table(cold_lds, rec, lane);
        const double w_a = fma((double)t_lo, step, t_first), w_b = fma((double)(t_hi - 1), step, t_first);
        // (a window the plan rejected is a static item of the redo list: the generic kernel writes that satellite's pieces)
        if (!az_plan_window(p, blockIdx.y, slot, w_a, w_b, k)) dead = true;
        az_fast_rec_hot(rec, k);
        az_wave_lds_fence();
        if (DELTA && k.tc_ != 0.0 && lane == 0) cold_lds[FCX_udot] = fma(2.0 * rec[FC_nl2], k.tc_, rec[FCX_udot]); // rate of U about tc
        az_wave_lds_fence();
        az_seed_fast_rec(rec, rec[FC_nl2], fma((double)(t_lo + lane) - 64.0, step, t_first), k.tc_, fc, McLds{cold_lds + FCX_NUM + RC_NUM});
    }
    // This thread's share of a tile flush, fixed for the whole kernel: pieces q = 1024 m + tid, m < 3, of the 3,072
    // sixteen-byte pieces of a full tile (positions: 64 time rows x 24 pieces, then velocities).  Per piece the LDS byte
    // offset inside a tile buffer, the output pointer of time row 0 of the CURRENT iteration's block minus that block's
    // offset (a uniform 64-bit add per iteration turns it into the address), and the time row (partial last iteration).
    unsigned f_lds[3], f_out[3], f_row = 0; // (f_out: byte offset of the piece inside the iteration's block of 64 time rows;
                                            // the host keeps 64 rows x stride x 24 bytes below 4 GB)
#pragma unroll
    for (unsigned m = 0; m < 3; ++m) {
        const unsigned q = 1024u * m + threadIdx.x, arr = q / 1536u, pq = q - arr * 1536u, row = pq / 24u, col = pq - row * 24u;
        f_lds[m] = arr < NA ? (arr * (64u * AZ_TILE_PITCH) + row * AZ_TILE_PITCH + col * 2u) * 8u : 0xffffffffu;
#if defined(AZ_TILE_ABLATE) && AZ_TILE_ABLATE == 5
        f_out[m] = (unsigned)(((size_t)row * p.stride_sats + (s_first & 511u)) * 24u + col * 16u);
#else
        f_out[m] = (unsigned)(((size_t)row * p.stride_sats + s_first) * 24u + col * 16u);
#endif
        f_row |= (row | (arr << 7)) << (8u * m);
        // write-enable bits of the piece's two doubles (bits 24 + 2 m, 25 + 2 m): double d of the run belongs to member d / 3
        f_row |= (((writable >> ((col * 2u) / 3u)) & 1u) | (((writable >> ((col * 2u + 1u) / 3u)) & 1u) << 1)) << (24u + 2u * m);
    }
    // time rows that start on a 128-byte boundary (a padded out_stride_sats: 16 satellites = 384 bytes): every 384-byte run
    // is three whole lines and leaves with the streaming hint (0.283 -> 0.270 ms); unaligned rows share their first and
    // last line with the neighbouring tile's run, which only merge in L2 without it (0.293 against 0.355 ms with the hint)
    const bool stream_out = ((p.stride_sats * 24u) & 127u) == 0 && ((reinterpret_cast<size_t>(p.pos) | (VEL ? reinterpret_cast<size_t>(p.vel) : 0)) & 127u) == 0;
    unsigned kiter = 0;
#pragma unroll 1
    for (unsigned base = t_lo; base < t_hi; base += 64, ++kiter) {
        const unsigned i = base + lane;
        const bool live = i < t_hi;
        double r[3], v[3];
        if (!dead) {
            double t = fma((double)i, step, t_first), dl = 0.0;
            if (DELTA) {
                dl = (double)dl_lds[i - t_lo]; // (segments are multiples of 64 points, at most AZ_DELTA_SEG)
                t += dl;
            }
            const double *cold_now = az_opaque_lds(cold_lds);
            k.cold = cold_now;
            const RotCoefLds rk{cold_now + FCX_NUM};
            const bool bad = ecc ? az_sgp4_fast_step<VEL, true, DELTA>(k, p.g, rk, t, fc, r, v, dl) : az_sgp4_fast_step<VEL, false, DELTA>(k, p.g, rk, t, fc, r, v, dl);
            if (ECEF) {
                const unsigned jj = min(i, t_hi - 1) - t_lo;
                const double sg = gst[2 * jj], cg = gst[2 * jj + 1];
                az_to_ecef(r, sg, cg);
                if (VEL) az_to_ecef(v, sg, cg);
                if (FRAME == 2) az_ecef_to_geodetic(r);
            }
            if (az_any(bad && live)) {
                dead = true;
                if (lane == 0) {
                    const unsigned it = atomicAdd(p.redo_count, 1u);
                    p.redo_items[3 * (size_t)it + 0] = slot;
                    p.redo_items[3 * (size_t)it + 1] = base;
                    p.redo_items[3 * (size_t)it + 2] = t_hi;
                }
            }
        }
        if (copy) {
            const size_t at = ((size_t)slot * p.n_times + min(i, t_hi - 1)) * 3; // the frame is already the output's
            r[0] = p.tmp_pos[at]; r[1] = p.tmp_pos[at + 1]; r[2] = p.tmp_pos[at + 2];
            if (VEL) { v[0] = p.tmp_vel[at]; v[1] = p.tmp_vel[at + 1]; v[2] = p.tmp_vel[at + 2]; }
        } else if (zero) {
            r[0] = r[1] = r[2] = 0.0;
            v[0] = v[1] = v[2] = 0.0;
        }
        double *buf = tile + (kiter & 1u) * (NA * 64 * AZ_TILE_PITCH);
        if (!dead || copy || zero) {
            double *q = buf + lane * AZ_TILE_PITCH + w * 3;
            q[0] = r[0]; q[1] = r[1]; q[2] = r[2];
            if (VEL) {
                q += 64 * AZ_TILE_PITCH;
                q[0] = v[0]; q[1] = v[1]; q[2] = v[2];
            }
        }
        __syncthreads();
#if defined(AZ_TILE_ABLATE) && AZ_TILE_ABLATE == 5 /* tuning experiment: every flush lands in the same 64 time rows (L2-resident) */
        const size_t gbase = 0;
#else
        const size_t gbase = (size_t)base * p.stride_sats * 3; // uniform
#endif
        const bool full = base + 64 <= t_hi;
        if (contig) {
            const char *bufc = reinterpret_cast<const char *>(buf);
            char *pos_b = reinterpret_cast<char *>(p.pos + gbase), *vel_b = VEL ? reinterpret_cast<char *>(p.vel + gbase) : nullptr; // uniform
#pragma unroll
            for (unsigned m = 0; m < 3; ++m) {
                if (f_lds[m] == 0xffffffffu || (!full && base + ((f_row >> (8u * m)) & 63u) >= t_hi)) continue;
                const az_d2s val = *reinterpret_cast<const az_d2s *>(bufc + f_lds[m]);
                az_d2s *g = reinterpret_cast<az_d2s *>((((f_row >> (8u * m)) & 128u) ? vel_b : pos_b) + f_out[m]);
                if (stream_out) __builtin_nontemporal_store(val, g);
                else *g = val;
            }
        } else {
            // the same pieces, each with its write-enable bits
            const char *bufc = reinterpret_cast<const char *>(buf);
            char *pos_b = reinterpret_cast<char *>(p.pos + gbase), *vel_b = VEL ? reinterpret_cast<char *>(p.vel + gbase) : nullptr; // uniform
#pragma unroll
            for (unsigned m = 0; m < 3; ++m) {
                const unsigned en = (f_row >> (24u + 2u * m)) & 3u;
                if (f_lds[m] == 0xffffffffu || en == 0u || (!full && base + ((f_row >> (8u * m)) & 63u) >= t_hi)) continue;
                const az_d2s val = *reinterpret_cast<const az_d2s *>(bufc + f_lds[m]);
                double *g = reinterpret_cast<double *>((((f_row >> (8u * m)) & 128u) ? vel_b : pos_b) + f_out[m]);
                if (en == 3u) *reinterpret_cast<az_d2s *>(g) = val;
                else if (en == 1u) g[0] = val.x;
                else g[1] = val.y;
            }
        }
    }
}

#include "cols_kernel.h"

// Staging of the packed kernel, DS instructions written by hand.  Component j of a lane's two grid points sits in
// one register pair while the row wants (x y z)(x y z): ds_write2_b32 puts the two halves three floats apart without
// a register move (the compiler pairs NEIGHBOURING floats into 64-bit writes instead: twelve v_mov per iteration).
// Writes and reads are both volatile asm, so they keep their program order; the hardware executes one wave's DS
// instructions in order.
__device__ __forceinline__ unsigned az_lds_address(const void *p)
{
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p;
}
__device__ __forceinline__ void az_stage_pair3(unsigned a, const az_f2 q[3])
{
    asm volatile("ds_write2_b32 %0, %1, %2 offset1:3\n\t"
                 "ds_write2_b32 %0, %3, %4 offset0:1 offset1:4\n\t"
                 "ds_write2_b32 %0, %5, %6 offset0:2 offset1:5"
                 :
                 : "v"(a), "v"(q[0].x), "v"(q[0].y), "v"(q[1].x), "v"(q[1].y), "v"(q[2].x), "v"(q[2].y));
}
// 1,536 staged bytes at LDS address `a` (this lane's piece: a + 16 lane) -> 96 aligned 16-byte pieces at out
__device__ __forceinline__ void az_flush_pair3(unsigned a, float *out, unsigned lane)
{
    az_f4 lo, hi;
    asm volatile("ds_read_b128 %0, %2\n\t"
                 "ds_read_b128 %1, %2 offset:1024\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"(a));
    az_f4 *dst = reinterpret_cast<az_f4 *>(out);
    __builtin_nontemporal_store(lo, dst + lane);
    if (lane < 32) __builtin_nontemporal_store(hi, dst + lane + 64);
}

// k_rows_fast in PACKED fp32 arithmetic (fast_step_f32.h) for fp32 outputs: near-circular members, TEME, uniform
// grid.  A lane carries two adjacent grid points (2i, 2i+1), a wave 128 per iteration: 1,536 contiguous bytes per
// array, staged through LDS like the fp64 rows.  All constants are wave-uniform scalars (SGPRs).
template <bool VEL, bool MIXED = false, bool DELTA = false> // MIXED: az_sgp4_fast_step_f32p (O(1) quantities in fp64: the default for fp32 outputs); DELTA: see k_rows_fast
__global__ void __launch_bounds__(64, MIXED ? AZ_ROWSF32P_WAVES : AZ_ROWSF32_WAVES) k_rows_fast32(PropArgs p)
{
    const unsigned lane = threadIdx.x;
    const unsigned row = az_list_slot(p); // XCD-aware row assignment, see k_rows
    if (row >= az_list_end(p)) return;
    const unsigned s = p.list[row];
    const unsigned fl = p.flags[s];
    if ((p.mask != nullptr && p.mask[s] == 0) || s < p.row_lo || s >= p.row_hi) return;
    const unsigned t_lo = blockIdx.y * p.tile; // the host keeps p.tile a multiple of 128 for this kernel
    const unsigned t_hi = min(t_lo + p.tile, p.n_times);
    unsigned base = t_lo;
    if (AZ_FLAG_ECLASS(fl) == 0) {
        __shared__ __attribute__((aligned(16))) float rows_stage[2 * 128 * 3];
        const double off = az_uniform(p.offsets ? p.offsets[s] : 0.0);
        float *prow = reinterpret_cast<float *>(p.pos) + (size_t)s * p.n_times * 3;
        float *vrow = VEL ? reinterpret_cast<float *>(p.vel) + (size_t)s * p.n_times * 3 : nullptr;
        const bool staged = AZ_ROWS_LDS_STORE &&
                            (((reinterpret_cast<size_t>(prow) | (VEL ? reinterpret_cast<size_t>(vrow) : 0)) & 15u) == 0);
        const double step = p.uniform_step;
        const double t_first = p.grid_t0 + off;
        __shared__ float once_lds[F32_NUM];
        FastK32Bcast k;
        FastCarry32 fc;
        bool window_ok;
        const unsigned stage_w = az_lds_address(rows_stage) + lane * 24, stage_r = az_lds_address(rows_stage) + lane * 16;
        {
            const double w_a = fma((double)t_lo, step, t_first), w_b = fma((double)(t_hi - 1), step, t_first);
            FastK k0, k1;
            az_load_fast(p.el, p.n_pad, s, fl, p.inc, 0, k0);   // increments of 64 grid steps ...
            az_double_increments(k0);                           // ... of 128: one lane step
            window_ok = az_plan_window(p, blockIdx.y, row + p.redo_slot0, w_a, w_b, k0);
            if (!window_ok) return;                             // (a static item of the redo list)
            az_fast_udot(k0);                                   // (DELTA: rate of U about the window's tc)
            az_load_fast(p.el, p.n_pad, s, fl, p.inc, 1, k1);   // increments of one grid step
            {
                const double *w = p.plan_win + ((size_t)blockIdx.y * p.plan_stride + row + p.redo_slot0) * AZ_PLAN_NUM;
                k1.sdU_ = w[AZ_PLAN_s1U]; k1.cdU_ = w[AZ_PLAN_c1U];
                k1.tc_ = k0.tc_; k1.tmid_ = k0.tmid_; k1.sOc_ = k0.sOc_; k1.cOc_ = k0.cOc_;
            }
            FastK32 kk;
            az_load_fast32(k0, k1, step, p.g, kk);
#define X(n) if (lane == 0) once_lds[F32_##n] = kk.n##_;
            AZ_F32_ONCE(X)
#undef X
#define X(n) k.n##_ = az_uniform32(kk.n##_);
            AZ_F32_MANY(X)
#undef X
#define U(n) k.n = az_uniform(kk.n);
            U(sab64) U(sdU) U(cdU) U(tc) U(tmid) U(s1U) U(c1U)
            if (MIXED) { U(abase_km64) U(rv0_64) U(sOc64) U(cOc64) U(sinio64) U(cosio64) }
#undef U
            az_wave_lds_fence();
            // seed one lane step (128 grid steps) BEFORE this lane's first even grid point
            FastCarry f0;
            az_seed_fast(p.el, p.n_pad, s, fma((double)(t_lo + 2 * lane) - 128.0, step, t_first), k.tc, f0);
            az_seed_fast32(f0, k1, fc);
        }
        __shared__ __attribute__((aligned(8))) float dl_lds[DELTA ? AZ_DELTA_SEG : 2];
        if (DELTA) {
#pragma unroll
            for (unsigned j = lane; j < AZ_DELTA_SEG; j += 64) dl_lds[j] = p.delta[t_lo + j]; // (the table is zero-padded by one segment)
            az_wave_lds_fence();
        }
#pragma unroll 1
        for (; window_ok && base < t_hi; base += 128) {
            const unsigned i = base + 2 * lane;
            const bool live_a = i < t_hi, live_b = i + 1 < t_hi;
            const double t = fma((double)i, step, t_first);
            // re-read the once-per-step constants where they are used (see k_rows_fast; the opaque zero lives in a
            // VGPR here: DS addresses are VGPRs, and a scalar one is copied once per read)
            unsigned zero = 0;
            asm volatile("" : "+v"(zero));
            k.once = once_lds + zero;
            az_f2 r[3], v[3];
#if defined(AZ_ABLATE) && AZ_ABLATE == 2 /* tuning experiment: stores only */
            r[0] = az_splat2((float)t); r[1] = r[0] + 1.0f; r[2] = r[0] + 2.0f; v[0] = r[0] + 3.0f; v[1] = r[0] + 4.0f; v[2] = r[0] + 5.0f;
#else
            az_f2 dl = az_splat2(0.0f);
            if (DELTA) dl = *reinterpret_cast<const az_f2 *>(dl_lds + (i - t_lo)); // (i - t_lo is even, below AZ_DELTA_SEG)
            if (MIXED) az_sgp4_fast_step_f32p<VEL, DELTA>(k, p.g, t, fc, r, v, dl);
            else az_sgp4_fast_step_f32<VEL, DELTA>(k, p.g, t, fc, r, v, dl);
#endif
#if defined(AZ_ABLATE) && AZ_ABLATE == 1 /* tuning experiment: arithmetic only (every component stays live) */
            if (!(live_a && (r[0].x + r[1].x + r[2].x + r[0].y + r[1].y + r[2].y +
                             (VEL ? v[0].x + v[1].x + v[2].x + v[0].y + v[1].y + v[2].y : 0.0f)) == 1.2345e30f)) continue;
#endif
            if (staged && base + 128 <= t_hi) {
                az_stage_pair3(stage_w, r);
                az_flush_pair3(stage_r, prow + (size_t)base * 3, lane);
                if (VEL) {
                    az_stage_pair3(stage_w + 1536, v);
                    az_flush_pair3(stage_r + 1536, vrow + (size_t)base * 3, lane);
                }
            } else {
                if (live_a) {
                    const az_f3s a = {r[0].x, r[1].x, r[2].x};
                    __builtin_nontemporal_store(a, reinterpret_cast<az_f3s *>(prow + (size_t)i * 3));
                    if (VEL) {
                        const az_f3s b = {v[0].x, v[1].x, v[2].x};
                        __builtin_nontemporal_store(b, reinterpret_cast<az_f3s *>(vrow + (size_t)i * 3));
                    }
                }
                if (live_b) {
                    const az_f3s a = {r[0].y, r[1].y, r[2].y};
                    __builtin_nontemporal_store(a, reinterpret_cast<az_f3s *>(prow + (size_t)(i + 1) * 3));
                    if (VEL) {
                        const az_f3s b = {v[0].y, v[1].y, v[2].y};
                        __builtin_nontemporal_store(b, reinterpret_cast<az_f3s *>(vrow + (size_t)(i + 1) * 3));
                    }
                }
            }
        }
    }
}

// Near-earth rows, any grid, any eccentricity: tier votes inside the step (az_sgp4_step).  Two ways in:
//   grid (rows padded to 8, time segments)  -- the whole list, when no fast path applies (irregular grid, fused
//                                               screen, fast path switched off);
//   REDO: grid (N, 4)                       -- the items k_rows_fast rejected; every workgroup takes items
//                                               blockIdx.x, +gridDim.x, ... and a quarter (blockIdx.y) of each,
//                                               so that the few items are spread over the whole chip.
template <bool VEL, bool FRAME, int SINK, bool REDO = false>
__global__ void __launch_bounds__(64, AZ_ROWS_WAVES) k_rows(PropArgs p)
{
    typedef typename std::conditional<SINK == AZ_SINK_F32, float, double>::type out_t;
    const unsigned lane = threadIdx.x;
    // (+ 8: the epoch angles az_sgp4_step re-reads when it re-seeds / rebuilds its carried pairs, laid out like one column of
    // the element table -- from LDS, not from global memory: on an irregular grid that is EVERY step, and a vector load in
    // this loop waits, through vmcnt, for every output store issued before it)
    __shared__ __attribute__((aligned(16))) double rows_lds[AZ_ROWS_TLDS + C_NUM_MAX + 2 + 8];
    static_assert(F_nodeo < 8 && F_argpo < 8 && F_mo < 8, "the epoch angles sit in the first eight rows of the element table");
    constexpr bool redo = REDO; // a separate instantiation: the item loop costs the whole-list form ~100 spilled SGPRs
    const unsigned n_items = redo ? *p.redo_count : 1u;
#pragma unroll 1
    for (unsigned item = redo ? blockIdx.x : 0u; item < n_items; item += redo ? gridDim.x : 1u) {
    unsigned row, t_lo, t_hi;
    if (redo) {
        row = p.redo_items[3 * (size_t)item];
        const unsigned b0 = p.redo_items[3 * (size_t)item + 1], b1 = p.redo_items[3 * (size_t)item + 2];
        const unsigned quarter = (((b1 - b0 + 63) / 64 + gridDim.y - 1) / gridDim.y) * 64; // whole iterations
        t_lo = b0 + blockIdx.y * quarter;
        t_hi = min(t_lo + quarter, b1);
        if (t_lo >= b1) continue;
    } else {
        // XCD-aware row assignment: workgroup b runs on XCD b % 8 (each XCD has its own L2), so the
        // rows are dealt out in eight contiguous ranges -- every XCD then reads one eighth of the SoA
        // element table (8 satellites share each 64-B line) instead of all of it
        row = az_list_slot(p); // (gridDim.x is a multiple of 8)
        if (row >= az_list_end(p)) return;
        // blockIdx.y: time segment of p.tile grid points (a multiple of 64) -- splits long rows so that
        // the grid has enough waves to fill the chip several times over
        t_lo = blockIdx.y * p.tile;
        t_hi = min(t_lo + p.tile, p.n_times);
    }
    const unsigned s = p.list[row]; // wave-uniform
    const unsigned fl = p.flags[s];
    if (SINK != AZ_SINK_SCREEN && ((p.mask != nullptr && p.mask[s] == 0) || s < p.row_lo || s >= p.row_hi)) continue;
    const double off = az_uniform(p.offsets ? p.offsets[s] : 0.0);
    const RotK rk = az_rotk();
#if defined(AZ_ABLATE) && AZ_ABLATE == 3 /* tuning experiment: all rows alias 64 rows (L2-resident window) */
    const size_t srow = s & 63u;
#else
    const size_t srow = s;
#endif
    out_t *prow = reinterpret_cast<out_t *>(p.pos) + srow * p.n_times * 3;
    out_t *vrow = VEL ? reinterpret_cast<out_t *>(p.vel) + srow * p.n_times * 3 : nullptr;
    double best_d2 = __builtin_inf();
    unsigned best_t = 0xffffffffu;
    az_wave_lds_fence();
    Sgp4Lane e;
    typedef ColdBroadcast ColdT;
    const ColdT cold{rows_lds + AZ_ROWS_TLDS};
    az_load_sgp4(p.el, p.n_pad, s, fl, e, cold);
    e.mdot = az_uniform(e.mdot); e.argpdot = az_uniform(e.argpdot); e.nodedot = az_uniform(e.nodedot);
    e.xnodcf = az_uniform(e.xnodcf); e.aycof = az_uniform(e.aycof); e.xlcof = az_uniform(e.xlcof);
    e.sinio = az_uniform(e.sinio); e.cosio = az_uniform(e.cosio); e.k_mrt = az_uniform(e.k_mrt);
    e.k_c2u = az_uniform(e.k_c2u); e.k_su = az_uniform(e.k_su); e.k_node = az_uniform(e.k_node);
    e.k_inc = az_uniform(e.k_inc); e.x1mth2 = az_uniform(e.x1mth2); e.k_rv = az_uniform(e.k_rv);
    double *seed_lds = rows_lds + AZ_ROWS_TLDS + C_NUM_MAX + 2;
    if (lane < 8) seed_lds[lane] = p.el[(size_t)lane * p.n_pad + s];
    az_wave_lds_fence();
    Sgp4Carry c;
    c.t_prev = 0.0;
    c.sW = c.sO = c.sA = 0.0;
    c.cW = c.cO = c.cA = 1.0;
    c.dt_c = -1.0e300;
    c.sdA = c.pW = c.qW = 0.0;
    c.cdA = 1.0;
    // Time values go through LDS (AZ_ROWS_TLDS grid points per refill): the loop then holds no
    // vector-memory LOAD at all.  vmcnt counts loads and stores in issue order, so a load's
    // s_waitcnt also waits for every output store issued before it -- one load per iteration
    // serialises the arithmetic against the write stream; ds_read waits on lgkmcnt only.
#pragma unroll 1
    for (unsigned base = t_lo; base < t_hi; base += 64) {
        const unsigned i = base + lane;
        const bool live = i < t_hi;
        const unsigned kk = (base - t_lo) & (AZ_ROWS_TLDS - 1u);
        if (kk == 0) {
            az_wave_lds_fence();
#pragma unroll
            for (unsigned j = 0; j < AZ_ROWS_TLDS; j += 64) rows_lds[j + lane] = p.times[min(i + j, t_hi - 1)];
            az_wave_lds_fence();
        }
        const double t = rows_lds[kk + lane] + off;
        double r[3], v[3];
        // full re-seed of the carried pairs at the start and every 64 iterations (4,096 grid points)
        const bool first = ((base - t_lo) & (64u * 64u - 1u)) == 0;
#if defined(AZ_ABLATE) && AZ_ABLATE == 2 /* tuning experiment: stores only */
        r[0] = t; r[1] = t + 1.0; r[2] = t + 2.0; v[0] = t + 3.0; v[1] = t + 4.0; v[2] = t + 5.0;
        (void)first;
#else
        az_sgp4_step<VEL, ColdT, true>(e, cold, seed_lds, 1, 0, p.g, rk, t, first, c, r, v, p.grid_exact_uniform != 0);
#endif
        if (SINK == AZ_SINK_SCREEN) {
            // distances are frame-independent (ECEF is a rotation of TEME about z), so the screen
            // works on the TEME vectors directly
            const double *q = p.screen_target + (size_t)(live ? i : t_hi - 1) * 3;
            const double dx = q[0] - r[0], dy = q[1] - r[1], dz = q[2] - r[2];
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (live && d2 < best_d2) { // NaN (failed target step) never wins
                best_d2 = d2;
                best_t = i;
            }
            continue;
        }
        if (FRAME) az_epilogue(r, v, p.mode, VEL, p.sin_g, p.cos_g, live ? i : t_hi - 1);
#if defined(AZ_ABLATE) && AZ_ABLATE == 1 /* tuning experiment: arithmetic only (every component stays live) */
        if (!(live && (r[0] + r[1] + r[2] + (VEL ? v[0] + v[1] + v[2] : 0.0)) == 1.2345e300)) continue;
#endif
        // direct 24-byte (12-byte) pieces per lane, contiguous across the wave: this kernel is paced by its
        // arithmetic, and for it the LDS transpose of k_rows_fast measures slower (0.269 vs 0.262 ms)
        if (live && redo && p.tm_rows) {
            // redo pass behind the time-major tile kernel (k_tiles_fast): this lane's 24 bytes of time row i
            const size_t ob = ((size_t)i * p.stride_sats + s) * 3;
            az_put3(reinterpret_cast<out_t *>(p.pos) + ob, r);
            if (VEL) az_put3(reinterpret_cast<out_t *>(p.vel) + ob, v);
        } else if (live) {
            az_put3_stream(prow + (size_t)i * 3, r);
            if (VEL) az_put3_stream(vrow + (size_t)i * 3, v);
        }
    }
    if (SINK == AZ_SINK_SCREEN) {
        az_wave_argmin(best_d2, best_t);
        if (lane == 0) {
            // whole-list form: one partial per (time segment, list slot).  Redo pass behind the fast screen kernels: four
            // partials (blockIdx.y = quarter of the item) per (segment of the member's own form, list slot), stored behind
            // the fast kernels' screen_nseg rows (p.tile / p.tile_e: the two forms' segment lengths)
            size_t part = blockIdx.y;
            if (redo) part = p.screen_nseg + 4u * (p.redo_items[3 * (size_t)item + 1] / (row < p.n_circ ? p.tile : p.tile_e)) + blockIdx.y;
            p.part_d2[part * p.n_list + row] = best_d2;
            p.part_t[part * p.n_list + row] = best_t;
        }
    }
    } // items
    // two item counters used alternately: this launch consumed redo_count, the NEXT launch's k_rows_fast (ordered
    // after this kernel on the stream) appends through redo_next, which is re-armed here.  (An arrival counter
    // with "last one out resets" serialises one atomic per workgroup on a single address: measured 100 us.)
    if (redo && lane == 0 && blockIdx.x == 0 && blockIdx.y == 0) *p.redo_next = *p.redo_static; // dynamic items go behind the plan's static ones
}

// Deep-space rows, satellite-major output (and the fused screen): ONE WAVE PER SATELLITE, lane = time,
// like k_rows.  All 61 per-satellite constants are wave-uniform and live in LDS (broadcast reads), so
// the kernel needs a third of the registers of the lane = satellite form (which holds 24 + carried
// state per lane and runs at 1-2 waves/SIMD).  Each 64-point iteration starts from the resonance
// state that k_deep_seed(nearest = 1) prepared for that chunk; a lane then advances at most a step
// or two of 720 minutes on its own.
#ifndef AZ_ROWSD_WAVES
#define AZ_ROWSD_WAVES 3
#endif
#define AZ_DEEP_SEED_MAX 64 /* chunk seeds of one time segment staged in LDS: segments of at most 64 x 64 grid points */
template <bool VEL, bool FRAME, int SINK>
__global__ void __launch_bounds__(64, AZ_ROWSD_WAVES) k_rows_deep(PropArgs p)
{
    typedef typename std::conditional<SINK == AZ_SINK_F32, float, double>::type out_t;
    constexpr unsigned TL = 512;
    const unsigned lane = threadIdx.x;
    const unsigned row = az_list_slot(p); // XCD-aware row assignment, see k_rows
    if (row >= az_list_end(p)) return;
    const unsigned s = p.list[row];
    const unsigned fl = p.flags[s];
    if (SINK != AZ_SINK_SCREEN && ((p.mask != nullptr && p.mask[s] == 0) || s < p.row_lo || s >= p.row_hi)) return;
    const unsigned t_lo = blockIdx.y * p.tile; // p.tile is a multiple of 64 (and at most 64 * AZ_DEEP_SEED_MAX: host)
    const unsigned t_hi = min(t_lo + p.tile, p.n_times);
    __shared__ double lds[TL + H_NUM + D_NUM + 3 * AZ_DEEP_SEED_MAX + MC_NUM];
    double *seed_lds = lds + TL + H_NUM + D_NUM;
    int irez;
    {
        Sdp4Bcast e0{lds + TL, 0};
        az_load_sdp4(p.el, p.n_pad, s, fl, e0, ColdBroadcastLit{lds + TL + H_NUM});
        irez = e0.irez;
        if (lane < MC_NUM) lds[TL + H_NUM + D_NUM + 3 * AZ_DEEP_SEED_MAX + lane] = az_mc_table[lane];
    }
    // The loop below holds NO vector-memory load: vmcnt counts loads and stores in issue order, so a load's s_waitcnt also
    // waits for every output store issued before it -- one load per iteration serialises the arithmetic against the write
    // stream (round 2's loop fetched the chunk seeds and, on spills, scratch words every iteration).  The resonance
    // states of all chunks of this segment are staged in LDS once; time is arithmetic on uniform grids and an LDS table
    // otherwise.
    const bool res = irez != 0; // wave-uniform
    const unsigned n_chunks = (t_hi - t_lo + 63u) >> 6;
    if (res && p.seeds) {
        for (unsigned j = lane; j < 3u * n_chunks; j += 64u) {
            const unsigned c = j / 3u, f = j - 3u * c;
            seed_lds[j] = p.seeds[((size_t)((t_lo >> 6) + c) * 3 + f) * p.n_list + row];
        }
    }
    const double off = az_uniform(p.offsets ? p.offsets[s] : 0.0);
    const bool uniform = p.uniform_step != 0.0 && p.delta == nullptr && p.delta64 == nullptr; // (a quasi-uniform grid: the time table, like any other grid)
    const double step = p.uniform_step, t_first = uniform ? p.times[0] + off : 0.0;
    const RotK rk = az_rotk();
    const size_t out_row = p.rows_compact ? row : s;
    out_t *prow = reinterpret_cast<out_t *>(p.pos) + out_row * p.n_times * 3;
    out_t *vrow = VEL ? reinterpret_cast<out_t *>(p.vel) + out_row * p.n_times * 3 : nullptr;
    double best_d2 = __builtin_inf();
    unsigned best_t = 0xffffffffu;
    Sdp4Acc acc; // resonance accelerations of this lane's current integrator state (no state yet)
    acc.atime = __builtin_nan("");
    acc.xndt = acc.xnddt = acc.xldot = 0.0;
    az_wave_lds_fence();
#pragma unroll 1
    for (unsigned base = t_lo; base < t_hi; base += 64) {
        const unsigned i = base + lane;
        const bool live = i < t_hi;
        double t;
        if (uniform) {
            t = fma((double)i, step, t_first);
        } else {
            const unsigned kk = (base - t_lo) & (TL - 1u);
            if (kk == 0) {
                az_wave_lds_fence();
#pragma unroll
                for (unsigned j = 0; j < TL; j += 64) lds[j + lane] = p.times[min(i + j, t_hi - 1)];
                az_wave_lds_fence();
            }
            t = lds[kk + lane] + off;
        }
        // the constants are re-read from LDS where they are used: an address the compiler cannot see through keeps it
        // from hoisting 61 loop-invariant loads into 122 VGPRs (a VGPR-held LDS address: reads are immediate offsets)
        const double *lds_now = az_opaque_lds(lds + TL);
        const Sdp4Bcast e{const_cast<double *>(lds_now), irez};
        const ColdBroadcast cold{const_cast<double *>(lds_now) + H_NUM, lds_now + H_NUM + D_NUM + 3 * AZ_DEEP_SEED_MAX};
        Sdp4Carry cy;
        if (res && p.seeds) {
            const double *sd = lds_now + H_NUM + D_NUM + 3u * ((base - t_lo) >> 6); // wave-uniform address
            cy.atime = sd[0];
            cy.xli = sd[1];
            cy.xni = sd[2];
        } else {
            cy.atime = 0.0;
            cy.xli = e(H_xlamo);
            cy.xni = e(H_no_unkozai);
        }
        double r[3], v[3];
        if (res) az_resonance_cached(e, cold, t, cy, acc);
        int rc = az_sdp4_step_pre<VEL>(e, cold, p.g, rk, t, cy, r, v, acc);
        if (SINK == AZ_SINK_SCREEN) {
            const double *q = p.screen_target + (size_t)(live ? i : t_hi - 1) * 3;
            const double dx = q[0] - r[0], dy = q[1] - r[1], dz = q[2] - r[2];
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (live && rc == 0 && d2 < best_d2) {
                best_d2 = d2;
                best_t = i;
            }
            continue;
        }
        if (FRAME) az_epilogue(r, v, p.mode, VEL, p.sin_g, p.cos_g, live ? i : t_hi - 1);
        if (rc != 0) {
            r[0] = r[1] = r[2] = 0.0;
            v[0] = v[1] = v[2] = 0.0;
            if (p.err && live) p.err[(size_t)s * p.n_times + i] = (unsigned char)rc;
        }
        if (live) {
            az_put3_stream(prow + (size_t)i * 3, r);
            if (VEL) az_put3_stream(vrow + (size_t)i * 3, v);
        }
    }
    if (SINK == AZ_SINK_SCREEN) {
        az_wave_argmin(best_d2, best_t);
        if (lane == 0) {
            p.part_d2[(size_t)blockIdx.y * p.n_list + row] = best_d2;
            p.part_t[(size_t)blockIdx.y * p.n_list + row] = best_t;
        }
    }
}

// Deep-space rows, TIME-major output: k_rows_deep writes its rows satellite-major into a compact scratch array (row =
// list slot; full-line streaming stores, like the satellite-major layout) and this pure-memory kernel turns 32 list slots
// x 64 time steps at a time into time-major rows through LDS.  When the 32 slots are consecutive catalog rows (catalogs
// grouped by regime: the reference's own Constellation puts its deep-space members last, src/Constellation.zig L101-200)
// every time step leaves as ONE 768-byte run; otherwise as per-satellite 24-byte pieces.  Round 2 stored those pieces
// straight from the arithmetic kernel for every deep-space row: partial-line writes n_sats x 24 bytes apart, which also
// halved the rate of the near-earth tile kernel running beside it (0.60 instead of 0.27 ms).
#define AZ_TR_ROWS 32
template <class T>
__global__ void __launch_bounds__(192) k_deep_transpose(const T *tmp_pos, const T *tmp_vel, T *out_pos, T *out_vel, const unsigned *list,
                                                        unsigned n_list, unsigned n_times, size_t stride_sats, const unsigned char *mask,
                                                        unsigned row_lo, unsigned row_hi)
{
    constexpr unsigned PITCH = AZ_TR_ROWS * 3 + 1; // odd word pitch: conflict-free column writes
    __shared__ T tile[64 * PITCH];
    const unsigned slot0 = blockIdx.x * AZ_TR_ROWS, t0 = blockIdx.y * 64u;
    const unsigned nr = min((unsigned)AZ_TR_ROWS, n_list - slot0), nt = min(64u, n_times - t0);
    const unsigned s_first = list[slot0];
    const bool run = list[slot0 + nr - 1] - s_first == nr - 1 && s_first >= row_lo && s_first + nr <= row_hi && mask == nullptr; // uniform
    const int n_arr = tmp_vel ? 2 : 1;
    // gather: thread = one of the 192 words of a slot's 64 x 3 block, the slots at a uniform stride -- 32 independent loads
    // in flight per thread (a loop that stores each value to LDS as it arrives waits for memory 32 times over)
    const unsigned col = threadIdx.x, tt_g = col / 3u, cp_g = col - tt_g * 3u;
    const size_t row_words = (size_t)n_times * 3;
    // scatter: thread = word j of the run of time step tt_s + 2 m
    const unsigned tt_s = threadIdx.x / 96u, j_s = threadIdx.x - tt_s * 96u, rw_s = j_s / 3u, cp_s = j_s - rw_s * 3u;
    const unsigned sj = list[min(slot0 + rw_s, n_list - 1)];
    const bool keep = rw_s < nr && (run || (sj >= row_lo && sj < row_hi && (mask == nullptr || mask[sj] != 0)));
    const size_t dcol = run ? (size_t)s_first * 3 + j_s : (size_t)sj * 3 + cp_s;
    for (int arr = 0; arr < n_arr; ++arr) {
        const T *src = (arr ? tmp_vel : tmp_pos) + ((size_t)slot0 * n_times + t0) * 3 + col;
        T *dst = (arr ? out_vel : out_pos) + dcol;
        T val[AZ_TR_ROWS];
#pragma unroll
        for (unsigned k = 0; k < AZ_TR_ROWS; ++k)
            val[k] = (k < nr && tt_g < nt) ? __builtin_nontemporal_load(src + k * row_words) : (T)0;
        if (arr) __syncthreads(); // the previous array has left the tile
#pragma unroll
        for (unsigned k = 0; k < AZ_TR_ROWS; ++k) tile[tt_g * PITCH + k * 3u + cp_g] = val[k];
        __syncthreads();
        if (keep) {
#pragma unroll 8
            for (unsigned m = 0; m < 32u; ++m) {
                const unsigned tt = tt_s + 2u * m;
                if (tt < nt) dst[(size_t)(t0 + tt) * stride_sats * 3] = tile[tt * PITCH + j_s];
            }
        }
    }
}

// screen: combine the partial minima of a launch list (k_screen_finalize2 below) and apply the reference's conventions: start
// from threshold^2 / index 0, strict '<', the target itself keeps the threshold, distances leave as sqrt
// (src/Constellation.zig L700-703, L733, L744-747, L752-754).  Smallest distance, earliest grid point among equals (the parts are
// not always in time order: the redo pass's partials follow the fast kernels'); a part that saw nothing carries +inf / 0xffffffff.

// Round 6: everything the screening kernels need before they start, in ONE launch (it was three: k_screen_fill, the target's
// track through k_one_satellite, k_screen_parts_clear -- each a dependent launch of a few microseconds in front of 0.14 ms of
// arithmetic).  Workgroups [0, nb_track): the target's track, one grid point per lane (the per-point step of k_one_satellite;
// NaN where the target's propagation fails: such a point never compares closer than the threshold); the next nb_clear
// workgroups reset the partial minima the generic pass only partly overwrites; the rest write the start values
// (threshold, index 0) of every catalog row.  track == nullptr: the caller supplies the track (azh_screen_track_device).
__global__ void __launch_bounds__(64) k_screen_prep(const double *__restrict__ el, const unsigned *__restrict__ flags, size_t n_pad, unsigned target,
                                                    const double *__restrict__ tsince, unsigned n_times, const double *__restrict__ offsets, AzGrav g,
                                                    double *track, unsigned nb_track, size_t n_clear, double *part_d2, unsigned *part_t,
                                                    unsigned nb_clear, unsigned n_rows, double threshold, double *out_d, unsigned *out_t)
{
    if (blockIdx.x < nb_track) {
        const unsigned i = blockIdx.x * 64 + threadIdx.x;
        const double t = tsince[i < n_times ? i : n_times - 1] + (offsets ? offsets[target] : 0.0);
        const unsigned fl = flags[target];
        double r[3], v[3];
        int rc = AZ_FLAG_ERR(fl);
        if (rc == 0) {
            if (fl & AZ_FLAG_DEEP) {
                Sdp4Lane e;
                Sdp4Carry c;
                __shared__ double cold_deep[D_NUM * 64];
                double *cold = cold_deep + threadIdx.x;
                az_load_sdp4(el, n_pad, target, fl, e, ColdLds{cold});
                c.atime = 0.0;
                c.xli = e(H_xlamo);
                c.xni = e(H_no_unkozai);
                rc = az_sdp4_step<true>(e, ColdLds{cold}, g, az_rotk(), t, c, r, v);
            } else {
                Sgp4Lane e;
                Sgp4Carry c;
                ColdRegs cold;
                az_load_sgp4(el, n_pad, target, fl, e, cold);
                c.t_prev = 0.0;
                az_sgp4_step<true>(e, cold, el, n_pad, target, g, az_rotk(), t, true, c, r, v);
            }
        }
        if (rc != 0) r[0] = r[1] = r[2] = __builtin_nan("");
        if (i < n_times) { track[(size_t)i * 3] = r[0]; track[(size_t)i * 3 + 1] = r[1]; track[(size_t)i * 3 + 2] = r[2]; }
        return;
    }
    if (blockIdx.x < nb_track + nb_clear) {
        // 4 elements per lane: 256 per workgroup
        const size_t base = (size_t)(blockIdx.x - nb_track) * 256 + threadIdx.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t i = base + 64 * k;
            if (i < n_clear) { part_d2[i] = __builtin_inf(); part_t[i] = 0xffffffffu; }
        }
        return;
    }
    const unsigned base = (blockIdx.x - nb_track - nb_clear) * 256 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned i = base + 64 * k;
        if (i < n_rows) { out_d[i] = threshold; out_t[i] = 0; }
    }
}

// ... and the finalisation of both launch lists (near-earth, deep-space) in one launch
__global__ void __launch_bounds__(256) k_screen_finalize2(const double *pd_a, const unsigned *pt_a, unsigned parts_a, const unsigned *list_a, unsigned n_a,
                                                          const double *pd_b, const unsigned *pt_b, unsigned parts_b, const unsigned *list_b, unsigned n_b,
                                                          double threshold_sq, unsigned target, double *out_d, unsigned *out_t)
{
    unsigned li = blockIdx.x * blockDim.x + threadIdx.x;
    const bool second = li >= n_a;
    if (second) li -= n_a;
    const unsigned n_list = second ? n_b : n_a, n_parts = second ? parts_b : parts_a;
    if (li >= n_list) return;
    const double *part_d2 = second ? pd_b : pd_a;
    const unsigned *part_t = second ? pt_b : pt_a;
    const unsigned s = (second ? list_b : list_a)[li];
    double best = threshold_sq;
    unsigned bt = 0;
    if (s != target) {
        for (unsigned k = 0; k < n_parts; ++k) {
            const double d = part_d2[(size_t)k * n_list + li];
            const unsigned t = part_t[(size_t)k * n_list + li];
            if (d < best || (d == best && t < bt && d < threshold_sq)) {
                best = d;
                bt = t;
            }
        }
    }
    out_d[s] = sqrt(best);
    out_t[s] = bt;
}

// ------------------------------------------------------------------------------------------------
// All-vs-all coarse conjunction screen on device-resident positions (coarseScreen,
// bindings/python/src/conjunction.zig L11-150): per time step a cell list with cell edge =
// threshold; every satellite probes the 27 cells around its own and reports partners with a larger
// index that are closer than the threshold.  The reference builds one chained hash table per time
// step on one CPU thread; here a chunk of time steps is processed at once, one table per step:
//   k_cells_build : lane = (step, satellite): cell coordinates -> bucket -> push on the bucket's
//                   chain (atomicExch on the head; chain order is irrelevant to the result set)
//   k_cells_probe : lane = (step, satellite): walk the 27 chains, exact cell match (hash collisions),
//                   distance test, append (s, other, t) through one atomic counter
// Positions are read as stored (either layout); with the time-major layout a wave's loads coalesce.
struct CellArgs {
    const double *pos;
    unsigned n_sats, n_times;
    int layout;          // AZ_LAYOUT_*: 0 sat-major (s*n_times+t)*3, 1 time-major (t*stride+s)*3
    size_t stride_sats;
    const unsigned char *valid; // per satellite, may be null
    double inv_cell, thr2;
    unsigned t0, n_steps; // chunk of time steps
    unsigned table_mask;  // buckets per step - 1
    unsigned *head;       // [n_steps][table]
    unsigned *occupied;   // [n_steps][table / 32]: one bit per bucket that holds at least one satellite (round 6)
    unsigned *next;       // [n_steps][n_sats]
    unsigned *out_pairs;  // [max][2]
    unsigned *out_t;      // [max]
    unsigned long long *count;
    unsigned long long max_results;
    int skip_zero; // rows the propagator zero-filled (failed init / failed step) are not satellites
};

__device__ __forceinline__ unsigned az_cell_hash(int cx, int cy, int cz)
{
    // conjunction.zig L152-160 (Knuth multiplicative hash, wrapping arithmetic)
    unsigned h = (unsigned)cx;
    h *= 2654435761u;
    h ^= (unsigned)cy;
    h *= 2654435761u;
    h ^= (unsigned)cz;
    h *= 2654435761u;
    return h;
}

__device__ __forceinline__ const double *az_cell_pos(const CellArgs &a, unsigned s, unsigned t)
{
    return a.pos + (a.layout == 0 ? ((size_t)s * a.n_times + t) * 3 : ((size_t)t * a.stride_sats + s) * 3);
}

__device__ __forceinline__ bool az_cell_of(const CellArgs &a, unsigned s, unsigned t, double r[3], int c[3])
{
    if (a.valid && a.valid[s] == 0) return false;
    const double *q = az_cell_pos(a, s, t);
    r[0] = q[0];
    if (!(fabs(r[0]) <= 1.79769313486231570815e308)) return false; // isFinite(x), conjunction.zig L63
    r[1] = q[1];
    r[2] = q[2];
    if (a.skip_zero && r[0] == 0.0 && r[1] == 0.0 && r[2] == 0.0) return false;
    c[0] = (int)floor(r[0] * a.inv_cell);
    c[1] = (int)floor(r[1] * a.inv_cell);
    c[2] = (int)floor(r[2] * a.inv_cell);
    return true;
}

// Workgroup -> (satellite group, step) with every STEP on ONE XCD (round 6): workgroups are dealt to the XCDs round-robin by their
// flat index, so with a (groups, steps) grid every XCD touched every step's bucket table -- 60 tables x 256 KB per chunk -- and
// none of them stayed in its 4-MB L2: k_cells_probe fetched 2.45 GB per dispatch for 38 MB of distinct data (profiles/
// r06_screen_all.txt, first block).  Flat index f: XCD = f mod 8 takes steps k = 8 m + XCD, one after the other, so the tables
// an XCD's L2 holds at any time are the one or two steps it is working on.  Grid: 8 * groups * ceil(n_steps / 8) workgroups.
__device__ __forceinline__ bool az_cells_slot(const CellArgs &a, unsigned &s, unsigned &k)
{
    const unsigned groups = (a.n_sats + 255u) / 256u;
    const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    k = (j / groups) * 8u + xcd;
    s = (j % groups) * 256u + threadIdx.x;
    return k < a.n_steps && s < a.n_sats;
}

__global__ void __launch_bounds__(256) k_cells_build(CellArgs a)
{
    unsigned s, k;
    if (!az_cells_slot(a, s, k)) return;
    double r[3];
    int c[3];
    if (!az_cell_of(a, s, a.t0 + k, r, c)) return;
    const unsigned h = az_cell_hash(c[0], c[1], c[2]) & a.table_mask;
    a.next[(size_t)k * a.n_sats + s] = atomicExch(&a.head[(size_t)k * (a.table_mask + 1u) + h], s);
}

// the occupancy bitmap of a chunk's bucket tables: one lane per bucket, one ballot per wave, two words per wave (an atomicOr per
// satellite in k_cells_build doubled that kernel: 6.6 inserts per 32-bucket word)
__global__ void __launch_bounds__(256) k_cells_bits(const unsigned *__restrict__ head, unsigned *__restrict__ occupied, size_t n_buckets)
{
    const size_t b = (size_t)blockIdx.x * 256 + threadIdx.x; // (n_buckets is a multiple of 2^16)
    const unsigned long long m = __builtin_amdgcn_ballot_w64(b < n_buckets && head[b] != 0xffffffffu);
    if ((threadIdx.x & 63u) == 0 && b < n_buckets) {
        occupied[b >> 5] = (unsigned)m;
        occupied[(b >> 5) + 1] = (unsigned)(m >> 32);
    }
}

// AZ_CELL_BITMAP_WORDS: the largest occupancy bitmap a workgroup stages in LDS (2^18 buckets = 32 KB); larger tables (catalogs
// beyond 131,072 satellites) read the bitmap words from global memory instead
#define AZ_CELL_BITMAP_WORDS 8192u
__global__ void __launch_bounds__(256) k_cells_probe(CellArgs a)
{
    // Round 6: the step's occupancy bitmap (one bit per bucket, 8 KB for 2^16 buckets) is staged in LDS first.  A satellite probes
    // 27 buckets and in a sparse shell four in five of them are empty: the bit test answers those from LDS, and only occupied
    // buckets cost a dependent random read of the head table (profiles/r06_experiments.txt D).
    extern __shared__ unsigned bits[]; // (dynamic: table / 32 words when the table is staged, nothing otherwise)
    const unsigned words = (a.table_mask + 1u) >> 5;
    const bool staged = words <= AZ_CELL_BITMAP_WORDS;
    unsigned s, k;
    {
        // (every thread of the workgroup takes part in the staging, also those beyond the catalog)
        const unsigned groups = (a.n_sats + 255u) / 256u;
        const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
        k = (j / groups) * 8u + xcd;
        s = (j % groups) * 256u + threadIdx.x;
        if (k >= a.n_steps) return; // (uniform per workgroup)
        if (staged) {
            const unsigned *src = a.occupied + (size_t)k * words;
            for (unsigned w = threadIdx.x; w < words; w += 256u) bits[w] = src[w];
            __syncthreads();
        }
    }
    if (s >= a.n_sats) return;
    const unsigned t = a.t0 + k;
    double r[3];
    int c[3];
    if (!az_cell_of(a, s, t, r, c)) return;
    const unsigned *head = a.head + (size_t)k * (a.table_mask + 1u);
    const unsigned *next = a.next + (size_t)k * a.n_sats;
    const unsigned *occ = a.occupied + (size_t)k * words;
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dz = -1; dz <= 1; ++dz) {
                const int nx = c[0] + dx, ny = c[1] + dy, nz = c[2] + dz;
                const unsigned h = az_cell_hash(nx, ny, nz) & a.table_mask;
                const unsigned word = staged ? bits[h >> 5] : occ[h >> 5];
                if (!((word >> (h & 31u)) & 1u)) continue; // an empty bucket
                unsigned idx = head[h];
                while (idx != 0xffffffffu) {
                    const unsigned other = idx;
                    idx = next[other];
                    if (other <= s) continue;
                    double q[3];
                    int oc[3];
                    az_cell_of(a, other, t, q, oc); // members of a chain are valid by construction
                    if (oc[0] != nx || oc[1] != ny || oc[2] != nz) continue; // another cell in this bucket
                    const double ex = r[0] - q[0], ey = r[1] - q[1], ez = r[2] - q[2];
                    if (ex * ex + ey * ey + ez * ez < a.thr2) {
                        const unsigned long long w = atomicAdd(a.count, 1ull);
                        if (w < a.max_results) {
                            a.out_pairs[2 * w] = s;
                            a.out_pairs[2 * w + 1] = other;
                            a.out_t[w] = t;
                        }
                    }
                }
            }
}

// rows of satellites whose init failed: zero state + the init error code at every time
__global__ void k_fill_bad(const unsigned *list, unsigned n_list, const unsigned *flags, unsigned n_times,
                           double *pos, double *vel, unsigned char *err, const unsigned char *mask, int layout,
                           size_t stride_sats, int f32, unsigned row_lo, unsigned row_hi)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x; // time
    const unsigned li = blockIdx.y;
    if (i >= n_times || li >= n_list) return;
    const unsigned s = list[li];
    if ((mask && !mask[s]) || s < row_lo || s >= row_hi) return;
    const size_t ob = (layout == 0) ? ((size_t)s * n_times + i) * 3 : ((size_t)i * stride_sats + s) * 3;
    if (f32) {
        float *p32 = reinterpret_cast<float *>(pos), *v32 = reinterpret_cast<float *>(vel);
        p32[ob] = p32[ob + 1] = p32[ob + 2] = 0.0f;
        if (vel) v32[ob] = v32[ob + 1] = v32[ob + 2] = 0.0f;
    } else {
        pos[ob] = pos[ob + 1] = pos[ob + 2] = 0.0;
        if (vel) vel[ob] = vel[ob + 1] = vel[ob + 2] = 0.0;
    }
    if (err) err[(size_t)s * n_times + i] = (unsigned char)AZ_FLAG_ERR(flags[s]);
}

// GMST table (WorldCoordinateSystem.julianToGmst, src/WorldCoordinateSystem.zig L146-154;
// Constellation.zig L573-581)
__global__ void k_gmst(const double *times, unsigned n, double reference_jd, double *sin_g, double *cos_g)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = (reference_jd + times[i] / 1440.0) - 2451545.0;
    const double tc = d / 36525.0;
    double gm = 280.46061837 + 360.98564736629 * d + 0.000387933 * tc * tc - tc * tc * tc / 38710000.0;
    gm = fmod(gm, 360.0);
    if (gm < 0) gm += 360.0;
    gm *= AZ_PI / 180.0;
    sin_g[i] = sin(gm);
    cos_g[i] = cos(gm);
}

// uniform time grid: (sin,cos) of the per-step increments of the mean anomaly and the argument of perigee, per satellite,
// for the two lane mappings (fast_step.h).  One lane per satellite, once per staged grid.
__global__ void __launch_bounds__(256) k_prep_inc(const double *el, size_t n, size_t n_pad, double step, double *inc)
{
    const size_t s = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const double rate[2] = {el[(size_t)F_mdot * n_pad + s], el[(size_t)F_argpdot * n_pad + s]};
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const double dt = which == 0 ? 64.0 * step : step;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            double sn, cs;
            az_sincos(rate[a] * dt, sn, cs);
            inc[(size_t)(AZ_INC_NUM * which + 2 * a) * n_pad + s] = sn;
            inc[(size_t)(AZ_INC_NUM * which + 2 * a + 1) * n_pad + s] = cs;
        }
    }
}

// ... and the record of folded constants the lane = time fast kernels read (fast_step.h, FastRec); after k_prep_inc on the
// same stream.  One lane per satellite, once per staged grid.
__global__ void __launch_bounds__(256) k_prep_rec(const double *el, const unsigned *flags, size_t n, size_t n_pad, const double *inc, double *rec)
{
    const size_t s = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    FastK k;
    az_load_fast(el, n_pad, s, flags[s], inc, 0, k);
    az_fast_rec_store(el, n_pad, s, k, rec + s * FR_NUM);
}

// scalar helpers of the c_api (coords_*, src/c_api/coordinates.zig; orbital_*, src/c_api/orbital_mechanics.zig over
// src/calculations.zig L83-125): op 0 julianToGmst(in[0]), op 1 eciToEcefGmst(in[0..3), gmst = in[3]), op 2
// ecefToGeodeticDeg(in[0..3)) [lat deg, lon deg, alt km], op 3 velocity / period / escape velocity, op 4 Hohmann transfer
__global__ void k_coords(int op, const double *in, double *out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (op == 0) {
        const double d = in[0] - 2451545.0;
        const double tc = d / 36525.0;
        double gm = 280.46061837 + 360.98564736629 * d + 0.000387933 * tc * tc - tc * tc * tc / 38710000.0;
        gm = fmod(gm, 360.0);
        if (gm < 0) gm += 360.0;
        out[0] = gm * (AZ_PI / 180.0);
        out[1] = out[2] = 0.0;
    } else if (op == 1) {
        double r[3] = {in[0], in[1], in[2]};
        az_to_ecef(r, sin(in[3]), cos(in[3]));
        out[0] = r[0]; out[1] = r[1]; out[2] = r[2];
    } else if (op == 2) {
        double r[3] = {in[0], in[1], in[2]};
        az_ecef_to_geodetic(r);
        out[0] = r[0] * (180.0 / AZ_PI); out[1] = r[1] * (180.0 / AZ_PI); out[2] = r[2];
    } else if (op == 3) {
        // orbital_velocity (vis-viva; sma = 0: circular), orbital_period, orbital_escape_velocity: in = mu, radius, sma
        const double mu = in[0], r = in[1], a = in[2];
        out[0] = sqrt(a != 0.0 ? mu * (2.0 / r - 1.0 / a) : mu / r);
        out[1] = 2.0 * AZ_PI * sqrt(a * a * a / mu);
        out[2] = sqrt(2.0 * mu / r);
    } else {
        // orbital_hohmann between circular orbits r1 -> r2: in = mu, r1, r2; out = sma, dv1, dv2, |dv1| + |dv2|, transfer time s
        const double mu = in[0], r1 = in[1], r2 = in[2];
        const double sma = 0.5 * (r1 + r2), v1c = sqrt(mu / r1), v2c = sqrt(mu / r2);
        const double dv1 = v1c * sqrt(2.0 * r2 / (r1 + r2)) - v1c, dv2 = v2c - v2c * sqrt(2.0 * r1 / (r1 + r2));
        out[0] = sma; out[1] = dv1; out[2] = dv2; out[3] = fabs(dv1) + fabs(dv2); out[4] = AZ_PI * sqrt(sma * sma * sma / mu);
    }
}

// known-answer hook for devmath.h on the device itself (azh_selftest_math)
__global__ void __launch_bounds__(64) k_math_kat(const double *x, unsigned n, double *out)
{
    const unsigned i = blockIdx.x * 64 + threadIdx.x;
    const double v = x[i < n ? i : n - 1];
    double sn, cs;
    az_sincos(v, sn, cs);
    const double rc = az_rcp(v), rs = az_rsqrt(fabs(v));
    double s0 = 0.6684330296514408, c0 = 0.7437723340317224; // (sin,cos)(0.7321)
    az_rotate(s0, c0, v, az_rotk());
    if (i < n) {
        out[i] = sn; out[n + i] = cs; out[2 * (size_t)n + i] = v * rc; out[3 * (size_t)n + i] = fabs(v) * rs * rs;
        out[4 * (size_t)n + i] = s0; out[5 * (size_t)n + i] = c0;
    }
}

// element initialisation: raw[k*n_pad + s] -> el rows + flags
__global__ void __launch_bounds__(64) k_init(const double *raw, size_t n, size_t n_pad, AzGrav g, double *el,
                                                   unsigned *flags)
{
    const size_t s = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (s >= n) return;
    double in[AZ_NUM_RAW];
#pragma unroll
    for (int k = 0; k < AZ_NUM_RAW; ++k) in[k] = raw[(size_t)k * n_pad + s];
    flags[s] = az_init_satellite(in, g, el, n_pad, s);
}

// cols_kernel.h -- k_cols_fast: TIME-MAJOR output, ONE LANE = ONE SATELLITE, branch-free step, time-major inner loop.
//
// BASELINE.json's north_star in its literal form ("one lane = one satellite, time-major inner loop, LDS-staged per-satellite
// constants") and the one structure tools/tile_store_probe.hip says can beat the 16-row tiles of
// k_tiles_fast: a wave is 64 consecutive CATALOG rows, so every step leaves as one 1,536-byte run per array (the tile kernel:
// 384-byte runs, one workgroup-wide barrier per 64 steps, sixteen waves that store at the same moment).  No barrier anywhere:
// a workgroup is one wave and waves drift apart like the row kernel's.
//
// MEASURED (round 5, profiles/r05_experiments.txt A): parity-green on every time-major test of the GPU tier, and SLOWER than the
// tiles on every box it ran on -- same box, config 2: 0.343-0.352 ms against k_tiles_fast's 0.280-0.284 even with the copied
// lanes switched off (0.41 with them), config 3 0.53-0.64 against 0.45-0.49.  Its store stream alone (no arithmetic) takes
// 0.20-0.27 ms depending on the box: 3,000 waves that each write one 1.5-KB run per array and step reach 3.5-4.6 TB/s, the
// 16-row tiles with their 64-row blocks do better.  It therefore stays OFF by default (azh_set_tile_kernel(c, 2) or
// ASTROZ_AMD_COLS=1 selects it): north_star's literal kernel exists, is tested, and is not the fast one on this machine.
//   * the step is az_sgp4_fast_step in its near-circular form, validated once per (time segment, satellite) by the window
//     plan (k_plan_windows) -- no compare, no vote, no branch in the loop;
//   * the 39 per-satellite constants of the step are per LANE here (the lane = time kernels hold them in SGPRs): the ones used
//     late in the step sit in a per-wave LDS table, one 512-byte column per constant (conflict-free ds_read_b64 at the point
//     of use), the rest in VGPRs; polynomial coefficients are wave-uniform literals (SGPRs);
//   * time is arithmetic (t_first[lane] + i step), the deviation of a quasi-uniform grid and the Greenwich angle of an ECEF
//     launch are wave-uniform: scalar loads, no vector-memory load in the loop for a wave of near-circular members;
//   * the lanes whose row this kernel does not compute -- eccentric members (k_rows_fast<ECC> ran just before), deep-space
//     members (k_rows_deep), failed members (zeros) -- take their 24 bytes per array from the compact satellite-major scratch
//     array (row map: AZ_ROW_*), loaded one step ahead, so that the wave still leaves full runs;
//   * windows the plan rejects are static items of the redo list: the generic pass writes their 24-byte pieces afterwards;
//   * stores: wave-private LDS transpose, three full 1-KB store instructions per two steps and array (az_tm_flush).
#pragma once

#ifndef AZ_COLS_WAVES
#define AZ_COLS_WAVES 3 /* waves per SIMD the register allocator must allow: 164 VGPRs, no scratch (forced to 4: 128 + 92-116 B of scratch, +50 %) */
#endif

// where each per-lane constant lives.  LDS: read once per step, late (J2 corrections, orientation): the read's latency hides
// behind the Kepler solve, and values that the step rotates in place (node and inclination pairs) arrive as fresh registers.
#ifndef AZ_COLSK_LDS
#define AZ_COLSK_LDS(X)                                                                                       \
    X(k_mrt) X(k_c2u) X(k_node) X(nodedot) X(tmid) X(xnodcf) X(sOc) X(cOc) X(sinio) X(cosio) X(k_su) X(k_inc) \
    X(x1mth2) X(k_rv)
#define AZ_COLSK_REG(X)                                                                                       \
    X(cc1) X(d2) X(d3) X(d4) X(nl2) X(nl3) X(nl4) X(nl5) X(eta) X(omgcof) X(xmcof) X(xd) X(bc4) X(bc5) X(ecb) \
    X(sab) X(aycof) X(xlcof) X(sdA) X(cdA) X(sdW) X(cdW) X(tc) X(sdU) X(cdU) X(mdot) X(argpdot) X(udot)
#endif
enum ColsLds {
#define X(n) CL_##n,
    AZ_COLSK_LDS(X)
#undef X
    CL_NUM
};
struct FastKCols {
    static constexpr bool SCALAR = false;
    const double *col; // this lane's word of column 0
#define X(n) double n##_;
    AZ_COLSK_REG(X)
#undef X
#define X(n) AZ_MEMBER double n() const { return n##_; }
    AZ_COLSK_REG(X)
#undef X
#define X(n) AZ_MEMBER double n() const { return col[CL_##n * 64]; }
    AZ_COLSK_LDS(X)
#undef X
};

// a wave-uniform double through a scalar load (constant address space: the compiler emits s_load, not a vector load that
// would wait, through vmcnt, for the output stores in flight)
__device__ __forceinline__ double az_sload(const double *q)
{
    return *(__attribute__((address_space(4))) const double *)(q);
}

// grid: x = groups of 64 catalog rows (padded to a multiple of 8: XCD-aware, az_xcd_row), y = time segments of p.tile points
template <bool VEL, int FRAME, int DELTA> // FRAME: 0 TEME, 1 ECEF, 2 geodetic positions (+ ECEF velocities); DELTA: 0 exact grid, 2 quasi-uniform (fp64 deviations)
__global__ void __launch_bounds__(64, (FRAME == 2 && VEL) ? 2 : AZ_COLS_WAVES) k_cols_fast(PropArgs p) // (geodetic + velocities: 2 waves/SIMD, no spill)
{
    constexpr unsigned NA = VEL ? 2u : 1u;
    __shared__ __attribute__((aligned(16))) double col_lds[CL_NUM * 64];
    __shared__ __attribute__((aligned(16))) double stage[NA * AZ_TM_ROW];
    const unsigned lane = threadIdx.x;
    const unsigned s_first = az_xcd_row() * 64u;
    if (s_first >= p.n_rows || s_first + 64u <= p.row_lo || s_first >= p.row_hi) return;
    const unsigned t_lo = blockIdx.y * p.tile;
    if (t_lo >= p.n_times) return;
    const unsigned t_hi = min(t_lo + p.tile, p.n_times);
    const bool in = s_first + lane < p.n_rows;
    const unsigned s = in ? s_first + lane : p.n_rows - 1u; // (lanes beyond the catalog shadow its last row; they never store)
    const unsigned rm = p.rowmap[s], kind = rm >> 30, slot = rm & 0x3fffffffu;
    const unsigned fl = p.flags[s];
    const bool wr = in && s >= p.row_lo && s < p.row_hi && (p.mask == nullptr || p.mask[s] != 0);
    if (!az_any(wr)) return;
    const bool near = kind == AZ_ROW_NEAR;
    const bool copy = wr && ((near && slot >= p.n_circ) || kind == AZ_ROW_COPY); // computed by the lane = time kernels before
    const bool zero = kind == AZ_ROW_ZERO;
    const size_t crow = (size_t)(kind == AZ_ROW_COPY ? slot : p.ecc_row0 + (slot - p.n_circ)) * p.n_times;
#if defined(AZ_COLS_ABLATE_NOCOPY) /* tuning experiment: the copied lanes keep whatever the step computes */
    const bool any_copy = false;
#else
    const bool any_copy = az_any(copy);
#endif
    const bool dense = !az_any(!wr); // every lane writes: the wave's 64 x 24 bytes of a step are one run
    const double step = p.uniform_step, t_first = p.grid_t0 + (p.offsets ? p.offsets[s] : 0.0);
    FastKCols k;
    FastCarry fc;
    {
        FastK k0;
        az_load_fast(p.el, p.n_pad, s, fl, p.inc, 1, k0); // (increments of ONE grid step)
        const double w_a = fma((double)t_lo, step, t_first), w_b = fma((double)(t_hi - 1), step, t_first);
        // window constants from the plan (per lane; a rejected window is a static item of the redo list and stays stale here)
        const size_t at = (size_t)blockIdx.y * p.plan_stride + (near ? slot : 0u);
        const double *w = p.plan_win + at * AZ_PLAN_NUM;
        k0.tmid_ = 0.5 * (w_a + w_b);
        // (only for the lanes this kernel computes: az_seed_fast votes on tc over the wave, and a copied lane's plan entry may
        // belong to another segmentation or not exist at all -- its bits must not decide how the other lanes are seeded)
        k0.tc_ = (near && slot < p.n_circ && (p.plan_flag[at] & AZ_PLAN_TC)) ? k0.tmid_ : 0.0;
        k0.sOc_ = w[AZ_PLAN_sOc]; k0.cOc_ = w[AZ_PLAN_cOc]; k0.sdU_ = w[AZ_PLAN_s1U]; k0.cdU_ = w[AZ_PLAN_c1U];
        if (DELTA) az_fast_udot(k0);
        double *colw = col_lds + lane;
#define X(n) colw[CL_##n * 64] = k0.n##_;
        AZ_COLSK_LDS(X)
#undef X
#define X(n) k.n##_ = k0.n##_;
        AZ_COLSK_REG(X)
#undef X
        az_seed_fast(p.el, p.n_pad, s, fma((double)t_lo - 1.0, step, t_first), k0.tc_, fc); // one step BEFORE the first
        az_wave_lds_fence();
    }
    double *lds_p = stage, *lds_v = stage + AZ_TM_ROW;
    const size_t pitch = (size_t)p.stride_sats * 3;
    // The rows other kernels computed.  A vector load in this loop waits, through vmcnt, for every output store issued before
    // it, so the loads of step i + 1 are issued at the end of step i's arithmetic -- AHEAD of step i's stores -- and consumed
    // one step later, into the same registers.  They are unconditional (lanes without a copied row read the scratch array's
    // first row: one broadcast line) and the loop exists with and without them, chosen per wave (a conditional load's values
    // merge at the join, and the compiler waits for them there).  Even so the wave can never have more than one step's stores
    // in flight, which costs this store-bound kernel a third of its rate (profiles/r05_experiments.txt A); scalar loads, which
    // know nothing of the stores, were tried in their place: their latency per copied lane and step is worse (0.70 ms).
    const double *cpos = p.tmp_pos + (copy ? crow * 3 : 0), *cvel = VEL ? p.tmp_vel + (copy ? crow * 3 : 0) : nullptr;
    auto run = [&](auto copy_tag, auto dense_tag) {
        constexpr bool COPY = decltype(copy_tag)::value, DENSE = decltype(dense_tag)::value;
        az_d2s npq = {0.0, 0.0}, nvq = {0.0, 0.0};
        double npz = 0.0, nvz = 0.0;
        if (COPY) {
            npq = *reinterpret_cast<const az_d2s *>(cpos + (size_t)t_lo * 3); npz = cpos[(size_t)t_lo * 3 + 2];
            if (VEL) { nvq = *reinterpret_cast<const az_d2s *>(cvel + (size_t)t_lo * 3); nvz = cvel[(size_t)t_lo * 3 + 2]; }
        }
        // every load of the set-up has landed before the loop starts: a load still pending at the loop header would put a
        // wait INSIDE the loop (the header merges the entry state with the back edge's), and that wait, through vmcnt, is a
        // wait for the previous step's stores
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
#pragma unroll 1
        for (unsigned i = t_lo; i < t_hi; ++i) {
            double t = fma((double)i, step, t_first), dl = 0.0;
            if (DELTA) {
                dl = az_sload(p.delta64 + i);
                t += dl;
            }
            k.col = az_opaque_lds(col_lds + lane);
            double r[3], v[3];
#if defined(AZ_ABLATE) && AZ_ABLATE == 2 /* tuning experiment: stores only */
            r[0] = t; r[1] = t + 1.0; r[2] = t + 2.0; v[0] = t + 3.0; v[1] = t + 4.0; v[2] = t + 5.0;
#else
            az_sgp4_fast_step<VEL, false, DELTA>(k, p.g, RotCoefLit(), t, fc, r, v, dl);
#endif
            if (FRAME) {
                const double sg = az_sload(p.sin_g + i), cg = az_sload(p.cos_g + i);
                az_to_ecef(r, sg, cg);
                if (VEL) az_to_ecef(v, sg, cg);
                if (FRAME == 2) az_ecef_to_geodetic(r);
            }
            if (COPY) {
                // this step's copied values, then -- into the same registers, ahead of this step's stores -- the next step's
                r[0] = copy ? npq.x : r[0]; r[1] = copy ? npq.y : r[1]; r[2] = copy ? npz : r[2];
                if (VEL) { v[0] = copy ? nvq.x : v[0]; v[1] = copy ? nvq.y : v[1]; v[2] = copy ? nvz : v[2]; }
                const size_t a1 = (size_t)min(i + 1u, t_hi - 1u) * 3;
                npq = *reinterpret_cast<const az_d2s *>(cpos + a1); npz = cpos[a1 + 2];
                if (VEL) { nvq = *reinterpret_cast<const az_d2s *>(cvel + a1); nvz = cvel[a1 + 2]; }
            }
            if (zero) {
                r[0] = r[1] = r[2] = 0.0;
                v[0] = v[1] = v[2] = 0.0;
            }
#if defined(AZ_ABLATE) && AZ_ABLATE == 1 /* tuning experiment: arithmetic only */
            if (r[0] + r[1] + r[2] + (VEL ? v[0] + v[1] + v[2] : 0.0) != 1.2345e300) continue;
#endif
            if (DENSE) {
                const unsigned par = (i - t_lo) & 1u; // wave-uniform
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
                az_tm_flush(lds_p, p.pos + ob, pitch, lane, par, i + 1 == t_hi);
                if (VEL) az_tm_flush(lds_v, p.vel + ob, pitch, lane, par, i + 1 == t_hi);
                az_wave_lds_fence();
            } else if (wr) {
                const size_t ob = ((size_t)i * p.stride_sats + s) * 3;
                p.pos[ob] = r[0]; p.pos[ob + 1] = r[1]; p.pos[ob + 2] = r[2];
                if (VEL) { p.vel[ob] = v[0]; p.vel[ob + 1] = v[1]; p.vel[ob + 2] = v[2]; }
            }
        }
    };
    if (dense) {
        if (any_copy) run(std::true_type{}, std::true_type{});
        else run(std::false_type{}, std::true_type{});
    } else { // (catalog tail, row windows)
        if (any_copy) run(std::true_type{}, std::false_type{});
        else run(std::false_type{}, std::false_type{});
    }
}

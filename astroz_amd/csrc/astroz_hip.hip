// astroz_hip.hip -- libastroz_hip.so: C ABI (include/astroz_hip.h) over the gfx950 kernels.
// gfx950 only; compiled with  hipcc --offload-arch=gfx950.
#include "../../include/astroz_hip.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h> // types and prototypes only: librccl is loaded on first use (azh_group_propagate_allgather)

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <map>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "kernels.h"
#include "tle_host.h"
#include "host_step.h"

namespace {

thread_local std::string g_last_error;

bool hip_ok(hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();
    return false;
}
#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        if (!hip_ok((expr), #expr)) return AZ_ERR_HIP;                                                      \
    } while (0)

AzGrav make_grav(int which)
{
    // src/constants.zig L41-64
    AzGrav g;
    if (which == AZ_WGS72) {
        g.radius_km = 6378.135;
        g.j2 = 0.001082616;
        g.j4 = -0.00000165597;
        g.xke = 0.0743669161331734132;
        g.j3oj2 = -0.00234506972242078;
    } else {
        g.radius_km = 6378.137;
        g.j2 = 0.00108262998905;
        g.j4 = -0.00000161098761;
        g.xke = 0.07436685316871385;
        g.j3oj2 = -0.00233899967218727;
    }
    g.vkmpersec = g.xke * g.radius_km / 60.0; // src/Sgp4.zig L177
    g.half_j2 = 0.5 * g.j2;
    return g;
}

const char *const kFieldNames[] = {
#define X(n) #n,
    AZ_SGP4_FIELDS(X) AZ_DEEP_FIELDS(X)
#undef X
};

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    int ensure(size_t n)
    {
        if (n <= cap) return AZ_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = std::max<size_t>(n, 64);
        if (!hip_ok(hipMalloc((void **)&p, want * sizeof(T)), "hipMalloc")) return AZ_ERR_HIP;
        cap = want;
        return AZ_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// pinned staging slots of the host-returning calls (copy_back_staged)
constexpr int kStageSlots = 3;
constexpr size_t kStageChunk = size_t(16) << 20;
struct HostStager {
    void *slot[kStageSlots] = {nullptr, nullptr, nullptr};
    hipEvent_t ev[kStageSlots] = {nullptr, nullptr, nullptr};
    bool ready = false;
    int32_t ensure()
    {
        if (ready) return AZ_OK;
        for (int k = 0; k < kStageSlots; ++k) {
            if (!slot[k] && !hip_ok(hipHostMalloc(&slot[k], kStageChunk, hipHostMallocDefault), "hipHostMalloc(stage)")) return AZ_ERR_HIP;
            if (!ev[k] && !hip_ok(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming), "hipEventCreate")) return AZ_ERR_HIP;
        }
        ready = true;
        return AZ_OK;
    }
    void release()
    {
        for (int k = 0; k < kStageSlots; ++k) {
            if (slot[k]) (void)hipHostFree(slot[k]);
            if (ev[k]) (void)hipEventDestroy(ev[k]);
            slot[k] = nullptr;
            ev[k] = nullptr;
        }
        ready = false;
    }
};
} // namespace

struct azh_constellation {
    int device = 0;
    size_t n = 0, n_pad = 0;
    AzGrav g{};
    double *d_el = nullptr;
    unsigned *d_flags = nullptr;
    std::vector<unsigned> h_flags;
    std::vector<double> h_epoch;
    std::vector<double> h_raw[AZ_NUM_RAW]; // the TLE-unit inputs (for azh_constellation_subset)
    // launch lists (table indices)
    DevBuf<unsigned> d_list; // [near-earth | deep (by irez) | bad]
    std::vector<unsigned> h_list; // host copy (row windows: the slot range of a window in the catalog-ordered sub-lists)
    unsigned n_sgp4 = 0, n_sdp4 = 0, n_bad = 0;
    // per-call scratch
    DevBuf<double> d_times, d_offsets, d_sin, d_cos, d_seeds;
    DevBuf<double> d_node_cache; // [3][n_pad]: per satellite, the resonance integrator node nearest to epoch the last seeding pass reached
    // window plans of the staged uniform grid (k_plan_windows), one per launch shape: [0] row kernels (64 grid points per
    // lane step), [1] the packed fp32 row kernel (128), [2] the time-major tile kernel.  redo: [0],[1] item counters
    // (alternating launches, re-armed to the number of static items), [2] number of static items (windows the validation
    // bounds reject), [4..] (list slot, first grid point, end) triples -- static items first, dynamic ones behind
    struct FastPlan {
        bool valid = false, mixed32 = false;
        unsigned tile_c = 0, tile_e = 0, n_list = 0, n_seg = 0, parity = 0;
        DevBuf<double> win;
        DevBuf<unsigned char> flag;
        DevBuf<unsigned> redo;
    } plan[4]; // ([3]: the lane = satellite time-major kernel, k_cols_fast)
    DevBuf<double> d_inc;       // uniform grids: per-satellite rotation increments (k_prep_inc), [2 * AZ_INC_NUM][n_pad]
    DevBuf<double> d_fast_rec;  // ... and the record of folded constants of the lane = time fast kernels (k_prep_rec), [n_pad][FR_NUM]
    double uniform_step = 0.0;  // step of the staged grid if it is uniform, else 0
    // quasi-uniform grid (the reference's own (jd, fr) arithmetic: uniform to ~4e-7 min): deviations from the ideal grid,
    // fp32, and their largest magnitude (0: exactly uniform, d_delta unused)
    DevBuf<float> d_delta;
    double delta_max = 0.0;
    std::vector<float> h_delta; // (source of the asynchronous upload)
    // ... or, WIDE form (jitter of seconds about a uniform grid, |delta| <= AZ_DELTA_WIDE_MAX): fp64 deviations from a fitted grid
    DevBuf<double> d_delta64;
    std::vector<double> h_delta64;
    bool delta_wide = false;
    double grid_t0 = 0.0;       // origin of the ideal grid (times[0] unless the grid is a fit)
    unsigned last_path = 0;     // AZH_PATH_* bits of the most recent launch set (azh_last_path)
    DevBuf<double> d_tgt, d_part_d2, d_out_d; // fused screen: target track, partial minima, results
    DevBuf<unsigned> d_part_t, d_out_t;
    // one satellite x many times (Satrec.sgp4 / sgp4_array / c_api sgp4_propagate*): persistent scratch so
    // that a scalar call costs one small H2D, one launch, one D2H and one synchronisation -- no allocation
    DevBuf<double> d_one_t, d_one_o;
    DevBuf<unsigned char> d_one_e;
    DevBuf<unsigned> d_one_items; // k_one_fast -> k_one_satellite hand-over list (count, then segment indices)
    unsigned one_segments = 0;     // segments of the most recent one-satellite call that k_one_fast was launched on (azh_last_one_stats)
    hipEvent_t ev_one = nullptr;   // recorded behind the kernels of the most recent k_one_fast call (azh_last_one_stats waits on it:
                                   // the caller's stream may be gone by then)
    void *h_stage = nullptr; // pinned host staging for small one-satellite calls: h_stage_cap points (grows to kOneStage)
    size_t h_stage_cap = 0;
    // the inputs of the staged grid (stage_inputs skips the staging of byte-identical ones)
    std::vector<double> h_times, h_offsets;
    std::vector<uint8_t> h_mask;
    double staged_ref_jd = 0.0;
    hipStream_t staged_stream = nullptr;
    bool staged_valid = false;
    void *h_small = nullptr; // pinned host buffer the kernels of a SMALL host-returning constellation call write into directly
    size_t h_small_cap = 0;
    HostStager stager;       // pinned staging slots of the host-returning calls (copy_back_staged), allocated on first use
    bool seeds_valid = false; // resonance seeds match the staged times/offsets and tile
    unsigned seeds_tile = 0;
    bool seeds_rows = false; // seed table laid out for the lane = time kernel (64-point chunks)
    DevBuf<unsigned char> d_mask;
    DevBuf<double> d_host_pos, d_host_vel; // azh_propagate_host: grow-only device-side result buffers
    DevBuf<unsigned char> d_host_err;
    bool have_offsets = false, have_mask = false;
    unsigned cached_n_times = 0;
    int cached_mode = 0;
    hipStream_t s_main = nullptr, s_deep = nullptr, s_ecc = nullptr;
    bool own_stream = true; // false: s_main is the device's shared stream of small handles (never destroyed with the handle)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_fork2 = nullptr, ev_join2 = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
    unsigned off_cat = 0; // d_list + off_cat: near-earth members in plain catalog order (k_tiles_fast: runs of consecutive rows)
    unsigned off_deep_cat = 0; // d_list + off_deep_cat: deep-space members in plain catalog order (lane = time kernels)
    unsigned off_rowmap = 0;   // d_list + off_rowmap: per catalog row, kind << 30 | slot (AZ_ROW_*: k_tiles_fast)
    unsigned off_rowmap2 = 0;  // ... the same with near-earth slots counted in the [class 0 | other classes] list (k_cols_fast)
    // hipGraph cache of the cached-input launch sets (azh_propagate_device_cached / _window): a launch set is 3-9 API calls
    // (kernels on up to three streams, fork / join events, a memset); captured once per (outputs, layout, stride, row window,
    // redo-counter parity) it replays as ONE hipGraphLaunch.  Dropped whenever new inputs are staged or a switch changes.
    struct LaunchGraph {
        const void *pos, *vel, *err;
        int layout, f32;
        size_t stride, row_lo, row_hi;
        hipStream_t st;
        unsigned sig_before, sig_after; // parity bits of the four window plans before / after the launch set
        unsigned path;
        unsigned seen;                  // calls with this key before it was captured
        unsigned fails;                 // failed captures of this key; from the second on the key runs eagerly for good
        hipGraph_t graph;
        hipGraphExec_t exec;
    };
    std::vector<LaunchGraph> graphs;
    int graphs_on = 0;         // azh_set_graphs / ASTROZ_AMD_GRAPHS (off by default: measured, it only pays for multi-window pipelines)
    bool capturing = false;    // inside a stream capture: anything that must allocate, synchronize or rebuild first returns AZ_RC_EAGER
    int cols_kernel = 0;       // time-major output on (quasi-)uniform grids through k_cols_fast (lane = satellite) instead of k_tiles_fast
    DevBuf<double> d_deep_tmp; // time-major output: the deep-space rows' compact satellite-major scratch (k_deep_transpose)
    bool tile_kernel = true; // time-major output through the tile kernels (azh_set_tile_kernel)
    unsigned off_circ = 0, n_circ = 0; // d_list + off_circ: near-earth members in catalog order, [n_circ of eccentricity class 0 | the rest]
    bool timed = false;
    bool timing = true; // record the ev_t0/ev_t1 pair around every launch set (azh_set_timing)
    unsigned tile_sgp4 = 0, tile_sdp4 = 0;
    // fp32 outputs (azh_set_f32_arithmetic): 0 (default) = the mixed-precision step where it applies (near-circular members,
    // TEME, uniform grid: every O(1) quantity in fp64, the small ones in packed fp32 -- storage-level accuracy, 0.4 m /
    // 0.4 mm/s), fp64 arithmetic rounded once at the store elsewhere; 1 = packed fp32 arithmetic (opt-in: metres / mm/s);
    // 2 = fp64 arithmetic rounded at the store everywhere
    int f32_mode = 0;
    bool fast_path = true; // use the branch-free uniform-grid step where it applies (azh_set_fast_path)
    // host route of calls of a few points (host_step.h): the element columns the DEVICE initialised, mirrored on the host.
    // Small handles (what Satrec / sgp4_init make): the whole table, fetched with the status words when the handle is made;
    // catalogs: one column per satellite asked for, fetched on first use (at most kHostCols of them, oldest replaced).
    std::vector<double> h_el;                            // [AZ_NUM_FIELDS][n_pad] or empty
    std::vector<std::pair<size_t, std::vector<double>>> h_cols; // (satellite, AZ_NUM_FIELDS contiguous doubles)
    size_t h_cols_next = 0;
};

namespace {

constexpr size_t kHostMirrorPad = 64; // handles of up to 64 satellites mirror their whole element table on the host
// (default point limit of the host route: 128, i.e. 128 near-earth / 64 deep-space points per call)
constexpr size_t kHostCols = 256;     // cached columns of larger handles
constexpr size_t kHostColsBeforeTable = 8;      // ... after this many single columns the whole table is mirrored instead,
constexpr size_t kHostTableMax = size_t(64) << 20; // if it is at most this large (64 MB: 98,000 satellites)
// calls of at most this many points take the host route (azh_set_host_points; 0 switches it off)
std::atomic<size_t> g_host_points{[] {
    const char *e = getenv("ASTROZ_AMD_HOST_POINTS");
    return e ? (size_t)strtoull(e, nullptr, 10) : size_t(128);
}()};
inline size_t host_points() { return azhost::cpu_ok() ? g_host_points.load(std::memory_order_relaxed) : 0; }

// internal return code: the launch set cannot be captured yet (a plan, the seeds or a scratch buffer must be built first)
constexpr int32_t AZ_RC_EAGER = 0x7a5eca9;
int set_device(const azh_constellation *c) { return hip_ok(hipSetDevice(c->device), "hipSetDevice") ? AZ_OK : AZ_ERR_HIP; }

// One launch stream per DEVICE for all handles of a few satellites (what sgp4_init / Satrec make, possibly by the thousand): a
// stream of its own per handle was 0.4-3.5 ms to create and 0.5-2.5 ms to destroy -- hipStreamCreate / hipStreamDestroy get
// slower with every live stream (3.5 / 2.5 ms each at 500) -- for handles that have nothing to overlap.  Created on first use,
// kept for the life of the process.  Launches of different handles interleave on it; a synchronize waits for all of them.
hipStream_t small_stream(int device)
{
    static std::mutex mu;
    static std::map<int, hipStream_t> streams;
    std::lock_guard<std::mutex> lk(mu);
    auto it = streams.find(device);
    if (it != streams.end()) return it->second;
    hipStream_t s = nullptr;
    if (!hip_ok(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate")) return nullptr;
    streams[device] = s;
    return s;
}

void drop_graphs(azh_constellation *c);
void destroy(azh_constellation *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    drop_graphs(c);
    if (c->d_el) (void)hipFree(c->d_el);
    if (c->d_flags) (void)hipFree(c->d_flags);
    c->d_list.release();
    c->d_times.release();
    c->d_offsets.release();
    c->d_sin.release();
    c->d_cos.release();
    c->d_seeds.release();
    c->d_node_cache.release();
    c->d_deep_tmp.release();
    c->d_inc.release();
    c->d_fast_rec.release();
    c->d_delta.release();
    c->d_delta64.release();
    for (auto &pl : c->plan) { pl.win.release(); pl.flag.release(); pl.redo.release(); }
    c->d_tgt.release();
    c->d_part_d2.release();
    c->d_out_d.release();
    c->d_part_t.release();
    c->d_out_t.release();
    c->d_one_t.release();
    c->d_one_o.release();
    c->d_one_e.release();
    c->d_one_items.release();
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_small) (void)hipHostFree(c->h_small);
    c->stager.release();
    c->d_mask.release();
    c->d_host_pos.release();
    c->d_host_vel.release();
    c->d_host_err.release();
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_fork2) (void)hipEventDestroy(c->ev_fork2);
    if (c->ev_join2) (void)hipEventDestroy(c->ev_join2);
    if (c->ev_one) (void)hipEventDestroy(c->ev_one);
    if (c->s_ecc && c->s_ecc != c->s_main) (void)hipStreamDestroy(c->s_ecc);
    if (c->ev_t0) (void)hipEventDestroy(c->ev_t0);
    if (c->ev_t1) (void)hipEventDestroy(c->ev_t1);
    if (c->s_deep && c->s_deep != c->s_main) (void)hipStreamDestroy(c->s_deep);
    if (c->s_main && c->own_stream) (void)hipStreamDestroy(c->s_main);
    delete c;
}

// raw columns (AzRawField order, each of length n) -> device table via the init kernel
int32_t build(const std::vector<double> (&cols)[AZ_NUM_RAW], size_t n, int grav, int device, azh_constellation **out)
{
    if (!out) return AZ_ERR_NULL_POINTER;
    *out = nullptr;
    if (n == 0) return AZ_ERR_VALUE;
    if (n > 0x7fffffffu) return AZ_ERR_VALUE;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) {
        g_last_error = "no such HIP device";
        return AZ_ERR_HIP;
    }
    azh_constellation *c = new (std::nothrow) azh_constellation();
    if (!c) return AZ_ERR_ALLOC_FAILED;
    c->device = device;
    c->n = n;
    c->n_pad = (n + 63) / 64 * 64;
    c->g = make_grav(grav);
    int32_t rc = AZ_OK;
    double *d_raw = nullptr;
    do {
        if (set_device(c) != AZ_OK) { rc = AZ_ERR_HIP; break; }
        // one launch stream + two side streams (deep-space launch; eccentric launch and redo pass beside the bulk).  A handle of
        // a few satellites -- what sgp4_init / Satrec create, possibly by the thousand -- has nothing to overlap: its side
        // streams ARE the launch stream (the fork / join events then order a stream with itself)
        const bool side_streams = n > 16;
        if (!side_streams) {
            c->s_main = small_stream(device);
            c->own_stream = false;
        }
        if ((side_streams ? !hip_ok(hipStreamCreateWithFlags(&c->s_main, hipStreamNonBlocking), "hipStreamCreate") : c->s_main == nullptr) ||
            (side_streams && (!hip_ok(hipStreamCreateWithFlags(&c->s_deep, hipStreamNonBlocking), "hipStreamCreate") ||
                              !hip_ok(hipStreamCreateWithFlags(&c->s_ecc, hipStreamNonBlocking), "hipStreamCreate"))) ||
            !hip_ok(hipEventCreateWithFlags(&c->ev_fork2, hipEventDisableTiming), "hipEventCreate") ||
            !hip_ok(hipEventCreateWithFlags(&c->ev_join2, hipEventDisableTiming), "hipEventCreate") ||
            !hip_ok(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming), "hipEventCreate") ||
            !hip_ok(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming), "hipEventCreate") ||
            !hip_ok(hipEventCreateWithFlags(&c->ev_one, hipEventDisableTiming), "hipEventCreate") ||
            !hip_ok(hipEventCreate(&c->ev_t0), "hipEventCreate") || !hip_ok(hipEventCreate(&c->ev_t1), "hipEventCreate")) {
            rc = AZ_ERR_HIP;
            break;
        }
        if (!side_streams) c->s_deep = c->s_ecc = c->s_main;
        const size_t np = c->n_pad;
        if (!hip_ok(hipMalloc((void **)&c->d_el, sizeof(double) * AZ_NUM_FIELDS * np), "hipMalloc(el)") ||
            !hip_ok(hipMalloc((void **)&c->d_flags, sizeof(unsigned) * np), "hipMalloc(flags)") ||
            !hip_ok(hipMalloc((void **)&d_raw, sizeof(double) * AZ_NUM_RAW * np), "hipMalloc(raw)")) {
            rc = AZ_ERR_HIP;
            break;
        }
        std::vector<double> staging((size_t)AZ_NUM_RAW * np, 0.0);
        for (int k = 0; k < AZ_NUM_RAW; ++k) memcpy(&staging[(size_t)k * np], cols[k].data(), sizeof(double) * n);
        if (!hip_ok(hipMemcpyAsync(d_raw, staging.data(), sizeof(double) * staging.size(), hipMemcpyHostToDevice, c->s_main), "H2D raw") ||
            !hip_ok(hipMemsetAsync(c->d_el, 0, sizeof(double) * AZ_NUM_FIELDS * np, c->s_main), "memset el") ||
            !hip_ok(hipMemsetAsync(c->d_flags, 0, sizeof(unsigned) * np, c->s_main), "memset flags")) {
            rc = AZ_ERR_HIP;
            break;
        }
        hipLaunchKernelGGL(k_init, dim3((unsigned)(np / 64)), dim3(64), 0, c->s_main, d_raw, n, np, c->g, c->d_el, c->d_flags);
        if (!hip_ok(hipGetLastError(), "k_init launch")) { rc = AZ_ERR_HIP; break; }
        c->h_flags.resize(n);
        if (np <= kHostMirrorPad) { // (host route of few-point calls: the initialised columns travel back with the status words)
            c->h_el.resize((size_t)AZ_NUM_FIELDS * np);
            if (!hip_ok(hipMemcpyAsync(c->h_el.data(), c->d_el, sizeof(double) * c->h_el.size(), hipMemcpyDeviceToHost, c->s_main), "D2H el")) {
                rc = AZ_ERR_HIP;
                break;
            }
        }
        if (!hip_ok(hipMemcpyAsync(c->h_flags.data(), c->d_flags, sizeof(unsigned) * n, hipMemcpyDeviceToHost, c->s_main), "D2H flags") ||
            !hip_ok(hipStreamSynchronize(c->s_main), "sync(init)")) {
            rc = AZ_ERR_HIP;
            break;
        }
        c->h_epoch = cols[R_epoch_jd];
        for (int k = 0; k < AZ_NUM_RAW; ++k) c->h_raw[k] = cols[k];
        // launch lists.
        //  [near-earth]  catalog order, except that inside every group of AZ_BLOCK consecutive
        //      members (= one workgroup = 4 waves) the members are ordered by eccentricity class.
        //      A single e = 0.2 satellite needs ~4 Kepler-Newton trips and the wide rotation tier and
        //      would drag the 63 other lanes of its wave along (4% such members touch 93% of the
        //      waves of an unsorted catalog); after the in-group ordering they share the group's last
        //      wave.  The group still writes the same contiguous span of rows.
        //  [deep-space by resonance class]  uniform integrator branch per wave.
        //  [failed inits]
        std::vector<unsigned> list;
        list.reserve(n);
        for (size_t s = 0; s < n; ++s)
            if (AZ_FLAG_ERR(c->h_flags[s]) == 0 && !(c->h_flags[s] & AZ_FLAG_DEEP)) list.push_back((unsigned)s);
        c->n_sgp4 = (unsigned)list.size();
        for (size_t g0 = 0; g0 < list.size(); g0 += AZ_BLOCK) {
            const size_t g1 = std::min(g0 + (size_t)AZ_BLOCK, list.size());
            std::stable_sort(list.begin() + g0, list.begin() + g1, [&](unsigned x, unsigned y) {
                return AZ_FLAG_ECLASS(c->h_flags[x]) < AZ_FLAG_ECLASS(c->h_flags[y]);
            });
        }
        {
            // deep space: order by (resonance class, near-equatorial [Lyddane branch], eccentricity
            // class) so that the integrator branch, the Lyddane branch and the number of Kepler-Newton
            // trips are as uniform as possible inside each wave
            std::vector<unsigned> deep;
            for (size_t s = 0; s < n; ++s)
                if (AZ_FLAG_ERR(c->h_flags[s]) == 0 && (c->h_flags[s] & AZ_FLAG_DEEP)) deep.push_back((unsigned)s);
            const std::vector<double> &incl = cols[R_incl_deg];
            auto key = [&](unsigned s) {
                const unsigned f = c->h_flags[s];
                const unsigned low = incl[s] < 12.5 ? 1u : 0u; // 0.2 rad = 11.46 deg, with margin for the periodics
                return (AZ_FLAG_IREZ(f) << 4) | (low << 2) | AZ_FLAG_ECLASS(f);
            };
            std::stable_sort(deep.begin(), deep.end(), [&](unsigned x, unsigned y) { return key(x) < key(y); });
            list.insert(list.end(), deep.begin(), deep.end());
        }
        c->n_sdp4 = (unsigned)list.size() - c->n_sgp4;
        for (size_t s = 0; s < n; ++s)
            if (AZ_FLAG_ERR(c->h_flags[s]) != 0) list.push_back((unsigned)s);
        c->n_bad = (unsigned)list.size() - c->n_sgp4 - c->n_sdp4;
        // the near-earth members once more, in catalog order, by eccentricity class: [class 0 | other classes]
        // (the two instantiations of k_rows_fast)
        c->off_circ = (unsigned)list.size();
        for (size_t s = 0; s < n; ++s)
            if (AZ_FLAG_ERR(c->h_flags[s]) == 0 && !(c->h_flags[s] & AZ_FLAG_DEEP) && AZ_FLAG_ECLASS(c->h_flags[s]) == 0)
                list.push_back((unsigned)s);
        c->n_circ = (unsigned)list.size() - c->off_circ;
        for (size_t s = 0; s < n; ++s)
            if (AZ_FLAG_ERR(c->h_flags[s]) == 0 && !(c->h_flags[s] & AZ_FLAG_DEEP) && AZ_FLAG_ECLASS(c->h_flags[s]) != 0)
                list.push_back((unsigned)s);
        // ... and once in plain catalog order (time-major tiles of 16 consecutive rows, k_tiles_fast)
        c->off_cat = (unsigned)list.size();
        for (size_t s = 0; s < n; ++s)
            if (AZ_FLAG_ERR(c->h_flags[s]) == 0 && !(c->h_flags[s] & AZ_FLAG_DEEP)) list.push_back((unsigned)s);
        // ... and the deep-space members in plain catalog order: one wave per satellite needs no grouping by branch, and
        // runs of consecutive rows leave the time-major transposer as single long stores
        c->off_deep_cat = (unsigned)list.size();
        for (size_t s = 0; s < n; ++s)
            if (AZ_FLAG_ERR(c->h_flags[s]) == 0 && (c->h_flags[s] & AZ_FLAG_DEEP)) list.push_back((unsigned)s);
        // ... and the row map of the time-major tile kernel: what every catalog row is and where its data comes from
        c->off_rowmap = (unsigned)list.size();
        {
            unsigned near = 0, deepn = 0;
            for (size_t s = 0; s < n; ++s) {
                const unsigned f = c->h_flags[s];
                if (AZ_FLAG_ERR(f) != 0) list.push_back((unsigned)AZ_ROW_ZERO << 30);
                else if (f & AZ_FLAG_DEEP) list.push_back(((unsigned)AZ_ROW_COPY << 30) | deepn++);
                else list.push_back(((unsigned)AZ_ROW_NEAR << 30) | near++);
            }
        }
        // ... and once more for the lane = satellite time-major kernel, whose plan and redo items count near-earth slots in
        // the [class 0 | other classes] list: slot < n_circ = computed there, the rest arrive through the scratch array
        c->off_rowmap2 = (unsigned)list.size();
        {
            unsigned circ = 0, eccn = 0, deepn = 0;
            for (size_t s = 0; s < n; ++s) {
                const unsigned f = c->h_flags[s];
                if (AZ_FLAG_ERR(f) != 0) list.push_back((unsigned)AZ_ROW_ZERO << 30);
                else if (f & AZ_FLAG_DEEP) list.push_back(((unsigned)AZ_ROW_COPY << 30) | deepn++);
                else if (AZ_FLAG_ECLASS(f) == 0) list.push_back(((unsigned)AZ_ROW_NEAR << 30) | circ++);
                else list.push_back(((unsigned)AZ_ROW_NEAR << 30) | (c->n_circ + eccn++));
            }
        }
        if (const char *e = getenv("ASTROZ_AMD_COLS")) c->cols_kernel = atoi(e);
        if (const char *e = getenv("ASTROZ_AMD_GRAPHS")) c->graphs_on = atoi(e);
        c->h_list = list;
        if (c->d_list.ensure(list.size()) != AZ_OK ||
            !hip_ok(hipMemcpy(c->d_list.p, list.data(), sizeof(unsigned) * list.size(), hipMemcpyHostToDevice), "H2D list")) {
            rc = AZ_ERR_HIP;
            break;
        }
    } while (0);
    if (d_raw) (void)hipFree(d_raw);
    if (rc != AZ_OK) {
        destroy(c);
        return rc;
    }
    *out = c;
    return AZ_OK;
}

int32_t build_from_records(const std::vector<azh::TleRecord> &recs, int grav, int device, azh_constellation **out)
{
    std::vector<double> cols[AZ_NUM_RAW];
    for (auto &v : cols) v.resize(recs.size());
    azh::parallel_ranges(recs.size(), azh::parse_threads_for(recs.size() * 140), [&](size_t a, size_t b) {
        for (size_t i = a; i < b; ++i) {
            const azh::TleRecord &r = recs[i];
            cols[R_epoch_jd][i] = r.epoch_jd;
            cols[R_mm_revday][i] = r.mm_revday;
            cols[R_ecc][i] = r.ecc;
            cols[R_incl_deg][i] = r.incl_deg;
            cols[R_raan_deg][i] = r.raan_deg;
            cols[R_argp_deg][i] = r.argp_deg;
            cols[R_ma_deg][i] = r.ma_deg;
            cols[R_bstar][i] = r.bstar;
        }
    });
    return build(cols, recs.size(), grav, device, out);
}

unsigned auto_tile(unsigned n_list, unsigned n_times, unsigned forced, unsigned min_tile)
{
    if (forced) return std::min(std::max(forced, 1u), std::max(n_times, 1u));
    // aim for ~16k waves (4 per SIMD x 4 rounds over 1,024 SIMDs) without making tiles so short
    // that the per-tile element loads and full-sincos seeding dominate
    const unsigned waves_x = (n_list + 63) / 64;
    unsigned n_tiles = (16384 + waves_x - 1) / std::max(waves_x, 1u);
    n_tiles = std::max(1u, std::min(n_tiles, n_times));
    unsigned tile = (n_times + n_tiles - 1) / n_tiles;
    tile = std::max(tile, std::min(min_tile, n_times));
    return std::max(tile, 1u);
}

template <bool DEEP, bool FRAME>
void launch_propagate2(const PropArgs &a, int layout, bool vel, hipStream_t st)
{
    // grid.x padded to a multiple of 8: XCD-aware placement (see k_propagate)
    dim3 grid(((a.n_list + AZ_BLOCK - 1) / AZ_BLOCK + 7) / 8 * 8, (a.n_times + a.tile - 1) / a.tile);
    dim3 block(AZ_BLOCK);
    if (layout == AZ_LAYOUT_TIME_MAJOR) {
        if (vel)
            hipLaunchKernelGGL((k_propagate<1, true, DEEP, FRAME>), grid, block, 0, st, a);
        else
            hipLaunchKernelGGL((k_propagate<1, false, DEEP, FRAME>), grid, block, 0, st, a);
    } else {
        if (vel)
            hipLaunchKernelGGL((k_propagate<0, true, DEEP, FRAME>), grid, block, 0, st, a);
        else
            hipLaunchKernelGGL((k_propagate<0, false, DEEP, FRAME>), grid, block, 0, st, a);
    }
}

// k_rows time segments: enough waves for ~4 rounds of 4 waves/SIMD, segments of >= 256 grid points,
// a multiple of 64 grid points each
unsigned rows_tile(unsigned n_list, unsigned n_times, unsigned forced)
{
    unsigned segs = std::max(1u, std::min((16384u + n_list - 1) / std::max(n_list, 1u), (n_times + 255) / 256));
    if (forced) segs = std::max(1u, (n_times + forced - 1) / forced);
    return ((n_times + segs - 1) / segs + 63) / 64 * 64;
}

bool use_rows(const PropArgs &a, int layout, bool deep)
{
    // satellite-major near-earth rows (and the fused screen, which stores nothing): one wave per
    // satellite, lane = time
    // both populations have a lane = time kernel (k_rows, k_rows_deep).  Deep-space rows use theirs for the
    // time-major layout too (the lane = satellite form needs 250+ VGPRs, 1-2 waves/SIMD): satellite-major into a
    // compact scratch array, then k_deep_transpose (launch_all).
    if (deep && layout == AZ_LAYOUT_TIME_MAJOR && a.n_times >= 32) return true;
    return (layout == AZ_LAYOUT_SAT_MAJOR || a.screen_target) && a.n_times >= 32;
}

// number of partial minima per list slot a screen launch produces
unsigned screen_parts(const PropArgs &a, bool deep)
{
    if (use_rows(a, AZ_LAYOUT_SAT_MAJOR, deep)) {
        unsigned tile = rows_tile(a.n_list, a.n_times, a.tile_forced);
        if (deep) tile = std::min(tile, 64u * (unsigned)AZ_DEEP_SEED_MAX); // as launch_propagate
        return (a.n_times + tile - 1) / tile;
    }
    return (a.n_times + a.tile - 1) / a.tile;
}

// second stream for the eccentric-member launch of the row kernels (runs beside the near-circular bulk)
struct EccSide {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};

inline unsigned cgrid_y(unsigned n_times, unsigned tile) { return (n_times + tile - 1) / tile; }
// launch shape of the fast kernels on a uniform grid: time-segment lengths of the near-circular and the eccentric launch,
// which kernel family (= which window plan: 0 rows, 1 packed fp32 rows, 2 time-major tiles)
struct FastShape {
    unsigned tile_c = 0, tile_e = 0;
    bool packed32 = false; // the two-points-per-lane fp32 kernel (k_rows_fast32) takes the near-circular rows ...
    bool mixed32 = false;  // ... in its mixed-precision form
    int kind = 0;
};
// a wave's time window must stay short enough for the window-centred constants of the fast step (the node moves
// ~6e-5 rad/min: +-1,500 minutes keep it inside the 1/8-rad rotation tier)
unsigned fast_window_cap(double step)
{
    const double span = 3000.0 / std::max(std::fabs(step), 1e-9);
    return span >= 4.0e9 ? 0xffffffc0u : std::max(64u, (unsigned)span / 64u * 64u);
}
FastShape fast_shape_rows(const PropArgs &a, unsigned n_sgp4, unsigned n_circ)
{
    FastShape f;
    const unsigned cap = fast_window_cap(a.uniform_step);
    const unsigned n_ecc = n_sgp4 - n_circ;
    f.tile_c = std::min(rows_tile(n_sgp4, a.n_times, a.tile_forced), cap);
    // eccentric members: few rows.  Beside a bulk launch four times their size they take its segment length (their waves
    // fill in wherever the bulk leaves room; fewer, longer segments = fewer seed sincos); on their own, finer time segments
    // so that they still fill the chip
    f.tile_e = n_circ >= 4u * n_ecc ? f.tile_c : std::min(rows_tile(std::max(n_ecc, 1u), a.n_times, 256), cap);
    // packed fp32 kernel: a lane carries two grid points, a wave iteration 128 (windows shorter than that -- grid
    // steps beyond ~23 minutes -- keep the fp64 kernel with rounded stores)
    // (ECEF / geodetic: the Greenwich-angle table of k_rows_fast holds 256 points and is refilled inside the loop)
    if (a.delta || a.delta64) { // quasi-uniform grid: a wave stages its segment's deviations in LDS
        f.tile_c = std::min(f.tile_c, (unsigned)AZ_DELTA_SEG);
        f.tile_e = std::min(f.tile_e, (unsigned)AZ_DELTA_SEG);
    }
    f.packed32 = a.f32 && a.mode == AZ_OUT_TEME && a.arith32 != 2 && cap >= 128u && !a.delta_wide; // (the wide form: fp64 kernels)
    f.mixed32 = f.packed32 && a.arith32 == 0;
    if (f.packed32) f.tile_c = std::max(128u, f.tile_c / 128u * 128u);
    f.kind = f.packed32 ? 1 : 0;
    return f;
}
FastShape fast_shape_tiles(const PropArgs &a, unsigned n_rows)
{
    FastShape f;
    unsigned tile = std::min(rows_tile(std::max((n_rows + 15u) / 16u, 1u) * 16u, a.n_times, a.tile_forced), fast_window_cap(a.uniform_step));
    if (a.mode != AZ_OUT_TEME) tile = std::min(tile, (unsigned)AZ_TILE_SEG_MAX); // the Greenwich-angle table of a time segment is staged in LDS
    if (a.delta || a.delta64) tile = std::min(tile, (unsigned)AZ_DELTA_SEG); // ... and the deviations of a quasi-uniform grid
    f.tile_c = f.tile_e = tile;
    f.kind = 2;
    return f;
}

// the fast kernels, exactly uniform or quasi-uniform grid (DELTA instantiations: fast_step.h)
template <bool VEL, int FRAME, int SINK, bool ECC>
void launch_rows_fast(const PropArgs &a, dim3 grid, hipStream_t st)
{
    if (a.delta_wide) hipLaunchKernelGGL((k_rows_fast<VEL, FRAME, SINK, ECC, 2>), grid, dim3(64), 0, st, a);
    else if (a.delta) hipLaunchKernelGGL((k_rows_fast<VEL, FRAME, SINK, ECC, 1>), grid, dim3(64), 0, st, a);
    else hipLaunchKernelGGL((k_rows_fast<VEL, FRAME, SINK, ECC, 0>), grid, dim3(64), 0, st, a);
}
template <bool VEL, bool MIXED>
void launch_rows_fast32(const PropArgs &a, dim3 grid, hipStream_t st)
{
    if (a.delta) hipLaunchKernelGGL((k_rows_fast32<VEL, MIXED, true>), grid, dim3(64), 0, st, a);
    else hipLaunchKernelGGL((k_rows_fast32<VEL, MIXED, false>), grid, dim3(64), 0, st, a);
}
template <bool VEL, int FRAME>
void launch_tiles_fast(const PropArgs &a, dim3 grid, hipStream_t st)
{
    // (both quasi-uniform forms run the WIDE instantiation here: it is a superset of the tight form's corrections, and in this
    // kernel -- at its register limit -- the tight instantiation comes out with 28 B of scratch against 12 and measures 8 %
    // slower, 0.318-0.322 against 0.296-0.299 ms same run)
    if (a.delta64) hipLaunchKernelGGL((k_tiles_fast<VEL, FRAME, 2>), grid, dim3(1024), 0, st, a);
    else hipLaunchKernelGGL((k_tiles_fast<VEL, FRAME, 0>), grid, dim3(1024), 0, st, a);
}

template <bool VEL, int FRAME> // FRAME: 0 TEME, 1 ECEF, 2 geodetic (the generic kernels take frame / no frame and p.mode)
void launch_rows2(const PropArgs &a, dim3 grid, bool deep, hipStream_t st, const EccSide &side = EccSide(), const FastShape *shape = nullptr)
{
    constexpr bool FR = FRAME != 0;
    if (deep) {
        if (a.f32) hipLaunchKernelGGL((k_rows_deep<VEL, FR, AZ_SINK_F32>), grid, dim3(64), 0, st, a);
        else hipLaunchKernelGGL((k_rows_deep<VEL, FR, AZ_SINK_F64>), grid, dim3(64), 0, st, a);
    } else if (a.redo_items != nullptr && shape != nullptr) {
        // uniform grid: the branch-free kernels.  The near-circular bulk runs alone on the launch stream; beside it, on the
        // side stream, the eccentric members (few rows) and then the generic kernel over the redo list -- the windows the
        // plan's validation bounds rejected (static, known before any kernel runs: well under one per cent of the segments)
        // plus whatever the eccentric form's Newton validation handed over.  Nothing follows the bulk launch on its stream:
        // round 2's redo pass (12 us + two launch gaps per step) waited for it.
        PropArgs e = a, c = a;
        e.list = a.list + a.n_circ;
        e.n_list = a.n_list - a.n_circ;
        c.n_list = a.n_circ;
        e.tile = shape->tile_e;
        c.tile = shape->tile_c;
        const bool packed32 = shape->packed32 && !FR;
        // a row window: both launches are cut to the list slots it covers (their lists are in catalog order)
        const bool win = a.row_lo > 0 || a.row_hi < a.n_rows;
        unsigned n_e = e.n_list, n_c = c.n_list;
        if (win) {
            c.slot_lo = a.win_circ_lo; c.slot_hi = a.win_circ_hi; n_c = c.slot_hi - c.slot_lo;
            e.slot_lo = a.win_ecc_lo; e.slot_hi = a.win_ecc_hi; n_e = e.slot_hi - e.slot_lo;
            if (c.slot_hi == 0) c.n_list = n_c = 0; // (slot_hi = 0 means "whole list" to the kernels: an empty window launches nothing)
            if (e.slot_hi == 0) e.n_list = n_e = 0;
        }
        dim3 egrid((n_e + 7) / 8 * 8, (a.n_times + e.tile - 1) / e.tile);
        dim3 cgrid((n_c + 7) / 8 * 8, (a.n_times + c.tile - 1) / c.tile);
        // the generic pass: every workgroup takes items b, b + gridDim.x, ... and a quarter of each; enough workgroups that a
        // large catalog's rejected windows (1 % of 125,000 x 14 segments in config 5's share) do not queue up behind 1,024 waves
        dim3 rgrid(std::min(8192u, std::max(256u, (a.n_list * cgrid_y(a.n_times, shape->tile_c) + 63u) / 64u)), 4);
        // redo items carry (list slot, first, end): slots of the eccentric launch are offset into the common list
        e.redo_slot0 = a.n_circ;
        const bool beside = side.stream != nullptr && n_c;
        hipStream_t se = beside ? side.stream : st;
        if (beside) {
            (void)hipEventRecord(side.fork, st);
            (void)hipStreamWaitEvent(se, side.fork, 0);
        }
        if (n_e) {
            if (a.f32) launch_rows_fast<VEL, FRAME, AZ_SINK_F32, true>(e, egrid, se);
            else launch_rows_fast<VEL, FRAME, AZ_SINK_F64, true>(e, egrid, se);
        }
        if (a.f32) hipLaunchKernelGGL((k_rows<VEL, FR, AZ_SINK_F32, true>), rgrid, dim3(64), 0, se, a);
        else hipLaunchKernelGGL((k_rows<VEL, FR, AZ_SINK_F64, true>), rgrid, dim3(64), 0, se, a);
        if (n_c) {
            if (packed32 && shape->mixed32) launch_rows_fast32<VEL, true>(c, cgrid, st);
            else if (packed32) launch_rows_fast32<VEL, false>(c, cgrid, st);
            else if (a.f32) launch_rows_fast<VEL, FRAME, AZ_SINK_F32, false>(c, cgrid, st);
            else launch_rows_fast<VEL, FRAME, AZ_SINK_F64, false>(c, cgrid, st);
        }
        if (beside) {
            (void)hipEventRecord(side.join, se);
            (void)hipStreamWaitEvent(st, side.join, 0);
        }
    } else {
        if (a.f32) hipLaunchKernelGGL((k_rows<VEL, FR, AZ_SINK_F32>), grid, dim3(64), 0, st, a);
        else hipLaunchKernelGGL((k_rows<VEL, FR, AZ_SINK_F64>), grid, dim3(64), 0, st, a);
    }
}

// time-major output of the near-earth members on a uniform grid: 16-satellite tiles of lane = time waves (k_tiles_fast),
// then the generic kernel on whatever their validation rejected (24-byte pieces, row by row)
void launch_tiles(const PropArgs &a0, bool vel, hipStream_t st, const FastShape &shape)
{
    PropArgs a = a0;
    a.tile = shape.tile_c;
    const bool ecef = a.mode != AZ_OUT_TEME, geo = a.mode == AZ_OUT_GEODETIC;
    dim3 grid(((a.n_rows + 15u) / 16u + 7u) / 8u * 8u, (a.n_times + a.tile - 1) / a.tile); // tiles of 16 catalog rows
    if (geo) {
        if (vel) launch_tiles_fast<true, 2>(a, grid, st);
        else launch_tiles_fast<false, 2>(a, grid, st);
    } else if (ecef) {
        if (vel) launch_tiles_fast<true, 1>(a, grid, st);
        else launch_tiles_fast<false, 1>(a, grid, st);
    } else {
        if (vel) launch_tiles_fast<true, 0>(a, grid, st);
        else launch_tiles_fast<false, 0>(a, grid, st);
    }
    a.tm_rows = 1;
    dim3 rgrid(256, 4);
    if (ecef) {
        if (vel) hipLaunchKernelGGL((k_rows<true, true, AZ_SINK_F64, true>), rgrid, dim3(64), 0, st, a);
        else hipLaunchKernelGGL((k_rows<false, true, AZ_SINK_F64, true>), rgrid, dim3(64), 0, st, a);
    } else {
        if (vel) hipLaunchKernelGGL((k_rows<true, false, AZ_SINK_F64, true>), rgrid, dim3(64), 0, st, a);
        else hipLaunchKernelGGL((k_rows<false, false, AZ_SINK_F64, true>), rgrid, dim3(64), 0, st, a);
    }
}

// time-major output through k_cols_fast (lane = satellite): segment length of the main launch.  A wave is 64 catalog rows x one
// segment and runs for tens of microseconds, so a second, partly filled generation of waves would cost a large part of the
// step: when one generation of AZ_COLS_WAVES waves per SIMD can hold the whole grid (config 2: 211 row groups x 19 segments of 76
// points = 4,009 waves for 4,096 places), the segments are cut for exactly that; larger launches take 128-point segments.
unsigned cols_tile(unsigned n_rows, unsigned n_times, unsigned forced, double step)
{
    const unsigned cap = fast_window_cap(step);
    if (forced) return std::min(std::max(forced, 1u), std::min(std::max(n_times, 1u), cap));
    const unsigned groups = (n_rows + 63u) / 64u, places = 1024u * (unsigned)AZ_COLS_WAVES;
    const unsigned segs = std::max(1u, places / std::max(groups, 1u));
    unsigned tile = (n_times + segs - 1u) / segs;
    if (tile < 32u) tile = std::min(128u, std::max(n_times, 1u)); // (several generations anyway)
    return std::max(1u, std::min(tile, cap));
}
FastShape fast_shape_cols(const PropArgs &a, unsigned n_rows, unsigned n_sgp4, unsigned n_circ)
{
    FastShape f = fast_shape_rows(a, n_sgp4, n_circ); // tile_e: the eccentric members' own lane = time launch
    f.tile_e = std::min(rows_tile(std::max(n_sgp4 - n_circ, 1u), a.n_times, a.tile_forced ? a.tile_forced : 256u), fast_window_cap(a.uniform_step));
    if (a.delta || a.delta64) f.tile_e = std::min(f.tile_e, (unsigned)AZ_DELTA_SEG);
    f.tile_c = cols_tile(n_rows, a.n_times, a.tile_forced, a.uniform_step);
    f.packed32 = f.mixed32 = false;
    f.kind = 3;
    return f;
}
template <bool VEL, int FRAME>
void launch_cols_fast(const PropArgs &a, dim3 grid, hipStream_t st)
{
    if (a.delta64) hipLaunchKernelGGL((k_cols_fast<VEL, FRAME, 2>), grid, dim3(64), 0, st, a);
    else hipLaunchKernelGGL((k_cols_fast<VEL, FRAME, 0>), grid, dim3(64), 0, st, a);
}
// a: list = [class 0 | other classes], rowmap = the matching row map, tmp_pos / tmp_vel = the scratch array (deep-space rows
// already in flight on their stream; the caller has made `st` wait for them)
void launch_cols(const PropArgs &a0, bool vel, hipStream_t st, const FastShape &shape, const EccSide &side, unsigned ecc_row0)
{
    PropArgs a = a0;
    const int frame = a.mode == AZ_OUT_GEODETIC ? 2 : (a.mode != AZ_OUT_TEME ? 1 : 0);
    // eccentric members: their own lane = time launch into the scratch array, beside whatever precedes the main launch
    PropArgs e = a;
    e.list = a.list + a.n_circ;
    e.n_list = a.n_list - a.n_circ;
    e.redo_slot0 = a.n_circ;
    e.tile = shape.tile_e;
    e.pos = const_cast<double *>(a.tmp_pos);
    e.vel = vel ? const_cast<double *>(a.tmp_vel) : nullptr;
    e.rows_compact = 1;
    e.ecc_row0 = ecc_row0;
    if (e.n_list) {
        dim3 egrid((e.n_list + 7) / 8 * 8, (a.n_times + e.tile - 1) / e.tile);
        hipStream_t se = side.stream ? side.stream : st;
        if (se != st) {
            (void)hipEventRecord(side.fork, st);
            (void)hipStreamWaitEvent(se, side.fork, 0);
        }
        if (frame == 2) { if (vel) launch_rows_fast<true, 2, AZ_SINK_F64, true>(e, egrid, se); else launch_rows_fast<false, 2, AZ_SINK_F64, true>(e, egrid, se); }
        else if (frame == 1) { if (vel) launch_rows_fast<true, 1, AZ_SINK_F64, true>(e, egrid, se); else launch_rows_fast<false, 1, AZ_SINK_F64, true>(e, egrid, se); }
        else { if (vel) launch_rows_fast<true, 0, AZ_SINK_F64, true>(e, egrid, se); else launch_rows_fast<false, 0, AZ_SINK_F64, true>(e, egrid, se); }
        if (se != st) {
            (void)hipEventRecord(side.join, se);
            (void)hipStreamWaitEvent(st, side.join, 0);
        }
    }
    a.tile = shape.tile_c;
    a.ecc_row0 = ecc_row0;
    dim3 grid(((a.n_rows + 63u) / 64u + 7u) / 8u * 8u, (a.n_times + a.tile - 1) / a.tile);
    if (frame == 2) { if (vel) launch_cols_fast<true, 2>(a, grid, st); else launch_cols_fast<false, 2>(a, grid, st); }
    else if (frame == 1) { if (vel) launch_cols_fast<true, 1>(a, grid, st); else launch_cols_fast<false, 1>(a, grid, st); }
    else { if (vel) launch_cols_fast<true, 0>(a, grid, st); else launch_cols_fast<false, 0>(a, grid, st); }
    // the generic pass over the windows the plan rejected and the eccentric form's hand-overs: 24-byte pieces, row by row
    a.tm_rows = 1;
    dim3 rgrid(256, 4);
    if (frame) {
        if (vel) hipLaunchKernelGGL((k_rows<true, true, AZ_SINK_F64, true>), rgrid, dim3(64), 0, st, a);
        else hipLaunchKernelGGL((k_rows<false, true, AZ_SINK_F64, true>), rgrid, dim3(64), 0, st, a);
    } else {
        if (vel) hipLaunchKernelGGL((k_rows<true, false, AZ_SINK_F64, true>), rgrid, dim3(64), 0, st, a);
        else hipLaunchKernelGGL((k_rows<false, false, AZ_SINK_F64, true>), rgrid, dim3(64), 0, st, a);
    }
}

// returns the AZH_PATH_* bit of the kernel family it launched
unsigned launch_propagate(const PropArgs &a, int layout, bool vel, bool deep, hipStream_t st, const EccSide &side = EccSide(),
                          const FastShape *shape = nullptr)
{
    const bool frame = a.mode != AZ_OUT_TEME;
    if (use_rows(a, layout, deep)) {
        PropArgs b = a;
        b.tile = rows_tile(a.n_list, a.n_times, a.tile_forced);
        if (deep) b.tile = std::min(b.tile, 64u * (unsigned)AZ_DEEP_SEED_MAX); // k_rows_deep stages a segment's chunk seeds in LDS
        const unsigned n_slots = a.slot_hi ? a.slot_hi - a.slot_lo : a.n_list; // (a row window's share of a catalog-ordered list)
        dim3 grid((n_slots + 7) / 8 * 8, (a.n_times + b.tile - 1) / b.tile);
        if (deep || a.screen_target || a.inc == nullptr) b.redo_items = nullptr; // k_rows_fast: near-earth rows on a uniform grid
        if (a.screen_target) {
            if (deep) hipLaunchKernelGGL((k_rows_deep<false, false, AZ_SINK_SCREEN>), grid, dim3(64), 0, st, b);
            else hipLaunchKernelGGL((k_rows<false, false, AZ_SINK_SCREEN>), grid, dim3(64), 0, st, b);
        } else if (a.mode == AZ_OUT_GEODETIC) {
            if (vel) launch_rows2<true, 2>(b, grid, deep, st, side, shape);
            else launch_rows2<false, 2>(b, grid, deep, st, side, shape);
        } else if (frame) {
            if (vel) launch_rows2<true, 1>(b, grid, deep, st, side, shape);
            else launch_rows2<false, 1>(b, grid, deep, st, side, shape);
        } else {
            if (vel) launch_rows2<true, 0>(b, grid, deep, st, side, shape);
            else launch_rows2<false, 0>(b, grid, deep, st, side, shape);
        }
        return deep ? AZH_PATH_DEEP_ROWS : (!a.screen_target && b.redo_items != nullptr && shape != nullptr ? AZH_PATH_ROWS_FAST : AZH_PATH_ROWS_GENERIC);
    }
    if (a.screen_target) {
        // fused screen, lane = satellite (deep-space members; near-earth on very short grids)
        dim3 grid(((a.n_list + AZ_BLOCK - 1) / AZ_BLOCK + 7) / 8 * 8, (a.n_times + a.tile - 1) / a.tile);
        if (deep) hipLaunchKernelGGL((k_propagate<0, false, true, false, true>), grid, dim3(AZ_BLOCK), 0, st, a);
        else hipLaunchKernelGGL((k_propagate<0, false, false, false, true>), grid, dim3(AZ_BLOCK), 0, st, a);
        return AZH_PATH_LANE_SAT;
    }
    if (deep) {
        if (frame) launch_propagate2<true, true>(a, layout, vel, st);
        else launch_propagate2<true, false>(a, layout, vel, st);
    } else {
        if (frame) launch_propagate2<false, true>(a, layout, vel, st);
        else launch_propagate2<false, false>(a, layout, vel, st);
    }
    return AZH_PATH_LANE_SAT;
}

// upload times / offsets / mask and (if needed) build the GMST table
int32_t stage_inputs(azh_constellation *c, const double *times, size_t n_times, const double *offsets,
                     const uint8_t *mask, int mode, double reference_jd, hipStream_t st)
{
    if (n_times > 0xffffffffu) return AZ_ERR_VALUE;
    // The same inputs as the staged ones, byte for byte, on the same stream: everything derived from them -- the device copies,
    // the grid's classification and deviation tables, the Greenwich table, increments, records, window plans, resonance seeds --
    // is still in the handle.  (A benchmark loop, or a caller asking for positions and then velocities, repeats a grid; for a
    // single satellite x 1,440 the staging is two thirds of the call: 57 of 86 us.  The comparison reads 119 KB for config 2.)
    {
        const bool same = c->staged_valid && st == c->staged_stream && n_times == c->h_times.size() && mode == c->cached_mode &&
                          (mode == AZ_OUT_TEME || reference_jd == c->staged_ref_jd) &&
                          (offsets != nullptr) == c->have_offsets && (mask != nullptr) == c->have_mask &&
                          (n_times == 0 || memcmp(times, c->h_times.data(), sizeof(double) * n_times) == 0) &&
                          (!offsets || (c->h_offsets.size() == c->n && memcmp(offsets, c->h_offsets.data(), sizeof(double) * c->n) == 0)) &&
                          (!mask || (c->h_mask.size() == c->n && memcmp(mask, c->h_mask.data(), c->n) == 0));
        if (same) return AZ_OK;
        c->staged_valid = false;
    }
    if (c->d_times.ensure(n_times) != AZ_OK) return AZ_ERR_HIP;
    HIP_TRY(hipMemcpyAsync(c->d_times.p, times, sizeof(double) * n_times, hipMemcpyHostToDevice, st));
    c->have_offsets = offsets != nullptr;
    if (offsets) {
        if (c->d_offsets.ensure(c->n) != AZ_OK) return AZ_ERR_HIP;
        HIP_TRY(hipMemcpyAsync(c->d_offsets.p, offsets, sizeof(double) * c->n, hipMemcpyHostToDevice, st));
    }
    c->have_mask = mask != nullptr;
    if (mask) {
        if (c->d_mask.ensure(c->n) != AZ_OK) return AZ_ERR_HIP;
        HIP_TRY(hipMemcpyAsync(c->d_mask.p, mask, c->n, hipMemcpyHostToDevice, st));
    }
    if (mode != AZ_OUT_TEME) {
        if (c->d_sin.ensure(n_times) != AZ_OK || c->d_cos.ensure(n_times) != AZ_OK) return AZ_ERR_HIP;
        hipLaunchKernelGGL(k_gmst, dim3((unsigned)((n_times + 255) / 256)), dim3(256), 0, st, c->d_times.p,
                           (unsigned)n_times, reference_jd, c->d_sin.p, c->d_cos.p);
        HIP_TRY(hipGetLastError());
    }
    c->cached_n_times = (unsigned)n_times;
    c->cached_mode = mode;
    c->seeds_valid = false; // new time grid / offsets
    drop_graphs(c);          // (captured launch sets hold the old grid's buffers and shapes)
    for (auto &pl : c->plan) pl.valid = false;
    // uniform grid?  times[i] == times[0] + i*step up to the rounding of the grid itself: the fast step
    // (fast_step.h) then advances its carried angles by per-satellite constant rotations
    c->uniform_step = 0.0;
    c->delta_max = 0.0;
    c->delta_wide = false;
    c->grid_t0 = n_times ? times[0] : 0.0;
    if (n_times >= 2) {
        // ... or QUASI-uniform: within AZ_DELTA_MAX minutes of such a grid.  That is what the reference's own API hands over,
        // times = ((jd + fr) - reference_jd) * 1440 (api.py L300-302, Constellation.zig L266-269): jd + fr at 2.46e6 days is
        // quantised to 2^-31 day = 6.7e-7 min.  The fast kernels then run along the ideal grid and correct every point to its
        // actual time to first order in the deviation (fast_step.h, DELTA = 1).
        double t0 = times[0], step = (times[n_times - 1] - t0) / (double)(n_times - 1);
        double tmax = std::max(std::fabs(t0), std::fabs(times[n_times - 1]));
        const double tol = 4.0 * 2.220446049250313e-16 * std::max(tmax, std::fabs(step));
        bool uni = std::isfinite(step) && step != 0.0;
        double dmax = 0.0;
        for (size_t i = 1; uni && i < n_times; ++i) {
            dmax = std::max(dmax, std::fabs(times[i] - std::fma((double)i, step, t0)));
            uni = dmax <= AZ_DELTA_MAX; // (false for a NaN)
        }
        if (uni && dmax > tol) {
            // (zero-padded by one segment: a wave stages AZ_DELTA_SEG values from its segment start without a bounds test)
            c->h_delta.assign(n_times + AZ_DELTA_SEG, 0.0f);
            for (size_t i = 0; i < n_times; ++i) c->h_delta[i] = (float)(times[i] - std::fma((double)i, step, t0));
            if (c->d_delta.ensure(c->h_delta.size()) != AZ_OK) return AZ_ERR_HIP;
            // (pageable source, like `times` itself: the runtime has read it when the call returns)
            HIP_TRY(hipMemcpyAsync(c->d_delta.p, c->h_delta.data(), sizeof(float) * c->h_delta.size(), hipMemcpyHostToDevice, st));
            // ... and as fp64 for the tile kernel (k_tiles_fast stages doubles in both forms)
            c->h_delta64.assign(n_times + AZ_DELTA_SEG, 0.0);
            for (size_t i = 0; i < n_times; ++i) c->h_delta64[i] = times[i] - std::fma((double)i, step, t0);
            if (c->d_delta64.ensure(c->h_delta64.size()) != AZ_OK) return AZ_ERR_HIP;
            HIP_TRY(hipMemcpyAsync(c->d_delta64.p, c->h_delta64.data(), sizeof(double) * c->h_delta64.size(), hipMemcpyHostToDevice, st));
            c->delta_max = dmax * (1.0 + 1e-6) + 1e-12;
        }
        if (!uni && n_times >= 64) {
            // ... or uniform up to a JITTER of seconds (time stamps of a periodic process): least-squares line through the
            // points, deviations up to AZ_DELTA_WIDE_MAX minutes, staged as fp64 (fast_step.h, DELTA = 2)
            const double nn = (double)n_times, im = 0.5 * (nn - 1.0);
            double tm = 0.0;
            for (size_t i = 0; i < n_times; ++i) tm += times[i];
            tm /= nn;
            double sxy = 0.0, sxx = 0.0;
            for (size_t i = 0; i < n_times; ++i) {
                sxy += ((double)i - im) * (times[i] - tm);
                sxx += ((double)i - im) * ((double)i - im);
            }
            step = sxy / sxx;
            t0 = tm - step * im;
            bool ok = std::isfinite(step) && std::isfinite(t0) && step != 0.0;
            dmax = 0.0;
            for (size_t i = 0; ok && i < n_times; ++i) {
                dmax = std::max(dmax, std::fabs(times[i] - std::fma((double)i, step, t0)));
                ok = dmax <= AZ_DELTA_WIDE_MAX;
            }
            // (a jitter as large as the step itself is not "a uniform grid with jitter": such grids stay with the generic kernels)
            if (ok && dmax <= 0.5 * std::fabs(step)) {
                c->h_delta64.assign(n_times + AZ_DELTA_SEG, 0.0);
                for (size_t i = 0; i < n_times; ++i) c->h_delta64[i] = times[i] - std::fma((double)i, step, t0);
                if (c->d_delta64.ensure(c->h_delta64.size()) != AZ_OK) return AZ_ERR_HIP;
                HIP_TRY(hipMemcpyAsync(c->d_delta64.p, c->h_delta64.data(), sizeof(double) * c->h_delta64.size(), hipMemcpyHostToDevice, st));
                c->delta_max = dmax * (1.0 + 1e-6) + 1e-12;
                c->delta_wide = true;
                c->grid_t0 = t0;
                uni = true;
            }
        }
        if (uni) {
            if (c->d_inc.ensure((size_t)2 * AZ_INC_NUM * c->n_pad) != AZ_OK) return AZ_ERR_HIP;
            hipLaunchKernelGGL(k_prep_inc, dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, st, c->d_el, c->n, c->n_pad,
                               step, c->d_inc.p);
            HIP_TRY(hipGetLastError());
            if (c->d_fast_rec.ensure((size_t)FR_NUM * c->n_pad) != AZ_OK) return AZ_ERR_HIP;
            hipLaunchKernelGGL(k_prep_rec, dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, st, c->d_el, c->d_flags, c->n, c->n_pad,
                               c->d_inc.p, c->d_fast_rec.p);
            HIP_TRY(hipGetLastError());
            c->uniform_step = step;
        }
    }
    try {
        c->h_times.assign(times, times + n_times);
        if (offsets) c->h_offsets.assign(offsets, offsets + c->n);
        if (mask) c->h_mask.assign(mask, mask + c->n);
        c->staged_ref_jd = reference_jd;
        c->staged_stream = st;
        c->staged_valid = true;
    } catch (const std::bad_alloc &) {
        c->staged_valid = false; // (the next call stages again)
    }
    return AZ_OK;
}

// deep-space launch arguments: list slice, tile, and the resonance state at every tile start --
// computed once per (time grid, offsets, tile) and kept in the handle, like the reference keeps its
// carries (src/Constellation.zig L88, L294)
int32_t prepare_deep(azh_constellation *c, PropArgs &d, hipStream_t st, bool rows, bool beside_bulk = false)
{
    const unsigned n_times = d.n_times;
    d.list = c->d_list.p + (rows ? c->off_deep_cat : c->n_sgp4); // lane = time: catalog order; lane = satellite: grouped by branch
    d.n_list = c->n_sdp4;
    d.tile = auto_tile(c->n_sdp4, n_times, c->tile_sdp4, 8);
    // beside a near-earth row launch four times their size the deep-space rows take its segment length (fewer waves, fewer
    // set-ups; the chip is full either way); on their own -- deep-space-only catalogs, or ahead of the time-major tile
    // kernel -- the finer automatic segments (a single generation of long waves is latency-bound: 80 -> 140 us)
    d.tile_forced = c->tile_sdp4 ? c->tile_sdp4 : (rows && beside_bulk && c->n_sgp4 >= 4u * c->n_sdp4 ? rows_tile(c->n_sgp4, n_times, c->tile_sgp4) : 0u);
    // lane = time kernel: one state per 64-point chunk, taken at the chunk's grid point nearest to epoch
    const unsigned seed_tile = rows ? 64u : d.tile;
    const unsigned n_tiles = (n_times + seed_tile - 1) / seed_tile;
    if (!c->seeds_valid || c->seeds_tile != seed_tile || c->seeds_rows != rows) {
        if (c->capturing) return AZ_RC_EAGER;
        drop_graphs(c); // (captured launch sets hold the old seed table / its layout)
        if (c->d_seeds.ensure((size_t)n_tiles * 3 * c->n_sdp4) != AZ_OK) return AZ_ERR_HIP;
        if (c->d_node_cache.p == nullptr) {
            if (c->d_node_cache.ensure(3 * c->n_pad) != AZ_OK) return AZ_ERR_HIP;
            HIP_TRY(hipMemsetAsync(c->d_node_cache.p, 0, sizeof(double) * 3 * c->n_pad, st)); // atime = 0: start from epoch
        }
        hipLaunchKernelGGL(k_deep_seed, dim3((c->n_sdp4 + 63) / 64), dim3(64), 0, st, c->d_el, c->d_flags, c->n_pad,
                           d.list, c->n_sdp4, d.times, n_times, d.offsets, seed_tile, c->d_seeds.p, rows ? 1 : 0, c->d_node_cache.p);
        HIP_TRY(hipGetLastError());
        c->seeds_valid = true;
        c->seeds_tile = seed_tile;
        c->seeds_rows = rows;
    }
    d.seeds = c->d_seeds.p;
    return AZ_OK;
}

// the window plan of the staged grid for one launch shape (k_plan_windows): built on first use, kept until the next staging
int32_t ensure_plan(azh_constellation *c, PropArgs &a, const FastShape &shape, hipStream_t st)
{
    azh_constellation::FastPlan &pl = c->plan[shape.kind];
    const unsigned n_list = a.n_list;
    const unsigned n_seg = (a.n_times + std::min(shape.tile_c, shape.tile_e) - 1) / std::min(shape.tile_c, shape.tile_e);
    if (!pl.valid || pl.tile_c != shape.tile_c || pl.tile_e != shape.tile_e || pl.n_list != n_list || pl.mixed32 != shape.mixed32) {
        if (c->capturing) return AZ_RC_EAGER;
        drop_graphs(c); // (captured launch sets hold this plan's buffers and counters)
        // static items: at most one per (slot, segment); dynamic ones: only waves of eccentric members file them, at most one
        // per 64-point iteration
        const size_t items = (size_t)n_list * n_seg + ((size_t)a.n_times + 63) / 64 * (c->n_sgp4 - c->n_circ);
        if (pl.redo.cap < 4 + 3 * items || pl.win.cap < (size_t)n_list * n_seg * AZ_PLAN_NUM) HIP_TRY(hipStreamSynchronize(st)); // (launches in flight use the old buffers)
        if (pl.redo.ensure(4 + 3 * items) != AZ_OK || pl.win.ensure((size_t)n_list * n_seg * AZ_PLAN_NUM) != AZ_OK ||
            pl.flag.ensure((size_t)n_list * n_seg) != AZ_OK)
            return AZ_ERR_HIP;
        HIP_TRY(hipMemsetAsync(pl.redo.p, 0, 4 * sizeof(unsigned), st));
        PlanArgs q{};
        q.el = a.el; q.flags = a.flags; q.n_pad = a.n_pad; q.list = a.list; q.n_list = n_list; q.n_circ = a.n_circ;
        q.n_times = a.n_times; q.tile_c = shape.tile_c; q.tile_e = shape.tile_e; q.by_flags = shape.kind == 2 ? 1u : 0u;
        q.times = a.times; q.offsets = a.offsets; q.inc = a.inc; q.step = a.uniform_step; q.dt_mult = shape.kind == 1 ? 128.0 : 64.0;
        q.delta_max = (a.delta || a.delta64) ? a.delta_max : 0.0;
        q.grid_t0 = a.grid_t0;
        q.f32_mixed = shape.mixed32 ? 1u : 0u;
        q.win = pl.win.p; q.flag = pl.flag.p;
        q.redo_static = pl.redo.p + 2; q.redo_c0 = pl.redo.p; q.redo_c1 = pl.redo.p + 1; q.redo_items = pl.redo.p + 4;
        q.g = a.g;
        hipLaunchKernelGGL(k_plan_windows, dim3((n_list + 255) / 256, n_seg), dim3(256), 0, st, q);
        hipLaunchKernelGGL(k_plan_arm, dim3(1), dim3(64), 0, st, q.redo_static, q.redo_c0, q.redo_c1);
        HIP_TRY(hipGetLastError());
        pl.valid = true;
        pl.tile_c = shape.tile_c; pl.tile_e = shape.tile_e; pl.n_list = n_list; pl.n_seg = n_seg; pl.parity = 0; pl.mixed32 = shape.mixed32;
    }
    a.plan_win = pl.win.p;
    a.plan_flag = pl.flag.p;
    a.plan_stride = n_list;
    a.redo_static = pl.redo.p + 2;
    a.redo_count = pl.redo.p + pl.parity;
    a.redo_next = pl.redo.p + (pl.parity ^ 1u);
    a.redo_items = pl.redo.p + 4;
    pl.parity ^= 1u;
    return AZ_OK;
}

void drop_graphs(azh_constellation *c)
{
    for (auto &g : c->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    c->graphs.clear();
}
unsigned plan_sig(const azh_constellation *c)
{
    unsigned s = 0;
    for (unsigned k = 0; k < 4; ++k) s |= (c->plan[k].parity & 1u) << k;
    return s;
}

// the launches proper; inputs already staged on the device
int32_t launch_all(azh_constellation *c, double *d_pos, double *d_vel, int layout, size_t stride, uint8_t *d_err,
                   hipStream_t st, int f32 = 0, size_t row_lo = 0, size_t row_hi = ~(size_t)0)
{
    const unsigned n_times = c->cached_n_times;
    if (n_times == 0) return AZ_OK;
    row_hi = std::min(row_hi, c->n);
    if (row_lo >= row_hi) return AZ_OK;
    if (stride == 0) stride = c->n;
    if (layout == AZ_LAYOUT_TIME_MAJOR && stride < c->n) return AZ_ERR_VALUE;

    PropArgs a{};
    a.el = c->d_el;
    a.flags = c->d_flags;
    a.n_pad = c->n_pad;
    a.times = c->d_times.p;
    a.n_times = n_times;
    a.offsets = c->have_offsets ? c->d_offsets.p : nullptr;
    a.pos = d_pos;
    a.vel = d_vel;
    a.sin_g = c->d_sin.p;
    a.cos_g = c->d_cos.p;
    a.mask = c->have_mask ? c->d_mask.p : nullptr;
    a.err = d_err;
    a.stride_sats = stride;
    a.mode = c->cached_mode;
    a.f32 = f32;
    a.arith32 = f32 ? c->f32_mode : 2;
    a.g = c->g;
    a.uniform_step = c->fast_path ? c->uniform_step : 0.0;
    a.inc = (a.uniform_step != 0.0) ? c->d_inc.p : nullptr;
    a.fast_rec = (a.uniform_step != 0.0) ? c->d_fast_rec.p : nullptr;
    a.delta = (a.uniform_step != 0.0 && c->delta_max > 0.0 && !c->delta_wide) ? c->d_delta.p : nullptr;
    a.delta64 = (a.uniform_step != 0.0 && c->delta_max > 0.0) ? c->d_delta64.p : nullptr; // (tight grids: the tile kernel's copy)
    a.delta_wide = c->delta_wide ? 1 : 0;
    a.grid_t0 = c->grid_t0;
    a.delta_max = c->delta_max;
    a.grid_exact_uniform = (c->uniform_step != 0.0 && c->delta_max == 0.0) ? 1 : 0;
    a.row_lo = (unsigned)row_lo;
    a.row_hi = (unsigned)row_hi;
    // a row window: the slots it covers in the catalog-ordered sub-lists (the lane = time launches are cut to them)
    const bool windowed = row_lo > 0 || row_hi < c->n;
    auto slots_of = [&](unsigned off, unsigned cnt, unsigned &lo, unsigned &hi) {
        const unsigned *b = c->h_list.data() + off, *e = b + cnt;
        lo = (unsigned)(std::lower_bound(b, e, (unsigned)row_lo) - b);
        hi = (unsigned)(std::lower_bound(b, e, (unsigned)row_hi) - b);
    };
    unsigned deep_lo = 0, deep_hi = 0;
    if (windowed) {
        slots_of(c->off_circ, c->n_circ, a.win_circ_lo, a.win_circ_hi);
        slots_of(c->off_circ + c->n_circ, c->n_sgp4 - c->n_circ, a.win_ecc_lo, a.win_ecc_hi);
        slots_of(c->off_deep_cat, c->n_sdp4, deep_lo, deep_hi);
    }

    unsigned path = 0;
    if (c->timing) HIP_TRY(hipEventRecord(c->ev_t0, st));
    if (d_err) HIP_TRY(hipMemsetAsync(d_err + row_lo * (size_t)n_times, 0, (row_hi - row_lo) * (size_t)n_times, st));
    // time-major output on a (quasi-)uniform grid: the near-earth members take the 16-satellite tile kernel, with or without a
    // satellite mask (round 5).  (Irregular grids stay on the lane = satellite kernel: a generic-step tile kernel was built and
    // measured in round 4 -- 0.40 ms against k_propagate's 0.35 -- see profiles/r04_experiments.txt.)
    const bool tiles = c->n_sgp4 > 0 && c->tile_kernel && a.inc != nullptr && layout == AZ_LAYOUT_TIME_MAJOR && !f32 &&
                       n_times >= 64 && // (masked launches too, round 5: the flush has write-enable bits per piece)
                       (size_t)stride * 64 * 24 < 0xf0000000ull; // (k_tiles_fast: 32-bit byte offsets inside a block of 64 time rows)
    // ... or, when switched on (azh_set_tile_kernel(c, 2) / ASTROZ_AMD_COLS), the lane = satellite kernel k_cols_fast
    const bool cols = tiles && c->cols_kernel != 0 && a.mask == nullptr;
    const unsigned n_ecc = c->n_sgp4 - c->n_circ;
    // compact satellite-major scratch rows of a time-major launch: [deep-space list slots | (k_cols_fast) eccentric list slots]
    const size_t scratch_rows = (size_t)c->n_sdp4 + (cols ? n_ecc : 0u);
    const size_t scratch_per = scratch_rows * n_times * 3, scratch_words = f32 ? (scratch_per + 1) / 2 : scratch_per; // in doubles
    auto ensure_scratch = [&](hipStream_t user) -> int32_t {
        if (c->capturing && c->d_deep_tmp.cap < scratch_words * (d_vel ? 2 : 1)) return AZ_RC_EAGER;
        if (c->d_deep_tmp.cap < scratch_words * (d_vel ? 2 : 1)) {
            HIP_TRY(hipStreamSynchronize(user));
            drop_graphs(c); // (captured launch sets hold the old scratch pointer: ADVICE r05)
        }
        if (c->d_deep_tmp.ensure(scratch_words * (d_vel ? 2 : 1)) != AZ_OK) return AZ_ERR_HIP;
        a.tmp_pos = c->d_deep_tmp.p;
        a.tmp_vel = d_vel ? c->d_deep_tmp.p + scratch_words : nullptr;
        return AZ_OK;
    };
    const bool fork = c->n_sdp4 > 0;
    if (cols && !fork && n_ecc)
        if (int32_t rc = ensure_scratch(st); rc != AZ_OK) return rc;
    if (fork) {
        // deep-space rows on their own stream, concurrent with the near-earth launch
        HIP_TRY(hipEventRecord(c->ev_fork, st));
        HIP_TRY(hipStreamWaitEvent(c->s_deep, c->ev_fork, 0));
        PropArgs d = a;
        const bool deep_rows = use_rows(d, layout, true);
        const bool deep_skip = windowed && deep_rows && deep_hi == deep_lo; // (no deep-space row inside the window)
        if (windowed && deep_rows) { d.slot_lo = deep_lo; d.slot_hi = deep_hi; }
        if (int32_t rc = prepare_deep(c, d, c->s_deep, deep_rows, /*beside_bulk=*/!tiles && a.inc != nullptr); rc != AZ_OK) return rc;
        if (deep_rows && layout == AZ_LAYOUT_TIME_MAJOR) {
            // time-major: the rows go satellite-major into a compact scratch array (row = list slot), then one pure-memory
            // kernel writes them out as time-major runs (k_deep_transpose)
            if (int32_t rc = ensure_scratch(c->s_deep); rc != AZ_OK) return rc;
            PropArgs t = d;
            t.pos = const_cast<double *>(a.tmp_pos);
            t.vel = const_cast<double *>(a.tmp_vel);
            t.rows_compact = 1;
            if (!deep_skip) path |= launch_propagate(t, AZ_LAYOUT_SAT_MAJOR, d_vel != nullptr, true, c->s_deep);
            HIP_TRY(hipGetLastError());
            dim3 tg((c->n_sdp4 + AZ_TR_ROWS - 1) / AZ_TR_ROWS, (n_times + 63) / 64);
            if (tiles || deep_skip) { /* the tile kernel copies the scratch rows through its own tiles / no deep-space row in the window */ }
            else if (f32)
                hipLaunchKernelGGL((k_deep_transpose<float>), tg, dim3(192), 0, c->s_deep, reinterpret_cast<const float *>(t.pos),
                                   reinterpret_cast<const float *>(t.vel), reinterpret_cast<float *>(d_pos), reinterpret_cast<float *>(d_vel),
                                   d.list, c->n_sdp4, n_times, stride, d.mask, d.row_lo, d.row_hi);
            else
                hipLaunchKernelGGL((k_deep_transpose<double>), tg, dim3(192), 0, c->s_deep, t.pos, t.vel, d_pos, d_vel, d.list, c->n_sdp4,
                                   n_times, stride, d.mask, d.row_lo, d.row_hi);
        } else if (!deep_skip) {
            path |= launch_propagate(d, layout, d_vel != nullptr, true, c->s_deep);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->ev_join, c->s_deep));
    }
    if (c->n_sgp4 > 0) {
        a.list = c->d_list.p;
        a.n_list = c->n_sgp4;
        a.tile = auto_tile(c->n_sgp4, n_times, c->tile_sgp4, 8);
        a.tile_forced = c->tile_sgp4;
        // The tile kernel copies the deep-space rows out of the scratch array: they go FIRST and it follows them.  (Beside
        // each other they would not overlap well anyway: a tile workgroup is 16 waves of 128 VGPRs and needs a whole CU's
        // register files at once, so any small workgroup of another kernel resident on the CU keeps it out.)
        if (tiles && fork) HIP_TRY(hipStreamWaitEvent(st, c->ev_join, 0));
        if (tiles && !cols) a.list = c->d_list.p + c->off_cat; // plain catalog order; the redo items index this list
        FastShape shape;
        const bool fast = a.inc != nullptr && (tiles || use_rows(a, layout, false));
        if (fast) {
            // uniform grid: every near-earth member -> the branch-free kernels (near-circular or eccentric Kepler form by
            // class), windows prepared and validated once per staged grid by the plan; what the plan rejects and what the
            // eccentric form's Newton validation hands over comes back through the redo list
            if (!tiles || cols) a.list = c->d_list.p + c->off_circ; // [class 0 | other classes], catalog order inside each
            a.n_list = c->n_sgp4;
            a.n_circ = c->n_circ;
            a.rowmap = c->d_list.p + (cols ? c->off_rowmap2 : c->off_rowmap);
            a.n_rows = (unsigned)c->n;
            shape = cols ? fast_shape_cols(a, (unsigned)c->n, c->n_sgp4, c->n_circ)
                         : (tiles ? fast_shape_tiles(a, (unsigned)c->n) : fast_shape_rows(a, c->n_sgp4, c->n_circ));
            if (int32_t rc = ensure_plan(c, a, shape, st); rc != AZ_OK) return rc;
        }
        if (a.n_list > 0 && cols) {
            launch_cols(a, d_vel != nullptr, st, shape, EccSide{c->s_ecc, c->ev_fork2, c->ev_join2}, c->n_sdp4);
            path |= AZH_PATH_COLS_FAST;
        } else if (a.n_list > 0 && tiles) {
            launch_tiles(a, d_vel != nullptr, st, shape);
            path |= AZH_PATH_TILES_FAST;
        } else if (a.n_list > 0) {
            path |= launch_propagate(a, layout, d_vel != nullptr, false, st, EccSide{c->s_ecc, c->ev_fork2, c->ev_join2}, fast ? &shape : nullptr);
        }
        if ((a.delta || a.delta64) && (path & (AZH_PATH_TILES_FAST | AZH_PATH_ROWS_FAST | AZH_PATH_COLS_FAST))) path |= AZH_PATH_QUASI_UNIFORM;
        HIP_TRY(hipGetLastError());
    }
    if (c->n_bad > 0) {
        hipLaunchKernelGGL(k_fill_bad, dim3((n_times + 255) / 256, c->n_bad), dim3(256), 0, st,
                           c->d_list.p + c->n_sgp4 + c->n_sdp4, c->n_bad, c->d_flags, n_times, d_pos, d_vel, d_err,
                           c->have_mask ? c->d_mask.p : nullptr, layout, stride, f32, a.row_lo, a.row_hi);
        HIP_TRY(hipGetLastError());
    }
    if (fork && !tiles) HIP_TRY(hipStreamWaitEvent(st, c->ev_join, 0));
    c->last_path = path;
    if (c->timing) {
        HIP_TRY(hipEventRecord(c->ev_t1, st));
        c->timed = true;
    }
    return AZ_OK;
}

// launch_all for the cached-input entry points, through the graph cache.  The first call with a key runs eagerly (it may
// have to build a plan, seeds or a scratch buffer); the second is captured -- the launch set on the caller's stream and the
// handle's side streams, joined back by the same events as in the eager form -- and every later one is one hipGraphLaunch.
int32_t launch_cached(azh_constellation *c, double *d_pos, double *d_vel, int layout, size_t stride, uint8_t *d_err, hipStream_t st,
                      int f32 = 0, size_t row_lo = 0, size_t row_hi = ~(size_t)0)
{
    // (a small handle launches on its device's shared stream: no capture there)
    if (!c->graphs_on || c->timing || c->cached_n_times == 0 || !c->own_stream) return launch_all(c, d_pos, d_vel, layout, stride, d_err, st, f32, row_lo, row_hi);
    const unsigned sig = plan_sig(c);
    azh_constellation::LaunchGraph *hit = nullptr, *pending = nullptr;
    for (auto &g : c->graphs)
        if (g.pos == d_pos && g.vel == d_vel && g.err == d_err && g.layout == layout && g.f32 == f32 && g.stride == stride &&
            g.row_lo == row_lo && g.row_hi == row_hi && g.st == st) {
            if (g.fails >= 2) return launch_all(c, d_pos, d_vel, layout, stride, d_err, st, f32, row_lo, row_hi); // (uncapturable key)
            if (g.exec && g.sig_before == sig) { hit = &g; break; }
            if (!g.exec && g.sig_before == sig) pending = &g;
        }
    if (hit) {
        HIP_TRY(hipGraphLaunch(hit->exec, st));
        for (unsigned k = 0; k < 4; ++k) c->plan[k].parity = (hit->sig_after >> k) & 1u;
        c->last_path = hit->path;
        return AZ_OK;
    }
    if (!pending) {
        // first sight of this key: eager, remembered
        if (c->graphs.size() >= 24) drop_graphs(c); // (a caller cycling through many output buffers: start over)
        const int32_t rc = launch_all(c, d_pos, d_vel, layout, stride, d_err, st, f32, row_lo, row_hi);
        if (rc == AZ_OK) {
            try {
                c->graphs.push_back({d_pos, d_vel, d_err, layout, f32, stride, row_lo, row_hi, st, sig, plan_sig(c), c->last_path, 1u, 0u, nullptr, nullptr});
            } catch (const std::bad_alloc &) {
            }
        }
        return rc;
    }
    // second sight: capture
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        return launch_all(c, d_pos, d_vel, layout, stride, d_err, st, f32, row_lo, row_hi);
    }
    unsigned parity0[4];
    for (unsigned k = 0; k < 4; ++k) parity0[k] = c->plan[k].parity;
    c->capturing = true;
    const int32_t rc = launch_all(c, d_pos, d_vel, layout, stride, d_err, st, f32, row_lo, row_hi);
    c->capturing = false;
    hipGraph_t graph = nullptr;
    const hipError_t e_end = hipStreamEndCapture(st, &graph);
    hipGraphExec_t exec = nullptr;
    if (rc != AZ_OK || e_end != hipSuccess || !graph || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (graph) (void)hipGraphDestroy(graph);
        for (unsigned k = 0; k < 4; ++k) c->plan[k].parity = parity0[k]; // (nothing ran)
        // the entry stays: one more attempt when this parity comes round again (a plan or the seeds may have had to be built
        // first), then the key runs eagerly for good -- no BeginCapture / EndCapture pair on every other call
        pending->fails += 1;
        if (rc != AZ_OK && rc != AZ_RC_EAGER) return rc;
        return launch_all(c, d_pos, d_vel, layout, stride, d_err, st, f32, row_lo, row_hi);
    }
    pending->graph = graph;
    pending->exec = exec;
    pending->sig_after = plan_sig(c);
    pending->path = c->last_path;
    HIP_TRY(hipGraphLaunch(exec, st));
    return AZ_OK;
}

// Pinned host memory the library hands out for RESULTS (azh_host_alloc / azh_host_free): a host-returning call whose output
// arrays live here needs no staging hop and no copy threads -- the device-to-host DMA lands in the caller's array at the link
// rate (57 GB/s: config 2's 932 MB in 16.4 ms; through the pinned staging slots into fresh pageable arrays: 20-25 ms; straight
// into fresh pageable arrays: 55 ms).  Pinning is the expensive part (hundreds of milliseconds per GB), so freed blocks stay
// pinned in a pool and the next result of that size takes them over; the pool keeps at most ASTROZ_AMD_HOST_POOL_MB (default
// 4,096) of free blocks.  What the reference's own Python layer does at this point is numpy.empty per call (api.py L304-314):
// the callee allocates -- here too, from memory the DMA engines can write.
class HostPool {
  public:
    int32_t alloc(size_t bytes, void **out)
    {
        if (!out) return AZ_ERR_NULL_POINTER;
        *out = nullptr;
        const size_t need = std::max<size_t>((bytes + kGrain - 1) / kGrain * kGrain, kGrain);
        {
            std::lock_guard<std::mutex> lock(mu_);
            auto it = free_.lower_bound(need);
            if (it != free_.end() && it->first <= need + need / 4 + kGrain) { // (a block up to 25 % larger serves)
                void *q = it->second;
                free_bytes_ -= it->first;
                live_[static_cast<char *>(q)] = it->first;
                free_.erase(it);
                *out = q;
                return AZ_OK;
            }
        }
        void *q = nullptr;
        if (hipHostMalloc(&q, need, hipHostMallocPortable) != hipSuccess || !q) {
            (void)hipGetLastError();
            trim(0); // give the runtime the pool's free blocks back and try once more
            if (!hip_ok(hipHostMalloc(&q, need, hipHostMallocPortable), "hipHostMalloc(result)")) return AZ_ERR_ALLOC_FAILED;
        }
        std::lock_guard<std::mutex> lock(mu_);
        live_[static_cast<char *>(q)] = need;
        *out = q;
        return AZ_OK;
    }
    void release(void *q)
    {
        if (!q) return;
        {
            std::lock_guard<std::mutex> lock(mu_);
            auto it = live_.find(static_cast<char *>(q));
            if (it == live_.end()) return; // not ours
            free_.emplace(it->second, q);
            free_bytes_ += it->second;
            live_.erase(it);
        }
        trim(cap());
    }
    // [q, q + len) lies inside one live block
    bool owns(const void *q, size_t len)
    {
        if (!q) return false;
        std::lock_guard<std::mutex> lock(mu_);
        auto it = live_.upper_bound(const_cast<char *>(static_cast<const char *>(q)));
        if (it == live_.begin()) return false;
        --it;
        return static_cast<const char *>(q) + len <= it->first + it->second;
    }
    void stats(size_t *live_bytes, size_t *free_bytes)
    {
        std::lock_guard<std::mutex> lock(mu_);
        size_t l = 0;
        for (auto &kv : live_) l += kv.second;
        if (live_bytes) *live_bytes = l;
        if (free_bytes) *free_bytes = free_bytes_;
    }
    void trim(size_t keep)
    {
        std::vector<void *> drop;
        {
            std::lock_guard<std::mutex> lock(mu_);
            while (free_bytes_ > keep && !free_.empty()) {
                auto it = std::prev(free_.end()); // largest first
                drop.push_back(it->second);
                free_bytes_ -= it->first;
                free_.erase(it);
            }
        }
        for (void *q : drop) (void)hipHostFree(q);
    }

  private:
    static constexpr size_t kGrain = size_t(2) << 20;
    static size_t cap()
    {
        static const size_t c = [] {
            const char *e = getenv("ASTROZ_AMD_HOST_POOL_MB");
            return (size_t)(e ? std::max(0L, atol(e)) : 4096L) << 20;
        }();
        return c;
    }
    std::mutex mu_;
    std::map<char *, size_t> live_;
    std::multimap<size_t, void *> free_;
    size_t free_bytes_ = 0;
};
HostPool &host_pool()
{
    static HostPool *p = new HostPool(); // (never destroyed: at process exit the HIP runtime may already be gone)
    return *p;
}

// Device -> caller's host arrays.  A D2H straight into pageable memory runs at the PCIe rate only when the runtime has seen
// the destination before: config 2's 932 MB take 17 ms into arrays a previous call wrote, but 55 ms into FRESH ones
// (numpy.empty every call -- what SatrecArray.sgp4 does), and mapping the pages beforehand does not help (measured with 0-16
// page-touch threads: 55 ms throughout; tools/host_path_probe3.py) -- the cost is the runtime's first-time pinning of the
// range.  So the copy goes through pinned staging slots that live in the handle (device -> slot at the link rate, slots
// cycling) and a few host threads move each landed chunk into the caller's array while the next chunk is on the link.
// K threads that copy [src, src + len) to dst in K pieces; jobs are handed over one chunk at a time
class CopyPool {
  public:
    explicit CopyPool(unsigned k)
    {
        for (unsigned i = 0; i < k; ++i) {
            try {
                th_.emplace_back([this, i] { work(i); });
            } catch (const std::system_error &) {
                break;
            }
        }
    }
    unsigned size() const { return (unsigned)th_.size(); }
    // start copying; returns a ticket
    size_t submit(char *dst, const char *src, size_t len)
    {
        std::lock_guard<std::mutex> lock(mu_);
        jobs_.push_back({dst, src, len, (unsigned)th_.size()});
        cv_.notify_all();
        return jobs_.size() - 1;
    }
    void wait(size_t ticket)
    {
        std::unique_lock<std::mutex> lock(mu_);
        done_cv_.wait(lock, [&] { return jobs_[ticket].left == 0; });
    }
    ~CopyPool()
    {
        {
            std::lock_guard<std::mutex> lock(mu_);
            stop_ = true;
            cv_.notify_all();
        }
        for (auto &t : th_) t.join();
    }

  private:
    struct Job {
        char *dst;
        const char *src;
        size_t len;
        unsigned left;
    };
    void work(unsigned me)
    {
        size_t next = 0;
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lock(mu_);
                cv_.wait(lock, [&] { return stop_ || next < jobs_.size(); });
                if (next >= jobs_.size()) return; // (stop, nothing left)
                j = jobs_[next];
            }
            // (ceil: with floor(len / k) a multiple of 64 and len % k != 0, k pieces of the floor would stop short of len)
            const size_t k = th_.size(), piece = ((j.len + k - 1) / k + 63) & ~size_t(63);
            const size_t lo = std::min(j.len, piece * me), hi = std::min(j.len, piece * (me + 1));
            if (hi > lo) memcpy(j.dst + lo, j.src + lo, hi - lo);
            {
                std::lock_guard<std::mutex> lock(mu_);
                if (--jobs_[next].left == 0) done_cv_.notify_all();
            }
            ++next;
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    std::deque<Job> jobs_;
    std::vector<std::thread> th_;
    bool stop_ = false;
};
std::atomic<int> g_host_copy_threads{-1}; // azh_set_host_copy_threads: -1 automatic, 0 = plain pageable copies

// arrays[k] (host) <- src[k] (device), bytes[k] each, in order, on `st`.  threads: host copy threads (0: the calling thread
// moves the chunks itself -- one copier thread per device of a group).  The data has landed when this returns.
int32_t copy_back_staged(HostStager &hs, void *const *dst, const void *const *src, const size_t *bytes, int n_arrays, hipStream_t st,
                         unsigned threads)
{
    if (int32_t rc = hs.ensure(); rc != AZ_OK) return rc;
    struct Chunk {
        char *dst;
        const char *src;
        size_t len;
    };
    std::vector<Chunk> chunks;
    for (int k = 0; k < n_arrays; ++k)
        for (size_t o = 0; o < bytes[k]; o += kStageChunk)
            chunks.push_back({static_cast<char *>(dst[k]) + o, static_cast<const char *>(src[k]) + o, std::min(kStageChunk, bytes[k] - o)});
    CopyPool pool(threads);
    const bool pooled = pool.size() > 0;
    std::vector<size_t> ticket(chunks.size(), 0);
    auto land = [&](size_t i) -> int32_t { // chunk i has been issued into slot i % S: wait for it, move it out
        HIP_TRY(hipEventSynchronize(hs.ev[i % kStageSlots]));
        if (pooled) ticket[i] = pool.submit(chunks[i].dst, static_cast<const char *>(hs.slot[i % kStageSlots]), chunks[i].len);
        else memcpy(chunks[i].dst, hs.slot[i % kStageSlots], chunks[i].len);
        return AZ_OK;
    };
    for (size_t i = 0; i < chunks.size(); ++i) {
        if (pooled && i >= (size_t)kStageSlots) pool.wait(ticket[i - kStageSlots]); // the slot is free again
        HIP_TRY(hipMemcpyAsync(hs.slot[i % kStageSlots], chunks[i].src, chunks[i].len, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(hs.ev[i % kStageSlots], st));
        // (inline mode: chunk i - 1 is moved out while chunk i is on the link; with S >= 2 slots its slot is not reused before)
        if (i >= 1)
            if (int32_t rc = land(i - 1); rc != AZ_OK) return rc;
    }
    if (!chunks.empty())
        if (int32_t rc = land(chunks.size() - 1); rc != AZ_OK) return rc;
    if (pooled)
        for (size_t i = chunks.size() > (size_t)kStageSlots ? chunks.size() - kStageSlots : 0; i < chunks.size(); ++i) pool.wait(ticket[i]);
    return AZ_OK;
}

unsigned host_copy_threads()
{
    const int want = g_host_copy_threads.load(std::memory_order_relaxed);
    if (want >= 0) return (unsigned)want;
    return std::min(6u, std::max(1u, std::thread::hardware_concurrency() / 2));
}

// one satellite x n times, times and outputs in device memory: long series of a near-earth member take k_one_fast (every wave
// tries the branch-free step on its 1,024 points) with k_one_satellite behind it for what that hands over; shorter ones, the
// c_api's interleaved layout and deep-space members take k_one_satellite (one generic step per point) directly
#ifndef AZ_ONE_FAST
#define AZ_ONE_FAST 1
#endif
// (a wave of k_one_fast works through its 1,024 points one 64-point iteration after the other: ~40 us whatever the length of
// the series, and a memset and a second launch come with it -- below about a million points the generic kernel, one point per
// lane and all of them at once, is done sooner: 10^6 points 32 us against 28)
constexpr size_t kOneFastMin = size_t(1) << 20;
int32_t launch_one(azh_constellation *c, size_t sat, const double *d_t, size_t n, double *d_p, double *d_v, unsigned char *d_e,
                   int interleaved, hipStream_t st)
{
    const unsigned f = sat < c->h_flags.size() ? c->h_flags[sat] : ~0u;
    const bool fast = AZ_ONE_FAST && !interleaved && n >= kOneFastMin && f != ~0u && AZ_FLAG_ERR(f) == 0 && !(f & AZ_FLAG_DEEP);
    c->one_segments = 0;
    if (fast) {
        const unsigned n_seg = (unsigned)((n + AZ_ONE_SEG - 1) / AZ_ONE_SEG);
        const unsigned cap = (n_seg + AZ_ONE_LISTS - 1) / AZ_ONE_LISTS; // capacity of each of the AZ_ONE_LISTS hand-over lists
        if (c->d_one_items.ensure((size_t)AZ_ONE_HEAD + (size_t)cap * AZ_ONE_LISTS) != AZ_OK) return AZ_ERR_HIP;
        c->one_segments = n_seg;
        HIP_TRY(hipMemsetAsync(c->d_one_items.p, 0, sizeof(unsigned) * AZ_ONE_HEAD, st));
        if (d_v)
            hipLaunchKernelGGL((k_one_fast<true>), dim3(n_seg), dim3(64), 0, st, c->d_el, c->d_flags, c->n_pad, (unsigned)sat, d_t,
                               (unsigned)n, d_p, d_v, d_e, c->g, (const double *)nullptr, c->d_one_items.p, cap);
        else
            hipLaunchKernelGGL((k_one_fast<false>), dim3(n_seg), dim3(64), 0, st, c->d_el, c->d_flags, c->n_pad, (unsigned)sat, d_t,
                               (unsigned)n, d_p, d_v, d_e, c->g, (const double *)nullptr, c->d_one_items.p, cap);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL((k_one_satellite<true>), dim3(4096), dim3(64), 0, st, c->d_el, c->d_flags, c->n_pad, (unsigned)sat, d_t,
                           (unsigned)n, d_p, d_v, d_e, 0, c->g, (const double *)nullptr, 0, (const unsigned *)c->d_one_items.p, cap);
        HIP_TRY(hipEventRecord(c->ev_one, st));
    } else {
        hipLaunchKernelGGL((k_one_satellite<false>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, c->d_el, c->d_flags, c->n_pad,
                           (unsigned)sat, d_t, (unsigned)n, d_p, d_v, d_e, interleaved, c->g, (const double *)nullptr, 0,
                           (const unsigned *)nullptr);
    }
    HIP_TRY(hipGetLastError());
    return AZ_OK;
}

// One-satellite calls of up to this many points go through a pinned staging buffer of the handle (it starts at 1,024 points
// and doubles on demand: a Satrec that only ever asks for one point holds 64 KB).  Pageable copies of a few KB cost ~20 us
// each: 1,440 points (BASELINE config 1's shape) took 68 us against 26 us for 1,024.
#ifndef AZ_ONE_ZERO_COPY
#define AZ_ONE_ZERO_COPY 1
#endif
#ifndef AZ_ONE_ZERO_COPY_MAX
#define AZ_ONE_ZERO_COPY_MAX 16384
#endif
constexpr size_t kOneStage = 16384;
constexpr size_t kSmallOut = size_t(512) << 10; // bytes of results a host-returning constellation call lets its kernels write to pinned host memory directly
constexpr size_t kOneZeroCopy = AZ_ONE_ZERO_COPY_MAX; // points the kernel exchanges with the pinned buffer directly

// the element column of one satellite on the host (table, padded length, column index), or nullptr when it cannot be fetched
const double *host_column(azh_constellation *c, size_t sat, size_t &n_pad, size_t &col)
{
    if (!c->h_el.empty()) {
        n_pad = c->n_pad;
        col = sat;
        return c->h_el.data();
    }
    for (auto &e : c->h_cols)
        if (e.first == sat) {
            n_pad = 1;
            col = 0;
            return e.second.data();
        }
    if (set_device(c) != AZ_OK) return nullptr;
    // a catalog whose members are asked for one after the other (a loop of scalar calls over Satrec objects that share this
    // handle): after a few single columns the whole table comes over in one copy (9 MB for 13,478 satellites: 0.3 ms, where
    // 13,478 strided column copies would be 0.4 s)
    if (c->h_cols.size() >= kHostColsBeforeTable && (size_t)AZ_NUM_FIELDS * c->n_pad * sizeof(double) <= kHostTableMax) {
        try {
            c->h_el.resize((size_t)AZ_NUM_FIELDS * c->n_pad);
        } catch (const std::bad_alloc &) {
            c->h_el.clear();
        }
        if (!c->h_el.empty()) {
            if (hip_ok(hipMemcpy(c->h_el.data(), c->d_el, sizeof(double) * c->h_el.size(), hipMemcpyDeviceToHost), "D2H table")) {
                c->h_cols.clear();
                n_pad = c->n_pad;
                col = sat;
                return c->h_el.data();
            }
            c->h_el.clear();
        }
    }
    n_pad = 1;
    col = 0;
    std::vector<double> f(AZ_NUM_FIELDS);
    if (!hip_ok(hipMemcpy2D(f.data(), sizeof(double), c->d_el + sat, sizeof(double) * c->n_pad, sizeof(double), AZ_NUM_FIELDS,
                            hipMemcpyDeviceToHost), "D2H column"))
        return nullptr;
    if (c->h_cols.size() < kHostCols) {
        c->h_cols.emplace_back(sat, std::move(f));
        return c->h_cols.back().second.data();
    }
    auto &slot = c->h_cols[c->h_cols_next++ % kHostCols];
    slot.first = sat;
    slot.second = std::move(f);
    return slot.second.data();
}

// one satellite x n times.  interleaved = 1: out6 is n x 6 (x,y,z,vx,vy,vz; c_api batch layout); otherwise
// pos (n x 3), vel (n x 3, optional), err (n, optional).
int32_t run_one_satellite(azh_constellation *c, size_t sat, const double *tsince, size_t n, int interleaved,
                          double *out6, double *pos, double *vel, uint8_t *err)
{
    if (n > 0xffffffffu) return AZ_ERR_VALUE;
    // a handful of points: the same step source on the calling thread, from the device-initialised column (host_step.h) -- no
    // launch, no synchronize.  Measured (tools/host_route_probe.py, EPYC 9575F): 0.07-0.14 us per near-earth point, 0.16-0.23 per
    // deep-space point, against 21-23 us per call through the kernel: the routes cross at ~230 / ~120 points, so deep-space
    // members take the host route up to half the limit
    if (sat < c->n && n <= host_points() / ((c->h_flags[sat] & AZ_FLAG_DEEP) ? 2 : 1)) {
        size_t np = 0, col = 0;
        if (const double *el = host_column(c, sat, np, col)) {
            azhost::propagate_points(el, np, col, c->h_flags[sat], c->g, tsince, n, interleaved, out6, pos, vel, err);
            c->last_path = AZH_PATH_HOST_STEP;
            c->one_segments = 0;
            return AZ_OK;
        }
    }
    if (set_device(c) != AZ_OK) return AZ_ERR_HIP;
    c->last_path = 0; // (none of the constellation kernel families, and not the host route)
    hipStream_t st = c->s_main;
    if (c->d_one_t.ensure(n) != AZ_OK || c->d_one_o.ensure(6 * n) != AZ_OK || c->d_one_e.ensure(n) != AZ_OK) return AZ_ERR_HIP;
    const bool staged = n <= kOneStage;
    if (staged && c->h_stage_cap < n) {
        size_t cap = std::max<size_t>(c->h_stage_cap, 1024);
        while (cap < n) cap *= 2;
        if (c->h_stage) (void)hipHostFree(c->h_stage);
        c->h_stage = nullptr;
        c->h_stage_cap = 0;
        HIP_TRY(hipHostMalloc(&c->h_stage, cap * (7 * sizeof(double) + 8), hipHostMallocDefault));
        c->h_stage_cap = cap;
    }
    const size_t scap = c->h_stage_cap;
    double *hs_t = static_cast<double *>(c->h_stage);
    double *hs_o = hs_t ? hs_t + scap : nullptr;
    uint8_t *hs_e = hs_o ? reinterpret_cast<uint8_t *>(hs_o + 6 * scap) : nullptr;
    double *d_p = c->d_one_o.p, *d_v = c->d_one_o.p + 3 * n;
    // A handful of points: the kernel reads the times from and writes its results into the pinned buffer ITSELF (host memory
    // the device can address: coherent by default) -- one launch and one synchronize instead of three copy commands around it
    // (a single point 26 -> 19 us; 1,440 points, BASELINE config 1's shape, 68 -> 21 us = 67 M propagations/s through
    // Satrec.sgp4_array; 16,000 points 109 -> 74 us).
    const bool zero_copy = AZ_ONE_ZERO_COPY && staged && n <= kOneZeroCopy;
    if (zero_copy) {
        memcpy(hs_t, tsince, sizeof(double) * n);
        double *dev_t = nullptr;
        HIP_TRY(hipHostGetDevicePointer((void **)&dev_t, hs_t, 0));
        double *dev_o = dev_t + (hs_o - hs_t);
        unsigned char *dev_e = reinterpret_cast<unsigned char *>(dev_t) + (reinterpret_cast<char *>(hs_e) - reinterpret_cast<char *>(hs_t));
        if (int32_t lrc = launch_one(c, sat, dev_t, n, dev_o, interleaved ? (double *)nullptr : dev_o + 3 * n,
                                     interleaved ? (unsigned char *)nullptr : dev_e, interleaved, st); lrc != AZ_OK)
            return lrc;
        if (!hip_ok(hipStreamSynchronize(st), "sync")) return AZ_ERR_HIP;
        if (interleaved) {
            memcpy(out6, hs_o, sizeof(double) * 6 * n);
        } else {
            memcpy(pos, hs_o, sizeof(double) * 3 * n);
            if (vel) memcpy(vel, hs_o + 3 * n, sizeof(double) * 3 * n);
            if (err) memcpy(err, hs_e, n);
        }
        return AZ_OK;
    }
    if (staged) {
        memcpy(hs_t, tsince, sizeof(double) * n);
        HIP_TRY(hipMemcpyAsync(c->d_one_t.p, hs_t, sizeof(double) * n, hipMemcpyHostToDevice, st));
    } else {
        HIP_TRY(hipMemcpyAsync(c->d_one_t.p, tsince, sizeof(double) * n, hipMemcpyHostToDevice, st));
    }
    if (int32_t lrc = launch_one(c, sat, c->d_one_t.p, n, d_p, interleaved ? (double *)nullptr : d_v,
                                 interleaved ? (unsigned char *)nullptr : c->d_one_e.p, interleaved, st); lrc != AZ_OK)
        return lrc;
    int32_t rc = AZ_OK;
    if (staged) {
        if (!hip_ok(hipMemcpyAsync(hs_o, c->d_one_o.p, sizeof(double) * 6 * n, hipMemcpyDeviceToHost, st), "D2H") ||
            (!interleaved && !hip_ok(hipMemcpyAsync(hs_e, c->d_one_e.p, n, hipMemcpyDeviceToHost, st), "D2H")) ||
            !hip_ok(hipStreamSynchronize(st), "sync"))
            rc = AZ_ERR_HIP;
        if (rc == AZ_OK) {
            if (interleaved) {
                memcpy(out6, hs_o, sizeof(double) * 6 * n);
            } else {
                memcpy(pos, hs_o, sizeof(double) * 3 * n);
                if (vel) memcpy(vel, hs_o + 3 * n, sizeof(double) * 3 * n);
                if (err) memcpy(err, hs_e, n);
            }
        }
    } else {
        // long time series: through the pinned staging slots, like the constellation's host-returning calls (copy_back_staged)
        void *const dst[3] = {interleaved ? (void *)out6 : (void *)pos, interleaved ? nullptr : (void *)vel, interleaved ? nullptr : (void *)err};
        const void *const src[3] = {interleaved ? (const void *)c->d_one_o.p : (const void *)d_p, d_v, c->d_one_e.p};
        const size_t len[3] = {sizeof(double) * (interleaved ? 6 : 3) * n, (!interleaved && vel) ? sizeof(double) * 3 * n : 0,
                               (!interleaved && err) ? n : 0};
        const unsigned thr = host_copy_threads();
        bool ok = true;
        const bool pinned_out = host_pool().owns(dst[0], len[0]) && (!len[1] || host_pool().owns(dst[1], len[1])); // azh_host_alloc
        if (!pinned_out && thr > 0 && len[0] + len[1] + len[2] >= (size_t(8) << 20)) {
            ok = copy_back_staged(c->stager, dst, src, len, 3, st, thr) == AZ_OK;
        } else {
            for (int k = 0; k < 3 && ok; ++k)
                if (len[k]) ok = hip_ok(hipMemcpyAsync(dst[k], src[k], len[k], hipMemcpyDeviceToHost, st), "D2H");
        }
        if (!ok || !hip_ok(hipStreamSynchronize(st), "sync")) rc = AZ_ERR_HIP;
    }
    if (rc != AZ_OK) (void)hipStreamSynchronize(st);
    return rc;
}

} // namespace

// host-side text entry points: nothing may unwind through the C boundary (a catalog-scale parse allocates hundreds of MB)
namespace {
template <class F>
int32_t guarded(F &&f)
{
    try {
        return f();
    } catch (const std::bad_alloc &) {
        g_last_error = "out of host memory";
        return AZ_ERR_ALLOC_FAILED;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        return AZ_ERR_UNKNOWN;
    } catch (...) {
        return AZ_ERR_UNKNOWN;
    }
}
} // namespace

// ======================================================================================= (B)
extern "C" {

int azh_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

const char *azh_last_error(void) { return g_last_error.c_str(); }

namespace {
void record_to_fields(const azh::TleRecord &r, double *o)
{
    o[0] = r.satnum; o[1] = r.epoch_year; o[2] = r.epoch_day; o[3] = r.epoch_jd; o[4] = r.ndot; o[5] = r.bstar;
    o[6] = r.incl_deg; o[7] = r.raan_deg; o[8] = r.ecc; o[9] = r.argp_deg; o[10] = r.ma_deg; o[11] = r.mm_revday;
    o[12] = r.elnum; o[13] = r.revnum; o[14] = (double)(unsigned char)r.classification; o[15] = 0.0;
}
int32_t records_out(const std::vector<azh::TleRecord> &recs, double *out16, size_t max_records, size_t *n_found)
{
    if (!n_found || (max_records && !out16)) return AZ_ERR_NULL_POINTER;
    *n_found = recs.size();
    const size_t n = std::min(recs.size(), max_records);
    azh::parallel_ranges(n, azh::parse_threads_for(n * 140), [&](size_t a, size_t b) {
        for (size_t i = a; i < b; ++i) record_to_fields(recs[i], out16 + 16 * i);
    });
    return AZ_OK;
}
} // namespace

int32_t azh_parse_tle_lines(const char *line1, const char *line2, double *o)
{
    if (!line1 || !line2 || !o) return AZ_ERR_NULL_POINTER;
    azh::TleRecord r;
    int rc = azh::parse_lines(line1, line2, r);
    if (rc == -1) return AZ_ERR_BAD_TLE_LENGTH;
    if (rc != 0) return AZ_ERR_UNKNOWN;
    record_to_fields(r, o);
    return AZ_OK;
}


int32_t azh_parse_tle_text(const char *text, size_t len, double *out16, size_t max_records, size_t *n_found)
{
    return guarded([&]() -> int32_t {
        if (!text) return AZ_ERR_NULL_POINTER;
        std::vector<azh::TleRecord> recs;
        azh::parse_all(std::string_view(text, len), recs);
        return records_out(recs, out16, max_records, n_found);
    });
}

void azh_set_parse_threads(int32_t n) { azh::set_parse_threads(n > 0 ? (unsigned)n : 0u); }

int32_t azh_parse_omm_json(const char *text, size_t len, double *out16, size_t max_records, size_t *n_found)
{
    return guarded([&]() -> int32_t {
        if (!text) return AZ_ERR_NULL_POINTER;
        std::vector<azh::TleRecord> recs;
        const int rc = azh::parse_omm_json(std::string_view(text, len), recs);
        if (rc == -1) return AZ_ERR_BAD_TLE_LENGTH;
        if (rc != 0) return AZ_ERR_VALUE;
        return records_out(recs, out16, max_records, n_found);
    });
}

int32_t azh_constellation_from_tle_text(const char *text, size_t len, int32_t grav, int32_t device,
                                        azh_constellation **out)
{
    return guarded([&]() -> int32_t {
        if (!text || !out) return AZ_ERR_NULL_POINTER;
        std::vector<azh::TleRecord> recs;
        azh::parse_all(std::string_view(text, len), recs);
        if (recs.empty()) return AZ_ERR_BAD_TLE_LENGTH;
        return build_from_records(recs, grav, device, out);
    });
}

int32_t azh_constellation_from_omm_json(const char *text, size_t len, int32_t grav, int32_t device,
                                        azh_constellation **out)
{
    return guarded([&]() -> int32_t {
        if (!text || !out) return AZ_ERR_NULL_POINTER;
        std::vector<azh::TleRecord> recs;
        const int rc = azh::parse_omm_json(std::string_view(text, len), recs);
        if (rc == -1) return AZ_ERR_BAD_TLE_LENGTH; // Tle.zig L199: epoch string too short
        if (rc != 0) return AZ_ERR_VALUE;
        if (recs.empty()) return AZ_ERR_BAD_TLE_LENGTH;
        return build_from_records(recs, grav, device, out);
    });
}

int32_t azh_constellation_from_tle_lines(const char *const *line1, const char *const *line2, size_t n,
                                         int32_t grav, int32_t device, azh_constellation **out)
{
    return guarded([&]() -> int32_t {
        if (!line1 || !line2 || !out) return AZ_ERR_NULL_POINTER;
        std::vector<azh::TleRecord> recs(n);
        for (size_t i = 0; i < n; ++i) {
            if (!line1[i] || !line2[i]) return AZ_ERR_NULL_POINTER;
            int rc = azh::parse_lines(line1[i], line2[i], recs[i]);
            if (rc == -1) return AZ_ERR_BAD_TLE_LENGTH;
            if (rc != 0) return AZ_ERR_UNKNOWN;
        }
        return build_from_records(recs, grav, device, out);
    });
}

int32_t azh_constellation_from_elements(size_t n, const double *epoch_jd, const double *mm, const double *ecc,
                                        const double *incl, const double *raan, const double *argp,
                                        const double *ma, const double *bstar, int32_t grav, int32_t device,
                                        azh_constellation **out)
{
    return guarded([&]() -> int32_t {
        if (!epoch_jd || !mm || !ecc || !incl || !raan || !argp || !ma || !bstar || !out) return AZ_ERR_NULL_POINTER;
        std::vector<double> cols[AZ_NUM_RAW];
        const double *src[AZ_NUM_RAW] = {epoch_jd, mm, ecc, incl, raan, argp, ma, bstar};
        for (int k = 0; k < AZ_NUM_RAW; ++k) cols[k].assign(src[k], src[k] + n);
        return build(cols, n, grav, device, out);
    });
}

int32_t azh_constellation_subset(const azh_constellation *c, const uint32_t *indices, size_t n, int32_t device,
                                 azh_constellation **out)
{
    return guarded([&]() -> int32_t {
        if (!c || !indices || !out) return AZ_ERR_NULL_POINTER;
        std::vector<double> cols[AZ_NUM_RAW];
        for (auto &v : cols) v.resize(n);
        for (size_t i = 0; i < n; ++i) {
            if (indices[i] >= c->n) return AZ_ERR_VALUE;
            for (int k = 0; k < AZ_NUM_RAW; ++k) cols[k][i] = c->h_raw[k][indices[i]];
        }
        const int grav = (c->g.radius_km == 6378.135) ? AZ_WGS72 : AZ_WGS84;
        return build(cols, n, grav, device < 0 ? c->device : device, out);
    });
}

void azh_constellation_free(azh_constellation *c) { destroy(c); }

size_t azh_num_satellites(const azh_constellation *c) { return c ? c->n : 0; }
size_t azh_num_sgp4(const azh_constellation *c) { return c ? c->n_sgp4 : 0; }
size_t azh_num_sdp4(const azh_constellation *c) { return c ? c->n_sdp4 : 0; }

int32_t azh_get_epochs(const azh_constellation *c, double *out)
{
    if (!c || !out) return AZ_ERR_NULL_POINTER;
    memcpy(out, c->h_epoch.data(), sizeof(double) * c->n);
    return AZ_OK;
}

int32_t azh_get_status(const azh_constellation *c, uint8_t *err, uint8_t *deep, uint8_t *irez)
{
    if (!c) return AZ_ERR_NULL_POINTER;
    for (size_t s = 0; s < c->n; ++s) {
        const unsigned f = c->h_flags[s];
        if (err) err[s] = (uint8_t)AZ_FLAG_ERR(f);
        if (deep) deep[s] = (f & AZ_FLAG_DEEP) ? 1 : 0;
        if (irez) irez[s] = (uint8_t)AZ_FLAG_IREZ(f);
    }
    return AZ_OK;
}

int32_t azh_get_field(const azh_constellation *c, const char *name, double *out)
{
    if (!c || !name || !out) return AZ_ERR_NULL_POINTER;
    for (int k = 0; k < AZ_NUM_FIELDS; ++k) {
        if (strcmp(kFieldNames[k], name) == 0) {
            if (set_device(c) != AZ_OK) return AZ_ERR_HIP;
            HIP_TRY(hipMemcpy(out, c->d_el + (size_t)k * c->n_pad, sizeof(double) * c->n, hipMemcpyDeviceToHost));
            return AZ_OK;
        }
    }
    return AZ_ERR_VALUE;
}

int32_t azh_set_timing(azh_constellation *c, int32_t enabled)
{
    if (!c) return AZ_ERR_NULL_POINTER;
    c->timing = enabled != 0;
    if (!c->timing) c->timed = false;
    return AZ_OK;
}

int32_t azh_set_f32_mode(azh_constellation *c, int32_t mode)
{
    if (!c) return AZ_ERR_NULL_POINTER;
    if (mode < 0 || mode > 2) return AZ_ERR_VALUE;
    c->f32_mode = mode;
    drop_graphs(c);
    return AZ_OK;
}

int32_t azh_set_f32_arithmetic(azh_constellation *c, int32_t enabled)
{
    return azh_set_f32_mode(c, enabled ? AZH_F32_PACKED : AZH_F32_FP64_ROUNDED);
}

void azh_set_host_copy_threads(int32_t n) { g_host_copy_threads.store(n < 0 ? -1 : n, std::memory_order_relaxed); }
void azh_set_host_points(size_t n) { g_host_points.store(n, std::memory_order_relaxed); }
size_t azh_get_host_points(void) { return host_points(); }

int32_t azh_host_alloc(size_t bytes, void **out)
{
    return guarded([&]() -> int32_t { return host_pool().alloc(bytes, out); });
}
void azh_host_free(void *p)
{
    (void)guarded([&]() -> int32_t { host_pool().release(p); return AZ_OK; });
}
void azh_host_pool_stats(size_t *live_bytes, size_t *free_bytes) { host_pool().stats(live_bytes, free_bytes); }
void azh_host_pool_trim(void) { host_pool().trim(0); }

int32_t azh_set_tile_kernel(azh_constellation *c, int32_t enabled)
{
    if (!c) return AZ_ERR_NULL_POINTER;
    c->tile_kernel = enabled != 0;
    c->cols_kernel = enabled == 2 ? 1 : (enabled == 1 ? 0 : c->cols_kernel); // 2: the lane = satellite kernel (k_cols_fast), 1: the 16-row tiles
    drop_graphs(c);
    return AZ_OK;
}

int32_t azh_set_fast_path(azh_constellation *c, int32_t enabled)
{
    if (!c) return AZ_ERR_NULL_POINTER;
    c->fast_path = enabled != 0;
    drop_graphs(c);
    return AZ_OK;
}

int32_t azh_set_time_tile(azh_constellation *c, uint32_t sgp4_tile, uint32_t sdp4_tile)
{
    if (!c) return AZ_ERR_NULL_POINTER;
    c->tile_sgp4 = sgp4_tile;
    c->tile_sdp4 = sdp4_tile;
    drop_graphs(c);
    return AZ_OK;
}

int32_t azh_set_graphs(azh_constellation *c, int32_t enabled)
{
    if (!c) return AZ_ERR_NULL_POINTER;
    c->graphs_on = enabled != 0;
    if (!c->graphs_on) drop_graphs(c);
    return AZ_OK;
}

static int32_t azh_propagate_device_impl(azh_constellation *c, const double *times, size_t n_times, const double *offsets,
                             double *d_pos, double *d_vel, int32_t mode, double reference_jd, const uint8_t *mask,
                             int32_t layout, size_t stride, uint8_t *d_err, void *stream)
{
    if (!c || !d_pos || (n_times && !times)) return AZ_ERR_NULL_POINTER;
    if (mode < 0 || mode > 2 || layout < 0 || layout > 1) return AZ_ERR_VALUE;
    if (n_times == 0) return AZ_OK; // an empty grid is valid and produces nothing
    if (set_device(c) != AZ_OK) return AZ_ERR_HIP;
    hipStream_t st = stream ? (hipStream_t)stream : c->s_main;
    int32_t rc = stage_inputs(c, times, n_times, offsets, mask, mode, reference_jd, st);
    if (rc != AZ_OK) return rc;
    return launch_all(c, d_pos, d_vel, layout, stride, d_err, st);
}
int32_t azh_propagate_device(azh_constellation *c, const double *times, size_t n_times, const double *offsets,
                             double *d_pos, double *d_vel, int32_t mode, double reference_jd, const uint8_t *mask,
                             int32_t layout, size_t stride, uint8_t *d_err, void *stream)
{
    return guarded([&]() -> int32_t { return azh_propagate_device_impl(c, times, n_times, offsets, d_pos, d_vel, mode, reference_jd, mask, layout, stride, d_err, stream); });
}

static int32_t azh_propagate_device_cached_impl(azh_constellation *c, double *d_pos, double *d_vel, int32_t layout,
                                    size_t stride, uint8_t *d_err, void *stream)
{
    if (!c || !d_pos) return AZ_ERR_NULL_POINTER;
    if (layout < 0 || layout > 1) return AZ_ERR_VALUE;
    if (c->cached_n_times == 0) return AZ_ERR_NOT_INITIALIZED;
    if (set_device(c) != AZ_OK) return AZ_ERR_HIP;
    return launch_cached(c, d_pos, d_vel, layout, stride, d_err, stream ? (hipStream_t)stream : c->s_main);
}
int32_t azh_propagate_device_cached(azh_constellation *c, double *d_pos, double *d_vel, int32_t layout,
                                    size_t stride, uint8_t *d_err, void *stream)
{
    return guarded([&]() -> int32_t { return azh_propagate_device_cached_impl(c, d_pos, d_vel, layout, stride, d_err, stream); });
}

static int32_t azh_propagate_device_window_impl(azh_constellation *c, size_t row_lo, size_t row_hi, double *d_pos, double *d_vel,
                                    int32_t layout, size_t stride, uint8_t *d_err, void *stream)
{
    if (!c || !d_pos) return AZ_ERR_NULL_POINTER;
    if (layout < 0 || layout > 1) return AZ_ERR_VALUE;
    if (c->cached_n_times == 0) return AZ_ERR_NOT_INITIALIZED;
    if (set_device(c) != AZ_OK) return AZ_ERR_HIP;
    return launch_cached(c, d_pos, d_vel, layout, stride, d_err, stream ? (hipStream_t)stream : c->s_main, 0, row_lo, row_hi);
}
int32_t azh_propagate_device_window(azh_constellation *c, size_t row_lo, size_t row_hi, double *d_pos, double *d_vel,
                                    int32_t layout, size_t stride, uint8_t *d_err, void *stream)
{
    return guarded([&]() -> int32_t { return azh_propagate_device_window_impl(c, row_lo, row_hi, d_pos, d_vel, layout, stride, d_err, stream); });
}

static int32_t azh_propagate_device_f32_impl(azh_constellation *c, const double *times, size_t n_times, const double *offsets,
                                 float *d_pos, float *d_vel, int32_t mode, double reference_jd, const uint8_t *mask,
                                 int32_t layout, size_t stride, uint8_t *d_err, void *stream)
{
    if (!c || !d_pos || (n_times && !times)) return AZ_ERR_NULL_POINTER;
    if (mode < 0 || mode > 2 || layout < 0 || layout > 1) return AZ_ERR_VALUE;
    if (n_times == 0) return AZ_OK;
    if (set_device(c) != AZ_OK) return AZ_ERR_HIP;
    hipStream_t st = stream ? (hipStream_t)stream : c->s_main;
    int32_t rc = stage_inputs(c, times, n_times, offsets, mask, mode, reference_jd, st);
    if (rc != AZ_OK) return rc;
    return launch_all(c, reinterpret_cast<double *>(d_pos), reinterpret_cast<double *>(d_vel), layout, stride, d_err, st, 1);
}
int32_t azh_propagate_device_f32(azh_constellation *c, const double *times, size_t n_times, const double *offsets,
                                 float *d_pos, float *d_vel, int32_t mode, double reference_jd, const uint8_t *mask,
                                 int32_t layout, size_t stride, uint8_t *d_err, void *stream)
{
    return guarded([&]() -> int32_t { return azh_propagate_device_f32_impl(c, times, n_times, offsets, d_pos, d_vel, mode, reference_jd, mask, layout, stride, d_err, stream); });
}

static int32_t azh_propagate_device_cached_f32_impl(azh_constellation *c, float *d_pos, float *d_vel, int32_t layout, size_t stride,
                                        uint8_t *d_err, void *stream)
{
    if (!c || !d_pos) return AZ_ERR_NULL_POINTER;
    if (layout < 0 || layout > 1) return AZ_ERR_VALUE;
    if (c->cached_n_times == 0) return AZ_ERR_NOT_INITIALIZED;
    if (set_device(c) != AZ_OK) return AZ_ERR_HIP;
    return launch_cached(c, reinterpret_cast<double *>(d_pos), reinterpret_cast<double *>(d_vel), layout, stride, d_err,
                         stream ? (hipStream_t)stream : c->s_main, 1);
}
int32_t azh_propagate_device_cached_f32(azh_constellation *c, float *d_pos, float *d_vel, int32_t layout, size_t stride,
                                        uint8_t *d_err, void *stream)
{
    return guarded([&]() -> int32_t { return azh_propagate_device_cached_f32_impl(c, d_pos, d_vel, layout, stride, d_err, stream); });
}

// Fused single-target conjunction screen (Constellation.screenConstellation, src/Constellation.zig
// L683-756): nothing but 12 bytes per satellite ever leaves the chip.
// The screen proper.  The target is either a member of c (`target` < c->n: its track is computed here, and the row reports
// threshold / 0 like the reference's) or an EXTERNAL track `d_track` (n_times x 3 TEME km on c's device: a satellite of another
// shard, or an object that is in no catalog), in which case `target` = kNoTarget or the member to leave out.
constexpr size_t kNoTarget = ~(size_t)0;
static int32_t screen_core(azh_constellation *c, const double *times, size_t n_times, const double *offsets, size_t target,
                           const double *d_track, double threshold_km, double *d_min_dist, uint32_t *d_min_t, void *stream)
{
    if (!c || !d_min_dist || !d_min_t || (n_times && !times)) return AZ_ERR_NULL_POINTER;
    if (!d_track && target >= c->n) return AZ_ERR_VALUE;
    if (d_track && target != kNoTarget && target >= c->n) return AZ_ERR_VALUE;
    if (set_device(c) != AZ_OK) return AZ_ERR_HIP;
    hipStream_t st = stream ? (hipStream_t)stream : c->s_main;
    int32_t rc = AZ_OK;
    if (n_times > 0 && (rc = stage_inputs(c, times, n_times, offsets, nullptr, AZ_OUT_TEME, 0.0, st)) != AZ_OK) return rc;
    const unsigned nt = (unsigned)n_times;
    if (c->timing) HIP_TRY(hipEventRecord(c->ev_t0, st));
    const unsigned nb_fill = ((unsigned)c->n + 255) / 256;
    if (nt == 0) {
        hipLaunchKernelGGL(k_screen_prep, dim3(nb_fill), dim3(64), 0, st, c->d_el, c->d_flags, c->n_pad, 0u, (const double *)nullptr, 0u,
                           (const double *)nullptr, c->g, (double *)nullptr, 0u, (size_t)0, (double *)nullptr, (unsigned *)nullptr, 0u,
                           (unsigned)c->n, threshold_km, d_min_dist, d_min_t);
        HIP_TRY(hipGetLastError());
    }
    if (nt > 0) {
        if (!d_track && c->d_tgt.ensure((size_t)nt * 3) != AZ_OK) return AZ_ERR_HIP;
        PropArgs a{};
        a.el = c->d_el;
        a.flags = c->d_flags;
        a.n_pad = c->n_pad;
        a.times = c->d_times.p;
        a.n_times = nt;
        a.offsets = c->have_offsets ? c->d_offsets.p : nullptr;
        a.stride_sats = c->n;
        a.mode = AZ_OUT_TEME;
        a.g = c->g;
        a.uniform_step = c->fast_path ? c->uniform_step : 0.0;
        a.inc = (a.uniform_step != 0.0) ? c->d_inc.p : nullptr;
        a.fast_rec = (a.uniform_step != 0.0) ? c->d_fast_rec.p : nullptr;
        a.delta = (a.uniform_step != 0.0 && c->delta_max > 0.0 && !c->delta_wide) ? c->d_delta.p : nullptr;
        a.delta64 = (a.uniform_step != 0.0 && c->delta_max > 0.0) ? c->d_delta64.p : nullptr;
        a.delta_wide = c->delta_wide ? 1 : 0;
        a.grid_t0 = c->grid_t0;
        a.delta_max = c->delta_max;
        a.grid_exact_uniform = (c->uniform_step != 0.0 && c->delta_max == 0.0) ? 1 : 0;
        a.row_lo = 0;
        a.row_hi = 0xffffffffu;
        a.screen_target = d_track ? d_track : c->d_tgt.p;
        PropArgs near = a, deep = a;
        unsigned parts_near = 0, parts_deep = 0;
        c->last_path = 0;
        // near-earth members on a uniform grid: the branch-free fast kernels with the screen as their sink (no stores at all:
        // the arithmetic-only rate), windows the plan rejects and Newton hand-overs through the generic pass like a propagation
        const bool fast_screen = c->n_sgp4 > 0 && a.inc != nullptr && nt >= 32;
        FastShape shape;
        if (c->n_sgp4 > 0) {
            near.list = c->d_list.p;
            near.n_list = c->n_sgp4;
            near.tile = auto_tile(c->n_sgp4, nt, c->tile_sgp4, 8);
            near.tile_forced = c->tile_sgp4;
            parts_near = screen_parts(near, false);
            if (fast_screen) {
                near.list = c->d_list.p + c->off_circ;
                near.n_circ = c->n_circ;
                shape = fast_shape_rows(near, c->n_sgp4, c->n_circ);
                if ((rc = ensure_plan(c, near, shape, st)) != AZ_OK) return rc;
                near.tile = shape.tile_c;
                near.tile_e = shape.tile_e;
                near.screen_nseg = cgrid_y(nt, std::min(shape.tile_c, shape.tile_e));
                parts_near = 5u * near.screen_nseg; // the fast kernels' rows + four per segment for the generic pass
            }
        }
        if (c->n_sdp4 > 0) {
            if ((rc = prepare_deep(c, deep, st, use_rows(deep, AZ_LAYOUT_SAT_MAJOR, true))) != AZ_OK) return rc;
            parts_deep = screen_parts(deep, true);
        }
        const size_t np_near = (size_t)parts_near * c->n_sgp4, np_deep = (size_t)parts_deep * c->n_sdp4;
        if (c->d_part_d2.ensure(np_near + np_deep) != AZ_OK || c->d_part_t.ensure(np_near + np_deep) != AZ_OK) return AZ_ERR_HIP;
        const double thr2 = threshold_km * threshold_km;
        near.part_d2 = c->d_part_d2.p;
        near.part_t = c->d_part_t.p;
        deep.part_d2 = c->d_part_d2.p + np_near;
        deep.part_t = c->d_part_t.p + np_near;
        // ONE preparation launch (round 6; it was three): the target's track, the reset of the partial minima the generic pass
        // only partly overwrites, the start values of every row
        {
            const unsigned nb_track = d_track ? 0u : (nt + 63) / 64;
            const size_t n_clear = fast_screen ? np_near : 0;
            const unsigned nb_clear = (unsigned)((n_clear + 255) / 256);
            hipLaunchKernelGGL(k_screen_prep, dim3(nb_track + nb_clear + nb_fill), dim3(64), 0, st, c->d_el, c->d_flags, c->n_pad,
                               (unsigned)(d_track ? 0 : target), c->d_times.p, nt, c->have_offsets ? c->d_offsets.p : (const double *)nullptr, c->g,
                               c->d_tgt.p, nb_track, n_clear, near.part_d2, near.part_t, nb_clear, (unsigned)c->n, threshold_km, d_min_dist, d_min_t);
            HIP_TRY(hipGetLastError());
        }
        if (c->n_sgp4 > 0) {
            if (fast_screen) {
                PropArgs e = near, cc = near;
                e.list = near.list + near.n_circ;
                e.n_list = near.n_list - near.n_circ;
                e.tile = shape.tile_e;
                e.redo_slot0 = near.n_circ;
                cc.n_list = near.n_circ;
                // as in a propagation: the bulk alone on the launch stream, the eccentric members and the generic pass beside it
                const bool beside = cc.n_list > 0;
                hipStream_t se = beside ? c->s_ecc : st;
                if (beside) {
                    HIP_TRY(hipEventRecord(c->ev_fork2, st));
                    HIP_TRY(hipStreamWaitEvent(se, c->ev_fork2, 0));
                }
                if (e.n_list) launch_rows_fast<false, 0, AZ_SINK_SCREEN, true>(e, dim3((e.n_list + 7) / 8 * 8, cgrid_y(nt, e.tile)), se);
                hipLaunchKernelGGL((k_rows<false, false, AZ_SINK_SCREEN, true>), dim3(256, 4), dim3(64), 0, se, near);
                if (cc.n_list) launch_rows_fast<false, 0, AZ_SINK_SCREEN, false>(cc, dim3((cc.n_list + 7) / 8 * 8, cgrid_y(nt, cc.tile)), st);
                if (beside) {
                    HIP_TRY(hipEventRecord(c->ev_join2, se));
                    HIP_TRY(hipStreamWaitEvent(st, c->ev_join2, 0));
                }
                c->last_path = AZH_PATH_ROWS_FAST | ((near.delta || near.delta64) ? AZH_PATH_QUASI_UNIFORM : 0u);
            } else {
                c->last_path = launch_propagate(near, AZ_LAYOUT_SAT_MAJOR, false, false, st);
            }
            HIP_TRY(hipGetLastError());
        }
        if (c->n_sdp4 > 0) {
            c->last_path |= launch_propagate(deep, AZ_LAYOUT_SAT_MAJOR, false, true, st);
            HIP_TRY(hipGetLastError());
        }
        // ... and ONE finalisation over both lists (it was one per list)
        const unsigned n_fin = c->n_sgp4 + c->n_sdp4;
        if (n_fin > 0) {
            hipLaunchKernelGGL(k_screen_finalize2, dim3((n_fin + 255) / 256), dim3(256), 0, st, near.part_d2, near.part_t, parts_near, near.list, c->n_sgp4,
                               deep.part_d2, deep.part_t, parts_deep, deep.list, c->n_sdp4, thr2, (unsigned)target, d_min_dist, d_min_t);
            HIP_TRY(hipGetLastError());
        }
    }
    if (c->timing) {
        HIP_TRY(hipEventRecord(c->ev_t1, st));
        c->timed = true;
    }
    return AZ_OK;
}
int32_t azh_screen_target_device(azh_constellation *c, const double *times, size_t n_times, const double *offsets,
                                 size_t target, double threshold_km, double reference_jd, double *d_min_dist,
                                 uint32_t *d_min_t, void *stream)
{
    (void)reference_jd; // the reference rotates both vectors to ECEF first (L719, L737); distances do not change
    return guarded([&]() -> int32_t {
        if (c && target >= c->n) return AZ_ERR_VALUE;
        return screen_core(c, times, n_times, offsets, target, nullptr, threshold_km, d_min_dist, d_min_t, stream);
    });
}
int32_t azh_screen_track_device(azh_constellation *c, const double *times, size_t n_times, const double *offsets, const double *d_track,
                                size_t exclude_index, double threshold_km, double *d_min_dist, uint32_t *d_min_t, void *stream)
{
    return guarded([&]() -> int32_t {
        if (!d_track) return AZ_ERR_NULL_POINTER;
        return screen_core(c, times, n_times, offsets, exclude_index, d_track, threshold_km, d_min_dist, d_min_t, stream);
    });
}

static int32_t azh_screen_target_host_impl(azh_constellation *c, const double *times, size_t n_times, const double *offsets,
                               size_t target, double threshold_km, double reference_jd, double *min_dist,
                               uint32_t *min_t)
{
    if (!c || !min_dist || !min_t) return AZ_ERR_NULL_POINTER;
    if (set_device(c) != AZ_OK) return AZ_ERR_HIP;
    // results of up to kSmallOut bytes (43,000 satellites): the finalisation kernel writes them into a pinned host buffer of the
    // handle ITSELF (device-addressable, coherent) and one synchronize ends the call -- two pageable device-to-host copies of
    // a hundred KB cost 15-20 us each behind a 0.18-ms screen
    const size_t small_total = c->n * (sizeof(double) + sizeof(uint32_t)) + 64;
    if (small_total <= kSmallOut) {
        if (c->h_small_cap < small_total) {
            HIP_TRY(hipStreamSynchronize(c->s_main));
            if (c->h_small) (void)hipHostFree(c->h_small);
            c->h_small = nullptr;
            c->h_small_cap = 0;
            size_t cap = 65536;
            while (cap < small_total) cap *= 2;
            HIP_TRY(hipHostMalloc(&c->h_small, cap, hipHostMallocDefault));
            c->h_small_cap = cap;
        }
        double *hd = static_cast<double *>(c->h_small);
        uint32_t *ht = reinterpret_cast<uint32_t *>(hd + ((c->n + 7) & ~size_t(7)));
        double *dd = nullptr;
        HIP_TRY(hipHostGetDevicePointer((void **)&dd, hd, 0));
        uint32_t *dt = reinterpret_cast<uint32_t *>(dd + (reinterpret_cast<double *>(ht) - hd));
        int32_t rc = azh_screen_target_device(c, times, n_times, offsets, target, threshold_km, reference_jd, dd, dt, nullptr);
        if (rc != AZ_OK) return rc;
        HIP_TRY(hipStreamSynchronize(c->s_main));
        memcpy(min_dist, hd, sizeof(double) * c->n);
        memcpy(min_t, ht, sizeof(uint32_t) * c->n);
        return AZ_OK;
    }
    if (c->d_out_d.ensure(c->n) != AZ_OK || c->d_out_t.ensure(c->n) != AZ_OK) return AZ_ERR_HIP;
    int32_t rc = azh_screen_target_device(c, times, n_times, offsets, target, threshold_km, reference_jd, c->d_out_d.p,
                                          c->d_out_t.p, nullptr);
    if (rc != AZ_OK) return rc;
    HIP_TRY(hipMemcpyAsync(min_dist, c->d_out_d.p, sizeof(double) * c->n, hipMemcpyDeviceToHost, c->s_main));
    HIP_TRY(hipMemcpyAsync(min_t, c->d_out_t.p, sizeof(uint32_t) * c->n, hipMemcpyDeviceToHost, c->s_main));
    HIP_TRY(hipStreamSynchronize(c->s_main));
    return AZ_OK;
}
int32_t azh_screen_target_host(azh_constellation *c, const double *times, size_t n_times, const double *offsets,
                               size_t target, double threshold_km, double reference_jd, double *min_dist,
                               uint32_t *min_t)
{
    return guarded([&]() -> int32_t { return azh_screen_target_host_impl(c, times, n_times, offsets, target, threshold_km, reference_jd, min_dist, min_t); });
}

// All-vs-all coarse screen of device-resident positions (coarseScreen, bindings/python/src/
// conjunction.zig L11-150).  Result triplets come back on the host sorted by (t, s, other) -- the
// reference's order within one (t, s) depends on its hash-chain order; the set is identical.
namespace {
int32_t coarse_screen(const double *d_pos, size_t n_sats, size_t n_times, int32_t layout, size_t stride,
                      double threshold_km, const uint8_t *valid_mask, uint32_t *out_pairs, uint32_t *out_t,
                      size_t max_results, size_t *n_found, void *stream, int skip_zero)
{
    if (!d_pos || !n_found || (max_results && (!out_pairs || !out_t))) return AZ_ERR_NULL_POINTER;
    *n_found = 0;
    if (layout < 0 || layout > 1 || !(threshold_km > 0.0)) return AZ_ERR_VALUE;
    if (n_sats == 0 || n_times == 0) return AZ_OK;
    if (n_sats > 0x7fffffffu || n_times > 0xffffffffu) return AZ_ERR_VALUE;
    if (stride == 0) stride = n_sats;
    hipStream_t st = (hipStream_t)stream;
    // stream == NULL: the positions may still be being written by launches on a constellation's own
    // (non-blocking) streams, which the null stream does not order against -- wait for the device
    if (st == nullptr) HIP_TRY(hipDeviceSynchronize());
    // one bucket table per time step of a chunk; >= 2 buckets per satellite, at least the reference's 2^16
    unsigned bits = 16;
    while (bits < 24 && ((size_t)1 << bits) < 2 * n_sats) ++bits;
    const size_t table = (size_t)1 << bits;
    const unsigned chunk = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(n_times, 64), ((size_t)1 << 25) / (table + n_sats)));
    // Scratch of the screen: grow-only buffers per device, kept between calls (round 6: five hipMalloc / hipFree pairs per call
    // -- 16 MB of tables, and 120 MB of result space for the Python default of 10^7 results -- were 1 ms of a 1.8-ms call,
    // profiles/r06_screen_all.txt).  One screen at a time per device (the mutex is held for the whole call).
    int dev_id = 0;
    HIP_TRY(hipGetDevice(&dev_id));
    struct Scratch {
        std::mutex mu;
        DevBuf<unsigned> head, occupied, next, pairs, t;
        DevBuf<unsigned long long> count;
        DevBuf<uint8_t> valid;
    };
    static std::mutex table_mu;
    static std::map<int, std::unique_ptr<Scratch>> per_device;
    Scratch *S = nullptr;
    {
        std::lock_guard<std::mutex> lk(table_mu);
        auto &slot = per_device[dev_id];
        if (!slot) slot.reset(new Scratch());
        S = slot.get();
    }
    std::lock_guard<std::mutex> lock(S->mu);
    int32_t rc = AZ_OK;
    std::vector<uint32_t> hp, ht;
    // result space on the device: room for 2^20 pairs to begin with (a day of a 13,478-satellite catalog at 10 km: ~2,500); a
    // screen that finds more runs once more with room for all of them
    size_t cap = std::max<size_t>(std::min<size_t>(max_results, (size_t)1 << 20), 1 << 16);
    unsigned long long total = 0;
    do {
        if (S->head.ensure(table * chunk) != AZ_OK || S->occupied.ensure(table / 32 * chunk) != AZ_OK || S->next.ensure(n_sats * chunk) != AZ_OK ||
            S->count.ensure(1) != AZ_OK) { rc = AZ_ERR_HIP; break; }
        uint8_t *d_valid = nullptr;
        if (valid_mask) {
            if (S->valid.ensure(n_sats) != AZ_OK ||
                !hip_ok(hipMemcpyAsync(S->valid.p, valid_mask, n_sats, hipMemcpyHostToDevice, st), "H2D valid")) { rc = AZ_ERR_HIP; break; }
            d_valid = S->valid.p;
        }
        for (int pass = 0; pass < 2 && rc == AZ_OK; ++pass) {
            if (S->pairs.ensure(2 * cap) != AZ_OK || S->t.ensure(cap) != AZ_OK ||
                !hip_ok(hipMemsetAsync(S->count.p, 0, sizeof(unsigned long long), st), "memset count")) { rc = AZ_ERR_HIP; break; }
            CellArgs a{};
            a.pos = d_pos;
            a.n_sats = (unsigned)n_sats;
            a.n_times = (unsigned)n_times;
            a.layout = layout;
            a.stride_sats = stride;
            a.valid = d_valid;
            a.inv_cell = 1.0 / threshold_km;
            a.thr2 = threshold_km * threshold_km;
            a.table_mask = (unsigned)(table - 1);
            a.head = S->head.p;
            a.occupied = S->occupied.p;
            a.next = S->next.p;
            a.out_pairs = S->pairs.p;
            a.out_t = S->t.p;
            a.count = S->count.p;
            a.max_results = cap;
            a.skip_zero = skip_zero;
            for (size_t t0 = 0; t0 < n_times && rc == AZ_OK; t0 += chunk) {
                a.t0 = (unsigned)t0;
                a.n_steps = (unsigned)std::min<size_t>(chunk, n_times - t0);
                if (!hip_ok(hipMemsetAsync(S->head.p, 0xff, sizeof(unsigned) * table * a.n_steps, st), "memset head")) { rc = AZ_ERR_HIP; break; }
                // every step's workgroups on one XCD (az_cells_slot): 8 x groups x ceil(n_steps / 8) workgroups
                dim3 grid(8u * (unsigned)((n_sats + 255) / 256) * ((a.n_steps + 7u) / 8u));
                hipLaunchKernelGGL(k_cells_build, grid, dim3(256), 0, st, a);
                hipLaunchKernelGGL(k_cells_bits, dim3((unsigned)(table * a.n_steps / 256)), dim3(256), 0, st, S->head.p, S->occupied.p, table * a.n_steps);
                hipLaunchKernelGGL(k_cells_probe, grid, dim3(256), (table / 32 <= AZ_CELL_BITMAP_WORDS) ? (unsigned)(table / 32 * sizeof(unsigned)) : 0u, st, a);
                if (!hip_ok(hipGetLastError(), "k_cells")) { rc = AZ_ERR_HIP; break; }
            }
            if (rc != AZ_OK) break;
            if (!hip_ok(hipMemcpyAsync(&total, S->count.p, sizeof(total), hipMemcpyDeviceToHost, st), "D2H count") ||
                !hip_ok(hipStreamSynchronize(st), "sync(screen)")) { rc = AZ_ERR_HIP; break; }
            if (total <= cap) break;
            cap = (size_t)total; // the result buffer overflowed: one more pass with room for everything
        }
        if (rc != AZ_OK) break;
        const size_t got = (size_t)std::min<unsigned long long>(total, cap);
        hp.resize(2 * got);
        ht.resize(got);
        if (got) {
            if (!hip_ok(hipMemcpyAsync(hp.data(), S->pairs.p, sizeof(uint32_t) * 2 * got, hipMemcpyDeviceToHost, st), "D2H pairs") ||
                !hip_ok(hipMemcpyAsync(ht.data(), S->t.p, sizeof(uint32_t) * got, hipMemcpyDeviceToHost, st), "D2H t") ||
                !hip_ok(hipStreamSynchronize(st), "sync(screen)")) { rc = AZ_ERR_HIP; break; }
        }
        std::vector<size_t> order(got);
        for (size_t i = 0; i < got; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](size_t x, size_t y) {
            if (ht[x] != ht[y]) return ht[x] < ht[y];
            if (hp[2 * x] != hp[2 * y]) return hp[2 * x] < hp[2 * y];
            return hp[2 * x + 1] < hp[2 * y + 1];
        });
        const size_t keep = std::min(got, max_results);
        for (size_t i = 0; i < keep; ++i) {
            out_pairs[2 * i] = hp[2 * order[i]];
            out_pairs[2 * i + 1] = hp[2 * order[i] + 1];
            out_t[i] = ht[order[i]];
        }
        *n_found = keep;
    } while (0);
    if (rc != AZ_OK) (void)hipStreamSynchronize(st);
    return rc;
}
} // namespace

int32_t azh_coarse_screen_device(const double *d_pos, size_t n_sats, size_t n_times, int32_t layout, size_t stride,
                                 double threshold_km, const uint8_t *valid_mask, uint32_t *out_pairs, uint32_t *out_t,
                                 size_t max_results, size_t *n_found, void *stream)
{
    return coarse_screen(d_pos, n_sats, n_times, layout, stride, threshold_km, valid_mask, out_pairs, out_t, max_results,
                         n_found, stream, 0);
}

int32_t azh_coarse_screen_host(const double *pos, size_t n_sats, size_t n_times, int32_t layout, size_t stride,
                               double threshold_km, const uint8_t *valid_mask, uint32_t *out_pairs, uint32_t *out_t,
                               size_t max_results, size_t *n_found, int32_t device)
{
    if (!pos || !n_found) return AZ_ERR_NULL_POINTER;
    HIP_TRY(hipSetDevice(device));
    if (stride == 0) stride = n_sats;
    const size_t rows = (layout == AZ_LAYOUT_TIME_MAJOR) ? stride : n_sats;
    const size_t bytes = rows * n_times * 3 * sizeof(double);
    double *d_pos = nullptr;
    if (bytes == 0) { *n_found = 0; return AZ_OK; }
    HIP_TRY(hipMalloc((void **)&d_pos, bytes));
    int32_t rc = AZ_OK;
    if (!hip_ok(hipMemcpy(d_pos, pos, bytes, hipMemcpyHostToDevice), "H2D pos")) rc = AZ_ERR_HIP;
    if (rc == AZ_OK)
        rc = azh_coarse_screen_device(d_pos, n_sats, n_times, layout, stride, threshold_km, valid_mask, out_pairs, out_t,
                                      max_results, n_found, nullptr);
    (void)hipFree(d_pos);
    return rc;
}

// screen() without a target (bindings/python/astroz/__init__.py L633-658): propagate every member
// to TEME on the device (time-major scratch, never copied to the host) and run the coarse screen on it
static int32_t azh_screen_all_host_impl(azh_constellation *c, const double *times, size_t n_times, const double *offsets,
                            double threshold_km, uint32_t *out_pairs, uint32_t *out_t, size_t max_results,
                            size_t *n_found)
{
    if (!c || !n_found || (n_times && !times)) return AZ_ERR_NULL_POINTER;
    *n_found = 0;
    if (n_times == 0) return AZ_OK;
    if (set_device(c) != AZ_OK) return AZ_ERR_HIP;
    // (the handle's grow-only result buffer: no 39-MB hipMalloc / hipFree pair per call)
    if (c->d_host_pos.cap < 3 * c->n * n_times) HIP_TRY(hipStreamSynchronize(c->s_main));
    if (c->d_host_pos.ensure(3 * c->n * n_times) != AZ_OK) return AZ_ERR_HIP;
    double *d_pos = c->d_host_pos.p;
    int32_t rc = azh_propagate_device(c, times, n_times, offsets, d_pos, nullptr, AZ_OUT_TEME, 0.0, nullptr,
                                      AZ_LAYOUT_TIME_MAJOR, 0, nullptr, nullptr);
    // rows the propagator zero-filled (failed init, failed deep-space step) are skipped: a satellite
    // is never at the geocentre, and two such rows must not be reported as a conjunction
    if (rc == AZ_OK)
        rc = coarse_screen(d_pos, c->n, n_times, AZ_LAYOUT_TIME_MAJOR, 0, threshold_km, nullptr, out_pairs, out_t,
                           max_results, n_found, c->s_main, 1);
    (void)hipStreamSynchronize(c->s_main);
    return rc;
}
int32_t azh_screen_all_host(azh_constellation *c, const double *times, size_t n_times, const double *offsets,
                            double threshold_km, uint32_t *out_pairs, uint32_t *out_t, size_t max_results,
                            size_t *n_found)
{
    return guarded([&]() -> int32_t { return azh_screen_all_host_impl(c, times, n_times, offsets, threshold_km, out_pairs, out_t, max_results, n_found); });
}

static int32_t azh_propagate_host_impl(azh_constellation *c, const double *times, size_t n_times, const double *offsets,
                           double *pos, double *vel, int32_t mode, double reference_jd, const uint8_t *mask,
                           int32_t layout, size_t stride, uint8_t *err)
{
    if (!c || !pos || (n_times && !times)) return AZ_ERR_NULL_POINTER;
    if (n_times == 0) return AZ_OK;
    if (set_device(c) != AZ_OK) return AZ_ERR_HIP;
    if (stride == 0) stride = c->n;
    const size_t rows = (layout == AZ_LAYOUT_TIME_MAJOR) ? stride : c->n;
    const size_t bytes = rows * n_times * 3 * sizeof(double);
    // device-side result buffers live in the handle and only ever grow: repeated calls (the Python propagate(),
    // SatrecArray.sgp4) do not pay a hipMalloc/hipFree pair of hundreds of megabytes each time
    const size_t words = bytes / sizeof(double);
    // ONE satellite (BASELINE config 1: SatrecArray([sat]).sgp4(jd, fr)), TEME, no mask: (n_times, 1, 3) and (1, n_times, 3) are the
    // same bytes as the one-satellite path's (n_times, 3), so the call IS azh_propagate_one_host on tsince = times + offset -- one
    // kernel that reads its times from and writes into a pinned buffer, no staging of a grid, no increment / record / plan
    // kernels: 30 us whether the grid repeats or not (through the constellation launch set: 49 us repeated, 109 us on a fresh grid)
    // A FEW satellites x a few times (SatrecArray of a handful of records at one instant, a c_api client's tiny batch): the host
    // route of the one-satellite calls, satellite by satellite (host_step.h; the table of a handle of <= 64 satellites is
    // mirrored on the host) -- no staging, no launch, no synchronize.  Cost counted in near-earth points (a deep-space point = 2).
    if (c->n > 1 && !c->h_el.empty() && mode == AZ_OUT_TEME && mask == nullptr && host_points() > 0) {
        size_t cost = 0;
        for (size_t sidx = 0; sidx < c->n; ++sidx) cost += n_times * ((c->h_flags[sidx] & AZ_FLAG_DEEP) ? 2 : 1);
        if (cost <= host_points()) {
            std::vector<double> ts(n_times), p(3 * n_times), v(3 * n_times);
            std::vector<uint8_t> e(n_times);
            for (size_t sidx = 0; sidx < c->n; ++sidx) {
                const double o = offsets ? offsets[sidx] : 0.0;
                for (size_t t = 0; t < n_times; ++t) ts[t] = times[t] + o;
                azhost::propagate_points(c->h_el.data(), c->n_pad, sidx, c->h_flags[sidx], c->g, ts.data(), n_times, 0, nullptr, p.data(),
                                         vel ? v.data() : nullptr, e.data());
                for (size_t t = 0; t < n_times; ++t) {
                    const size_t at = (layout == AZ_LAYOUT_TIME_MAJOR ? t * stride + sidx : sidx * n_times + t) * 3;
                    memcpy(pos + at, &p[3 * t], 3 * sizeof(double));
                    if (vel) memcpy(vel + at, &v[3 * t], 3 * sizeof(double));
                    if (err) err[sidx * n_times + t] = e[t];
                }
            }
            c->last_path = AZH_PATH_HOST_STEP;
            c->cached_n_times = 0; // (nothing is staged on this route)
            c->staged_valid = false;
            c->timed = false;
            return AZ_OK;
        }
    }
    if (c->n == 1 && mode == AZ_OUT_TEME && mask == nullptr && (layout == AZ_LAYOUT_SAT_MAJOR || stride <= 1) && n_times <= kOneStage &&
        AZ_FLAG_ERR(c->h_flags[0]) == 0) {
        std::vector<double> ts(times, times + n_times);
        if (offsets)
            for (auto &t : ts) t += offsets[0];
        c->last_path = 0; // (none of the constellation kernel families)
        // nothing is staged on this route: the cached-input entry points must not find an older grid
        c->cached_n_times = 0;
        c->staged_valid = false;
        c->timed = false;
        return run_one_satellite(c, 0, ts.data(), n_times, 0, nullptr, pos, vel, err);
    }
    // A few satellites (what Satrec / SatrecArray([sat]) / a c_api client make): the kernels write straight into a pinned host
    // buffer of the handle (device-addressable, coherent) and one synchronize ends the call -- three pageable device-to-host
    // copies of a few KB cost ~15 us each.
    const size_t err_bytes = err ? c->n * n_times : 0, small_total = bytes * (vel ? 2 : 1) + ((err_bytes + 63) & ~size_t(63));
    if (small_total <= kSmallOut && mask == nullptr && !(layout == AZ_LAYOUT_TIME_MAJOR && stride > c->n)) {
        if (c->h_small_cap < small_total) {
            if (c->h_small) (void)hipHostFree(c->h_small);
            c->h_small = nullptr;
            c->h_small_cap = 0;
            size_t cap = 65536;
            while (cap < small_total) cap *= 2;
            HIP_TRY(hipHostMalloc(&c->h_small, cap, hipHostMallocDefault));
            c->h_small_cap = cap;
        }
        char *dev = nullptr;
        HIP_TRY(hipHostGetDevicePointer((void **)&dev, c->h_small, 0));
        const size_t o_vel = bytes, o_err = bytes * (vel ? 2 : 1);
        int32_t rc = azh_propagate_device(c, times, n_times, offsets, reinterpret_cast<double *>(dev), vel ? reinterpret_cast<double *>(dev + o_vel) : nullptr,
                                          mode, reference_jd, mask, layout, stride, err ? reinterpret_cast<uint8_t *>(dev + o_err) : nullptr, nullptr);
        if (rc == AZ_OK && !hip_ok(hipStreamSynchronize(c->s_main), "sync")) rc = AZ_ERR_HIP;
        if (rc != AZ_OK) {
            (void)hipStreamSynchronize(c->s_main);
            return rc;
        }
        const char *h = static_cast<const char *>(c->h_small);
        memcpy(pos, h, bytes);
        if (vel) memcpy(vel, h + o_vel, bytes);
        if (err) memcpy(err, h + o_err, err_bytes);
        return AZ_OK;
    }
    if (c->d_host_pos.ensure(words) != AZ_OK || (vel && c->d_host_vel.ensure(words) != AZ_OK) ||
        (err && c->d_host_err.ensure(c->n * n_times) != AZ_OK))
        return AZ_ERR_HIP;
    double *d_pos = c->d_host_pos.p, *d_vel = vel ? c->d_host_vel.p : nullptr;
    uint8_t *d_err = err ? c->d_host_err.p : nullptr;
    int32_t rc = AZ_OK;
    do {
        // rows the kernels do not touch (masked satellites, stride padding) must come back unchanged
        const bool partial = mask != nullptr || (layout == AZ_LAYOUT_TIME_MAJOR && stride > c->n);
        if (partial) {
            if (!hip_ok(hipMemcpyAsync(d_pos, pos, bytes, hipMemcpyHostToDevice, c->s_main), "H2D pos")) { rc = AZ_ERR_HIP; break; }
            if (vel && !hip_ok(hipMemcpyAsync(d_vel, vel, bytes, hipMemcpyHostToDevice, c->s_main), "H2D vel")) { rc = AZ_ERR_HIP; break; }
        }
        rc = azh_propagate_device(c, times, n_times, offsets, d_pos, d_vel, mode, reference_jd, mask, layout, stride, d_err, nullptr);
        if (rc != AZ_OK) break;
        void *const dst[3] = {pos, vel, err};
        const void *const src[3] = {d_pos, d_vel, d_err};
        const size_t len[3] = {bytes, vel ? bytes : 0, err ? c->n * n_times : 0};
        const unsigned thr = host_copy_threads();
        // result arrays from azh_host_alloc are pinned: the DMA goes straight into them (no staging hop, no copy threads)
        const bool pinned_out = host_pool().owns(pos, bytes) && (!vel || host_pool().owns(vel, bytes));
        if (!pinned_out && thr > 0 && len[0] + len[1] + len[2] >= (size_t(8) << 20)) {
            if ((rc = copy_back_staged(c->stager, dst, src, len, 3, c->s_main, thr)) != AZ_OK) break;
        } else {
            for (int k = 0; k < 3 && rc == AZ_OK; ++k)
                if (len[k] && !hip_ok(hipMemcpyAsync(dst[k], src[k], len[k], hipMemcpyDeviceToHost, c->s_main), "D2H")) rc = AZ_ERR_HIP;
            if (rc != AZ_OK) break;
        }
        if (!hip_ok(hipStreamSynchronize(c->s_main), "sync")) { rc = AZ_ERR_HIP; break; }
    } while (0);
    if (rc != AZ_OK) (void)hipStreamSynchronize(c->s_main);
    return rc;
}
int32_t azh_propagate_host(azh_constellation *c, const double *times, size_t n_times, const double *offsets,
                           double *pos, double *vel, int32_t mode, double reference_jd, const uint8_t *mask,
                           int32_t layout, size_t stride, uint8_t *err)
{
    return guarded([&]() -> int32_t { return azh_propagate_host_impl(c, times, n_times, offsets, pos, vel, mode, reference_jd, mask, layout, stride, err); });
}

static int32_t azh_propagate_jd_host_impl(azh_constellation *c, const double *jd, const double *fr, size_t n_times, double *pos,
                              double *vel, int32_t mode, int32_t layout, uint8_t *err)
{
    if (!c || !jd || !fr || !pos) return AZ_ERR_NULL_POINTER;
    // Constellation.propagate (src/Constellation.zig L266-269): tsinceBase = (jd+fr - refEpoch)*1440,
    // offsets = (refEpoch - epoch)*1440 (L153); GMST at jd+fr
    // reference epoch: the first near-earth member's (Constellation.zig L139-140); a constellation without one
    // uses its first member's
    double ref = c->h_epoch.empty() ? 0.0 : c->h_epoch[0];
    for (size_t s = 0; s < c->n; ++s)
        if (AZ_FLAG_ERR(c->h_flags[s]) == 0 && !(c->h_flags[s] & AZ_FLAG_DEEP)) { ref = c->h_epoch[s]; break; }
    std::vector<double> times(n_times), offs(c->n);
    for (size_t t = 0; t < n_times; ++t) times[t] = ((jd[t] + fr[t]) - ref) * 1440.0;
    for (size_t s = 0; s < c->n; ++s) offs[s] = (ref - c->h_epoch[s]) * 1440.0;
    return azh_propagate_host(c, times.data(), n_times, offs.data(), pos, vel, mode, ref, nullptr, layout, 0, err);
}
int32_t azh_propagate_jd_host(azh_constellation *c, const double *jd, const double *fr, size_t n_times, double *pos,
                              double *vel, int32_t mode, int32_t layout, uint8_t *err)
{
    return guarded([&]() -> int32_t { return azh_propagate_jd_host_impl(c, jd, fr, n_times, pos, vel, mode, layout, err); });
}

int32_t azh_synchronize(azh_constellation *c)
{
    if (!c) return AZ_ERR_NULL_POINTER;
    if (set_device(c) != AZ_OK) return AZ_ERR_HIP;
    HIP_TRY(hipStreamSynchronize(c->s_deep));
    HIP_TRY(hipStreamSynchronize(c->s_ecc));
    HIP_TRY(hipStreamSynchronize(c->s_main));
    return AZ_OK;
}

uint32_t azh_last_path(const azh_constellation *c) { return c ? c->last_path : 0u; }

int32_t azh_last_one_stats(azh_constellation *c, uint32_t *n_segments, uint32_t *n_handed_over)
{
    if (!c || !n_segments || !n_handed_over) return AZ_ERR_NULL_POINTER;
    *n_segments = c->one_segments;
    *n_handed_over = 0;
    if (c->one_segments == 0) return AZ_OK;
    if (set_device(c) != AZ_OK) return AZ_ERR_HIP;
    HIP_TRY(hipEventSynchronize(c->ev_one));
    unsigned head[AZ_ONE_HEAD]; // (8 KB: the list heads sit on their own 128-byte lines)
    HIP_TRY(hipMemcpy(head, c->d_one_items.p, sizeof(head), hipMemcpyDeviceToHost));
    unsigned cnt = 0;
    for (unsigned k = 0; k < AZ_ONE_LISTS; ++k) cnt += head[32u * k];
    *n_handed_over = cnt;
    return AZ_OK;
}

double azh_last_kernel_ms(azh_constellation *c)
{
    if (!c || !c->timed) return -1.0;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, c->ev_t0, c->ev_t1) != hipSuccess) {
        (void)hipGetLastError();
        return -1.0;
    }
    return (double)ms;
}

static int32_t azh_propagate_one_host_impl(azh_constellation *c, size_t sat, const double *tsince, size_t n, double *pos,
                               double *vel, uint8_t *err)
{
    if (!c || !tsince || !pos) return AZ_ERR_NULL_POINTER;
    if (sat >= c->n) return AZ_ERR_VALUE;
    if (n == 0) return AZ_OK;
    return run_one_satellite(c, sat, tsince, n, 0, nullptr, pos, vel, err);
}
int32_t azh_propagate_one_host(azh_constellation *c, size_t sat, const double *tsince, size_t n, double *pos,
                               double *vel, uint8_t *err)
{
    return guarded([&]() -> int32_t { return azh_propagate_one_host_impl(c, sat, tsince, n, pos, vel, err); });
}

static int32_t azh_propagate_one_device_impl(azh_constellation *c, size_t sat, const double *d_tsince, size_t n, double *d_pos,
                                 double *d_vel, uint8_t *d_err, void *stream)
{
    if (!c || !d_tsince || !d_pos) return AZ_ERR_NULL_POINTER;
    if (sat >= c->n || n > 0xffffffffu) return AZ_ERR_VALUE;
    if (n == 0) return AZ_OK;
    if (set_device(c) != AZ_OK) return AZ_ERR_HIP;
    hipStream_t st = stream ? (hipStream_t)stream : c->s_main;
    if (c->timing) HIP_TRY(hipEventRecord(c->ev_t0, st));
    if (int32_t lrc = launch_one(c, sat, d_tsince, n, d_pos, d_vel, d_err, 0, st); lrc != AZ_OK) return lrc;
    if (c->timing) {
        HIP_TRY(hipEventRecord(c->ev_t1, st));
        c->timed = true;
    }
    return AZ_OK;
}
int32_t azh_propagate_one_device(azh_constellation *c, size_t sat, const double *d_tsince, size_t n, double *d_pos,
                                 double *d_vel, uint8_t *d_err, void *stream)
{
    return guarded([&]() -> int32_t { return azh_propagate_one_device_impl(c, sat, d_tsince, n, d_pos, d_vel, d_err, stream); });
}

// device-side known-answer hook for the element math of the kernels (devmath.h): out[0..n) sin, [n..2n) cos,
// [2n..3n) x * az_rcp(x), [3n..4n) x * az_rsqrt(x)^2, [4n..6n) (sin,cos)(0.7321 + x) by az_rotate from (sin,cos)(0.7321)
int32_t azh_selftest_math(const double *x, size_t n, double *out6n, int32_t device)
{
    if (!x || !out6n) return AZ_ERR_NULL_POINTER;
    if (n == 0) return AZ_OK;
    HIP_TRY(hipSetDevice(device));
    double *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, sizeof(double) * 7 * n));
    int32_t rc = AZ_OK;
    if (!hip_ok(hipMemcpy(d, x, sizeof(double) * n, hipMemcpyHostToDevice), "H2D")) rc = AZ_ERR_HIP;
    if (rc == AZ_OK) {
        hipLaunchKernelGGL(k_math_kat, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, nullptr, d, (unsigned)n, d + n);
        if (!hip_ok(hipGetLastError(), "k_math_kat") ||
            !hip_ok(hipMemcpy(out6n, d + n, sizeof(double) * 6 * n, hipMemcpyDeviceToHost), "D2H"))
            rc = AZ_ERR_HIP;
    }
    (void)hipFree(d);
    return rc;
}

int32_t azh_selftest_host_step(const double *el, size_t n_pad, size_t sat, uint32_t flags, int32_t grav, const double *tsince_min,
                               size_t n, double *out6n, uint8_t *err)
{
    if (!el || !tsince_min || !out6n) return AZ_ERR_NULL_POINTER;
    if (sat >= n_pad) return AZ_ERR_VALUE;
    if (!azhost::cpu_ok()) return AZ_ERR_UNKNOWN;
    azhost::propagate_points(el, n_pad, sat, flags, make_grav(grav), tsince_min, n, 1, out6n, nullptr, nullptr, err);
    return AZ_OK;
}

// ======================================================================================= multi-GPU group
// One process, N devices (a Zig / C host has no torch.distributed): block-cyclic satellite shards -- the same plan
// as astroz_amd/distributed.py's ShardPlan -- one azh_constellation per device.  A host-memory result needs no
// collective at all (every device copies its blocks straight into the caller's catalog-ordered array over its own
// PCIe link); a result that must be resident on EVERY device is re-assembled by RCCL all-gathers over xGMI,
// chunk-pipelined against the propagation of the next chunk.  The reference has no counterpart
// (single process, std.Thread: src/Constellation.zig L327-385).
struct azh_group {
    int n_dev = 0;
    size_t n = 0, rows = 0, n_chunks = 1;
    std::vector<int> devices;
    std::vector<azh_constellation *> shard;
    std::vector<std::vector<uint32_t>> members; // catalog rows of every shard, ascending (== local order)
    std::vector<std::vector<double>> off_stage; // per-shard epoch offsets of the call in flight (source of an async H2D)
    std::vector<double> track_stage;            // the target's track of the group screen in flight (source of async H2Ds)
    std::vector<ncclComm_t> comms;              // created on the first all-gather
    std::vector<hipStream_t> s_comm;
    std::vector<hipEvent_t> ev;
    size_t cell_lo(size_t chunk, int d) const { return std::min((chunk * n_dev + d) * rows, n); }
    size_t cell_hi(size_t chunk, int d) const { return std::min((chunk * n_dev + d) * rows + rows, n); }
    size_t padded() const { return n_chunks * n_dev * rows; }
};

namespace {

struct Rccl {
    void *h = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy already mapped into the process (PyTorch ships its own) is picked up by SONAME
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.h) break;
        }
        if (!r.h) return;
#define AZ_SYM(f) r.f = reinterpret_cast<decltype(r.f)>(dlsym(r.h, "nccl" #f))
        AZ_SYM(CommInitAll); AZ_SYM(CommDestroy); AZ_SYM(AllGather); AZ_SYM(GroupStart); AZ_SYM(GroupEnd); AZ_SYM(GetErrorString);
#undef AZ_SYM
        r.ok = r.CommInitAll && r.CommDestroy && r.AllGather && r.GroupStart && r.GroupEnd;
    });
    return r;
}

bool nccl_ok(ncclResult_t e, const char *what)
{
    if (e == ncclSuccess) return true;
    g_last_error = std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(e) : "RCCL error");
    return false;
}

void group_destroy(azh_group *g)
{
    if (!g) return;
    for (size_t d = 0; d < g->comms.size(); ++d)
        if (g->comms[d]) (void)rccl().CommDestroy(g->comms[d]);
    for (int d = 0; d < (int)g->s_comm.size(); ++d) {
        (void)hipSetDevice(g->devices[d]);
        if (g->s_comm[d]) (void)hipStreamDestroy(g->s_comm[d]);
        if (d < (int)g->ev.size() && g->ev[d]) (void)hipEventDestroy(g->ev[d]);
    }
    for (azh_constellation *c : g->shard) destroy(c);
    delete g;
}

int32_t group_build(const std::vector<azh::TleRecord> &recs, int grav, const int32_t *devices, int32_t n_devices,
                    int32_t n_chunks, azh_group **out)
{
    if (!devices || !out) return AZ_ERR_NULL_POINTER;
    *out = nullptr;
    if (n_devices < 1 || n_chunks < 1 || recs.empty()) return AZ_ERR_VALUE;
    azh_group *g = new (std::nothrow) azh_group();
    if (!g) return AZ_ERR_ALLOC_FAILED;
    g->n_dev = n_devices;
    g->n = recs.size();
    g->devices.assign(devices, devices + n_devices);
    // ShardPlan: rows per (chunk, device) cell, a multiple of 64; chunks that would be pure padding are dropped
    const size_t cells = (size_t)n_devices * (size_t)n_chunks;
    g->rows = ((g->n + cells - 1) / cells + 63) / 64 * 64;
    g->n_chunks = std::max<size_t>(1, std::min<size_t>((size_t)n_chunks, (g->n + n_devices * g->rows - 1) / (n_devices * g->rows)));
    g->shard.assign(n_devices, nullptr);
    g->members.resize(n_devices);
    g->off_stage.resize(n_devices);
    int32_t rc = AZ_OK;
    for (int d = 0; d < n_devices && rc == AZ_OK; ++d) {
        std::vector<azh::TleRecord> mine;
        for (size_t c = 0; c < g->n_chunks; ++c)
            for (size_t r = g->cell_lo(c, d); r < g->cell_hi(c, d); ++r) {
                mine.push_back(recs[r]);
                g->members[d].push_back((uint32_t)r);
            }
        if (mine.empty()) continue; // more devices than 64-satellite cells
        rc = build_from_records(mine, grav, devices[d], &g->shard[d]);
    }
    if (rc != AZ_OK) {
        group_destroy(g);
        return rc;
    }
    *out = g;
    return AZ_OK;
}

// stage one shard's inputs and (optionally) launch all of it into its grow-only local blocks
int32_t group_stage(azh_group *g, int d, const double *times, size_t n_times, const double *offsets, int32_t mode,
                    double reference_jd, bool want_vel, bool want_err)
{
    azh_constellation *c = g->shard[d];
    if (!c) return AZ_OK;
    if (set_device(c) != AZ_OK) return AZ_ERR_HIP;
    std::vector<double> &off = g->off_stage[d]; // outlives the asynchronous upload (the callers synchronise before returning)
    if (offsets) {
        off.resize(c->n);
        for (size_t i = 0; i < c->n; ++i) off[i] = offsets[g->members[d][i]];
    }
    const size_t cap = g->n_chunks * g->rows * n_times * 3; // cells padded to `rows`: the all-gather needs equal blocks
    if (c->d_host_pos.ensure(cap) != AZ_OK || (want_vel && c->d_host_vel.ensure(cap) != AZ_OK) ||
        (want_err && c->d_host_err.ensure(c->n * n_times) != AZ_OK))
        return AZ_ERR_HIP;
    return stage_inputs(c, times, n_times, offsets ? off.data() : nullptr, nullptr, mode, reference_jd, c->s_main);
}

} // namespace

int32_t azh_group_create_from_tle_text(const char *text, size_t len, int32_t grav, const int32_t *devices,
                                       int32_t n_devices, int32_t n_chunks, azh_group **out)
{
    return guarded([&]() -> int32_t {
        if (!text) return AZ_ERR_NULL_POINTER;
        std::vector<azh::TleRecord> recs;
        azh::parse_all(std::string_view(text, len), recs);
        if (recs.empty()) return AZ_ERR_BAD_TLE_LENGTH;
        return group_build(recs, grav, devices, n_devices, n_chunks, out);
    });
}

int32_t azh_group_create_from_omm_json(const char *text, size_t len, int32_t grav, const int32_t *devices,
                                       int32_t n_devices, int32_t n_chunks, azh_group **out)
{
    return guarded([&]() -> int32_t {
        if (!text) return AZ_ERR_NULL_POINTER;
        std::vector<azh::TleRecord> recs;
        const int rc = azh::parse_omm_json(std::string_view(text, len), recs);
        if (rc == -1 || (rc == 0 && recs.empty())) return AZ_ERR_BAD_TLE_LENGTH;
        if (rc != 0) return AZ_ERR_VALUE;
        return group_build(recs, grav, devices, n_devices, n_chunks, out);
    });
}

void azh_group_free(azh_group *g) { group_destroy(g); }
size_t azh_group_num_satellites(const azh_group *g) { return g ? g->n : 0; }
int32_t azh_group_num_devices(const azh_group *g) { return g ? g->n_dev : 0; }
size_t azh_group_padded_rows(const azh_group *g) { return g ? g->padded() : 0; }

int32_t azh_group_get_epochs(const azh_group *g, double *out)
{
    if (!g || !out) return AZ_ERR_NULL_POINTER;
    for (int d = 0; d < g->n_dev; ++d)
        if (g->shard[d])
            for (size_t i = 0; i < g->shard[d]->n; ++i) out[g->members[d][i]] = g->shard[d]->h_epoch[i];
    return AZ_OK;
}

static int32_t azh_group_propagate_host_impl(azh_group *g, const double *times, size_t n_times, const double *offsets, size_t n_offsets,
                                 double *pos, double *vel, int32_t mode, double reference_jd, uint8_t *err)
{
    if (!g || !pos || (n_times && !times)) return AZ_ERR_NULL_POINTER;
    if (mode < 0 || mode > 2 || (offsets && n_offsets < g->n)) return AZ_ERR_VALUE;
    if (n_times == 0) return AZ_OK;
    const size_t row = n_times * 3;
    int32_t rc = AZ_OK;
    // enqueue everything on every device first (asynchronous), then wait: the devices run side by side
    for (int d = 0; d < g->n_dev && rc == AZ_OK; ++d) {
        azh_constellation *c = g->shard[d];
        if (!c) continue;
        if ((rc = group_stage(g, d, times, n_times, offsets, mode, reference_jd, vel != nullptr, err != nullptr)) != AZ_OK) break;
        rc = launch_all(c, c->d_host_pos.p, vel ? c->d_host_vel.p : nullptr, AZ_LAYOUT_SAT_MAJOR, 0,
                        err ? c->d_host_err.p : nullptr, c->s_main);
        if (rc != AZ_OK) break;
    }
    // Copies: a cell = consecutive catalog rows = consecutive local rows, one copy per cell and array.  Every device gets its
    // OWN copier thread, which stages its cells through that device's pinned slots while its share of the host copy threads
    // moves each landed chunk into the caller's array (copy_back_staged): the N PCIe links run side by side (issued from one
    // thread, host-synchronous pageable copies would take turns) and at least N threads write the host arrays.
    if (rc == AZ_OK) {
        std::vector<int32_t> rcs(g->n_dev, AZ_OK);
        const size_t arr_bytes = g->n * row * sizeof(double);
        // (result arrays from azh_host_alloc are pinned: every device's DMA goes straight into them)
        const bool staged = host_copy_threads() > 0 && !(host_pool().owns(pos, arr_bytes) && (!vel || host_pool().owns(vel, arr_bytes)));
        const unsigned per_dev = std::max(1u, host_copy_threads() / (unsigned)std::max(1, g->n_dev)); // host threads behind each device
        auto copy_cells = [&](int d) {
            azh_constellation *c = g->shard[d];
            if (!c) return;
            if (hipSetDevice(c->device) != hipSuccess) { rcs[d] = AZ_ERR_HIP; return; }
            std::vector<void *> dst;
            std::vector<const void *> src;
            std::vector<size_t> len;
            for (int arr = 0; arr < 3; ++arr) {
                if ((arr == 1 && !vel) || (arr == 2 && !err)) continue;
                const size_t unit = arr == 2 ? n_times : row * sizeof(double); // bytes per satellite row
                const char *base = arr == 0 ? (const char *)c->d_host_pos.p : arr == 1 ? (const char *)c->d_host_vel.p : (const char *)c->d_host_err.p;
                char *out = arr == 0 ? (char *)pos : arr == 1 ? (char *)vel : (char *)err;
                size_t local = 0;
                for (size_t k = 0; k < g->n_chunks; ++k) {
                    const size_t lo = g->cell_lo(k, d), cnt = g->cell_hi(k, d) - lo;
                    if (!cnt) continue;
                    dst.push_back(out + lo * unit); src.push_back(base + local * unit); len.push_back(cnt * unit);
                    local += cnt;
                }
            }
            if (staged) {
                rcs[d] = copy_back_staged(c->stager, dst.data(), src.data(), len.data(), (int)dst.size(), c->s_main, per_dev);
            } else {
                for (size_t k = 0; k < dst.size(); ++k)
                    if (hipMemcpyAsync(dst[k], src[k], len[k], hipMemcpyDeviceToHost, c->s_main) != hipSuccess) { rcs[d] = AZ_ERR_HIP; return; }
            }
            if (hipStreamSynchronize(c->s_main) != hipSuccess) rcs[d] = AZ_ERR_HIP;
        };
        // (it runs in std::thread: an exception that left it -- a vector growing out of memory -- would be std::terminate)
        auto copier = [&](int d) {
            try {
                copy_cells(d);
            } catch (const std::bad_alloc &) {
                rcs[d] = AZ_ERR_ALLOC_FAILED;
            } catch (...) {
                rcs[d] = AZ_ERR_UNKNOWN;
            }
        };
        std::vector<std::thread> th;
        for (int d = 1; d < g->n_dev; ++d) {
            try {
                th.emplace_back(copier, d);
            } catch (const std::system_error &) {
                copier(d);
            }
        }
        copier(0);
        for (auto &t : th) t.join();
        for (int d = 0; d < g->n_dev; ++d)
            if (rcs[d] != AZ_OK) { rc = rcs[d]; if (g_last_error.empty()) g_last_error = "device-to-host copy failed"; (void)hipGetLastError(); }
    }
    for (int d = 0; d < g->n_dev; ++d) {
        azh_constellation *c = g->shard[d];
        if (!c) continue;
        if (set_device(c) != AZ_OK || !hip_ok(hipStreamSynchronize(c->s_main), "sync") || !hip_ok(hipStreamSynchronize(c->s_deep), "sync"))
            rc = rc == AZ_OK ? AZ_ERR_HIP : rc;
    }
    return rc;
}
int32_t azh_group_propagate_host(azh_group *g, const double *times, size_t n_times, const double *offsets, size_t n_offsets,
                                 double *pos, double *vel, int32_t mode, double reference_jd, uint8_t *err)
{
    return guarded([&]() -> int32_t { return azh_group_propagate_host_impl(g, times, n_times, offsets, n_offsets, pos, vel, mode, reference_jd, err); });
}

static int32_t azh_group_propagate_allgather_impl(azh_group *g, const double *times, size_t n_times, const double *offsets, size_t n_offsets,
                                      double *const *d_pos, double *const *d_vel)
{
    if (!g || !d_pos || (n_times && !times)) return AZ_ERR_NULL_POINTER;
    if (offsets && n_offsets < g->n) return AZ_ERR_VALUE;
    if (n_times == 0) return AZ_OK;
    for (int d = 0; d < g->n_dev; ++d)
        if (!g->shard[d]) { g_last_error = "more devices than 64-satellite cells"; return AZ_ERR_VALUE; }
    Rccl &R = rccl();
    if (!R.ok) { g_last_error = "librccl.so could not be loaded"; return AZ_ERR_HIP; }
    if (g->comms.empty()) {
        g->comms.assign(g->n_dev, nullptr);
        if (!nccl_ok(R.CommInitAll(g->comms.data(), g->n_dev, g->devices.data()), "ncclCommInitAll")) { g->comms.clear(); return AZ_ERR_HIP; }
        g->s_comm.assign(g->n_dev, nullptr);
        g->ev.assign(g->n_dev, nullptr);
        for (int d = 0; d < g->n_dev; ++d) {
            HIP_TRY(hipSetDevice(g->devices[d]));
            HIP_TRY(hipStreamCreateWithFlags(&g->s_comm[d], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&g->ev[d], hipEventDisableTiming));
        }
    }
    const size_t row = n_times * 3, cell = g->rows * row;
    int32_t rc = AZ_OK;
    for (int d = 0; d < g->n_dev && rc == AZ_OK; ++d)
        rc = group_stage(g, d, times, n_times, offsets, AZ_OUT_TEME, 0.0, d_vel != nullptr, false);
    // chunk pipeline: chunk k is propagated on every device's own stream; its all-gather runs on the communication
    // streams while chunk k+1 is being propagated
    for (size_t k = 0; k < g->n_chunks && rc == AZ_OK; ++k) {
        for (int d = 0; d < g->n_dev && rc == AZ_OK; ++d) {
            azh_constellation *c = g->shard[d];
            if (set_device(c) != AZ_OK) { rc = AZ_ERR_HIP; break; }
            const size_t cnt = g->cell_hi(k, d) - g->cell_lo(k, d);
            if (cnt < g->rows) {
                // a partial (or empty) cell: the all-gather still moves `rows` rows of it -- its padding rows are zeros,
                // not whatever an earlier call left in the grow-only buffers
                const size_t at = (k * g->rows + cnt) * row, len = (g->rows - cnt) * row * sizeof(double);
                if (!hip_ok(hipMemsetAsync(c->d_host_pos.p + at, 0, len, c->s_main), "memset pad") ||
                    (d_vel && !hip_ok(hipMemsetAsync(c->d_host_vel.p + at, 0, len, c->s_main), "memset pad"))) { rc = AZ_ERR_HIP; break; }
            }
            if (cnt) rc = launch_all(c, c->d_host_pos.p, d_vel ? c->d_host_vel.p : nullptr, AZ_LAYOUT_SAT_MAJOR, 0, nullptr,
                                     c->s_main, 0, k * g->rows, k * g->rows + cnt);
            if (rc != AZ_OK) break;
            if (!hip_ok(hipEventRecord(g->ev[d], c->s_main), "record") || !hip_ok(hipStreamWaitEvent(g->s_comm[d], g->ev[d], 0), "wait"))
                rc = AZ_ERR_HIP;
        }
        if (rc != AZ_OK) break;
        for (int arr = 0; arr < (d_vel ? 2 : 1) && rc == AZ_OK; ++arr) {
            if (!nccl_ok(R.GroupStart(), "ncclGroupStart")) { rc = AZ_ERR_HIP; break; }
            for (int d = 0; d < g->n_dev; ++d) {
                azh_constellation *c = g->shard[d];
                const double *src = (arr ? c->d_host_vel.p : c->d_host_pos.p) + k * cell;
                double *dst = (arr ? d_vel[d] : d_pos[d]) + k * g->n_dev * cell;
                if (!nccl_ok(R.AllGather(src, dst, cell, ncclDouble, g->comms[d], g->s_comm[d]), "ncclAllGather")) rc = AZ_ERR_HIP;
            }
            if (!nccl_ok(R.GroupEnd(), "ncclGroupEnd")) rc = AZ_ERR_HIP;
        }
    }
    for (int d = 0; d < g->n_dev; ++d) {
        azh_constellation *c = g->shard[d];
        if (set_device(c) != AZ_OK || !hip_ok(hipStreamSynchronize(c->s_main), "sync") || !hip_ok(hipStreamSynchronize(c->s_deep), "sync") ||
            !hip_ok(hipStreamSynchronize(g->s_comm[d]), "sync"))
            rc = rc == AZ_OK ? AZ_ERR_HIP : rc;
    }
    return rc;
}
int32_t azh_group_propagate_allgather(azh_group *g, const double *times, size_t n_times, const double *offsets, size_t n_offsets,
                                      double *const *d_pos, double *const *d_vel)
{
    return guarded([&]() -> int32_t { return azh_group_propagate_allgather_impl(g, times, n_times, offsets, n_offsets, d_pos, d_vel); });
}

// The sharded consumer ("a consumer that only needs its own shard"): the fused single-target screen
// (Constellation.screenConstellation, src/Constellation.zig L683-756) over a group.  Every satellite's minimum distance to the
// target is independent of every other satellite's, so each device screens ITS rows; the only thing they share is the
// target's own track (n_times x 24 bytes: 35 KB for a day of minutes), computed once on the device that owns the target and
// handed to the others through the host.  No collective, nothing but 12 bytes per satellite leaves any device.
namespace {
int32_t group_screen(azh_group *g, const double *times, size_t n_times, const double *offsets, size_t n_offsets, size_t target,
                     double threshold_km, double *const *d_min_dist, uint32_t *const *d_min_t, double *h_min_dist, uint32_t *h_min_t)
{
    if (!g || (n_times && !times)) return AZ_ERR_NULL_POINTER;
    if (target >= g->n || (offsets && n_offsets < g->n)) return AZ_ERR_VALUE;
    // owner of the target and its local row
    int owner = -1;
    size_t t_local = 0;
    for (int d = 0; d < g->n_dev && owner < 0; ++d) {
        const auto &m = g->members[d];
        const auto it = std::lower_bound(m.begin(), m.end(), (uint32_t)target);
        if (it != m.end() && *it == (uint32_t)target) { owner = d; t_local = (size_t)(it - m.begin()); }
    }
    if (owner < 0) return AZ_ERR_VALUE;
    int32_t rc = AZ_OK;
    auto shard_offsets = [&](int d) -> const double * {
        azh_constellation *c = g->shard[d];
        if (!offsets || !c) return nullptr;
        std::vector<double> &off = g->off_stage[d];
        off.resize(c->n);
        for (size_t i = 0; i < c->n; ++i) off[i] = offsets[g->members[d][i]];
        return off.data();
    };
    // the target's track on its owner ...
    std::vector<double> &track = g->track_stage;
    const unsigned nt = (unsigned)n_times;
    if (nt > 0 && g->n_dev > 1) {
        azh_constellation *c = g->shard[owner];
        if (set_device(c) != AZ_OK) return AZ_ERR_HIP;
        if ((rc = stage_inputs(c, times, n_times, shard_offsets(owner), nullptr, AZ_OUT_TEME, 0.0, c->s_main)) != AZ_OK) return rc;
        if (c->d_tgt.ensure((size_t)nt * 3) != AZ_OK) return AZ_ERR_HIP;
        hipLaunchKernelGGL((k_one_satellite<false>), dim3((nt + 63) / 64), dim3(64), 0, c->s_main, c->d_el, c->d_flags, c->n_pad,
                           (unsigned)t_local, c->d_times.p, nt, c->d_tgt.p, (double *)nullptr, (unsigned char *)nullptr, 0, c->g,
                           c->have_offsets ? c->d_offsets.p : (const double *)nullptr, 1);
        HIP_TRY(hipGetLastError());
        track.resize((size_t)nt * 3);
        HIP_TRY(hipMemcpyAsync(track.data(), c->d_tgt.p, sizeof(double) * track.size(), hipMemcpyDeviceToHost, c->s_main));
        HIP_TRY(hipStreamSynchronize(c->s_main));
    }
    // ... then every device screens its own rows.  One host thread per device (N > 1): a screen is half a dozen launches of a few
    // microseconds each, and issued from ONE thread the eighth device would start 0.2 ms after the first -- longer than its
    // whole share takes (1,685 rows: ~30 us of kernels).  Each thread stages, launches, and (host results) copies its cells
    // into the caller's catalog-ordered arrays and waits for its own device.
    std::vector<int32_t> rcs(g->n_dev, AZ_OK);
    for (int d = 0; d < g->n_dev; ++d) (void)shard_offsets(d); // (fills g->off_stage[d] on this thread: no allocation races below)
    auto work = [&](int d) -> int32_t {
        azh_constellation *c = g->shard[d];
        if (!c) return AZ_OK;
        if (hipSetDevice(c->device) != hipSuccess) return AZ_ERR_HIP;
        double *out_d = d_min_dist ? d_min_dist[d] : nullptr;
        uint32_t *out_t = d_min_t ? d_min_t[d] : nullptr;
        if (!out_d || !out_t) {
            if (c->d_out_d.ensure(c->n) != AZ_OK || c->d_out_t.ensure(c->n) != AZ_OK) return AZ_ERR_HIP;
            out_d = c->d_out_d.p;
            out_t = c->d_out_t.p;
        }
        const double *off_d = offsets ? g->off_stage[d].data() : nullptr;
        int32_t r = AZ_OK;
        if (d == owner) {
            r = screen_core(c, times, n_times, off_d, t_local, nullptr, threshold_km, out_d, out_t, nullptr);
        } else {
            if (nt > 0) {
                if (c->d_tgt.ensure((size_t)nt * 3) != AZ_OK) return AZ_ERR_HIP;
                if (hipMemcpyAsync(c->d_tgt.p, track.data(), sizeof(double) * track.size(), hipMemcpyHostToDevice, c->s_main) != hipSuccess) return AZ_ERR_HIP;
            }
            r = screen_core(c, times, n_times, off_d, kNoTarget, c->d_tgt.p, threshold_km, out_d, out_t, nullptr);
        }
        if (r != AZ_OK) return r;
        if (h_min_dist && h_min_t) {
            // host results: a cell = consecutive catalog rows = consecutive local rows, one small copy per cell and array
            size_t local = 0;
            for (size_t k = 0; k < g->n_chunks; ++k) {
                const size_t lo = g->cell_lo(k, d), cnt = g->cell_hi(k, d) - lo;
                if (!cnt) continue;
                if (hipMemcpyAsync(h_min_dist + lo, c->d_out_d.p + local, sizeof(double) * cnt, hipMemcpyDeviceToHost, c->s_main) != hipSuccess ||
                    hipMemcpyAsync(h_min_t + lo, c->d_out_t.p + local, sizeof(uint32_t) * cnt, hipMemcpyDeviceToHost, c->s_main) != hipSuccess)
                    return AZ_ERR_HIP;
                local += cnt;
            }
            if (hipStreamSynchronize(c->s_main) != hipSuccess) return AZ_ERR_HIP;
        }
        return AZ_OK;
    };
    auto guarded_work = [&](int d) {
        try {
            rcs[d] = work(d);
        } catch (const std::bad_alloc &) {
            rcs[d] = AZ_ERR_ALLOC_FAILED;
        } catch (...) {
            rcs[d] = AZ_ERR_UNKNOWN;
        }
    };
    {
        std::vector<std::thread> th;
        for (int d = 1; d < g->n_dev; ++d) {
            try {
                th.emplace_back(guarded_work, d);
            } catch (const std::system_error &) {
                guarded_work(d);
            }
        }
        guarded_work(0);
        for (auto &t : th) t.join();
    }
    for (int d = 0; d < g->n_dev; ++d)
        if (rcs[d] != AZ_OK && rc == AZ_OK) rc = rcs[d];
    if (rc != AZ_OK) {
        if (g_last_error.empty()) g_last_error = "sharded screen failed on a device";
        (void)hipGetLastError();
        for (int d = 0; d < g->n_dev; ++d) {
            azh_constellation *c = g->shard[d];
            if (c && set_device(c) == AZ_OK) (void)hipStreamSynchronize(c->s_main);
        }
    }
    return rc;
}
} // namespace

int32_t azh_group_screen_target_host(azh_group *g, const double *times, size_t n_times, const double *offsets, size_t n_offsets,
                                     size_t target, double threshold_km, double reference_jd, double *min_dist, uint32_t *min_t)
{
    (void)reference_jd; // (signature parity with azh_screen_target_host: a rotation to ECEF does not change a distance)
    return guarded([&]() -> int32_t {
        if (!min_dist || !min_t) return AZ_ERR_NULL_POINTER;
        return group_screen(g, times, n_times, offsets, n_offsets, target, threshold_km, nullptr, nullptr, min_dist, min_t);
    });
}
int32_t azh_group_screen_target_device(azh_group *g, const double *times, size_t n_times, const double *offsets, size_t n_offsets,
                                       size_t target, double threshold_km, double reference_jd, double *const *d_min_dist,
                                       uint32_t *const *d_min_t)
{
    (void)reference_jd;
    return guarded([&]() -> int32_t {
        if (!d_min_dist || !d_min_t) return AZ_ERR_NULL_POINTER;
        if (g)
            for (int d = 0; d < g->n_dev; ++d)
                if (g->shard[d] && (!d_min_dist[d] || !d_min_t[d])) return AZ_ERR_NULL_POINTER;
        return group_screen(g, times, n_times, offsets, n_offsets, target, threshold_km, d_min_dist, d_min_t, nullptr, nullptr);
    });
}
size_t azh_group_shard_size(const azh_group *g, int32_t d) { return (g && d >= 0 && d < g->n_dev) ? g->members[d].size() : 0; }
int32_t azh_group_shard_rows(const azh_group *g, int32_t d, uint32_t *out)
{
    if (!g || !out) return AZ_ERR_NULL_POINTER;
    if (d < 0 || d >= g->n_dev) return AZ_ERR_VALUE;
    std::copy(g->members[d].begin(), g->members[d].end(), out);
    return AZ_OK;
}
int32_t azh_group_synchronize(azh_group *g)
{
    if (!g) return AZ_ERR_NULL_POINTER;
    int32_t rc = AZ_OK;
    for (int d = 0; d < g->n_dev; ++d) {
        azh_constellation *c = g->shard[d];
        if (!c) continue;
        if (set_device(c) != AZ_OK || !hip_ok(hipStreamSynchronize(c->s_main), "sync") || !hip_ok(hipStreamSynchronize(c->s_deep), "sync") ||
            !hip_ok(hipStreamSynchronize(c->s_ecc), "sync"))
            rc = AZ_ERR_HIP;
    }
    return rc;
}

// ======================================================================================= (A)
// The reference's c_api surface (src/c_api/root.zig).

uint32_t astroz_version(void) { return (0u << 16) | (3u << 8) | 0u; }
void astroz_init(void) {}
void astroz_deinit(void) {}

struct TleHandle {
    azh::TleRecord rec;
};

int32_t tle_parse(const char *str, void **out)
{
    if (!str || !out) return AZ_ERR_NULL_POINTER;
    TleHandle *h = new (std::nothrow) TleHandle();
    if (!h) return AZ_ERR_ALLOC_FAILED;
    int rc = azh::parse_first(std::string_view(str, strlen(str)), h->rec);
    if (rc != 0) {
        delete h;
        return rc == -1 ? AZ_ERR_BAD_TLE_LENGTH : AZ_ERR_UNKNOWN;
    }
    *out = h;
    return AZ_OK;
}
void tle_free(void *h) { delete static_cast<TleHandle *>(h); }
uint32_t tle_get_satellite_number(void *h) { return static_cast<TleHandle *>(h)->rec.satnum; }
double tle_get_epoch(void *h) { return (static_cast<TleHandle *>(h)->rec.epoch_jd - 2451545.0) * 86400.0; }
double tle_get_inclination(void *h) { return static_cast<TleHandle *>(h)->rec.incl_deg; }
double tle_get_eccentricity(void *h) { return static_cast<TleHandle *>(h)->rec.ecc; }
double tle_get_mean_motion(void *h) { return static_cast<TleHandle *>(h)->rec.mm_revday; }

struct Sgp4Handle {
    azh_constellation *c;
};

int32_t sgp4_init(void *tle, int32_t grav, void **out)
{
    if (!tle || !out) return AZ_ERR_NULL_POINTER;
    std::vector<azh::TleRecord> recs(1, static_cast<TleHandle *>(tle)->rec);
    azh_constellation *c = nullptr;
    int32_t rc = build_from_records(recs, grav == 1 ? AZ_WGS72 : AZ_WGS84, 0, &c);
    if (rc != AZ_OK) return rc;
    const unsigned f = c->h_flags[0];
    int32_t e = AZ_OK;
    // same precedence as Sgp4.initElements (src/Sgp4.zig L111-123)
    if (AZ_FLAG_ERR(f) == 1)
        e = AZ_ERR_INVALID_ECCENTRICITY;
    else if (AZ_FLAG_ERR(f) == 6)
        e = AZ_ERR_SATELLITE_DECAYED;
    else if (f & AZ_FLAG_DEEP)
        e = AZ_ERR_DEEP_SPACE_NOT_SUPPORTED;
    if (e != AZ_OK) {
        destroy(c);
        return e;
    }
    Sgp4Handle *h = new (std::nothrow) Sgp4Handle{c};
    if (!h) {
        destroy(c);
        return AZ_ERR_ALLOC_FAILED;
    }
    *out = h;
    return AZ_OK;
}

void sgp4_free(void *h)
{
    if (!h) return;
    destroy(static_cast<Sgp4Handle *>(h)->c);
    delete static_cast<Sgp4Handle *>(h);
}

int32_t sgp4_propagate(void *h, double tsince, double pos[3], double vel[3])
{
    if (!h || !pos || !vel) return AZ_ERR_NULL_POINTER;
    return azh_propagate_one_host(static_cast<Sgp4Handle *>(h)->c, 0, &tsince, 1, pos, vel, nullptr);
}

int32_t sgp4_propagate_batch(void *h, const double *times, double *results, uint32_t count)
{
    if (!h || !times || !results) return AZ_ERR_NULL_POINTER;
    if (count == 0) return AZ_OK;
    return run_one_satellite(static_cast<Sgp4Handle *>(h)->c, 0, times, count, 1, results, nullptr, nullptr, nullptr);
}

// the device side of azh_selftest_coords: one tiny launch on device 0 through the very device functions of the kernels' epilogue
// (k_gmst's formula, az_to_ecef, az_ecef_to_geodetic)
namespace {
int32_t coords_call(int op, const double in[4], double *out, int n_out = 3)
{
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    static double *d_buf = nullptr;
    const double nan = std::nan("");
    for (int i = 0; i < n_out; ++i) out[i] = nan;
    if (!hip_ok(hipSetDevice(0), "hipSetDevice")) return AZ_ERR_HIP;
    if (!d_buf && !hip_ok(hipMalloc((void **)&d_buf, sizeof(double) * 12), "hipMalloc")) return AZ_ERR_HIP;
    HIP_TRY(hipMemset(d_buf + 4, 0, sizeof(double) * 8));
    HIP_TRY(hipMemcpy(d_buf, in, sizeof(double) * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_coords, dim3(1), dim3(64), 0, nullptr, op, d_buf, d_buf + 4);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_buf + 4, sizeof(double) * n_out, hipMemcpyDeviceToHost));
    return AZ_OK;
}
} // namespace

// The seven closed-form scalars of the reference's c_api (src/c_api/root.zig L60-81): pure host functions there, pure host
// functions here -- a link-time client of libastroz_c.so gets them without a device (rounds 2-4 evaluated them in a
// one-thread kernel behind a mutex: an H2D copy, a launch and a D2H copy for a square root, and NaN without a GPU).  They are
// not on the propagation path; the rule "no CPU fallback" is about that path (tests/test_host_cpu.py::test_no_cpu_fallback).
// The device evaluations stay as known-answer tests of the device code they share with the output frames of the kernels
// (az_to_ecef, az_ecef_to_geodetic): azh_selftest_coords.
//
// orbital_*: src/c_api/orbital_mechanics.zig over src/calculations.zig L83-125
int32_t orbital_hohmann(double mu, double r1, double r2, azh_hohmann_result *out)
{
    if (!out) return AZ_ERR_NULL_POINTER;
    if (r1 <= 0 || r2 <= 0 || std::fabs(r1 - r2) < 1000) return AZ_ERR_VALUE;
    const double sma = 0.5 * (r1 + r2), v1c = std::sqrt(mu / r1), v2c = std::sqrt(mu / r2);
    const double dv1 = v1c * std::sqrt(2.0 * r2 / (r1 + r2)) - v1c, dv2 = v2c - v2c * std::sqrt(2.0 * r1 / (r1 + r2));
    out->semi_major_axis = sma; out->delta_v1 = dv1; out->delta_v2 = dv2; out->total_delta_v = std::fabs(dv1) + std::fabs(dv2);
    out->transfer_time = AZ_PI * std::sqrt(sma * sma * sma / mu);
    out->transfer_time_days = out->transfer_time / 86400.0;
    return AZ_OK;
}
double orbital_velocity(double mu, double radius, double sma)
{
    if (radius <= 0 || sma < 0) return -1.0;
    return std::sqrt(sma != 0.0 ? mu * (2.0 / radius - 1.0 / sma) : mu / radius); // vis-viva; sma = 0: circular
}
double orbital_period(double mu, double sma)
{
    if (sma <= 0) return -1.0;
    return 2.0 * AZ_PI * std::sqrt(sma * sma * sma / mu);
}
double orbital_escape_velocity(double mu, double radius)
{
    if (radius <= 0) return -1.0;
    return std::sqrt(2.0 * mu / radius);
}

// coords_*: src/WorldCoordinateSystem.zig L98-154 (julianToGmst, eciToEcef, ecefToGeodeticDeg)
double coords_julian_to_gmst(double jd)
{
    const double d = jd - 2451545.0, tc = d / 36525.0;
    double gm = 280.46061837 + 360.98564736629 * d + 0.000387933 * tc * tc - tc * tc * tc / 38710000.0;
    gm = std::fmod(gm, 360.0);
    if (gm < 0) gm += 360.0;
    return gm * (AZ_PI / 180.0);
}

void coords_eci_to_ecef(const double eci[3], double gmst, double ecef[3])
{
    if (!eci || !ecef) return;
    const double sg = std::sin(gmst), cg = std::cos(gmst);
    const double x = eci[0] * cg + eci[1] * sg, y = eci[1] * cg - eci[0] * sg;
    ecef[0] = x; ecef[1] = y; ecef[2] = eci[2];
}

void coords_ecef_to_geodetic(const double ecef[3], double lla[3])
{
    if (!ecef || !lla) return;
    // the reference's fixed-point iteration, as it stands (WorldCoordinateSystem.zig L98-121): at most ten trips, exit once the
    // latitude moves by less than 1e-12 rad; degrees out (ecefToGeodeticDeg)
    const double f = 1.0 / 298.257223563, e2 = 2.0 * f - f * f, a = 6378.137;
    const double x = ecef[0], y = ecef[1], z = ecef[2];
    const double lon = std::atan2(y, x), rho = std::sqrt(x * x + y * y);
    double lat = std::atan2(z, rho * (1.0 - e2));
    for (int it = 0; it < 10; ++it) {
        const double prev = lat, sl = std::sin(lat);
        const double N = a / std::sqrt(1.0 - e2 * sl * sl);
        lat = std::atan2(z + e2 * N * sl, rho);
        if (std::fabs(lat - prev) < 1e-12) break;
    }
    const double sl = std::sin(lat), cl = std::cos(lat);
    const double N = a / std::sqrt(1.0 - e2 * sl * sl);
    lla[0] = lat * (180.0 / AZ_PI); lla[1] = lon * (180.0 / AZ_PI); lla[2] = rho / cl - N;
}

// known-answer hook: the same seven quantities evaluated ON THE DEVICE (k_coords: the kernels' own frame code).  op 0: GMST of
// in[0]; 1: ECI in[0..2] -> ECEF at GMST in[3]; 2: ECEF -> (lat deg, lon deg, alt km); 3: (velocity, period, escape velocity)
// of (mu, radius, sma); 4: Hohmann (sma, dv1, dv2, |dv1| + |dv2|, transfer time) of (mu, r1, r2).  AZ_ERR_HIP without a device.
int32_t azh_selftest_coords(int32_t op, const double in[4], double out[5])
{
    if (!in || !out) return AZ_ERR_NULL_POINTER;
    if (op < 0 || op > 4) return AZ_ERR_VALUE;
    return guarded([&]() -> int32_t { return coords_call(op, in, out, 5); });
}

} // extern "C"

// Development probe (not product code): VERDICT r05 item 1 -- time-major output by TWO passes through the 256-MiB
// Infinity Cache.  A producer with the sat-major row kernel's store pattern (one wave per (row, time segment), lane = time,
// 1,536 contiguous bytes per array and 64-step iteration, optional fp64 FMA work per iteration and per wave set-up) writes
// chunks of a RING of scratch slots; a pure-memory transposer (LDS tile of TS satellites x TT steps, 16-byte accesses both
// sides) reads a finished chunk and writes the time-major arrays.  Questions: does the transposer read its chunk out of the
// Infinity Cache (FETCH_SIZE ~ 0)?  is the scratch written back to HBM although the ring reuses its addresses (WRITE_SIZE
// above the 932 MB of output)?  what does the transposer reach alone, and what does the pipeline take end to end, against
// 0.28-0.30 ms for k_tiles_fast and 0.20 ms for the row kernel?
//   hipcc --offload-arch=gfx950 -O3 tools/mall_transpose_probe.hip -o tools/mall_transpose_probe.bin
//   tools/mall_transpose_probe.bin                (timing table)
//   tools/mall_transpose_probe.bin <case index>   (one case, three repetitions: for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// Producer: rows [r0, r0 + n_rows) of the catalog, steps [0, tc) of the chunk; scratch slot is [row_in_chunk][tc][3] doubles
// per array.  seg = steps per wave (multiple of 64).  work = dependent fp64 FMAs per lane and iteration on each of 6 values
// (6 * work VALU per iteration); setup = the same once per wave.
template <bool NT>
__global__ void __launch_bounds__(256) k_prod(double *sp, double *sv, unsigned n_rows, unsigned tc, unsigned t_end, unsigned seg, unsigned work,
                                              unsigned setup, double seed)
{
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const unsigned row = blockIdx.x * 4 + wave;
    if (row >= n_rows) return;
    const unsigned t_lo = blockIdx.y * seg, t_hi = min(t_lo + seg, t_end);
    double a0 = seed + lane, a1 = seed * 2 + row, a2 = seed * 3, b0 = 1.0 + seed, b1 = 2.0 + lane, b2 = 3.0 + row;
    const double m = 0.999999, c = 1e-9;
    for (unsigned i = 0; i < setup; ++i) { a0 = a0 * m + c; a1 = a1 * m + c; a2 = a2 * m + c; b0 = b0 * m + c; b1 = b1 * m + c; b2 = b2 * m + c; }
    const size_t rb = (size_t)row * tc * 3;
    for (unsigned t0 = t_lo; t0 < t_hi; t0 += 64) {
        for (unsigned i = 0; i < work; ++i) { a0 = a0 * m + c; a1 = a1 * m + c; a2 = a2 * m + c; b0 = b0 * m + c; b1 = b1 * m + c; b2 = b2 * m + c; }
        // 64 steps x 24 B = 1,536 B per array: lanes 0..63 write 16 B each (1 KB), lanes 0..31 the remaining 512 B -- the
        // shape of az_flush_stage's stores (values are whatever the lane holds: the pattern is what is measured)
        const unsigned n = min(64u, t_hi - t0);
        double *p = sp + rb + (size_t)t0 * 3, *v = sv + rb + (size_t)t0 * 3;
        const d2 x = {a0 + a1, a2}, y = {b0 + b1, b2};
        if (2 * lane + 1 < n * 3) {
            if (NT) { __builtin_nontemporal_store(x, reinterpret_cast<d2 *>(p) + lane); __builtin_nontemporal_store(y, reinterpret_cast<d2 *>(v) + lane); }
            else { reinterpret_cast<d2 *>(p)[lane] = x; reinterpret_cast<d2 *>(v)[lane] = y; }
        }
        if (lane < 32 && 2 * (lane + 64) + 1 < n * 3) {
            if (NT) { __builtin_nontemporal_store(x, reinterpret_cast<d2 *>(p) + 64 + lane); __builtin_nontemporal_store(y, reinterpret_cast<d2 *>(v) + 64 + lane); }
            else { reinterpret_cast<d2 *>(p)[64 + lane] = x; reinterpret_cast<d2 *>(v)[64 + lane] = y; }
        }
    }
}

// Transposer: one workgroup = TS satellites x TT steps of one array (blockIdx.z = array).  Chunk rows [r0, r0 + n_rows) ->
// catalog rows, chunk steps [0, tc) -> grid points [t_base, t_base + tc).  Output time-major (t * n_sats + s) * 3 doubles.
template <unsigned TS, unsigned TT, bool NT_OUT, bool NT_IN>
__global__ void __launch_bounds__(256) k_xpose(const double *sp, const double *sv, double *op, double *ov, unsigned n_rows, unsigned tc, unsigned r0,
                                               unsigned t_base, unsigned n_sats, unsigned n_times)
{
    constexpr unsigned PITCH = TS * 3 + 2;
    __shared__ double tile[TT * PITCH];
    const double *src = blockIdx.z ? sv : sp;
    double *dst = blockIdx.z ? ov : op;
    const unsigned s0 = blockIdx.x * TS, t0 = blockIdx.y * TT;
    if (s0 >= n_rows || t0 >= tc) return;
    const unsigned ns = min(TS, n_rows - s0), nt = min(TT, min(tc - t0, n_times - min(n_times, t_base + t0)));
    constexpr unsigned PR = TT * 3 / 2;                       // 16-byte pieces per source row of the tile
    for (unsigned q = threadIdx.x; q < TS * PR; q += 256) {
        const unsigned s = q / PR, p = q - s * PR;
        if (s >= ns || 2 * p + 1 >= nt * 3 + 1) continue;
        const d2 *g = reinterpret_cast<const d2 *>(src + ((size_t)(s0 + s) * tc + t0) * 3) + p;
        const d2 x = NT_IN ? __builtin_nontemporal_load(g) : *g;
        const unsigned j0 = 2 * p, j1 = 2 * p + 1;
        tile[(j0 / 3) * PITCH + s * 3 + j0 % 3] = x.x;
        tile[(j1 / 3) * PITCH + s * 3 + j1 % 3] = x.y;
    }
    __syncthreads();
    constexpr unsigned PO = TS * 3 / 2;                       // 16-byte pieces per output run
    for (unsigned q = threadIdx.x; q < TT * PO; q += 256) {
        const unsigned t = q / PO, p = q - t * PO;
        if (t >= nt || 2 * p + 1 >= ns * 3 + 1) continue;
        const d2 x = *reinterpret_cast<const d2 *>(&tile[t * PITCH + 2 * p]);
        d2 *g = reinterpret_cast<d2 *>(dst + ((size_t)(t_base + t0 + t) * n_sats + (r0 + s0)) * 3) + p;
        if (NT_OUT) __builtin_nontemporal_store(x, g); else *g = x;
    }
}

// Does the Infinity Cache serve a buffer a kernel has just WRITTEN?  k_fill writes n 16-byte pieces, k_read reads them
// (16 B per lane, 8 loads in flight per lane) and folds them into one value per workgroup.
__global__ void __launch_bounds__(256) k_fill(d2 *p, size_t n, double v)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = d2{v, v + 1.0};
}
__global__ void __launch_bounds__(256) k_read(const d2 *p, size_t n, double *sink)
{
    double acc = 0.0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n; i += 8 * stride) {
        d2 x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = p[i + k * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += x[k].x + x[k].y;
    }
    for (; i < n; i += stride) acc += p[i].x + p[i].y;
    if (acc == 12345.678) sink[blockIdx.x] = acc; // (never true: keeps the loads)
}

struct Case {
    const char *name;
    unsigned rows;      // rows per chunk (0 = whole catalog)
    unsigned tc;        // steps per chunk (0 = whole grid)
    unsigned seg;       // steps per producer wave
    unsigned slots;     // ring slots (0 = full-size scratch: every chunk its own memory)
    unsigned work, setup;
    bool nt_scratch, nt_out, nt_in;
    unsigned tt;        // transposer tile steps: 16 / 32 / 64
    unsigned streams;   // 1 = producer and transposer serial on one stream; 2 = transposer on its own stream beside the next producer
    unsigned mode;      // 0 = both, 1 = producer only, 2 = transposer only (source = whatever the scratch holds)
};

static void launch_xpose(const Case &c, hipStream_t st, const double *sp, const double *sv, double *op, double *ov, unsigned n_rows, unsigned tc,
                         unsigned r0, unsigned t_base, unsigned n_sats, unsigned n_times)
{
    dim3 g((n_rows + 63) / 64, (tc + c.tt - 1) / c.tt, 2);
#define XP(TT_, A, B) hipLaunchKernelGGL((k_xpose<64, TT_, A, B>), g, dim3(256), 0, st, sp, sv, op, ov, n_rows, tc, r0, t_base, n_sats, n_times)
#define XPT(TT_) do { if (c.nt_out && c.nt_in) XP(TT_, true, true); else if (c.nt_out) XP(TT_, true, false); else if (c.nt_in) XP(TT_, false, true); else XP(TT_, false, false); } while (0)
    if (c.tt == 16) XPT(16); else if (c.tt == 32) XPT(32); else XPT(64);
}

int main(int argc, char **argv)
{
    const unsigned n_sats = 13478, n_times = 1440;
    const unsigned W = 34, S = 31;   // 6 * 34 = 204 VALU per iteration, 6 * 31 = 186 per wave set-up: the row kernel's counts
    std::vector<Case> cases = {
        // name                                                              rows   tc   seg slots work setup nts   nto    nti   tt str mode
        {"P0 producer alone, monolithic 768+672 segments, no work",              0,    0, 768, 0, 0, 0, true,  false, false, 32, 1, 1},
        {"P1 producer alone, monolithic, work 204/iter + 186 set-up",            0,    0, 768, 0, W, S, true,  false, false, 32, 1, 1},
        {"X0 transposer alone, 932 MB scratch from HBM, tile 64x32",             0,    0, 768, 0, 0, 0, false, false, false, 32, 1, 2},
        {"X1 transposer alone, ..., tile 64x64",                                 0,    0, 768, 0, 0, 0, false, false, false, 64, 1, 2},
        {"X2 transposer alone, ..., tile 64x16",                                 0,    0, 768, 0, 0, 0, false, false, false, 16, 1, 2},
        {"X3 transposer alone, tile 64x32, nt output",                           0,    0, 768, 0, 0, 0, false, true,  false, 32, 1, 2},
        {"X4 transposer alone, tile 64x32, nt output + nt input",                0,    0, 768, 0, 0, 0, false, true,  true,  32, 1, 2},
        // serial chunk -> transposer on one stream, no work: what the cache does
        {"A0 serial, chunks all rows x 128 steps (83 MB), ring 2, plain",        0,  128, 128, 2, 0, 0, false, false, false, 32, 1, 0},
        {"A1 serial, same, nt output",                                           0,  128, 128, 2, 0, 0, false, true,  false, 32, 1, 0},
        {"A2 serial, same, nt output, nt scratch stores",                        0,  128, 128, 2, 0, 0, true,  true,  false, 32, 1, 0},
        {"A3 serial, chunks all rows x 64 steps (41 MB), ring 2, nt output",     0,   64,  64, 2, 0, 0, false, true,  false, 32, 1, 0},
        {"A4 serial, chunks 1,728 rows x 384 steps (32 MB), ring 2, nt output", 1728, 384, 384, 2, 0, 0, false, true,  false, 32, 1, 0},
        {"A5 serial, chunks 1,024 rows x 1,440 steps (71 MB), ring 2, nt out",  1024,   0, 768, 2, 0, 0, false, true,  false, 32, 1, 0},
        {"A6 serial, chunks all rows x 128 steps, FULL scratch (no ring), nt o",   0,  128, 128, 0, 0, 0, false, true,  false, 32, 1, 0},
        // the pipeline with the row kernel's arithmetic: transposer of chunk k beside producer of chunk k + 1
        {"B0 pipeline, all rows x 128 steps, ring 2, work, nt out",              0,  128, 128, 2, W, S, false, true,  false, 32, 2, 0},
        {"B1 pipeline, all rows x 384 steps (248 MB), ring 2, work, nt out",     0,  384, 384, 2, W, S, false, true,  false, 32, 2, 0},
        {"B2 pipeline, 1,728 rows x 384 steps, ring 4, work, nt out",         1728,  384, 384, 4, W, S, false, true,  false, 32, 2, 0},
        {"B3 pipeline, 1,024 rows x 1,440 steps, ring 3, work, nt out",       1024,    0, 768, 3, W, S, false, true,  false, 32, 2, 0},
        {"B4 pipeline, 3,456 rows x 384 steps (64 MB), ring 3, work, nt out", 3456,  384, 384, 3, W, S, false, true,  false, 32, 2, 0},
        {"B5 serial (one stream), all rows x 128 steps, ring 2, work, nt out",   0,  128, 128, 2, W, S, false, true,  false, 32, 1, 0},
        {"B6 pipeline, all rows x 128 steps, ring 2, work, plain out",           0,  128, 128, 2, W, S, false, false, false, 32, 2, 0},
        {"B7 pipeline, all rows x 128 steps, ring 2, work, nt out, tile 64x64",  0,  128, 128, 2, W, S, false, true,  false, 64, 2, 0},
        {"B8 pipeline, monolithic producer then transposer (0 chunks), work",    0,    0, 768, 0, W, S, false, true,  false, 32, 1, 0},
    };
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    const int reps = only >= 0 ? 3 : 5;
    const size_t full = (size_t)n_sats * n_times * 3;
    double *sp, *sv, *op, *ov;
    const size_t scratch = (size_t)n_sats * 1536 * 3; // (the un-ringed chunk cases round the grid up to whole chunks)
    CK(hipMalloc(&sp, scratch * 8 + 4096)); CK(hipMalloc(&sv, scratch * 8 + 4096));
    CK(hipMalloc(&op, full * 8 + 4096)); CK(hipMalloc(&ov, full * 8 + 4096));
    CK(hipMemset(sp, 0, full * 8)); CK(hipMemset(sv, 0, full * 8));
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<hipEvent_t> ev_prod(64), ev_x(64);
    for (auto &e : ev_prod) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : ev_x) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (only < 0 || only == 100) {
        // read-after-write and read-after-read by working-set size: the second read of a set that fits the 256-MiB cache should
        // beat HBM if reads allocate; the read of a freshly WRITTEN set beats HBM only if writes allocate too
        const size_t mbs[] = {16, 32, 64, 83, 128, 192, 384}; // (the scratch array holds 497 MB)
        for (size_t mb : mbs) {
            const size_t n = mb * 1000000 / 16;
            float t_w = 1e30f, t_rw = 1e30f, t_rr = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                float ms;
                // evict: stream 1.8 GB of other memory through the caches first
                hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, s0, reinterpret_cast<const d2 *>(op), full / 2, ov);
                hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, s0, reinterpret_cast<const d2 *>(ov), full / 2, op);
                CK(hipEventRecord(e0, s0));
                hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, s0, reinterpret_cast<d2 *>(sp), n, 1.0 + rep);
                CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < t_w) t_w = ms;
                CK(hipEventRecord(e0, s0));
                hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, s0, reinterpret_cast<const d2 *>(sp), n, ov);
                CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < t_rw) t_rw = ms;
                CK(hipEventRecord(e0, s0));
                hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, s0, reinterpret_cast<const d2 *>(sp), n, ov);
                CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < t_rr) t_rr = ms;
            }
            printf("[R] working set %4zu MB: write %7.1f us (%5.2f TB/s)   read after write %7.1f us (%5.2f TB/s)   read after read %7.1f us (%5.2f TB/s)\n",
                   mb, t_w * 1e3, mb / t_w / 1e3, t_rw * 1e3, mb / t_rw / 1e3, t_rr * 1e3, mb / t_rr / 1e3);
            fflush(stdout);
        }
    }
    for (size_t ci = 0; ci < cases.size(); ++ci) {
        if (only >= 0 && (int)ci != only) continue;
        const Case &c = cases[ci];
        const unsigned R = c.rows ? c.rows : n_sats, TC = c.tc ? c.tc : n_times;
        const unsigned n_rc = (n_sats + R - 1) / R, n_tc = (n_times + TC - 1) / TC, n_chunks = n_rc * n_tc;
        const size_t slot = (size_t)R * TC * 3;   // doubles per array and slot
        if (n_chunks > 64) { printf("%s: too many chunks\n", c.name); continue; }
        float best = 1e30f;
        for (int rep = 0; rep < reps; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, s0));
            for (unsigned k = 0; k < n_chunks; ++k) {
                // time-outer order: all row chunks of a time chunk, then the next time chunk
                const unsigned kt = k / n_rc, kr = k % n_rc;
                const unsigned r0 = kr * R, rows = R < n_sats - r0 ? R : n_sats - r0;
                const unsigned tb = kt * TC, tcs = TC < n_times - tb ? TC : n_times - tb;
                const unsigned sl = c.slots ? k % c.slots : k;
                double *cp = sp + (c.slots ? sl * slot : (size_t)k * slot), *cv = sv + (c.slots ? sl * slot : (size_t)k * slot);
                if (c.mode != 2) {
                    // a slot may be overwritten once its previous transposer is done
                    if (c.streams == 2 && c.slots && k >= c.slots) CK(hipStreamWaitEvent(s0, ev_x[k - c.slots], 0));
                    dim3 g((rows + 3) / 4, (tcs + c.seg - 1) / c.seg);
                    if (c.nt_scratch) hipLaunchKernelGGL(k_prod<true>, g, dim3(256), 0, s0, cp, cv, rows, TC, tcs, c.seg, c.work, c.setup, 1.0 + rep);
                    else hipLaunchKernelGGL(k_prod<false>, g, dim3(256), 0, s0, cp, cv, rows, TC, tcs, c.seg, c.work, c.setup, 1.0 + rep);
                }
                if (c.mode != 1) {
                    hipStream_t xs = c.streams == 2 ? s1 : s0;
                    if (c.streams == 2) { CK(hipEventRecord(ev_prod[k], s0)); CK(hipStreamWaitEvent(s1, ev_prod[k], 0)); }
                    launch_xpose(c, xs, cp, cv, op, ov, rows, TC, r0, tb, n_sats, n_times);
                    if (c.streams == 2) CK(hipEventRecord(ev_x[k], s1));
                }
            }
            if (c.streams == 2 && c.mode != 1) CK(hipStreamWaitEvent(s0, ev_x[n_chunks - 1], 0));
            CK(hipEventRecord(e1, s0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double bytes = 2.0 * n_sats * 24.0 * n_times;
        printf("[%2zu] %-72s chunks=%-3u %8.3f ms  %6.2f TB/s of output\n", ci, c.name, n_chunks, best, bytes / best / 1e9);
        fflush(stdout);
    }
    return 0;
}

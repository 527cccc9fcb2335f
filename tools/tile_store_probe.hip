// Development probe (not product code): the store side of a time-major tile kernel on its own.  A workgroup owns a tile of
// S consecutive satellites and walks through the time rows in blocks of R rows; per block it writes, for each of two arrays,
// R runs of S*24 bytes at a row pitch of n_sats*24 bytes -- exactly what k_tiles_fast / k_tiles2_fast flush -- with an
// optional busy-wait between blocks (the arithmetic).  Which (S, R, threads, lane mapping, XCD mapping) moves 932 MB fastest?
//   hipcc --offload-arch=gfx950 -O3 tools/tile_store_probe.hip -o tools/tile_store_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4), aligned(8)));

template <bool NT>
__global__ void k_tile_store(char *a0, char *a1, unsigned n_sats, unsigned n_times, unsigned S, unsigned R, unsigned seg_rows, unsigned xcd_map,
                             unsigned spin, size_t pitch)
{
    const unsigned nt = blockDim.x;
    unsigned tile = blockIdx.x;
    if (xcd_map) tile = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const unsigned s_first = tile * S;
    if (s_first >= n_sats) return;
    const unsigned ppr = S * 24 / 16;                 // 16-byte pieces per run
    const unsigned t_lo = blockIdx.y * seg_rows, t_hi = min(t_lo + seg_rows, n_times);
    const f4 val = {1.0f * threadIdx.x, 2.0f, 3.0f, (float)tile};
    for (unsigned base = t_lo; base < t_hi; base += R) {
        if (spin) {
            const unsigned long long t0 = __builtin_readcyclecounter();
            while (__builtin_readcyclecounter() - t0 < spin) { }
        }
        __syncthreads();
        for (unsigned q = threadIdx.x; q < R * ppr; q += nt) {
            const unsigned row = q / ppr, col = q - row * ppr;
            if (base + row >= t_hi) continue;
            const size_t off = (size_t)(base + row) * pitch + (size_t)s_first * 24 + col * 16;
            if (NT) { __builtin_nontemporal_store(val, reinterpret_cast<f4 *>(a0 + off)); __builtin_nontemporal_store(val, reinterpret_cast<f4 *>(a1 + off)); }
            else { *reinterpret_cast<f4 *>(a0 + off) = val; *reinterpret_cast<f4 *>(a1 + off) = val; }
        }
    }
}

struct Case { const char *name; unsigned S, R, threads, seg_rows, xcd_map, spin; bool nt; unsigned pad; };

int main()
{
    const unsigned n_sats = 13478, n_times = 1440;
    std::vector<Case> cases = {
        {"16 sats x 64 rows, 1024 thr, xcd map (k_tiles_fast)", 16, 64, 1024, 768, 1, 0, false, 0},
        {"16 sats x 64 rows,  512 thr, xcd map (k_tiles2_fast)", 16, 64, 512, 768, 1, 0, false, 0},
        {"16 sats x 64 rows,  512 thr, xcd map, spin 6000", 16, 64, 512, 768, 1, 6000, false, 0},
        {"16 sats x 64 rows, 1024 thr, xcd map, spin 6000", 16, 64, 1024, 768, 1, 6000, false, 0},
        {"16 sats x 64 rows,  512 thr, NO xcd map", 16, 64, 512, 768, 0, 0, false, 0},
        {"32 sats x 64 rows, 1024 thr, xcd map", 32, 64, 1024, 768, 1, 0, false, 0},
        {"32 sats x 64 rows, 1024 thr, xcd map, spin 12000", 32, 64, 1024, 768, 1, 12000, false, 0},
        {"64 sats x 64 rows, 1024 thr, xcd map", 64, 64, 1024, 768, 1, 0, false, 0},
        {"64 sats x 16 rows, 1024 thr, xcd map", 64, 16, 1024, 768, 1, 0, false, 0},
        {"16 sats x 64 rows,  512 thr, rows padded to 16 sats, nt", 16, 64, 512, 768, 1, 0, true, 16},
        {"16 sats x 64 rows,  512 thr, rows padded, nt, spin 6000", 16, 64, 512, 768, 1, 6000, true, 16},
        {"16 sats x 64 rows,  512 thr, rows padded, plain", 16, 64, 512, 768, 1, 0, false, 16},
        {"16 sats x 64 rows,  512 thr, unpadded, nt", 16, 64, 512, 768, 1, 0, true, 0},
        {"16 sats x 64 rows,  512 thr, 1 segment of 1440", 16, 64, 512, 1440, 1, 0, false, 0},
        {"16 sats x 64 rows,  512 thr, segments of 256", 16, 64, 512, 256, 1, 0, false, 0},
    };
    const size_t maxpitch = (size_t)(n_sats + 64) * 24;
    char *a0, *a1;
    hipMalloc(&a0, maxpitch * n_times + 4096);
    hipMalloc(&a1, maxpitch * n_times + 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto &c : cases) {
        const unsigned stride = c.pad ? (n_sats + c.pad - 1) / c.pad * c.pad : n_sats;
        const size_t pitch = (size_t)stride * 24;
        const unsigned tiles = (n_sats + c.S - 1) / c.S;
        dim3 grid((tiles + 7) / 8 * 8, (n_times + c.seg_rows - 1) / c.seg_rows);
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0, 0);
            if (c.nt) hipLaunchKernelGGL(k_tile_store<true>, grid, dim3(c.threads), 0, 0, a0, a1, n_sats, n_times, c.S, c.R, c.seg_rows, c.xcd_map, c.spin, pitch);
            else hipLaunchKernelGGL(k_tile_store<false>, grid, dim3(c.threads), 0, 0, a0, a1, n_sats, n_times, c.S, c.R, c.seg_rows, c.xcd_map, c.spin, pitch);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double bytes = 2.0 * n_sats * 24.0 * n_times;
        printf("%-62s %8.3f ms  %6.2f TB/s\n", c.name, best, bytes / best / 1e9);
    }
    return 0;
}

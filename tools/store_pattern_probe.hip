// Development probe (not product code): how fast can wave64s write one output array (466 MB) when a wave's consecutive
// bursts are `pitch` bytes apart?  Satellite-major rows are written sequentially; a time-major array of 13,478 satellites
// puts a wave's consecutive time steps 323,472 bytes apart.
//   hipcc --offload-arch=gfx950 -O3 tools/store_pattern_probe.hip -o /tmp/store_probe && /tmp/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

// wave w -> group g = w % G, tile tl = w / G; burst k of the wave starts at g * goff + (tl * iters + k) * pitch.
// A burst is `burst` bytes: full 1,024-byte stores (16 B per lane) plus, for 1,536, one half-wave store.
template <bool NT>
__global__ void k_store(char *base, unsigned G, unsigned n_waves, size_t goff, size_t pitch, unsigned iters, unsigned burst)
{
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave >= n_waves) return;
    const unsigned g = wave % G, tl = wave / G;
    const f4 val = {1.0f * lane, 2.0f, 3.0f, (float)wave};
    char *p = base + (size_t)g * goff + (size_t)tl * iters * pitch;
    const unsigned pieces = burst / 16; // 16-byte pieces per burst: lanes 0..pieces-1 (and a second store beyond 64)
    for (unsigned k = 0; k < iters; ++k, p += pitch) {
        f4 *q = reinterpret_cast<f4 *>(p) + lane;
        if (lane < pieces) {
            if (NT) __builtin_nontemporal_store(val, q);
            else *q = val;
        }
        if (lane + 64 < pieces) {
            if (NT) __builtin_nontemporal_store(val, q + 64);
            else q[64] = val;
        }
    }
}

struct Case {
    const char *name;
    unsigned G, tiles, iters, burst, waves_per_block;
    size_t goff, pitch;
    bool nt;
};

int main()
{
    const size_t row = (size_t)13478 * 24; // 323,472 B: one time row of one array
    std::vector<Case> cases = {
        // satellite-major: wave = (satellite, half row): 13478 x 2 waves, 11 sequential bursts of 1,536 B (k_rows_fast)
        {"sat-major: sequential 1.5-KB bursts, nt", 13478 * 2, 1, 11, 1536, 1, 17280, 1536, true},
        {"sat-major: sequential 1.5-KB bursts, plain", 13478 * 2, 1, 11, 1536, 1, 17280, 1536, false},
        // time-major as k_propagate writes it: 211 groups of 64 satellites x 23 tiles of 64 time steps
        {"time-major: pitch = row (323,472 B), plain", 211, 23, 64, 1536, 4, 1536, row, false},
        {"time-major: pitch = row, nt", 211, 23, 64, 1536, 4, 1536, row, true},
        {"time-major: pitch = row, 1-KB bursts only", 211, 23, 64, 1024, 4, 1536, row, false},
        {"time-major: rows padded to 128-B pitch", 211, 23, 64, 1536, 4, 1536, (row + 127) / 128 * 128, false},
        {"time-major: rows padded to 4-KB pitch", 211, 23, 64, 1536, 4, 1536, (row + 4095) / 4096 * 4096, false},
        {"time-major: every row in its own 2-MB page", 211, 23, 64, 1536, 4, 1536, (size_t)2 << 20, false},
        // fewer, longer tiles (fewer rows in flight at once) and more, shorter ones
        {"time-major: 6 tiles of 240 steps", 211, 6, 240, 1536, 4, 1536, row, false},
        {"time-major: 90 tiles of 16 steps", 211, 90, 16, 1536, 4, 1536, row, false},
        // what a 16-satellite x 64-time transposed tile would write: 384-B runs, one per time row
        {"time-major: 842 groups of 16 sats (384-B runs), 23 tiles", 842, 23, 64, 384, 4, 384, row, false},
        // tile flushes of other heights: 32 satellites (768-B runs), 8 (192-B), 12 (288-B); 16 with rows padded to whole lines
        {"time-major: 421 groups of 32 sats (768-B runs), 23 tiles", 421, 23, 64, 768, 4, 768, row, false},
        {"time-major: 421 groups of 32 sats (768-B runs), nt", 421, 23, 64, 768, 4, 768, row, true},
        {"time-major: 1685 groups of 8 sats (192-B runs)", 1685, 23, 64, 192, 4, 192, row, false},
        {"time-major: 1123 groups of 12 sats (288-B runs)", 1123, 23, 64, 288, 4, 288, row, false},
        {"time-major: 16 sats, rows padded to 128 B, plain", 843, 23, 64, 384, 4, 384, (row + 127) / 128 * 128, false},
        {"time-major: 16 sats, rows padded to 128 B, nt", 843, 23, 64, 384, 4, 384, (row + 127) / 128 * 128, true},
        {"time-major: 16 sats (384-B runs), nt", 842, 23, 64, 384, 4, 384, row, true},
        {"time-major: 16 sats (384-B runs), 16 waves per workgroup", 842, 23, 64, 384, 16, 384, row, false},
        {"time-major: 32 sats, rows padded to 128 B, nt", 421, 23, 64, 768, 4, 768, (row + 127) / 128 * 128, true},
        // a 1024-lane workgroup of 16 lane = satellite waves: same bursts, 24 KB of a row per workgroup
        {"time-major: pitch = row, 16 waves per workgroup", 211, 23, 64, 1536, 16, 1536, row, false},
    };
    size_t need = 0;
    for (auto &c : cases) {
        const size_t end = (size_t)(c.G - 1) * c.goff + ((size_t)c.tiles * c.iters) * c.pitch + 4096;
        if (end > need) need = end;
    }
    char *buf;
    if (hipMalloc(&buf, need) != hipSuccess) { printf("hipMalloc(%zu) failed\n", need); return 1; }
    hipMemset(buf, 0, need);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (auto &c : cases) {
        const unsigned waves = c.G * c.tiles;
        const unsigned blocks = (waves + c.waves_per_block - 1) / c.waves_per_block;
        float best = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0, 0);
            if (c.nt) hipLaunchKernelGGL(k_store<true>, dim3(blocks), dim3(64 * c.waves_per_block), 0, 0, buf, c.G, waves, c.goff, c.pitch, c.iters, c.burst);
            else hipLaunchKernelGGL(k_store<false>, dim3(blocks), dim3(64 * c.waves_per_block), 0, 0, buf, c.G, waves, c.goff, c.pitch, c.iters, c.burst);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double bytes = (double)waves * c.iters * c.burst;
        printf("%-62s %8.3f ms  %6.2f TB/s\n", c.name, best, bytes / best / 1e9);
    }
    return 0;
}

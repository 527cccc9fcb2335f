// host_step.cpp -- see host_step.h.  The device headers are plain C++ (devmath.h): AZ_HOST_EMUL swaps the four gfx950
// intrinsics (v_rcp_f64 / v_rsq_f64 seeds, the wave vote, v_rndne) for scalar stand-ins and a wave becomes one lane.
// Compiled with -mfma -mavx2 (every fma() of the step is one instruction; x86-64-v3: any EPYC).
#define AZ_HOST_EMUL 1
#define AZ_DEVICE static inline
#define AZ_COLD_STRIDE 1
#include "host_step.h"
#include <cstring>
#include "propagate_device.h"

namespace azhost {

int num_fields() { return AZ_NUM_FIELDS; }

bool cpu_ok()
{
#if defined(__x86_64__)
    return __builtin_cpu_supports("fma") && __builtin_cpu_supports("avx2");
#else
    return true;
#endif
}

void propagate_points(const double *el, size_t n_pad, size_t sat, unsigned flags, const AzGrav &g, const double *tsince, size_t n, int interleaved,
                      double *out6, double *pos, double *vel, uint8_t *err)
{
    const int init_rc = AZ_FLAG_ERR(flags);
    const bool deep = (flags & AZ_FLAG_DEEP) != 0;
    // the constants of the satellite, loaded once per call (the kernel loads them once per lane)
    Sgp4Lane e4{};
    ColdRegs cold4{};
    Sdp4Lane ed{};
    double cold_store[D_NUM];
    const ColdLds coldd{cold_store};
    if (init_rc == 0) {
        if (deep) az_load_sdp4(el, n_pad, sat, flags, ed, coldd);
        else az_load_sgp4(el, n_pad, sat, flags, e4, cold4);
    }
    const RotK rk = az_rotk();
    for (size_t i = 0; i < n; ++i) {
        double r[3], v[3];
        int rc = init_rc;
        if (rc == 0) {
            if (deep) {
                Sdp4Carry c;
                c.atime = 0.0;
                c.xli = ed(H_xlamo);
                c.xni = ed(H_no_unkozai);
                rc = az_sdp4_step<true>(ed, coldd, g, rk, tsince[i], c, r, v);
            } else {
                Sgp4Carry c;
                c.t_prev = 0.0;
                az_sgp4_step<true>(e4, cold4, el, n_pad, sat, g, rk, tsince[i], true, c, r, v);
            }
        }
        if (rc != 0) r[0] = r[1] = r[2] = v[0] = v[1] = v[2] = 0.0;
        if (interleaved) {
            double *o = out6 + 6 * i;
            o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = v[0]; o[4] = v[1]; o[5] = v[2];
        } else {
            memcpy(pos + 3 * i, r, sizeof r);
            if (vel) memcpy(vel + 3 * i, v, sizeof v);
        }
        if (err) err[i] = (uint8_t)rc;
    }
}

} // namespace azhost

// host_step.h -- the product's OWN per-point step (propagate_device.h: az_sgp4_step / az_sdp4_step, the code k_one_satellite
// runs per lane) compiled for the host, for calls of a handful of points.
//
// Why: the reference's scalar call (`Satrec.sgp4(jd, fr)`, bindings/python/src/satrec.zig L169-201; `sgp4_propagate`,
// src/c_api/root.zig L13-81) is 0.4 us; a kernel launch + a synchronize is 20 us whatever the kernel does.  A call of at most
// azh_set_host_points() points (default 128 near-earth / 64 deep-space: where the two routes cross) therefore evaluates the SAME step source on the calling thread, from the element
// column the DEVICE initialised (k_init; mirrored to the host when the handle is made or on first use).  It is not a fallback:
// without a device no handle exists (creation returns AZ_ERR_HIP), nothing above the point limit ever takes it, and it is this
// library's own step source, not the test tier's CPU checker -- tests/test_round6_cpu.py and tests/test_gpu_round6.py hold
// these statements.
#pragma once
#include <cstddef>
#include <cstdint>

struct AzGrav;
namespace azhost {
// el / n_pad / sat: a host copy of the device element table (elem[field * n_pad + sat]; one column: n_pad = 1, sat = 0);
// flags: the satellite's status word.  interleaved = 1: out6 is n x 6 (x, y, z, vx, vy, vz); otherwise pos (n x 3), vel (n x 3, optional),
// err (n, optional).  Same per-point semantics as k_one_satellite (fresh seeds every point, zeros on error).
void propagate_points(const double *el, size_t n_pad, size_t sat, unsigned flags, const AzGrav &g, const double *tsince, size_t n, int interleaved,
                      double *out6, double *pos, double *vel, uint8_t *err);
int num_fields();
bool cpu_ok(); // the host cores have the FMA units this translation unit is compiled for
} // namespace azhost

"""cost of making a Satrec's device handle (parse + device init + mirror), of the first scalar call and of freeing it: one handle
per record (each record initialised on its own: what a C client's sgp4_init / sgp4_free pays too) against the records made
together sharing one handle"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
from astroz_amd import synth, _native
from astroz_amd.api import Satrec, WGS72
pairs = synth.synth_catalog(2000, 200, seed=3)
warm = [Satrec.twoline2rv(a, b, WGS72) for a, b in pairs[:10]]
for s in warm: s.sgp4(s.jdsatepoch, s.jdsatepochF)
n = len(pairs)
# (1) one handle per record: every record is made AND used before the next one exists
t0 = time.perf_counter()
hs = [_native.DeviceConstellation.from_tle_lines([p], WGS72, 0) for p in pairs]
t1 = time.perf_counter()
for h in hs: h.close()
t2 = time.perf_counter()
print("one handle per record (azh_constellation_from_tle_lines of ONE record; sgp4_init makes the same): create %.1f us, free %.1f us each" % (
    (t1 - t0) / n * 1e6, (t2 - t1) / n * 1e6))
# (2) records made together
t0 = time.perf_counter()
sats = [Satrec.twoline2rv(a, b, WGS72) for a, b in pairs]
t1 = time.perf_counter()
for s in sats: s.sgp4(s.jdsatepoch, s.jdsatepochF + 0.1)
t2 = time.perf_counter()
for s in sats: s.sgp4(s.jdsatepoch, s.jdsatepochF + 0.2)
t3 = time.perf_counter()
print("records made together: twoline2rv (text only) %.2f us each; first sgp4 (shared handle + device init + mirror) %.1f us each; second sgp4 %.2f us each" % (
    (t1 - t0) / n * 1e6, (t2 - t1) / n * 1e6, (t3 - t2) / n * 1e6))
t4 = time.perf_counter()
del sats, s
gc.collect()
print("destruction: %.2f us each" % ((time.perf_counter() - t4) / n * 1e6))

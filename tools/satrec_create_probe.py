"""cost of making a Satrec's one-satellite handle (parse + device init + mirror) and of the first scalar call"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
from astroz_amd import synth
from astroz_amd.api import Satrec, WGS72
pairs = synth.synth_catalog(2000, 200, seed=3)
sats = [Satrec.twoline2rv(a, b, WGS72) for a, b in pairs[:10]]
for s in sats: s.sgp4(s.jdsatepoch, s.jdsatepochF)
t0 = time.perf_counter()
sats = [Satrec.twoline2rv(a, b, WGS72) for a, b in pairs]
t1 = time.perf_counter()
for s in sats: s.sgp4(s.jdsatepoch, s.jdsatepochF + 0.1)
t2 = time.perf_counter()
for s in sats: s.sgp4(s.jdsatepoch, s.jdsatepochF + 0.2)
t3 = time.perf_counter()
n = len(sats)
print("twoline2rv (text only): %.2f us each; first sgp4 (handle creation + device init + mirror): %.1f us each; second sgp4: %.2f us each" % (
    (t1 - t0) / n * 1e6, (t2 - t1) / n * 1e6, (t3 - t2) / n * 1e6))
t4 = time.perf_counter()
del sats
print("destruction: %.1f us each" % ((time.perf_counter() - t4) / n * 1e6))

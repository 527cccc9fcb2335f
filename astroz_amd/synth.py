"""Deterministic synthetic TLE catalogs.

The reference ships no TLE catalog (its 13,478-satellite figure came from a live CelesTrak
download, benchmarks/sgp4_compat_test.py L79-97) and there is no network here, so benchmarks
and catalog-scale tests use a seeded generator that renders 69-column TLE text with valid
checksums; the text is then parsed by the library's own parser like any other input.
"""
import os

import numpy as np

START_JD = 2460800.5      # 2025-05-05 00:00 UTC
_JD_2025_JAN1 = 2460676.5
_XKE72 = 0.0743669161331734132
_RE72 = 6378.135


def _checksum(line68):
    return sum(int(c) if c.isdigit() else (1 if c == "-" else 0) for c in line68) % 10


def _fmt_exp(val):
    """TLE 'implied decimal point' exponent field, 8 chars: ' 12345-3' = 0.12345e-3."""
    if val == 0.0:
        return " 00000+0"
    sign = "-" if val < 0 else " "
    av = abs(val)
    exp = int(np.floor(np.log10(av))) + 1
    mant = int(round(av * 10.0 ** (5 - exp)))
    if mant >= 100000:
        mant //= 10
        exp += 1
    return "%s%05d%+d" % (sign, mant, exp)


def format_tle(satnum, epoch_jd, incl_deg, raan_deg, ecc, argp_deg, ma_deg, mm_revday, bstar,
               ndot=0.0, revnum=1):
    """Render one element set as (line1, line2), 69 columns each, checksums valid."""
    year = 2025
    doy = epoch_jd - _JD_2025_JAN1 + 1.0
    while doy < 1.0:  # epochs before 2025-01-01
        year -= 1
        ylen = 366.0 if (year % 4 == 0 and (year % 100 != 0 or year % 400 == 0)) else 365.0
        doy += ylen
    ndot_s = ("-" if ndot < 0 else " ") + ("%.8f" % abs(ndot))[1:]
    sn = "%05d" % (satnum % 100000)
    l1 = "1 %sU 25001A   %02d%012.8f %s  00000+0 %s 0  999" % (
        sn, year % 100, doy, ndot_s, _fmt_exp(bstar))
    l1 = l1[:68].ljust(68)
    l1 += str(_checksum(l1))
    ecc_s = ("%.7f" % ecc)[2:]
    l2 = "2 %s %8.4f %8.4f %s %8.4f %8.4f %11.8f%5d" % (
        sn, incl_deg, raan_deg % 360.0, ecc_s, argp_deg % 360.0, ma_deg % 360.0, mm_revday,
        revnum % 100000)
    l2 = l2[:68].ljust(68)
    l2 += str(_checksum(l2))
    return l1, l2


def _n_from_alt(alt_km, ecc):
    """rev/day for a perigee altitude (km) and eccentricity."""
    a = (1.0 + alt_km / _RE72) / (1.0 - ecc)
    n_radmin = _XKE72 / a ** 1.5
    return n_radmin * 1440.0 / (2.0 * np.pi), a


def _a_from_n(n_revday):
    return (_XKE72 / (n_revday * 2.0 * np.pi / 1440.0)) ** (2.0 / 3.0)


def near_earth_elements(n, seed=20260926, start_jd=START_JD):
    """Config-2 shell mix.  Returns dict of arrays (TLE units)."""
    rng = np.random.default_rng(seed)
    kind = rng.choice(6, size=n, p=[0.62, 0.05, 0.20, 0.08, 0.04, 0.01])
    if os.environ.get("AZ_SYNTH_NO_ECC"):  # tuning experiment only: no eccentric members
        kind[kind == 4] = 0
    incl = np.empty(n)
    alt = np.empty(n)
    ecc = np.exp(rng.uniform(np.log(1e-5), np.log(3e-3), n))
    mm = np.empty(n)
    shells = np.array([53.05, 53.2, 43.0, 70.0, 97.6])
    k = kind == 0
    incl[k] = shells[rng.integers(0, 5, k.sum())] + rng.uniform(-0.05, 0.05, k.sum())
    alt[k] = rng.uniform(340, 570, k.sum())
    k = kind == 1
    incl[k] = 87.9 + rng.uniform(-0.05, 0.05, k.sum())
    alt[k] = 1200.0 + rng.uniform(-10, 10, k.sum())
    k = kind == 2
    incl[k] = rng.uniform(96.5, 99.0, k.sum())
    alt[k] = rng.uniform(450, 850, k.sum())
    k = kind == 3
    incl[k] = rng.uniform(0.0, 100.0, k.sum())
    alt[k] = rng.uniform(300, 2000, k.sum())
    k = kind == 5  # simplified-drag branch: perigee below 220 km
    incl[k] = rng.uniform(28.0, 98.0, k.sum())
    alt[k] = rng.uniform(170, 215, k.sum())
    ecc[k] = np.exp(rng.uniform(np.log(1e-4), np.log(5e-3), k.sum()))
    circ = kind != 4
    mm[circ], _ = _n_from_alt(alt[circ], ecc[circ])
    # eccentric: Vanguard-like
    k = kind == 4
    mm[k] = rng.uniform(6.5, 13.0, k.sum())
    a = _a_from_n(mm[k])
    emax = np.minimum(0.25, 1.0 - (1.0 + 230.0 / _RE72) / a)
    ecc[k] = rng.uniform(0.01, np.maximum(emax, 0.0101))
    incl[k] = rng.uniform(20.0, 70.0, k.sum())
    bstar = np.exp(rng.uniform(np.log(1e-6), np.log(1e-3), n))
    bstar[rng.random(n) < 0.05] *= -1.0
    return dict(
        epoch_jd=start_jd - rng.uniform(0.0, 5.0, n), incl=incl, raan=rng.uniform(0, 360, n), ecc=ecc,
        argp=rng.uniform(0, 360, n), ma=rng.uniform(0, 360, n), mm=mm, bstar=bstar)


def deep_space_elements(n, seed=20260928, start_jd=START_JD):
    """Config-3 deep-space mix, proportions 600 GEO / 250 GNSS / 60 Molniya / 400 GTO / 212 other
    per 1,522."""
    rng = np.random.default_rng(seed)
    kind = rng.choice(5, size=n, p=np.array([600, 250, 60, 400, 212]) / 1522.0)
    incl = np.empty(n)
    ecc = np.empty(n)
    mm = np.empty(n)
    k = kind == 0  # GEO: synchronous resonance, includes the Lyddane (i < 0.2 rad) branch
    mm[k] = 1.0027 + rng.uniform(-0.002, 0.002, k.sum())
    incl[k] = rng.uniform(0.01, 15.0, k.sum())
    ecc[k] = rng.uniform(1e-5, 1e-3, k.sum())
    k = kind == 1  # GNSS
    mm[k] = rng.choice([2.0057, 2.13, 1.70, 1.86], k.sum()) + rng.uniform(-0.001, 0.001, k.sum())
    incl[k] = rng.uniform(55.0, 65.0, k.sum())
    ecc[k] = rng.uniform(1e-4, 0.02, k.sum())
    k = kind == 2  # Molniya: half-day resonance
    mm[k] = 2.006 + rng.uniform(-0.003, 0.003, k.sum())
    incl[k] = 63.4 + rng.uniform(-0.5, 0.5, k.sum())
    ecc[k] = rng.uniform(0.6, 0.74, k.sum())
    k = kind == 3  # GTO / HEO debris
    mm[k] = rng.uniform(2.0, 6.0, k.sum())
    a = _a_from_n(mm[k])
    emax = np.minimum(0.73, 1.0 - (1.0 + 250.0 / _RE72) / a)
    ecc[k] = rng.uniform(0.3, np.maximum(emax, 0.3001))
    incl[k] = rng.uniform(3.0, 30.0, k.sum())
    k = kind == 4
    mm[k] = rng.uniform(3.0, 6.3, k.sum())
    ecc[k] = rng.uniform(1e-4, 0.1, k.sum())
    incl[k] = rng.uniform(0.0, 110.0, k.sum())
    bstar = np.exp(rng.uniform(np.log(1e-6), np.log(1e-4), n))
    bstar[rng.random(n) < 0.3] = 0.0
    return dict(
        epoch_jd=start_jd - rng.uniform(0.0, 5.0, n), incl=incl, raan=rng.uniform(0, 360, n), ecc=ecc,
        argp=rng.uniform(0, 360, n), ma=rng.uniform(0, 360, n), mm=mm, bstar=bstar)


REFERENCE_DEEP_TLES = [  # the three deep-space TLEs of the reference's tests (src/Sdp4.zig L1423-1468)
    ("1 20413U 90005A   24186.00000000  .00000012  00000+0  10000-3 0  9992",
     "2 20413  55.4408  61.4858 0112981 129.5765 231.5553  2.00561730104446"),
    ("1 28626U 05004A   24186.00000000 -.00000098  00000+0  00000+0 0  9998",
     "2 28626   0.0163 279.8379 0003069  20.3251 343.1766  1.00270142 70992"),
    ("1 09880U 77021B   24186.00000000  .00000023  00000+0  00000+0 0  9999",
     "2 09880  63.4300  75.8891 7318036 269.8735  16.7549  2.00611684 54321"),
]


def elements_to_pairs(el, first_satnum=1):
    n = len(el["mm"])
    return [format_tle(first_satnum + i, el["epoch_jd"][i], el["incl"][i], el["raan"][i], el["ecc"][i],
                       el["argp"][i], el["ma"][i], el["mm"][i], el["bstar"][i]) for i in range(n)]


def synth_catalog(n_near=13478, n_deep=0, seed=20260926, start_jd=START_JD, interleave=True):
    """(line1, line2) pairs: config 2 (n_deep=0) or config 3 (n_near=13478, n_deep=1522).

    With deep-space members the two populations are interleaved pseudo-randomly (a real catalog
    is not sorted by regime) and the reference's three deep-space test TLEs are included."""
    pairs = elements_to_pairs(near_earth_elements(n_near, seed, start_jd), 1)
    if n_deep > 0:
        n_synth = max(n_deep - len(REFERENCE_DEEP_TLES), 0)
        deep = elements_to_pairs(deep_space_elements(n_synth, seed + 2, start_jd), n_near + 1)
        deep = (REFERENCE_DEEP_TLES + deep)[:n_deep]
        if interleave:
            rng = np.random.default_rng(seed + 7)
            pos = np.sort(rng.choice(n_near + n_deep, size=n_deep, replace=False))
            out, di, ni = [], 0, 0
            posset = set(pos.tolist())
            for i in range(n_near + n_deep):
                if i in posset:
                    out.append(deep[di]); di += 1
                else:
                    out.append(pairs[ni]); ni += 1
            pairs = out
        else:
            pairs = pairs + deep
    return pairs


def pairs_to_text(pairs):
    return "\n".join(a + "\n" + b for a, b in pairs) + "\n"

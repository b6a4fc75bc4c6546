#!/usr/bin/env python3
"""Independent pin for the UTM stage (pyproj/PROJ are absent: "parity unpinned vs pyproj").

Evaluates the transverse-Mercator projection from its DEFINITION with 50-digit mpmath -- no series:
the Gauss-Krueger map is the analytic continuation  N + iE = k0 * M(phi(psi + i*lambda))  of the
meridian arc M as a function of the isometric latitude psi.  phi(.) is inverted by complex Newton,
M by numerical quadrature of a(1-e^2)(1-e^2 sin^2 t)^(-3/2) along the complex segment [0, phi_c].
Writes tests/golden/utm_mpmath.npz (inputs + float64-rounded exact outputs).
"""
import os

import mpmath as mp
import numpy as np

mp.mp.dps = 50
A = mp.mpf(6378137)
F = 1 / mp.mpf("298.257223563")
E2 = F * (2 - F)
E = mp.sqrt(E2)
K0 = mp.mpf("0.9996")


def psi(phi):
    return mp.asinh(mp.tan(phi)) - E * mp.atanh(E * mp.sin(phi))


def tm_exact(lat_deg, lon_deg, zone, south):
    phi = mp.radians(mp.mpf(float(lat_deg)))
    lam = mp.radians(mp.mpf(float(lon_deg)) - (6 * zone - 183))
    w = psi(phi) + 1j * lam
    pc = mp.mpc(phi, lam * mp.cos(phi))          # start
    for _ in range(60):
        f = psi(pc) - w
        d = (1 - E2) / ((1 - E2 * mp.sin(pc) ** 2) * mp.cos(pc))
        step = f / d
        pc -= step
        if abs(step) < mp.mpf(10) ** -45:
            break
    M = A * (1 - E2) * mp.quad(lambda t: (1 - E2 * mp.sin(t) ** 2) ** mp.mpf(-1.5), [0, pc])
    z = K0 * M
    return 500000 + z.imag, z.real + (10000000 if south else 0)


def main():
    rng = np.random.default_rng(5)
    here = os.path.dirname(os.path.abspath(__file__))
    pts = []
    # bundled data (both column readings) + KITTI-like cloud + global spread
    pts += [(49.033603440345, 8.3950031909457, 32, 0), (49.03360622, 8.39500533, 32, 0), (8.39500533, 49.03360622, 39, 0)]
    for _ in range(12):
        pts.append((49.03 + rng.uniform(-0.02, 0.02), 8.395 + rng.uniform(-0.02, 0.02), 32, 0))
    for _ in range(25):
        lat, lon = rng.uniform(-80, 84), rng.uniform(-180, 180)
        zone = int((lon + 180) // 6 + 1)
        pts.append((lat, lon, min(zone, 60), int(lat < 0)))
    # zone edges / off-zone use (mean-lon zone pick puts points up to a few degrees outside their own zone)
    pts += [(45.0, 9.0, 32, 0), (45.0, 12.0, 32, 0), (45.0, 5.9, 32, 0), (-33.9, 18.4, 34, 1), (0.0, 9.0, 32, 0),
            (1e-9, 3.0, 31, 0), (70.0, 15.0, 32, 0), (60.0, 9.0 + 7.5, 32, 0)]
    lat, lon, zone, south, e, n = [], [], [], [], [], []
    for la, lo, z, s in pts:
        ee, nn = tm_exact(la, lo, z, s)
        lat.append(la); lon.append(lo); zone.append(z); south.append(s); e.append(float(ee)); n.append(float(nn))
    np.savez_compressed(os.path.join(here, "utm_mpmath.npz"), lat=np.array(lat), lon=np.array(lon),
                        zone=np.array(zone, np.int32), south=np.array(south, np.int32), E=np.array(e), N=np.array(n),
                        meta=np.array(f"mpmath {mp.__version__} dps=50; definition-level TM (analytic continuation of meridian arc)"))
    print("wrote utm_mpmath.npz", len(pts), "points")


def main_zones():
    """Round 5: every one of the 60 zones, both hemispheres, latitudes from the equator to 84 N / 80 S, the central meridian, the zone
    edges and points half a degree outside them (the reference picks ONE zone per log from the mean longitude, ref :131-133, so a log
    that straddles an edge is projected partly off-zone) -> tests/golden/utm_zones_mpmath.npz.  Pins the forward series to the
    definition and the INVERSE (direct Gaussian-latitude series) to the same points read backwards: a float64 (E, N) pair is within
    1e-9 m of the exact image, i.e. within 1e-14 degree of the exact pre-image."""
    here = os.path.dirname(os.path.abspath(__file__))
    lats = [1e-7, 0.5, 12.0, 30.0, 49.0336, 60.0, 72.0, 84.0, -0.5, -23.0, -45.0, -66.0, -80.0]
    dlon = [0.0, 3.0, -3.0, 1.7, -2.2, 3.5, -3.5]
    lat, lon, zone, south, e, n = [], [], [], [], [], []
    k = 0
    for z in range(1, 61):
        lon0 = 6 * z - 183
        for j in range(4):                                    # four points per zone, walking through both lists
            la, dl = lats[k % len(lats)], dlon[(k // 2) % len(dlon)]
            k += 1
            lo = lon0 + dl
            s_ = int(la < 0)
            ee, nn = tm_exact(la, lo, z, s_)
            lat.append(la); lon.append(lo); zone.append(z); south.append(s_); e.append(float(ee)); n.append(float(nn))
    np.savez_compressed(os.path.join(here, "utm_zones_mpmath.npz"), lat=np.array(lat), lon=np.array(lon),
                        zone=np.array(zone, np.int32), south=np.array(south, np.int32), E=np.array(e), N=np.array(n),
                        meta=np.array(f"mpmath {mp.__version__} dps=50; definition-level TM; 60 zones x 4 points"))
    print("wrote utm_zones_mpmath.npz", len(lat), "points")


if __name__ == "__main__":
    import sys
    if "--zones" in sys.argv:
        main_zones()
    else:
        main()

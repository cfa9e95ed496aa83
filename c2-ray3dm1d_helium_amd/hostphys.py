"""Host-side set-up helpers a Python host needs in place of the reference's Fortran set-up
modules (they are NOT part of the hot path; the Fortran host keeps using its own):

  reccoef(T)          ini_rec_colion_factors (cgsconstants.f90:140-266) -- the twelve module-global
                      coefficients that mat_ini fixes once for an isothermal run
                      (mat_ini_test.F90:168)
  test_density(z)     the uniform density of the reference's test problem (mat_ini_test.F90:238-254)
  test_grid(N, z)     cell size / volume of the 10 Mpc/h test box (grid.F90, test.F90:47)

REAL(4) literals of the reference are reproduced with numpy.float32.
"""
from __future__ import annotations

import math

import numpy as np


def _f(x):
    return float(np.float32(x))


eth0 = _f(13.598)
ethe = (_f(24.587), _f(54.416))
ev2k = float(np.float32(1.0) / np.float32(8.617e-05))
temph0 = eth0 * ev2k
temphe = (ethe[0] * ev2k, ethe[1] * ev2k)
colh0 = _f(1.3e-8) * _f(0.83) * _f(1.0) / (eth0 * eth0)
colhe = (_f(1.3e-8) * _f(0.63) * _f(2.0) / (ethe[0] * ethe[0]), _f(1.3e-8) * _f(1.30) * _f(1.0) / (ethe[1] * ethe[1]))


def reccoef(T: float) -> np.ndarray:
    """arech0, brech0, areche0, breche0, oreche0, areche1, breche1, treche1, colli_HI, colli_HeI,
    colli_HeII, v at temperature T (cgsconstants.f90:140-266)."""
    p = math.pow
    lam = 2.0 * (temph0 / T)
    arech0 = _f(1.269e-13) * p(lam, 1.503) / p(1.0 + p(lam / _f(0.522), _f(0.470)), _f(1.923))
    brech0 = _f(2.753e-14) * p(lam, 1.500) / p(1.0 + p(lam / _f(2.740), _f(0.407)), _f(2.242))
    if T < 9.0e3:
        areche0 = 1.269e-13 * p(lam, 1.503) / p(1.0 + p(lam / _f(0.522), _f(0.470)), _f(1.923))
        breche0 = 2.753e-14 * p(lam, 1.500) / p(1.0 + p(lam / _f(2.740), _f(0.407)), _f(2.242))
    else:
        lam = 2.0 * (temphe[0] / T)
        diel = 1.9e-3 * p(T, -1.5) * math.exp(-4.7e5 / T) * (1.0 + 0.3 * math.exp(-9.4e4 / T))
        areche0 = 3.000e-14 * p(lam, 0.654) + diel
        # flang -O2 evaluates x**0.750 as sqrt(x)*sqrt(sqrt(x))
        breche0 = 1.260e-14 * (math.sqrt(lam) * math.sqrt(math.sqrt(lam))) + diel
    oreche0 = areche0 - breche0
    lam = 2.0 * (temphe[1] / T)
    breche1 = 5.5060e-14 * p(lam, 1.5) / p(1.0 + p(lam / 2.740, 0.407), 2.242)
    areche1 = _f(2.538e-13) * p(lam, 1.503) / p(1.0 + p(lam / 0.522, 0.470), 1.923)
    treche1 = 3.4e-13 * p(T / 1.0e4, -0.6)
    v = 0.285 * p(T / 1.0e4, 0.119)
    sq = math.sqrt(T)
    colli = (colh0 * sq * math.exp(-temph0 / T), colhe[0] * sq * math.exp(-temphe[0] / T),
             colhe[1] * sq * math.exp(-temphe[1] / T))
    return np.array([arech0, brech0, areche0, breche0, oreche0, areche1, breche1, treche1, *colli, v])


# cosmoparms.f90:28-42 (WMAP3+), abundances.f90, cgsconstants.f90
h = _f(0.7)
Omega0 = _f(0.27)
Omega_B = _f(0.044)
H0 = 2.26830837024227824e-18      # as evaluated by the reference build
rho_crit_0 = 9.20346643016612840e-30
mu = (1.0 - _f(0.074)) + 4.0 * _f(0.074)
m_p = 1.672661e-24
Mpc = 3.08600011031262003e24
YEAR = 3.15576e7


def test_density(zred: float) -> float:
    """avg_dens = rho_crit_0*Omega_B/(mu*m_p)*(1+z)**3 (mat_ini_test.F90:241)."""
    return rho_crit_0 * Omega_B / (mu * m_p) * (1.0 + zred) ** 3


def test_grid(mesh: int, zred: float, boxsize_mpc_h: float = 10.0):
    """Proper cell size and volume of the test box at redshift z (grid.F90:37-149, cosmology.f90:159-202)."""
    dr = boxsize_mpc_h / h * Mpc / mesh / (1.0 + zred)
    return (dr, dr, dr), dr * dr * dr

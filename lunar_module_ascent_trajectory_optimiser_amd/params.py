"""Problem parameters (mirrors /root/reference/Launch_Optimiser.py:38-75,107-109) and the
synthetic sweeps of BASELINE.json configs 3 and 4."""
from __future__ import annotations

import dataclasses
import math

import numpy as np

PARAM_FIELDS = ("G", "M", "R0", "Ft", "M0", "mdot", "fuel_mass", "mass_scalar", "ang_acc_max",
                "r_peri", "r_apo", "T_scale", "angle_ub", "tf_lb", "tf_ub", "dcost")
G0 = 9.80665


@dataclasses.dataclass
class AscentParams:
    """One ascent NLP, SI units.  Defaults = Apollo 11 as in the reference script."""
    G: float = 6.674e-11            # Launch_Optimiser.py:50
    M: float = 7.346e22             # :51
    R0: float = 1738100.0           # :52
    Ft: float = 15346.0             # :61
    M0: float = 4821.0              # :62
    mdot: float = 5.053             # :63
    fuel_mass: float = 2376.0       # :64
    mass_scalar: float = 2376.0     # :108
    ang_acc_max: float = 5e-4       # :66
    r_peri: float = 17703.0         # :70
    r_apo: float = 88615.0          # :71
    T_scale: float = 470.0          # :38
    angle_ub: float = math.pi / 3   # :94
    tf_lb: float = 0.0              # :39
    tf_ub: float = 1.0              # :39
    dcost: float = 0.0              # :99

    def as_row(self) -> np.ndarray:
        return np.array([getattr(self, f) for f in PARAM_FIELDS], dtype=np.float64)

    @property
    def periapsis_v(self) -> float:      # Launch_Optimiser.py:75
        return math.sqrt(self.G * self.M / (self.R0 + 0.5 * (self.r_peri + self.r_apo)))


def pack(params) -> np.ndarray:
    """AscentParams | sequence of them | (batch,16) array -> contiguous (batch,16) float64."""
    if isinstance(params, AscentParams):
        return params.as_row()[None, :].copy()
    if isinstance(params, np.ndarray):
        a = np.ascontiguousarray(np.atleast_2d(params), dtype=np.float64)
        if a.shape[1] != len(PARAM_FIELDS):
            raise ValueError(f"parameter array must have {len(PARAM_FIELDS)} columns")
        return a
    return np.ascontiguousarray(np.stack([p.as_row() for p in params]), dtype=np.float64)


def sweep_isp_drymass(n_isp=64, n_dry=64, isp=(300.0, 320.0), dry=(2345.0, 2545.0), base=None,
                      tf_ub=1.2) -> np.ndarray:
    """BASELINE.json config 3: Isp x dry-mass grid (4096 NLPs by default), SURVEY.md 8d.
    mdot = Ft/(Isp*g0); M0 = dry + fuel_mass; tf_ub widened because high-Isp burns outlast 470 s."""
    base = base or AscentParams()
    row = base.as_row()
    out = np.tile(row, (n_isp * n_dry, 1))
    I, D = np.meshgrid(np.linspace(*isp, n_isp), np.linspace(*dry, n_dry), indexing="ij")
    out[:, PARAM_FIELDS.index("mdot")] = base.Ft / (I.ravel() * G0)
    out[:, PARAM_FIELDS.index("M0")] = D.ravel() + base.fuel_mass
    out[:, PARAM_FIELDS.index("tf_ub")] = tf_ub
    return out


def sweep_config4(n_isp=64, n_dry=64, n_apo=8, n_acc=8, apo_km=(70.0, 105.0), acc=(3e-4, 1e-3),
                  base=None) -> np.ndarray:
    """BASELINE.json config 4: config-3 grid x target apoapsis x angular-acceleration cap
    (262 144 NLPs by default).  Problem index = ((i_isp*n_dry + i_dry)*n_apo + i_apo)*n_acc + i_acc."""
    g3 = sweep_isp_drymass(n_isp, n_dry, base=base)
    apo = np.linspace(*apo_km, n_apo) * 1e3
    cap = np.geomspace(*acc, n_acc)
    out = np.repeat(g3, n_apo * n_acc, axis=0)
    A, C = np.meshgrid(apo, cap, indexing="ij")
    out[:, PARAM_FIELDS.index("r_apo")] = np.tile(A.ravel(), g3.shape[0])
    out[:, PARAM_FIELDS.index("ang_acc_max")] = np.tile(C.ravel(), g3.shape[0])
    return out

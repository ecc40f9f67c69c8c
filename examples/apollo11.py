#!/usr/bin/env python3
"""Apollo-11 lunar-module ascent to the 87 x 17 km orbit through the GEKKO-style surface.

This is this repository's own counterpart of the reference script: the same problem, declared with
the same modelling calls, solved by libascent on an MI355X instead of GEKKO/APMonitor/IPOPT.  It prints
the quantities the reference prints (/root/reference/Launch_Optimiser.py:178-194) and writes the same
three figures (:208-242).  Usage:  python examples/apollo11.py [--no-plots] [--outdir DIR]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "compat"))
from gekko import GEKKO  # noqa: E402  (the compatibility package; resolves to gekko_shim.GEKKO)

# ---- physical data (SI) -----------------------------------------------------------------------------
MOON = dict(G=6.674e-11, M=7.346e22, R0=1738100.0)
VEHICLE = dict(Ft=15346.0, M0=4821.0, M_dot=5.053, fuel=2376.0, ang_acc_max=5e-4)
ORBIT = dict(periapsis=17703.0, apoapsis=88615.0)
BURN_LIMIT = 470.0          # s, fuel-limited maximum burn: time scale of the free final time
NT = 200


def build(nt=NT, solver=None):
    m = GEKKO(solver=solver) if solver is not None else GEKKO()
    m.time = np.linspace(0, 1, nt)
    m.options.NODES, m.options.SOLVER, m.options.IMODE = 2, 3, 6
    m.options.MAX_ITER, m.options.MV_TYPE = 20000, 0
    m.options.OTOL = m.options.RTOL = 1e-3

    tf = m.FV(value=0, lb=0, ub=1)
    tf.STATUS = 1
    G, M, R0 = (m.Const(MOON[k], name=k) for k in ("G", "M", "R0"))
    Ft, M0 = m.Const(VEHICLE["Ft"], name="Ft"), m.Const(VEHICLE["M0"], name="M0")
    m.Const(VEHICLE["M_dot"], name="M_dot")
    mflow = VEHICLE["M_dot"] / VEHICLE["fuel"]
    S = m.Const(ORBIT["periapsis"], name="distance Scale")       # lengths are scaled by the insertion altitude
    m.Const(ORBIT["periapsis"], name="Rfmin")
    mS = m.Const(VEHICLE["fuel"], name="mass Scale")
    aS = m.Const(VEHICLE["ang_acc_max"] / 3)
    v_ins = np.sqrt(MOON["G"] * MOON["M"] / (MOON["R0"] + 0.5 * (ORBIT["periapsis"] + ORBIT["apoapsis"])))

    mass = m.Var(value=0, lb=0, ub=1, name="mass")
    y, ydot, ydd = m.Var(value=0, name="y"), m.Var(name="ydot"), m.Var(name="ydoubledot")
    x, xdot, xdd = m.Var(value=0, name="x"), m.Var(name="xdot"), m.Var(name="xdoubledot")
    angle, angledot = m.Var(value=0, lb=0, ub=np.pi / 3, name="angle"), m.Var(name="angledot")
    u = m.MV(name="angledoubledot", lb=-1, ub=1)
    u.STATUS, u.DCOST = 1, 1e-5

    scale = tf * BURN_LIMIT                      # d/dtau = tf * T * d/dt
    for var, rate in ((y, ydot), (ydot, ydd), (x, xdot), (xdot, xdd), (angle, angledot)):
        m.Equation(var.dt() == scale * rate)
    m.Equation(angledot.dt() == scale * u * aS)
    m.Equation(mass.dt() == mflow * BURN_LIMIT * tf)

    X, Y = x * S, y * S + R0                     # metres, Moon-centred; launch site on the +Y axis
    r = (X ** 2 + Y ** 2) ** (1 / 2)
    thrust = Ft / ((M0 - mS * mass) * r)         # thrust acceleration / r
    grav = G * M / (X ** 2 + Y ** 2) ** (3 / 2)
    m.Equation(ydd == (thrust * (Y * m.cos(3 * angle) + X * m.sin(3 * angle)) - Y * grav) / S)
    m.Equation(xdd == (thrust * (X * m.cos(3 * angle) - Y * m.sin(3 * angle)) - X * grav) / S)

    for v in (y, x, ydot, xdot, angle, mass):
        m.fix(v, pos=0, val=0)

    only_last = np.zeros(nt); only_last[-1] = 1                     # terminal constraints act on the last node
    slack_elsewhere = np.full(nt, S + R0 + 1.0); slack_elsewhere[-1] = 0
    p_rad, p_vel = m.Param(value=slack_elsewhere), m.Param(value=only_last)
    m.Equation(((y + R0 / S) ** 2 + x ** 2) ** (1 / 2) + p_rad >= (R0 + S) / S)
    m.Equation(xdot ** 2 + ydot ** 2 >= (v_ins / S) ** 2 * p_vel)
    m.Equation((Y * (ydot * S) + X * (xdot * S)) * p_vel == 0)
    m.Minimize(tf)
    return m, dict(tf=tf, x=x, y=y, xdot=xdot, ydot=ydot, xdd=xdd, ydd=ydd, angle=angle, mass=mass), v_ins


def report(m, v, v_ins):
    """The reference's prints, in its order and wording."""
    S, T = ORBIT["periapsis"], BURN_LIMIT
    tfv = v["tf"].value[0]
    print("Optimal Solution (final time): " + str(tfv * T))
    print(v_ins)
    for label, var in (("final y", "y"), ("final x", "x"), ("final ydot", "ydot"), ("final xdot", "xdot"),
                       ("final ydoubledot", "ydd"), ("final xdoubledot", "xdd")):
        print(label, v[var].value[-1] * S)
    print("final time", tfv * T)
    return tfv * T


def plots(m, v, outdir):
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    S, R0 = ORBIT["periapsis"], MOON["R0"]
    xs = -np.asarray(v["x"].value) * S                    # downrange positive, as the reference plots it
    ys = np.asarray(v["y"].value) * S + R0
    theta = 3 * np.asarray(v["angle"].value) * 180 / np.pi
    t = m.time * v["tf"].value[0] * BURN_LIMIT
    fig, ax = plt.subplots()
    ax.add_patch(plt.Circle((0, 0), R0))
    ax.plot(xs, ys, color=(0.9, 0.4, 0))
    ax.set(ylim=(R0 - 30000, R0 + 20000), xlim=(-5000, 300000), title="Position", xlabel="x", ylabel="y", aspect="equal")
    ax.grid()
    fig.savefig(os.path.join(outdir, "takeoff_contextualized.png"), dpi=300)
    fig, ax = plt.subplots()
    ax.plot(t, theta)
    ax.set(title="Angle", xlabel="time", ylabel="Angle / degrees")
    ax.grid()
    fig.savefig(os.path.join(outdir, "Angle_vs_Time.png"), dpi=300)
    fig, ax = plt.subplots()
    ax.plot(xs, ys)
    ax.set(title="Position", xlabel="x", ylabel="y", ylim=(R0 - 10000, R0 + 20000), aspect="equal")
    ax.grid()
    fig.savefig(os.path.join(outdir, "takeoff_trajectory.png"), dpi=300)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-plots", action="store_true")
    ap.add_argument("--outdir", default=".")
    ap.add_argument("--no-dcost", action="store_true", help="ignore the MV's DCOST = 1e-5 (Launch_Optimiser.py:99; applied as an l1 move penalty by default)")
    ap.add_argument("--scheme", type=int, default=0, help="0 backward Euler (the reference's NODES=2), 1 trapezoid, 2 Hermite-Simpson")
    a = ap.parse_args()
    model, variables, v_ins = build()
    if a.no_dcost:
        model.options.ASCENT_DCOST = 0
    if a.scheme:
        model.options.ASCENT_SCHEME = a.scheme
    model.solve(disp=True)
    report(model, variables, v_ins)
    if not a.no_plots:
        plots(model, variables, a.outdir)

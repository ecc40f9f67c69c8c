import sys, numpy as np
sys.path.insert(0,'/root/repo')
import lunar_module_ascent_trajectory_optimiser_amd as A
ref = {(60,0):0.0, (200,0):439.635822, (200,2):440.841977}
for nt,scheme in ((60,0),(200,0),(200,2)):
    r1=A.solve_batch(A.AscentParams(), nt, tol=1e-10, scheme=scheme, terminal="ellipse", max_iter=500)
    r2=A.solve_batch(A.AscentParams(), nt, tol=1e-10, scheme=scheme, terminal="ellipse_free", max_iter=500)
    o=r2.orbit()
    print(nt,scheme,"status",r1.status[0],r2.status[0],"iters",r1.iters[0],r2.iters[0],"tf1 %.6f tf2 %.6f s"%(r1.final_time()[0], r2.final_time()[0]), "orbit", o["periapsis_alt"][0], o["apoapsis_alt"][0], "fpa", o["flight_path_angle"][0], "kernel ms", r2.kernel_ms)
S=A.sweep_isp_drymass(4,4)
r=A.solve_batch(S, 200, tol=1e-9, scheme=2, terminal="ellipse_free", max_iter=500)
o=r.orbit(); print("sweep status", r.status, "iters", r.iters, "peri", np.abs(o["periapsis_alt"]-17703).max(), "apo", np.abs(o["apoapsis_alt"]-88615).max())
c=r.coast(coast_nodes=64); print("coast apo", np.abs(c["apoapsis_alt"]-88615).max(), "tf", c["tf"][:3]*470)

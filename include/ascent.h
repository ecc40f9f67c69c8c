/* libascent -- C ABI of the MI355X-native batched lunar-ascent NLP solver.
 *
 * Drop-in boundary.  The reference has no FFI of its own: its hot path is
 * the single call  m.solve(disp=True)  at /root/reference/Launch_Optimiser.py:177, which ships the
 * model declared at Launch_Optimiser.py:19-176 to GEKKO -> APMonitor -> IPOPT.  These entry points
 * are what a GEKKO-compatible front end binds instead of that call (see INTEGRATION.md for the
 * ctypes stub); lunar_module_ascent_trajectory_optimiser_amd/gekko_shim.py is such a front end.
 *
 * Conventions
 *   - every entry point returns 0 on success or a negative ascent_status code (usage / HIP error,
 *     text via ascent_strerror); per-problem solver outcomes go to status_out, never the return code
 *   - plain pointers and sizes only; the caller owns every buffer; the library allocates only its
 *     private per-device workspace
 *   - all arrays are double precision, structure-of-arrays with the PROBLEM index fastest:
 *       element (row r, problem p) of an array with `batch` problems lives at  a[r*batch + p]
 *   - scaled units exactly as the reference's GEKKO variables (Launch_Optimiser.py:83-109):
 *       lengths / Scalar (= r_peri), mass = burnt fraction of fuel_mass, angle = physical/3,
 *       u = angular acceleration / ang_acc_max, tf = final time / T_scale
 */
#ifndef ASCENT_H
#define ASCENT_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One NLP's physical parameters, SI units.  Mirrors Launch_Optimiser.py:38-75,107-109. */
typedef struct ascent_params {
  double G;            /* :50  gravitational constant                                   */
  double M;            /* :51  mass of the Moon, kg                                     */
  double R0;           /* :52  lunar radius, m                                          */
  double Ft;           /* :61  thrust, N                                                */
  double M0;           /* :62  wet mass, kg                                             */
  double mdot;         /* :63,65 propellant mass flow, kg/s (mflow = mdot/fuel_mass)    */
  double fuel_mass;    /* :64  kg                                                       */
  double mass_scalar;  /* :108 mass scale in (M0 - mass_scalar*mass); = fuel_mass in the
                               current script, 2576 in the v1 script (PDF p26)          */
  double ang_acc_max;  /* :66  angular-acceleration cap, rad/s^2                         */
  double r_peri;       /* :70  target periapsis altitude, m (= Rfmin = Scalar, :73,107) */
  double r_apo;        /* :71  target apoapsis altitude, m                              */
  double T_scale;      /* :38  final_time = 470 s (time scale; tf in [tf_lb, tf_ub])    */
  double angle_ub;     /* :94  upper bound of angle (= physical/3), pi/3                */
  double tf_lb;        /* :39                                                            */
  double tf_ub;        /* :39                                                            */
  double dcost;        /* :99  MV movement penalty (applied with ascent_opts.move_penalty = 1) */
} ascent_params;

typedef struct ascent_opts {
  int32_t n_nodes;     /* :20  nt, number of grid points (tau_k = k/(nt-1), :21), 3 .. 65536 */
  int32_t scheme;      /* 0 = NODES=2 two-point collocation = backward Euler (:25);
                          1 = trapezoid, control held over the step (not a reference scheme);
                          2 = Hermite-Simpson (compressed form, control held over the step; the method
                              source the reference's report cites, PDF p3/p25) -- a persistent kernel of its
                              own (h_solve); with move_penalty = 1 the dense-block solver path             */
  int32_t max_iter;    /* :28  interior-point iteration cap                              */
  int32_t warm_start;  /* 0 = built-in cold-start guess, 1 = primal part of `guess`,
                          2 = full primal-dual `guess` (multipliers kept)                */
  double tol;          /* KKT error tolerance (the reference's OTOL/RTOL, :31-32)        */
  double mu_init;      /* initial barrier parameter (<=0: 0.1 cold, 1e-4 warm)           */
  int32_t formulation; /* 0 = current script: u = angledoubledot is the MV (:96-100);
                          1 = v1 script (PDF p26-28): the angle itself is the MV -- carried in
                              the same arrays: angle = (angle_ub/2)(u+1), angledot = 0, and the
                              `angledoubledot` field holds that normalised control u            */
  int32_t coarse_nodes; /* nested iteration for cold starts (warm_start == 0): the NLP is first solved on a coarse
                           grid, that primal-dual solution is prolonged to the n_nodes grid and warm-starts it.
                           0 = automatic (grids of >= 40 nodes; coarse grid = max(14, (3 n_nodes + 5)/10) nodes,
                           recursively: 201 -> 60 -> 17 -- a grid one to three intervals beyond a multiple of 16 gives them up; coarse levels are solved to max(tol, 1e-3); a level
                           warm-started from the cold-started coarsest grid begins at mu = 1e-6, one warm-started
                           from a warm-started grid at mu = max(1e-9, tol/100); with move_penalty = 1: 1e-5 and
                           max(1e-8, 10 tol)), -1 = off (single grid),
                           > 0 = that many coarse nodes (two levels).
                           iters_out counts the iterations of all levels.                              */
  int32_t terminal;     /* 0 = the reference's terminal speed (:72-78: circular speed of the MEAN radius, imposed at
                               r_peri with r.v = 0, :158-173);
                           1 = the (r_peri, r_apo) ellipse proper (README.md:7): same three constraints with the vis-viva
                               speed at the periapsis of that ellipse, so that the burnout orbit is the target ellipse and
                               ascent_coast_batch's coast arc ends at its apoapsis;
                           2 = burnout ANYWHERE on that ellipse -- the burn--coast problem of BASELINE config 5 with the coast
                               arc eliminated exactly (two-body motion): two conditions, angular momentum >= and specific energy
                               <= those of the ellipse (an orbit nested in the target annulus; both active at the optimum), no
                               r.v = 0; ascent_coast_batch continues from whatever true anomaly the burn ends at.
                               Persistent kernels (every scheme) and dense-block path, formulation 0           */
  int32_t solver_path;  /* 0 = automatic (the persistent kernels: p_solve for schemes 0/1, h_solve for scheme 2; a handful of NLPs on a long
                               grid and scheme 2 with the move penalty: the dense-block path; ascent_default_path tells);
                           ASCENT_PATH_DENSE = the dense-block path for any scheme (formulation 0 only)       */
  int32_t move_penalty; /* 0 = ascent_params.dcost is ignored (a sweep's parameter sets may carry none);
                           1 = :99 angledoubledot.DCOST applied -- the model the reference declares: the objective is
                               tf + dcost * sum_k |u_k - u_{k-1}| (u_{-1} = 0, the MV's initial value; an l1 term with a slack pair
                               per step, as APMonitor documents DCOST).  Schemes 0 / 1: inside the persistent kernel (the control
                               becomes the eighth state of a stage, the pair reduces to one pivot); scheme 2: dense-block path.
                               With formulation 1 (the v1 script's angle.DCOST, PDF p26; scheme 0): the penalty is on the angle,
                               i.e. weight dcost * angle_ub / 2 on the normalised control, which starts from -1 (angle 0).
                               Needs dcost > 0 for every problem                                                     */
  int32_t reserved;     /* 0 */
} ascent_opts;

enum ascent_status {           /* function return codes */
  ASCENT_OK = 0,
  ASCENT_E_ARG = -1,           /* null pointer / bad size / unsupported option           */
  ASCENT_E_HIP = -2,           /* a HIP runtime call failed (see ascent_strerror)        */
  ASCENT_E_NODEVICE = -3,      /* no such device                                         */
  ASCENT_E_NOMEM = -4,         /* workspace allocation failed                            */
  ASCENT_E_NOTERM = -5         /* the host-steered pipeline exceeded its round budget (a solver
                                  condition that the per-problem statuses could not express)   */
};

enum ascent_problem_status {   /* values written to status_out[] */
  ASCENT_CONVERGED = 0,
  ASCENT_MAX_ITER = 1,
  ASCENT_LINESEARCH_FAILED = 2,
  ASCENT_REGULARISATION_FAILED = 3   /* numerical breakdown: inertia could not be corrected */
};

/* Row counts of the SoA arrays, K = n_nodes-1 (node 0 is fixed by the initial conditions,
 * Launch_Optimiser.py:145-151):
 *   iterate / step / guess "blob":  21*K + 10 rows
 *       rows [0,7K)     z_k  : node k=1..K, fields x y xdot ydot angle angledot mass  (row 7(k-1)+f)
 *       rows [7K,8K)    u_k  : angledoubledot
 *       rows [8K,15K)   lambda_k : multipliers of the 7 collocation defects of step k
 *       rows [15K,21K)  bound multipliers zL_angle zU_angle zL_mass zU_mass zL_u zU_u per node
 *       rows 21K..21K+9 tf, zL_tf, zU_tf, s1, s2, z_s1, z_s2, nu3, nu1, nu2
 *                       (s_i: slacks of the two terminal inequalities :161,:169; nu: multipliers of
 *                        the terminal r.v = 0 (:173) and of the two slack equations)
 *   traj_out: 10*n_nodes rows, row f*n_nodes + k, fields in the order of the reference's .value
 *       lists: x y xdot ydot xdoubledot ydoubledot angle angledot angledoubledot mass  (:187-202)
 */
#define ASCENT_BLOB_ROWS(n_nodes) (21 * ((n_nodes) - 1) + 10)
#define ASCENT_TRAJ_FIELDS 10

int ascent_version(void);
int ascent_device_count(void);
const char *ascent_strerror(int code);

/* Solve `batch` independent ascent NLPs (replaces m.solve, Launch_Optimiser.py:177).
 * p: AoS [batch] parameter structs (host or device per ptr_is_device, like every other pointer).
 * guess_or_null: blob [21K+10][batch] when o->warm_start != 0.
 * traj_out [10*n_nodes][batch], tf_out/status_out/iters_out [batch]; sol_blob_out_or_null
 * [21K+10][batch] receives the full primal-dual solution (usable as a warm start).
 * stream: hipStream_t or NULL.  With host pointers the call returns after the results are in the
 * caller's buffers.  With ptr_is_device != 0 and a stream, the persistent kernels (schemes 0 and 1, both
 * formulations, with or without the move penalty; scheme 2 without it; every terminal mode: the default at every batch size,
 * ascent_default_path) are only enqueued -- a handful of launches per grid level, no host involvement (with the NULL stream the
 * call waits for the solve); the split pipeline (ASCENT_PIPELINE=split) and the dense-block path (scheme 2 with the move
 * penalty, a few NLPs on long grids) synchronise the stream once per burst of interior-point rounds, because the host steers the
 * rounds, and return with the last kernels enqueued.
 * Concurrency: host-side, calls on one device are serialised by a mutex.  Device-side, the library keeps a workspace per
 * caller stream (up to three non-default streams per device; the default stream, the parity surfaces and any further
 * stream share workspace 0): solves enqueued on different streams with device pointers overlap on the device -- the
 * wavefronts of one fill the SIMDs the stragglers of the other leave idle (bench.py: pipelined_two_streams) -- while
 * a call whose predecessor in the SAME workspace is still executing makes its stream wait for that predecessor's last
 * kernel first (hipStreamWaitEvent), so two solves never share a workspace in flight.  ascent_last_kernel_ms reports
 * the most recently enqueued solve of the device. */
int ascent_solve_batch(const ascent_params *p, int64_t batch, const ascent_opts *o,
                       const double *guess_or_null, double *traj_out, double *tf_out,
                       int32_t *status_out, int32_t *iters_out, double *sol_blob_out_or_null,
                       int device_id, void *hip_stream_or_null, int ptr_is_device);

/* Per-node pieces of the path, exposed for parity testing (Launch_Optimiser.py:114-136):
 * iterate: blob [21K+10][batch] (z, u, lambda, tf are read).
 * defects [7K][batch]: z_k - z_{k-1} - h*T*tf*f(z_k,u_k), row 7(k-1)+f.
 * jac_blocks [8K][batch]: d(xdoubledot)/d(x,y,angle,mass), d(ydoubledot)/d(x,y,angle,mass).
 * hess_blocks [10K][batch]: upper triangle (xx xy xa xm yy ya ym aa am mm) of the Hessian of
 *   -h*T*tf*(lambda_xdot*xdoubledot + lambda_ydot*ydoubledot), the node's Lagrangian block.
 * Host pointers. */
int ascent_eval_nodes(const ascent_params *p, int64_t batch, const ascent_opts *o,
                      const double *iterate, double *defects, double *jac_blocks,
                      double *hess_blocks, int device_id);

/* One Newton step of the barrier problem: factorises and solves the bordered block-tridiagonal
 * KKT system at `iterate` with barrier parameter mu[p] and primal regularisation delta_w[p].
 * step [21K+10][batch]; inertia_out[p] = 0 if the KKT matrix had the correct inertia, 1 if not
 * (step then undefined).  Host pointers. */
int ascent_kkt_step(const ascent_params *p, int64_t batch, const ascent_opts *o,
                    const double *iterate, const double *mu, const double *delta_w, double *step,
                    int32_t *inertia_out, int device_id);

/* The same two parity surfaces through a chosen solver path.  ascent_eval_nodes / ascent_kkt_step above take the path
 * ascent_solve_batch would take for that batch (batch-size rule, ASCENT_PIPELINE / ASCENT_FACTOR environment
 * overrides); these run one round of exactly the kernels of the named path at the given iterate:
 *   ASCENT_PATH_FUSED       k_eval_nodes / the passes of k_solve (scheme 0, formulation 0 only)
 *   ASCENT_PATH_SPLIT_LANE  q_trial_eval -> q_decide_factor -> q_forward -> q_local -> q_adjoint
 *   ASCENT_PATH_SPLIT_WIDE  q_trial_eval -> q_factor_wide -> q_forward_wide -> q_local -> q_adjoint_wide
 *   ASCENT_PATH_PERSIST     p_solve / h_solve (the default of ascent_solve_batch): see the enum below
 *   ASCENT_PATH_DENSE       d_eval -> d_newton (ascent_kkt_step_path; with move_penalty = 1 as well, like ASCENT_PATH_PERSIST)
 * The split paths take schemes 0/1 and formulations 0/1.  For scheme 1 (trapezoid) `defects` is the trapezoid
 * defect and the Hessian block of node k is weighted by -(h*T*tf/2)*(lambda_k + lambda_{k+1}). */
enum ascent_path { ASCENT_PATH_AUTO = 0, ASCENT_PATH_FUSED = 1, ASCENT_PATH_SPLIT_LANE = 2, ASCENT_PATH_SPLIT_WIDE = 3,
                   ASCENT_PATH_DENSE = 4, /* d_eval -> d_newton, one wavefront per NLP on dense 8x8 blocks: schemes 0/1/2 */
                   ASCENT_PATH_PERSIST = 5 /* one round of the persistent kernel -- p_solve (schemes 0 / 1, formulation 1 with scheme 0) or
                                              h_solve (scheme 2, Hermite-Simpson, without the move penalty): ascent_kkt_step_path returns its
                                              Newton step; ascent_eval_nodes_path (schemes 0 / 1) the node rows it stages in LDS for the
                                              factorisation sweep, copied out before the sweep would read them */ };
/* Which kernels ascent_solve_batch runs for a batch of this size with these options (and the environment overrides):
 * an ascent_path value, never ASCENT_PATH_AUTO.  No device work. */
int ascent_default_path(int64_t batch, const ascent_opts *o);
/* Diagnostic (no device work): the persistent kernel's device workspace for a solve of `batch` NLPs with these options.
 * Returns the number of grid levels of the nested iteration (>= 1) or a negative code; out4[0] = bytes allocated,
 * out4[1] = bytes the finest level's kernels use from offset 0, out4[2] = offset of the second region (the two regions
 * alternate between the levels; 0 with one level), out4[3] = bytes the largest level living in the second region uses. */
int ascent_workspace_layout(int64_t batch, const ascent_opts *o, int64_t *out4);
int ascent_eval_nodes_path(const ascent_params *p, int64_t batch, const ascent_opts *o,
                           const double *iterate, double *defects, double *jac_blocks,
                           double *hess_blocks, int device_id, int path);
int ascent_kkt_step_path(const ascent_params *p, int64_t batch, const ascent_opts *o,
                         const double *iterate, const double *mu, const double *delta_w, double *step,
                         int32_t *inertia_out, int device_id, int path);

/* Dense stage records as the dense-block path's node kernel (d_eval) leaves them at `iterate` (parity surface for the
 * Hermite-Simpson derivatives): records[batch][K][6][64] doubles, six row-major 8x8 grids per step (7 states + one
 * padding slot): 0 = d c_k/d z_{k-1}, 1 = d c_k/d z_k, 2/3/4 = the (z_{k-1},z_{k-1}) / (z_{k-1},z_k) / (z_k,z_k) blocks of the
 * Hessian of lambda_k'c_k, 5 = vectors by row: c_k, d c_k/du_k, d c_k/d tf, and the (z_{k-1},tf), (z_k,tf) Hessian
 * columns.  Host pointers. */
int ascent_dense_records(const ascent_params *p, int64_t batch, const ascent_opts *o, const double *iterate,
                         double *records, int device_id);

/* The coast arc after the burn (second phase of BASELINE config 5; the reference's v1 script propagated it with
 * explicit Euler, PDF p28-29): Kepler-exact two-body propagation of every problem's final state to the next
 * apoapsis of its orbit, sampled uniformly in time.
 * final_state [4][batch]: scaled x, y, xdot, ydot of the last node (rows 0..3 of traj_out at node n_nodes-1);
 * coast_traj [4][coast_nodes+1][batch] (same scaled units; node 0 = the burnout state);
 * coast_tf [batch]: coast duration / T_scale;  apsides [2][batch]: periapsis and apoapsis altitude above R0, m.
 * Host or device pointers (ptr_is_device), optional stream. */
int ascent_coast_batch(const ascent_params *p, int64_t batch, const double *final_state, int32_t coast_nodes,
                       double *coast_traj, double *coast_tf, double *apsides, int device_id,
                       void *hip_stream_or_null, int ptr_is_device);

/* Generic bordered block-tridiagonal solve (parity surface of the linear algebra):
 *     [ T   B ] [x]   [r]        T: n_nodes x n_nodes blocks of size bs (<= 16): diag[i] on the diagonal, lower[i] = block
 *     [ B'  d ] [y] = [s]           (i, i-1) (lower[0] ignored), upper[i] = block (i, i+1) (upper[n-1] ignored);
 *                                B: nb border columns (1 + nb <= 16), d: nb x nb (border_diag, row-major).
 * Layouts (host pointers, row-major, system index slowest): diag / lower / upper [batch][n_nodes][bs][bs],
 * border [batch][n_nodes][bs][nb], border_diag [batch][nb][nb], rhs and sol [batch][n_nodes*bs + nb].
 * algo 0: block elimination serial in the node index, one wavefront per system; algo 1: parallel cyclic reduction over
 * the nodes (one wavefront per node, log2(n_nodes) levels).  Blocks are padded to 16x16 and multiplied with
 * v_mfma_f64_16x16x4_f64; no pivoting inside blocks (returns ASCENT_E_ARG "singular pivot" if one vanishes); cyclic reduction exposes no inertia -- the interior-point solver uses it for a handful of NLPs only
 * and guards it with a curvature test along the step (a weaker guarantee than the exact inertia of the Riccati recursions); the
 * border is closed by a Schur complement on the host.  The interior-point solver does not call this (it uses a
 * Riccati recursion in the 7x7 value function); ascent_last_kernel_ms reports the device time of the solve. */
int ascent_kkt_solve(int64_t batch, int32_t n_nodes, int32_t bs, int32_t nb, const double *diag, const double *lower,
                     const double *upper, const double *border, const double *border_diag, const double *rhs,
                     double *sol, int device_id, int algo);

/* Device time (ms) of the solve kernel of the most recent ascent_solve_batch on this device,
 * measured with HIP events recorded on the launch stream around the kernel; waits for that
 * kernel to finish; < 0 if there was none. */
double ascent_last_kernel_ms(int device_id);

#ifdef __cplusplus
}
#endif
#endif /* ASCENT_H */

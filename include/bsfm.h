/* include/bsfm.h -- C ABI of libbsfm_hip.so (MI355X / gfx950 sparse bundle-adjustment core + SIFT matcher).
 *
 * Everything here is `extern "C"`, plain pointers and sizes.  Paths in comments are relative to the
 * reference tree (snavely/bundler_sfm); each entry point cites the reference interface it replaces.
 *
 *   1. Drop-in boundary      run_sfm                  <- lib/sfm-driver/sfm.h:68-86 (sfm.c:592-1003)
 *   2. Extended boundary     bsfm_run_sfm_ex          <- same, but returns SBA's info[10]
 *                                                        (lib/sba-1.5/sba_levmar.c:512-527) and takes options
 *   3. Resident-problem API  bsfm_problem_* / bsfm_lm_*   sparse (CRS) boundary of SURVEY section 8(f).2:
 *                                                        the visibility map is passed the way SBA builds it
 *                                                        internally (lib/sba-1.5/sba_levmar.c:653-663)
 *                                                        instead of the dense vmask; state stays in HBM
 *   4. Matcher               bsfm_match_keys_l2       <- MatchKeys, src/keys2a.h:99-107 (keys2a.cpp:347-372)
 *                            bsfm_key_match_full      <- KeyMatchFull main loop, src/KeyMatchFull.cpp:105-151
 *
 * Error behaviour mirrors the reference: nothing throws across the boundary, run_sfm returns void and
 * prints the same two summary lines (sfm.c:872-873); the *_ex / bsfm_* entries return SBA's code
 * (iterations >= 0, or BSFM_ERROR = SBA_ERROR = -1, lib/sba-1.5/sba.h:57) and fill info[].
 * There is NO CPU fallback: when no HIP device is usable every compute entry fails loudly
 * (message on stderr, BSFM_ERROR / NULL).
 */
#ifndef BSFM_H
#define BSFM_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSFM_ERROR (-1)
#define BSFM_INFOSZ 10
#define BSFM_NUM_CAMERA_PARAMS 9
#define BSFM_POLY_INVERSE_DEGREE 6

/* Layout-identical to v3_t / v2_t (lib/matrix/vector.h:59-67). */
typedef struct { double p[3]; } bsfm_v3_t;
typedef struct { double p[2]; } bsfm_v2_t;

/* Layout-identical to camera_params_t (lib/sfm-driver/sfm.h:32-51); sizeof == 504 on LP64. */
typedef struct {
    double R[9];
    double t[3];
    double f;
    double k[2];
    double k_inv[BSFM_POLY_INVERSE_DEGREE];
    char constrained[BSFM_NUM_CAMERA_PARAMS];
    double constraints[BSFM_NUM_CAMERA_PARAMS];
    double weights[BSFM_NUM_CAMERA_PARAMS];
    double K_known[9];
    double k_known[5];
    char fisheye;
    char known_intrinsics;
    double f_cx, f_cy;
    double f_rad, f_angle;
    double f_focal;
    double f_scale, k_scale;
} bsfm_camera_params_t;

/* Jacobian used by the LM loop. */
enum {
    BSFM_JAC_FD = 0,       /* forward differences with the reference's steps
                              (lib/sba-1.5/sba_levmar_wrap.c:203-256): reference-identical semantics */
    BSFM_JAC_ANALYTIC = 1  /* closed-form derivative of the same model (no counterpart in Bundler,
                              which passes projac=NULL, sfm.c:820-828) */
};

enum { BSFM_SOLVER_DENSE = 0, BSFM_SOLVER_AUTO = 1, BSFM_SOLVER_ENVELOPE = 2 };

typedef struct {
    int jacobian;        /* BSFM_JAC_FD (default for run_sfm) or BSFM_JAC_ANALYTIC */
    int itmax;           /* default 150  (MAX_ITERS, sfm.c:814) */
    int verbose;         /* 0 silent; >=1 prints the two run_sfm summary lines; >=2 per-iteration lines */
    double opts[6];      /* tau, eps1, eps2, eps3, eps4, eps5 (sfm.c:705-714); eps2 is overwritten by the
                            run_sfm argument */
    int potrf_backend;   /* 0 = own MFMA-f64 tiled Cholesky (default), 1 = rocSOLVER cross-check (dlopen) */
    int reduced_solver;  /* BSFM_SOLVER_DENSE (default): Cholesky of the whole dense S, the reference's algorithm
                            (sba_Axb_Chol); BSFM_SOLVER_AUTO: when the cameras fall into groups that share no point
                            (S block diagonal up to a permutation, every group <= 128 unknowns) solve group by group,
                            otherwise the envelope solver.  BSFM_SOLVER_ENVELOPE: the cameras are renumbered (reverse
                            Cuthill-McKee on the co-visibility graph) and the tiled Cholesky skips the 128 x 128 tiles outside
                            the envelope of the reordered S -- exact (Cholesky without pivoting fills nothing outside the
                            envelope), dpotrf's info then counts in the reordered system.  Same solution up to rounding.
                            Env: BSFM_REDUCED_SOLVER=auto|dense|envelope. */
    int num_gpus;        /* run_sfm / bsfm_run_sfm_ex only: 0 or 1 = one GPU (default; env BSFM_NUM_GPUS overrides), n > 1 = shard the
                            points over the first n visible devices inside this one process (one host thread per GPU, RCCL
                            all-reduce of the camera system over xGMI, comm.hip), -1 = all visible devices */
} bsfm_options_t;

void bsfm_default_options(bsfm_options_t *opt);

/* ---- 1. drop-in boundary: identical signature to lib/sfm-driver/sfm.h:68-86 ------------------------- */
void run_sfm(int num_pts, int num_cameras, int ncons,
             char *vmask, double *projections,
             int est_focal_length, int const_focal_length,
             int undistort, int explicit_camera_centers,
             bsfm_camera_params_t *init_camera_params, bsfm_v3_t *init_pts,
             int use_constraints, int use_point_constraints,
             bsfm_v3_t *points_constraints, double point_constraint_weight,
             int fix_points, int optimize_for_fisheye, double eps2,
             double *Vout, double *Sout, double *Uout, double *Wout);

/* ---- 2. same, returning SBA's code and info[10]; opt may be NULL ------------------------------------ */
int bsfm_run_sfm_ex(int num_pts, int num_cameras, int ncons,
                    char *vmask, double *projections,
                    int est_focal_length, int const_focal_length,
                    int undistort, int explicit_camera_centers,
                    bsfm_camera_params_t *init_camera_params, bsfm_v3_t *init_pts,
                    int use_constraints, int use_point_constraints,
                    bsfm_v3_t *points_constraints, double point_constraint_weight,
                    int fix_points, int optimize_for_fisheye, double eps2,
                    double *Vout, double *Sout, double *Uout, double *Wout,
                    const bsfm_options_t *opt, double info[BSFM_INFOSZ]);

/* ---- 2b. SBA-level sibling ---------------------------------------------------------------------------
 * Same argument list as the reference's sba_motstr_levmar (lib/sba-1.5/sba.h:95-108 as extended by Bundler:
 * constraints and the V/S/U/W outputs, sba_levmar_wrap.c:599-698), except that the two host callbacks + adata
 * (proj, projac, adata) -- which a GPU core cannot honour per observation -- are replaced by a camera-model id and its
 * data block.  p (m*cnp camera + n*pnp point parameters) is refined in place; returns the number of iterations or
 * BSFM_ERROR (SBA_ERROR = -1), info[10] as SBA.  Restrictions, all failing loudly: pnp = 3, mnp = 2, covx = NULL,
 * omega (p[j*cnp+3..5]) zero on entry as in run_sfm (sfm.c:652-696), point-constraint weights uniform. */
#define BSFM_MODEL_SNAVELY 1     /* lib/sfm-driver/sfm.c:503-552: centre, incremental rotation, f*0.001, k*5.0 */
typedef struct {
    int est_focal_length, undistort, explicit_camera_centers;   /* cnp = 6 + est + 2*undistort */
    const double *R_init;        /* 9*m row-major: rotation of camera j at omega = 0 (init_params[j].R) */
    const double *f_init;        /* m: focal lengths used when est_focal_length == 0 (may be NULL otherwise) */
    const double *points;        /* 3*n: the fixed points of bsfm_sba_mot_levmar (globs->points of sfm.c:554-561); else NULL */
} bsfm_snavely_model_t;
typedef struct { char *constrained; double *constraints; double *weights; } bsfm_camera_constraints_t;   /* sba.h:80-84 */
typedef struct { char constrained; double constraints[3]; double weight; } bsfm_point_constraints_t;     /* sba.h:86-90 */
int bsfm_sba_motstr_levmar(int n, int m, int mcon, char *vmask, double *p, int cnp, int pnp,
                           double *x, double *covx, int mnp,
                           int camera_model, const void *model_data,
                           int itmax, int verbose, double opts[6], double info[BSFM_INFOSZ],
                           int use_constraints, bsfm_camera_constraints_t *constraints,
                           int use_point_constraints, bsfm_point_constraints_t *point_constraints,
                           double *Vout, double *Sout, double *Uout, double *Wout);

/* Camera-only sibling: sba_mot_levmar's argument list (lib/sba-1.5/sba_levmar_wrap.c:707-768) with the model id + data
 * block in place of proj/projac/adata; p holds the m*cnp camera parameters, the points come from model->points. */
int bsfm_sba_mot_levmar(int n, int m, int mcon, char *vmask, double *p, int cnp,
                        double *x, double *covx, int mnp,
                        int camera_model, const void *model_data,
                        int itmax, int verbose, double opts[6], double info[BSFM_INFOSZ],
                        int use_constraints, bsfm_camera_constraints_t *constraints);

/* ---- 3. resident-problem API ------------------------------------------------------------------------ */
typedef struct bsfm_problem bsfm_problem_t;

#define BSFM_ARRAYS_ON_DEVICE 1
#define BSFM_INDEX_ON_DEVICE 2
typedef struct {
    int n, m, mcon;              /* points, cameras, leading fixed cameras (SBA's mcon) */
    const int *rowptr;           /* n+1: CRS row pointers, one row per point (sba_levmar.c:653-663) */
    const int *colidx;           /* nvis: camera index of each observation, ascending within a row */
    const double *projections;   /* 2*nvis measurements in CRS order (== vmask row-major order) */
    int est_focal_length, undistort, explicit_camera_centers;
    const bsfm_camera_params_t *cameras;   /* m */
    const double *points;        /* 3*n */
    int use_constraints;         /* camera constraints taken from cameras[j].constrained/constraints/weights */
    int use_point_constraints;
    const double *point_constraints;       /* 3*n or NULL; all-zero row = unconstrained (sfm.c:757-781) */
    double point_constraint_weight;
    /* multi-GPU: this rank owns the points listed above (all their observations); cameras
     * are replicated.  nvis_global is the job-wide observation count (point-constraint weights scale by
     * it, sba_levmar.c:1017-1028); 0 means "same as local". */
    int world_size, rank;
    long long nvis_global;
    long long nvars_global;      /* m*cnp + 3*n_global, for the nobs<nvars check; 0 = local */
    /* SBA-level callers (bsfm_sba_motstr_levmar) hand over the LM vector itself: m*cnp + 3*n doubles already in the
     * scaled parametrisation of sfm.c:652-703 (then cameras[j] only supplies R, and f when the focal is not estimated),
     * and constraint values/weights already in that parametrisation (the rescaling of sfm.c:721-754 is skipped). */
    const double *p_packed;      /* NULL = pack from cameras/points */
    int constraints_prescaled;
    /* Camera-only refinement (run_sfm's fix_points != 0 -> sba_mot_levmar, lib/sfm-driver/sfm.c:839-846): the points are
     * constants, the normal equations decouple into one cnp x cnp system per camera (sba_levmar.c:2090-2690). */
    int fix_points;
    /* run_sfm's optimize_for_fisheye (sfm.c:819-851): project with sfm_project_point2_fisheye (sfm.c:448-492) -- pinhole
     * without the radial term, then the equidistant map of cameras[j].fisheye/f_cx/f_cy/f_rad/f_angle/f_focal. */
    int optimize_for_fisheye;
    /* 1 (BSFM_ARRAYS_ON_DEVICE): rowptr, colidx, projections and points are DEVICE pointers (data already resident in HBM, e.g.
     * produced by another kernel); point constraints cannot be combined with it.  2 (BSFM_INDEX_ON_DEVICE): only rowptr and colidx
     * are device pointers (run_sfm builds them there from the dense mask), everything else comes from the host.  cameras always
     * come from the host (504 bytes each). */
    int arrays_on_device;
} bsfm_problem_desc_t;

/* Sum-reduce `count` doubles in place across ranks (device pointer); op 0 = sum, 1 = max.
 * Called by the LM loop at its exchange steps when world_size > 1.  Must return 0 on success. */
typedef int (*bsfm_allreduce_fn)(void *device_buf, size_t count, int op, void *ctx);

/* ---- library-side collective (north star: RCCL reduce over xGMI behind the C boundary) ----------------------------
 * A communicator is one rank's handle.  bsfm_comm_create_from_env: one rank per PROCESS, RANK / WORLD_SIZE / LOCAL_RANK /
 * MASTER_PORT from the environment as torch.distributed.run or an mpirun wrapper export them (selects device LOCAL_RANK; the
 * ncclUniqueId travels through a file under /dev/shm, single node).  bsfm_comm_create_all: one rank per THREAD of this process
 * over the listed devices (ncclCommInitAll); when a device is listed twice the ranks use an in-process loopback transport
 * instead (test rigs with one GPU).  All-reduces are in place on device memory, op 0 = sum, 1 = max, enqueued on `stream`. */
typedef struct bsfm_comm bsfm_comm_t;
bsfm_comm_t *bsfm_comm_create_from_env(void);
int bsfm_comm_create_all(int ndev, const int *devs, bsfm_comm_t **comms);
void bsfm_comm_destroy(bsfm_comm_t *c);
int bsfm_comm_rank(const bsfm_comm_t *c);
int bsfm_comm_world(const bsfm_comm_t *c);
const char *bsfm_comm_transport(const bsfm_comm_t *c);      /* "rccl", "loopback" or "none" (world 1) */
int bsfm_comm_allreduce(bsfm_comm_t *c, void *device_buf, size_t count, int op, void *stream);
int bsfm_comm_allreduce_host(bsfm_comm_t *c, double *vals, int count, int op);     /* count <= 256, synchronous */
int bsfm_comm_barrier(bsfm_comm_t *c);
/* Collective: every rank passes one device allocation of its own (the BASE pointer of a hipMalloc) and receives pointers through which it can
 * address every rank's allocation (its own at peers[rank]).  "ipc" transport (hipIpc handles: processes sharing a device, or the devices of a
 * node over xGMI peer access) and "loopback" (threads of one process); BSFM_ERROR on "rccl".  What the distributed reduced-camera
 * factorisation (replaces the replicated sba_Axb_Chol, lib/sba-1.5/sba_levmar.c:1368) maps its panel tiles and counters with. */
int bsfm_comm_share(bsfm_comm_t *c, void *mine, void **peers);
int bsfm_comm_unshare(bsfm_comm_t *c, void **peers);
/* Test hook (no device, no RCCL): the id hand-over of bsfm_comm_create_from_env on its own (csrc/idfile.h).  rank 0 publishes
 * id[128] at `path`; any other rank waits up to timeout_s for a record that was written by THIS user with mode 0600, is no symbolic
 * link, carries this world size and is not older than the calling process by more than grace_s (120 in production), and receives it
 * in id[128].  Returns 0 or BSFM_ERROR (message on stderr). */
int bsfm_comm_idfile_exchange(const char *path, int rank, int world, double timeout_s, double grace_s, unsigned char *id);

bsfm_problem_t *bsfm_problem_create(const bsfm_problem_desc_t *desc, const bsfm_options_t *opt);
void bsfm_problem_destroy(bsfm_problem_t *pb);
void bsfm_problem_set_allreduce(bsfm_problem_t *pb, bsfm_allreduce_fn fn, void *ctx);
/* Use a library-side communicator (above) for every exchange step; takes precedence over the callback. */
void bsfm_problem_set_comm(bsfm_problem_t *pb, bsfm_comm_t *comm);
/* Use an externally owned HIP stream (e.g. torch's current stream) for every launch; NULL = own stream. */
void bsfm_problem_set_stream(bsfm_problem_t *pb, void *hip_stream);
/* Grow the resident problem between the rounds of an incremental reconstruction (the reference rebuilds vmask,
 * projections and every SBA work array from scratch for each run_sfm call, src/BundleFast.cpp:263-438 -> src/Bundle.cpp:597-637):
 * `num_new_cameras` cameras are appended after the existing ones (indices m .. m + num_new_cameras - 1), `num_new_points` points
 * after the existing ones, and `nadd` new observations (add_pt[q], add_cam[q], add_xy[2q], add_xy[2q+1]) -- of old or new points by
 * old or new cameras, in any order -- are merged into the measurement order.  Only the new data crosses PCIe: the observations,
 * the points and the current estimate of every old parameter stay in HBM (old cameras: rotation increment folded into R exactly as
 * run_sfm hands them back, sfm.c:876-922, so the next LM run starts like a fresh run_sfm call on the grown scene).  New points are
 * unconstrained.  The handle stays valid; on failure (BSFM_ERROR) the problem is unchanged.  Single-rank problems only. */
int bsfm_problem_append(bsfm_problem_t *pb, int num_new_cameras, const bsfm_camera_params_t *new_cameras,
                        int num_new_points, const double *new_points,
                        int nadd, const int *add_pt, const int *add_cam, const double *add_xy);
/* Re-upload parameters (cameras: centre/rotation/focal/k; points) without rebuilding the index. */
int bsfm_problem_reset_params(bsfm_problem_t *pb, const bsfm_camera_params_t *cameras, const double *points);

/* LM driver (restates lib/sba-1.5/sba_levmar.c:457-2081): begin computes the initial cost;
 * iterate runs up to `iters` outer iterations (fewer if a stop rule fires) and returns the stop code
 * (0 = still running); finish fills info[10] like the reference and returns iterations or BSFM_ERROR. */
int bsfm_lm_begin(bsfm_problem_t *pb);
int bsfm_lm_iterate(bsfm_problem_t *pb, int iters);
int bsfm_lm_finish(bsfm_problem_t *pb, double info[BSFM_INFOSZ]);
int bsfm_lm_solve_attempts(const bsfm_problem_t *pb);   /* linear systems solved so far (info[9]) */
double bsfm_lm_last_kernel_ms(const bsfm_problem_t *pb, const char *phase); /* HIP-event time of a phase in the last iteration;
                                                    "groups": camera groups solved separately (0 = dense reduced solve) */

/* Download results: packed parameter vector p (m*cnp + 3n, reference layout sfm.c:652-703), and/or
 * updated cameras (R <- dR(w) R, t <- c, f, k as sfm.c:876-922) and points. Any pointer may be NULL. */
int bsfm_problem_download(bsfm_problem_t *pb, double *p_out, bsfm_camera_params_t *cameras, double *points);
/* Index bookkeeping as the kernels see it (test entries, SURVEY 8 rows a7 / a20: "bit-exact"): the CRS of the visibility mask
 * (struct sba_crsm, lib/sba-1.5/sba.h:70-78, filled as lib/sba-1.5/sba_levmar.c:653-663) and the camera-major traversal that
 * sba_crsm_col_elmidxs (lib/sba-1.5/sba_crsm.c:183-212) re-derives for every camera: camobs[camptr[j] ..] = observation indices
 * of camera j in ascending point order, campos = its inverse, cam_pt / cam_cam = point / camera of a camera-major position.
 * Downloaded from HBM; any pointer may be NULL. */
int bsfm_problem_export_index(bsfm_problem_t *pb, int *rowptr, int *colidx, int *obs_pt, int *camptr, int *camobs, int *campos,
                              int *cam_pt, int *cam_cam);
/* Co-visibility structure of the Schur complement (replaces the pair search of lib/sba-1.5/sba_levmar.c:1218-1268): triples
 * (2 ints each: camera-major positions of (i,j) and (i,k)) grouped by block (j <= k) in (j,k) order and in point order inside a
 * block; tri_pt = point of a triple; blk_j / blk_k (nblk), blk_task0 (nblk + 1); tasks (4 ints each: start, count, diag, slot;
 * nslots entries in launch order, slot -1 = padding). */
int bsfm_problem_schur_sizes(const bsfm_problem_t *pb, int *ntriples, int *nblk, int *ntasks, int *nslots);
/* co-visibility triples per Schur task (a multiple of 16; default 192, BSFM_SCHUR_CHUNK overrides it before the first problem) */
int bsfm_schur_chunk(void);
/* Resident problems take their device buffers from a process-wide cache of blocks and give them back to it when they are destroyed
 * (hipFree synchronises the device and costs 40-60 us per buffer: a fifth of a 14-camera run_sfm call).  The cache is bounded
 * (BSFM_DEVCACHE_MB, default 6144; 0 = no cache) and empties itself when an allocation fails; this call empties it on request,
 * e.g. before another library needs the memory. */
void bsfm_device_cache_trim(void);
int bsfm_problem_export_schur(bsfm_problem_t *pb, int *triples, int *tri_pt, int *blk_j, int *blk_k, int *blk_task0, int *tasks);
/* dense vmask -> CRS exactly as run_sfm does it (host only); returns nvis, rowptr / colidx may be NULL. */
int bsfm_crs_from_vmask(int n, int m, const char *vmask, int *rowptr, int *colidx);
/* The same CRS built ON THE DEVICE (what run_sfm does for masks of at least 1 MB, BSFM_VMASK_DEVICE_MIN: upload, count / scan /
 * compaction kernels; replaces the two single-thread passes over the n*m bytes of lib/sba-1.5/sba_levmar.c:642-663) and copied
 * back; bit-identical to bsfm_crs_from_vmask.  ms_out (3 doubles or NULL): upload / kernels / total wall ms.  Needs a HIP device. */
int bsfm_crs_from_vmask_device(int n, int m, const char *vmask, int *rowptr, int *colidx, double *ms_out);
/* Wall milliseconds of the phases of the calling thread's last run_sfm / bsfm_run_sfm_ex call: "total", "crs" ("crs_upload",
 * "crs_kernels", "crs_on_device"), "create", "lm", "download"; -1 for an unknown name. */
double bsfm_run_sfm_last_ms(const char *phase);
/* Shrinking a resident problem -- the other half of the RunSFM_SBA outlier loop (src/Bundle.cpp:784-913: the points flagged by
 * the per-camera thresholds are dropped with all their views, then run_sfm runs again; bsfm_problem_outlier_stats delivers the
 * flags): removes the points with remove[i] != 0 (host array, n entries) and every observation of them ON THE DEVICE; the
 * remaining points keep their order, cameras, parameters and constraints stay in HBM, the index is rebuilt there.  remap_out
 * (may be NULL, n entries) receives the new index of every old point or -1 (Bundler's remap table).  The next bsfm_lm_begin
 * starts from the resident parameters.  Returns the number of points removed, < 0 on error. */
int bsfm_problem_remove_points(bsfm_problem_t *pb, const unsigned char *remove, int *remap_out);
int bsfm_problem_cnp(const bsfm_problem_t *pb);
int bsfm_problem_num_cameras(const bsfm_problem_t *pb);      /* grows with bsfm_problem_append */
int bsfm_problem_num_points(const bsfm_problem_t *pb);
long long bsfm_problem_nvis(const bsfm_problem_t *pb);

/* Component entries used by the parity tests (each one launches the production kernels):
 *  residuals: e = x - proj(p) in CRS order, returns ||e||^2 (+ constraint terms) in *cost
 *  normal equations at the current p with damping mu: any output may be NULL
 *    U (m*cnp*cnp, row-major blocks incl. constraints, diagonal + mu), ea (m*cnp),
 *    V (n*9 full symmetric, diagonal + mu), eb (n*3), J (nvis*(2*cnp+6): A_ij row-major then B_ij),
 *    S ((m-mcon)*cnp squared, dense symmetric), E ((m-mcon)*cnp)                                  */
int bsfm_eval_residuals(bsfm_problem_t *pb, double *e_out, double *cost);
/* Post-solve statistics of RunSFM_SBA (src/Bundle.cpp:659-913) at the problem's current parameters, without a round
 * trip of cameras/points: reprojection distance d = |x - proj| per observation; per camera the observation count, mean
 * distance, kth_element_copy(n, iround(0.8 n)) and (n, iround(0.5 n)) (lib/imagelib/qsort.c:152-203; 0.0 when k >= n) and
 * the outlier threshold clamp(1.2 * 2.0 * kth80, min_thr, max_thr) (Bundle.cpp:761-771; Bundler passes
 * m_min_proj_error_threshold = 8, m_max_proj_error_threshold = 16); per point the outlier flag (an observation above its
 * camera's threshold; points whose constraint has a non-zero x component are exempt, Bundle.cpp:800-804) and the error
 * of the first flagged observation in camera order (what "[RunSFM] Removing outlier" prints).  Any output may be NULL. */
int bsfm_problem_outlier_stats(bsfm_problem_t *pb, double min_thr, double max_thr,
                               int *cam_nobs, double *cam_mean, double *cam_kth80, double *cam_kth50, double *cam_thresh,
                               unsigned char *point_outlier, double *point_err, double *global_mean);
/* Ray-angle pruning of BundlerApp::RemoveBadPointsAndCameras (src/Bundle.cpp:4190-4261) at the resident parameters: per
 * point the largest angle (degrees) between the rays X_i - c_j to any two of its cameras; prune[i] = 1 when the point has
 * views and that angle is below 0.5 * ray_angle_threshold (Bundler's m_ray_angle_threshold = 2.0, BundlerApp.h:83).
 * Camera centres are parameters 0..2 (explicit_camera_centers, as Bundler always runs).  Any output may be NULL. */
int bsfm_problem_ray_angles(bsfm_problem_t *pb, double ray_angle_threshold, double *max_angle_deg, unsigned char *prune,
                            int *num_pruned);
int bsfm_eval_normal_equations(bsfm_problem_t *pb, double mu, double *U, double *ea, double *V, double *eb,
                               double *J, double *S, double *E);
/* Dense SPD solve on the device with the production Cholesky: A (n x n, symmetric, row-major, host),
 * b (n) -> x (n). Returns 0, or k>0 if the leading minor k is not positive definite (dpotrf's info). */
int bsfm_dense_chol_solve(int n, const double *A, const double *b, double *x, int backend);
/* The same solve `reps` times (A is uploaded again before each): ms_out[reps] = device time of every repetition (factorisation +
 * both substitutions, HIP events on the solve's stream), *flow_ms_out = mean HIP-event time of the k_chol_flow launches,
 * *flow_gflop_out = the flops (1e9) the library scheduled for one launch.  bench.py's `dense_valued_S` leg: the reference's
 * dpotrf + dpotrs (lib/sba-1.5/sba_lapack.c:374-485) on a matrix whose tiles all hold numbers.  Outputs may be NULL. */
int bsfm_dense_chol_solve_timed(int n, const double *A, const double *b, double *x, int backend, int reps, double *ms_out,
                                double *flow_ms_out, double *flow_gflop_out);
/* Test / diagnostic hook (no device needed): the static task order of the tile-dataflow Cholesky (csrc/chol_flow_sched.h) for a
 * system of nblk tile columns.  last[k] (NULL = dense) = last tile row of column k's envelope.  tasks_out (NULL to query the
 * count) receives 40-byte records { u8 type, np, part, nwait; u16 i, j, p0, pad; u32 sig; { u32 idx, thr } w[3] }: task types
 * 0 POTRF, 1 TRSM32 (16 parts), 2 TRSM64 (2), 3 UPD32 (10), 4 UPD64 (2), 5 UPD128, 6 FTRSM, 7 FUPD; a task waits until counter
 * w[q].idx >= w[q].thr for its nwait conditions and increments counter sig when done; counter of tile (i, j) = i * nblk + j, row
 * nblk = the right-hand side.  np_max / slots <= 0 select the defaults.  Returns the number of tasks, or -1 when the builder
 * fails its own dependency check (every wait must be satisfiable by tasks that come EARLIER in the order). */
int bsfm_chol_flow_schedule(int nblk, const int *last, int np_max, int slots, void *tasks_out, int capacity, double *sim_us);
/* bsfm_dense_chol_solve by the ranks of a communicator TOGETHER: the distributed tile-dataflow factorisation (tile column j belongs to rank
 * j mod world; panel tiles, inverse diagonal factors, y, x and the hand-off counters through peer-mapped windows, bsfm_comm_share) that replaces the
 * REPLICATED sba_Axb_Chol of the multi-GPU path (lib/sba-1.5/sba_levmar.c:1368; SURVEY 8(e) "what does not shard").  COLLECTIVE: every rank
 * passes the same A and b and receives the same x -- bit-identical to bsfm_dense_chol_solve's -- and the same return value.  Transports "ipc" and
 * "loopback"; a hand-off time-out on any rank makes every rank repeat the solve on its own (stream-ordered schedule).  Inside run_sfm / bsfm_lm_*:
 * BSFM_DIST_CHOL=1 with such a communicator. */
int bsfm_dense_chol_solve_dist(bsfm_comm_t *comm, int n, const double *A, const double *b, double *x, int backend);
/* Test / diagnostic hook (no device needed): the host-side plan of the DYNAMIC tile-dataflow Cholesky (csrc/chol_dyn_plan.h, the
 * round-6 default; replaces sba_Axb_Chol, lib/sba-1.5/sba_lapack.c:374-485).  chain_out / potrf_out receive the two static queues
 * (40-byte records as above; a wait whose thr has bit 31 set compares the low 10 bits of the word), init_out the initial image of
 * the launch's state words.  meta[8] = { chain tasks, POTRF tasks, words, offset of the TRSM32 counters, of the UPD32 counters, of
 * the "inverse diagonal factor exists" counters, of the rowdone pairs, of the half-tile state words (column-major pairs, nblk + 1
 * rows per column) }.  Buffers may be NULL / capacities 0 to query the sizes.  Returns 0, -1 on a size the format cannot hold. */
int bsfm_chol_dyn_plan(int nblk, const int *last, void *chain_out, int chain_cap, void *potrf_out, int potrf_cap,
                       unsigned *init_out, int init_cap, int *meta);

/* ---- 3b. batched multi-view triangulation -------------------------------------------- */
/* npoints independent points; point i owns views view_ptr[i] .. view_ptr[i+1]-1.  View v observes the normalised image
 * point p[2v], p[2v+1] in the camera with rotation R (9, row-major) and translation t (3): x = (R X + t).xy / (R X + t).z.
 * view_cam == NULL: R / t hold one entry per VIEW (the argument layout of triangulate_n, lib/imagelib/triangulate.h:38-43);
 * otherwise view_cam[v] indexes R / t of ncams cameras.  Modes:
 *   BSFM_TRI_N         triangulate_n         lib/imagelib/triangulate.c:181-272  linear least squares + lmdif polish (tol 1e-5)
 *   BSFM_TRI_N_REFINE  triangulate_n_refine  lib/imagelib/triangulate.c:133-178  lmdif polish of the point passed in X
 *   BSFM_TRI_PAIR      triangulate           lib/imagelib/triangulate.c:281-338  exactly two views, tol 1e-10
 * X (3*npoints, in/out), error (npoints or NULL: rms reprojection error; BSFM_TRI_PAIR: sum of squares),
 * info (npoints or NULL: MINPACK's lmdif1 code).  Host pointers.  Returns 0 or BSFM_ERROR (nothing written). */
#define BSFM_TRI_N 0
#define BSFM_TRI_N_REFINE 1
#define BSFM_TRI_PAIR 2
int bsfm_triangulate_batch(int mode, int npoints, const int *view_ptr, const double *p, const int *view_cam, int ncams,
                           const double *R, const double *t, double *X, double *error, int *info);

/* ---- 3c. batched epipolar geometry ---------------------------------------------------- */
/* glibc's rand() restated (random(), TYPE_3 additive feedback; srand(seed) == bsfm_rand_seed): the reference draws its
 * RANSAC samples with rand() (lib/imagelib/fmatrix.c:352), so a caller that wants the reference's samples seeds this the
 * way it seeds rand() (Bundler never calls srand: seed 1) and hands the state from call to call. */
typedef struct { unsigned int s[31]; int fi, ri; } bsfm_rand_t;
void bsfm_rand_seed(bsfm_rand_t *st, unsigned int seed);
int bsfm_rand_next(bsfm_rand_t *st);
/* estimate_fmatrix_ransac_matches (lib/imagelib/fmatrix.c:293-475, essential = 0) for npairs image pairs in order: pair p
 * owns matches match_ptr[p] .. match_ptr[p+1]-1; match q pairs the point (a_xy[2q], a_xy[2q+1], 1) of the FIRST point
 * argument with (b_xy[2q], b_xy[2q+1], 1) of the second (EstimateFMatrix passes k2 first, src/Epipolar.cpp:149).
 * Per pair: F (9, row-major; untouched when the pair has fewer than 8 matches or no trial finds an inlier) and the inlier
 * count of the best trial.  `rng` advances exactly as rand() would in the reference, early exits included. */
int bsfm_fmatrix_ransac_batch(int npairs, const int *match_ptr, const double *a_xy, const double *b_xy, int num_trials,
                              double threshold, double success_ratio, bsfm_rand_t *rng, double *F, int *inliers_max);
/* EstimateFMatrix (src/Epipolar.cpp:118-237, essential = false) for npairs image pairs in order: pairs with fewer than 20
 * matches are turned away; RANSAC as above with (k2, k1) and success ratio 0.95; inliers of its matrix; non-linear
 * refinement on them (refine_fmatrix_nonlinear_matches, lib/imagelib/fmatrix.c:637-659: lmdif, tol 1e-12, rank-2
 * projection inside the residual); inliers of the refined matrix.  k1_xy / k2_xy: keypoint positions of the two images,
 * 2 doubles per match.  Out: F (9 per pair; untouched where nothing was estimated), num_inliers (per pair), inlier
 * (1 byte per match), lm_info (per pair, MINPACK's code; may be NULL). */
int bsfm_estimate_fmatrix_batch(int npairs, const int *match_ptr, const double *k1_xy, const double *k2_xy, int num_trials,
                                double threshold, bsfm_rand_t *rng, double *F, int *num_inliers, unsigned char *inlier,
                                int *lm_info);

/* ---- 3d. track building ------------------------------------------------------------------- */
/* BundlerApp::ComputeTracks (src/ComputeTracks.cpp:36-313) on the match table as BaseApp::LoadMatchTable holds it (src/BundleIO.cpp:112-166:
 * one list per image pair pair_i[p] < pair_j[p], matches[2q] = key index in image pair_i, matches[2q+1] = key index in image pair_j), made
 * symmetric as MakeMatchListsSymmetric does (src/MatchTracks.cpp:337-392).  Tracks come back in the reference's numbering: track t owns
 * views track_ptr[t] .. track_ptr[t+1]-1, each (image, key), in the order the reference's breadth-first search claimed them.  Inside a pair a
 * key may occur at most once on either side (what PruneDoubleMatches leaves, src/MatchTracks.cpp:394-440); otherwise BSFM_ERROR.  Returns
 * the number of tracks (*num_views = total views); with track_ptr or views NULL only the counts are computed.  new_image_start is accepted
 * for signature parity and, as in the reference, has no effect. */
int bsfm_compute_tracks(int num_images, const int *num_keys, int num_pairs, const int *pair_i, const int *pair_j,
                        const int *match_ptr, const int *matches, int new_image_start,
                        int *track_ptr, int *views, int max_tracks, int max_views, int *num_views);

/* ---- 4. matcher ------------------------------------------------------------------------------------- */
/* Exact 2-NN ratio test between two descriptor sets (128-D uchar, squared L2 in int32):
 * keeps (i, nn0) iff (double)d0 < ratio*ratio*(double)d1 (src/keys2a.cpp:362). out_pairs gets up to
 * max_out (idx1, idx2) pairs in ascending idx1 order; returns the number of matches (may exceed max_out),
 * or BSFM_ERROR.  n2 < 2 is an error (the reference's ANN aborts in that case). */
int bsfm_match_keys_l2(int n1, const unsigned char *k1, int n2, const unsigned char *k2, double ratio,
                       int *out_pairs, int max_out);
/* All-pairs driver with KeyMatchFull's loop structure and output format (src/KeyMatchFull.cpp:105-151):
 * keys[i] -> num_keys[i] x 128 uchar (host); writes the text to `out_path` ("j i\nN\nidx_j idx_i\n...")
 * for pairs with >= 16 matches; window_radius <= 0 means all pairs.  Returns the number of pair blocks written. */
int bsfm_key_match_full(int num_images, const int *num_keys, const unsigned char *const *keys,
                        double ratio, int window_radius, const char *out_path);
/* Multi-GPU form (pair-parallel, no collective): rank r of world_size handles the database images i with
 * i % world_size == r (each against all j < i) and writes its own file; bsfm_merge_match_files then restores the single-run
 * file byte for byte (k-way merge of the "j i" blocks on i). */
int bsfm_key_match_full_sharded(int num_images, const int *num_keys, const unsigned char *const *keys,
                                double ratio, int window_radius, const char *out_path, int rank, int world_size);
int bsfm_merge_match_files(int count, const char *const *paths, const char *out_path);
/* Resident key set: the descriptors of all images (and their per-key statistics) stay in HBM across runs, so that repeated
 * matching passes (different window radius / ratio, or a timed benchmark pass) do not re-upload 128 bytes per key.
 * bsfm_key_match_full_sharded == create + run + destroy.  After a run, bsfm_match_set_stats reports the HIP-event time of the
 * brute-force kernels' launches (the union of their [start, end] intervals: consecutive launches run on two streams so that the
 * tail of one overlaps the head of the next), the number of descriptor distances they evaluated (x 128 MAC each), the image
 * pairs searched and the launch count. */
typedef struct bsfm_match_set bsfm_match_set_t;
bsfm_match_set_t *bsfm_match_set_create(int num_images, const int *num_keys, const unsigned char *const *keys);
int bsfm_match_set_run(bsfm_match_set_t *ms, double ratio, int window_radius, const char *out_path, int rank, int world_size);
int bsfm_match_set_stats(const bsfm_match_set_t *ms, double *kernel_ms, double *distances, long long *pairs, int *launches);
/* Scan kernel of the matcher: 0 = auto (default; per launch, from the share of accepted matches the last finished launch had),
 * 1 = k_match_l2, exact running top-2 (cost independent of the data), 2 = k_match_bound, one running maximum per slot + bounds +
 * exact rescan of the winning slot (faster when few queries pass the ratio test, slower when many do).  All three give the same matches (keys2a.cpp:347-372).  Returns the
 * previous setting; an out-of-range value only queries.  Environment: BSFM_MATCH_KERNEL=auto|top2|rescan. */
int bsfm_match_kernel(int mode);
/* launches of the last bsfm_match_set_run* that used the rescan kernel (of bsfm_match_set_stats' `launches`) */
int bsfm_match_set_rescan_launches(const bsfm_match_set_t *ms);
/* The same search with the match table in memory instead of text: pair p = images pair_i[p] < pair_j[p] in the
 * order of the text file, its matches matches[2q] (key of pair_i) / matches[2q+1] (key of pair_j) for q in match_ptr[p] ..
 * match_ptr[p+1]-1 -- the layout bsfm_compute_tracks and BaseApp::LoadMatchTable (src/BundleIO.cpp:112-166) use.  Only pairs
 * with >= 16 matches appear (KeyMatchFull.cpp:131).  The four arrays are allocated by the library: release with bsfm_free.
 * Returns the number of pairs. */
int bsfm_match_set_run_table(bsfm_match_set_t *ms, double ratio, int window_radius, int rank, int world_size,
                             int **pair_i, int **pair_j, int **match_ptr, int **matches);
void bsfm_free(void *p);
void bsfm_match_set_destroy(bsfm_match_set_t *ms);

/* ---- utilities --------------------------------------------------------------------------------------- */
int bsfm_device_count(void);                 /* 0 when no usable HIP device */
const char *bsfm_version(void);
int bsfm_device_synchronize(void);           /* hipDeviceSynchronize on the current device */
/* Deterministic synthetic BA scene: ring of m cameras, n points, `deg` views per point.
 * Fills rowptr(n+1), colidx(n*deg), projections(2*n*deg), cameras(m), points(3n) (already perturbed),
 * banded != 0 draws each point's cameras from a window of 50 neighbours. */
int bsfm_synth_ba(int m, int n, int deg, unsigned long long seed, int banded,
                  int *rowptr, int *colidx, double *projections,
                  bsfm_camera_params_t *cameras, double *points);
/* Deterministic synthetic SIFT-like descriptors: num x 128 uchar; `dup_from` (may be NULL, n_from keys)
 * donates ~20 % near-duplicates so that true matches exist. */
int bsfm_synth_keys(int num, unsigned long long seed, const unsigned char *dup_from, int n_from,
                    unsigned char *keys_out);

#ifdef __cplusplus
}
#endif
#endif /* BSFM_H */

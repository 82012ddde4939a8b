/*
 * splpak_hip.h -- C ABI of the MI355X (gfx950) implementation of SPLPAK's
 * least-squares cubic-spline fit / evaluate hot path.
 *
 * This is the drop-in boundary: a Fortran `splpak_module` (splpak_amd/fortran/
 * splpak_module.F90) binds these symbols with ISO_C_BINDING and keeps the
 * reference's public surface (`splpak_type%initialize/evaluate/destroy`,
 * `splpak_wp`).  Citations are into the reference, /root/reference/src/splpak.F90.
 *
 * Conventions
 *   - status return (`int32_t`) is the reference's `ierror`:
 *       0, 101..107 for the fit (:674-686), 0, 101..104 for evaluation (:1155-1161).
 *     Infrastructure failures (no GPU, HIP error, out of memory) are NEGATIVE
 *     (SPLPAK_E_*) and `splpak_last_error_message` explains them.  Nothing is
 *     printed by the library: the Fortran layer prints the reference's messages
 *     (cfaerr, :399-407) so stdout ordering matches.
 *   - `xdata` is the reference's column-major `xdata(l1xdat, ndata)` (:537-550),
 *     i.e. point i is the `ndim` doubles at `xdata + i*l1xdat`.
 *   - `coef` is ordered with the leftmost node index fastest (:657-673).
 *   - pointers are HOST pointers in the one-shot entry points and DEVICE pointers
 *     (same GPU as the plan) in the `_dev` / plan entry points.
 *   - there is NO CPU fallback: without a usable HIP device every compute entry
 *     point fails with SPLPAK_E_NODEVICE.
 */
#ifndef SPLPAK_HIP_H
#define SPLPAK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPLPAK_E_NODEVICE   (-1)  /* no HIP device / HIP runtime failure        */
#define SPLPAK_E_NOMEM      (-2)  /* device allocation failed                   */
#define SPLPAK_E_BADARG     (-3)  /* null pointer / inconsistent plan argument  */
#define SPLPAK_E_UNSUPPORTED (-4) /* ndim > 4 (the reference documents 1..4, :1099) */
#define SPLPAK_E_COMM       (-5)  /* the all-reduce callback reported failure   */

#define SPLPAK_MAXDIM 4

/* ---------------------------------------------------------------------------
 * One-shot host entry points -- what `splpak_module` binds.
 * ------------------------------------------------------------------------- */

/* Replaces splcw (:512-513) and, with wdata == NULL, splcc (:421-422; the
 * reference passes the sentinel wdata=[-1], :440).  A non-NULL wdata whose first
 * element is negative is treated like NULL (:581-588, :796).
 * `nwrk` is only used for the reference's 106 check (:772-781); the GPU path does
 * not use the caller's `work` array, except that when `hist_out` != NULL and
 * xtrap != 0 it receives the sparse-area histogram the reference leaves in
 * work(1:ncol) (:879-907).  `info` (optional, 10 doubles) receives diagnostics:
 *   [0] data rows used, [1] constraint rows, [2] refinement steps taken,
 *   [3] |last correction|_inf / |coef|_inf, [4] min Cholesky pivot (0 when no factorisation ran: the iterative solve answered),
 *   [5] seconds in assembly, [6] seconds in factorisation, [7] seconds in solve+refine (a 4-D plan with the iterative solve in
 *       front of a factorisation assembles the normal equations only when the factorisation is going to run: that assembly,
 *       and an attempt of the iteration that gave up, are then counted in [6]),
 *   [8] residual norm ||rows*coef - rhs||_2 over data and constraint rows -- the `reserr` that the
 *       reference computes (suprls :1693) and drops (splcw :690, :1052),
 *   [9] measured optimality residual of the returned coefficients: the gradient
 *       rho = A^T W (W y - W A x) - C^T C x of the least-squares functional (data rows A, constraint rows C),
 *       recomputed from the rows, as the componentwise backward error
 *       max_i |rho_i| / ((|N| |x|)_i + |A^T W^2 y|_i), N = A^T W^2 A + C^T C; 0 at the minimiser the reference
 *       computes, ~1e-14 when the fit is converged, ~cond(N) eps for plain normal equations.
 * The band-Cholesky solution is refined against the rows until the estimated remaining error
 * |dx|/|x| is below 1e-11 (2 steps at 64^3: corrections 9e-5, 1e-8, then an estimated 2e-12); a solve that is still contracting after the nominal
 * number of steps continues (up to 30), and one that then still misses 1e-10, or whose corrections
 * stop contracting while above 1e-8, is reported as 107 ("suprls failure") with an explanatory
 * splpak_last_error_message -- never as a silent success. */
int32_t splpak_fit_f64(int32_t ndim, const double *xdata, int32_t l1xdat,
                       const double *ydata, const double *wdata, int64_t ndata,
                       const double *xmin, const double *xmax, const int32_t *nodes,
                       double xtrap, double *coef, int64_t ncf, int64_t nwrk,
                       double *hist_out, double *info);

/* real32 twin (reference built with -DREAL32, :33-34).  Storage is f32, the
 * arithmetic is f64 (the arrays are widened when they are staged for upload), so it
 * is at least as accurate as the REAL32 reference. */
int32_t splpak_fit_f32(int32_t ndim, const float *xdata, int32_t l1xdat,
                       const float *ydata, const float *wdata, int64_t ndata,
                       const float *xmin, const float *xmax, const int32_t *nodes,
                       float xtrap, float *coef, int64_t ncf, int64_t nwrk,
                       float *hist_out, double *info);

/* Replaces a loop of splde (:1089) calls; nderiv == NULL gives splfe (:1258).
 * Query i is the `ndim` doubles at xq + i*ldxq (ldxq >= ndim, else SPLPAK_E_BADARG).  Error semantics per query are
 * the reference's: 101/102/103 return without computing (out is set to 0),
 * 104 (nderiv outside 0..2) is reported but the values are still computed with
 * nderiv clamped to 0..2 (the reference computes on, :1190-1194). */
int32_t splpak_eval_f64(int32_t ndim, int64_t nq, const double *xq, int32_t ldxq,
                        const int32_t *nderiv, const double *coef,
                        const double *xmin, const double *xmax, const int32_t *nodes,
                        double *out);
int32_t splpak_eval_f32(int32_t ndim, int64_t nq, const float *xq, int32_t ldxq,
                        const int32_t *nderiv, const float *coef,
                        const float *xmin, const float *xmax, const int32_t *nodes,
                        float *out);

/* ---------------------------------------------------------------------------
 * Resident-data (device pointer) API: plans, used by bench.py, by batched
 * callers and by the multi-GPU fit.  All work is enqueued on `stream`
 * (a hipStream_t passed as void*, NULL = the default stream).
 * ------------------------------------------------------------------------- */

typedef struct splpak_plan splpak_plan;

/* Sum-all-reduce hook for the sharded fit.  Called by
 * splpak_plan_fit_dev with a device pointer; must sum `count` doubles in place
 * across all ranks ON `stream` (or synchronise itself) and return 0.  With
 * torch.distributed/RCCL this is `all_reduce(tensor_view)`.  NULL => single rank.
 * The pointer lies in the plan's communication buffer for the histogram, the normal
 * equations and the refinement residuals.  A hook installed with
 * splpak_plan_set_allreduce_ex(.., SPLPAK_AR_ANY_POINTER) is also handed front panels,
 * Schur buffers and solve vectors that live in the library's own device allocations
 * (the nested-dissection factorisation distributed by subtrees, csrc/ndchol.hip). */
typedef int32_t (*splpak_allreduce_fn)(void *dev_buf, int64_t count, void *stream, void *user);

/* Validates exactly like splcw (:716-781; 105/106 are checked at fit time) and
 * allocates every device buffer the fit of a grid needs (band factor, stencil
 * normal equations, sort scratch for up to `max_ndata` points per call).
 * `max_ndata` (points per call on THIS GPU) is limited to 2^31 - 1025 (32-bit binning offsets);
 * larger values return SPLPAK_E_UNSUPPORTED.
 * `comm_buf_dev`/`comm_len`: optional caller-owned device buffer (doubles) the
 * all-reduced quantities live in -- pass a torch tensor's data_ptr so the
 * callback can all-reduce views of it; NULL lets the plan allocate it.  It must be device memory
 * the kernels can read and write at full speed (hipMalloc / a torch CUDA tensor): the assembly writes
 * the half stencil, right-hand side and histogram into it with plain stores (every sum has one owner and
 * a fixed order since round 2; the only floating-point atomic left is the histogram bump of a point so far
 * outside the grid that its nearest-node address is not a node of its window, src/splpak.F90:899).
 * `splpak_plan_comm_len` tells the required length. */
int64_t splpak_plan_comm_len(int32_t ndim, const int32_t *nodes);
int32_t splpak_plan_create(int32_t ndim, const int32_t *nodes, const double *xmin,
                           const double *xmax, double xtrap, int64_t max_ndata,
                           void *comm_buf_dev, int64_t comm_len, splpak_plan **plan);
void    splpak_plan_destroy(splpak_plan *plan);
void    splpak_plan_set_allreduce(splpak_plan *plan, splpak_allreduce_fn fn, void *user,
                                  int32_t rank, int32_t world);
/* The same with the hook's capabilities declared (round 4).  flags = 0 is splpak_plan_set_allreduce: the hook is only
 * ever called with windows of the plan's communication buffer (histogram, normal equations, refinement residuals) and
 * the factorisation is replicated on every rank, as in rounds 1-2.  SPLPAK_AR_ANY_POINTER: the hook accepts ANY device
 * pointer of the calling process (ncclAllReduce does; splpak_plan_set_rccl installs such a hook; the Python shim wraps
 * foreign pointers through the CUDA array interface) -- the nested-dissection factorisation of a sharded fit is then
 * distributed by subtrees (csrc/ndchol.hip; SPLPAK_ND_DIST=0 turns that off) and sums front panels, Schur buffers and
 * solve vectors that live in the library's own allocations.  Returns 0 or a negative status (a failure while the
 * per-rank job tables are rebuilt; the plan's next fit returns it on every rank). */
#define SPLPAK_AR_ANY_POINTER 1
#define SPLPAK_AR_ALWAYS      2   /* call the hook with one rank too (smoke tests of a one-rank communicator) */
#define SPLPAK_AR_STREAM_ORDERED 4 /* the hook only ENQUEUES its work on the stream it is handed (ncclAllReduce): the library then
                                     does not synchronise the stream before and after the call */
int32_t splpak_plan_set_allreduce_ex(splpak_plan *plan, splpak_allreduce_fn fn, void *user,
                                     int32_t rank, int32_t world, int32_t flags);
/* tuning / test knobs: nominal refinement steps (default 4; 0 = none; a solve that still contracts goes on
 * up to max(steps, 30)) and the tolerance on the (estimated) remaining relative error |dx|/|x| after a
 * step (default 1e-11; the parity bar is 1e-10) */
/* A native RCCL hook (round 4): the plan's sum-all-reduce becomes ncclAllReduce(.., ncclDouble, ncclSum, comm, stream) on the
 * fit's stream, with SPLPAK_AR_ANY_POINTER declared -- what a C / Fortran process-per-GPU caller uses instead of the Python
 * callback.  `nccl_comm` is the caller's ncclComm_t (one per process, made on the GPU the plan lives on).  librccl is opened at
 * run time (an RCCL the process already carries is reused; SPLPAK_RCCL_LIB names another); SPLPAK_E_COMM if it cannot be.
 * For callers without RCCL headers the communicator can be made here: rank 0 draws the 128-byte ncclUniqueId
 * (splpak_rccl_unique_id) and passes it to the other processes by any means -- splpak_rccl_comm_create_from_file[_ex] does it
 * through a file (no MPI needed on one node) -- and every process calls splpak_rccl_comm_create with the CURRENT device set
 * to its GPU.  The id file (round 5): { "SPLPAKID", job tag, publish time, id } = 152 bytes.  Rank 0 removes whatever lies at
 * `path`, writes its file atomically and removes it again once ncclCommInitRank has returned (collective: every rank has the
 * id by then); the others wait up to timeout_s seconds (<= 0: 60) for a file that carries THEIR job's tag and is no older than
 * that wait -- a file left by another run is ignored (a stale id would hang ncclCommInitRank), and the call times out with
 * SPLPAK_E_COMM.  `job` names the run (any string all ranks of ONE run share and other runs do not); NULL / "" = the environment
 * variable SPLPAK_RCCL_JOB, else what the launcher exports (TORCHELASTIC_RUN_ID, MASTER_ADDR, MASTER_PORT, SLURM_JOB_ID, ...). */
int32_t splpak_plan_set_rccl(splpak_plan *plan, void *nccl_comm, int32_t rank, int32_t world);
int32_t splpak_rccl_unique_id(char *id128);
int32_t splpak_rccl_comm_create(const char *id128, int32_t rank, int32_t world, void **nccl_comm);
int32_t splpak_rccl_comm_create_from_file(const char *path, int32_t rank, int32_t world, double timeout_s, void **nccl_comm);
int32_t splpak_rccl_comm_create_from_file_ex(const char *path, const char *job, int32_t rank, int32_t world, double timeout_s, void **nccl_comm);
void    splpak_rccl_comm_destroy(void *nccl_comm);
void    splpak_plan_set_refine(splpak_plan *plan, int32_t max_steps, double tol);

/* The fit on resident data.  xdata_dev/ydata_dev/wdata_dev (wdata_dev may be
 * NULL) hold THIS rank's `ndata` points (with more than one rank a rank may hold none: ndata = 0,
 * pointers ignored); coef_dev receives ncol coefficients (identical on every rank).  A rank whose
 * arguments are rejected still takes part in the first reduction, which carries an error flag: that
 * rank returns its status, the others SPLPAK_E_COMM -- nobody is left waiting in a collective.  Synchronises `stream` before returning (the error
 * flag and the refinement's convergence test are read back).  info as above. */
int32_t splpak_plan_fit_dev(splpak_plan *plan, const double *xdata_dev, int32_t l1xdat,
                            const double *ydata_dev, const double *wdata_dev,
                            int64_t ndata, double *coef_dev, void *stream, double *info);
/* device pointer to the (all-reduced) sparse-area histogram of the last fit */
const double *splpak_plan_hist_dev(const splpak_plan *plan);
/* device memory the plan holds (bytes) */
int64_t splpak_plan_device_bytes(const splpak_plan *plan);
/* Which factorisation of the normal equations the plan uses in place of suprls' triangularisation
 * (src/splpak.F90:1516-1619): returns 0 band Cholesky (four-stream pipeline), 1 its narrow form, 2 two-ended band,
 * 3 band distributed over several GPUs, 4 nested-dissection multifrontal (2-D / 3-D grids of >= 4 096 columns, 4-D grids of
 * >= 20 000), 5 the same distributed over the GPUs of a one-process multi-GPU plan (subtrees per GPU, the fronts above them by
 * block columns; the description names the elimination schedule's cut -- chosen from the free device memory unless the option
 * nd_cut fixes it -- and run-to-run bit reproducibility of a large grid's coefficients is promised for the SAME cut: the order in
 * which two siblings with different numbers of block steps add into their parent follows the schedule), 6 NO factorisation: the iterative solve below (grids whose factor does not fit the device, or by request);
 * a description is copied into buf. */
int32_t splpak_plan_factorisation(const splpak_plan *plan, char *buf, int32_t buflen);
/* Options (round 6).  Every switch of the library is a named option; a plan takes a snapshot of them when it is created --
 * the process defaults set here over the SPLPAK_<NAME> variables of the environment -- and a fit reads nothing else (no getenv
 * on the fit path; the one-shot entry's cached plan is keyed by the snapshot).  `name` is "nd_kb", "ND_KB" or "SPLPAK_ND_KB";
 * `value` is the text the environment variable would hold, NULL removes the setting.  Returns 0, SPLPAK_E_BADARG for an
 * unknown name, and (plan_set_option) SPLPAK_E_UNSUPPORTED for an option that shapes the plan's storage or job tables and is
 * therefore consumed at creation (solver, nd, nd_split, nd_cut, nd_kb, nd_res_cus, no_reorder, gram_scratch_mb, pcg_maxit,
 * mplan_rccl, rccl_lib): set those as defaults before splpak_plan_create.  The documented options are listed in
 * INTEGRATION.md; the other names are A/B switches of the test suite and may disappear.
 * splpak_plan_get_option: 1 if set (value copied into buf), 0 if not. */
int32_t splpak_set_default_option(const char *name, const char *value);
int32_t splpak_plan_set_option(splpak_plan *plan, const char *name, const char *value);
int32_t splpak_plan_get_option(const splpak_plan *plan, const char *name, char *buf, int32_t buflen);

/* The iterative solve (round 6; csrc/pcg.hip): preconditioned conjugate gradients on the normal equations with the operator
 * applied from the rows (data rows :788-855, constraint rows :921-1046) and a separable preconditioner, inside the same
 * refinement against the rows as the factorisations.  It replaces suprls (:1375-1695) for the grids the reference accepts
 * (any grid, :512-534) and no factorisation fits: BASELINE config 5's 4-D 32^4 grid on one GPU.  Selected automatically when
 * the factor storage cannot be allocated and, in front of the factorisation, for 4-D grids of 160 000 columns or more; or with
 * the option solver = pcg (iteration only) / pcg+direct (iteration first, the factorisation when it stagnates) / direct.  A fit whose iteration stagnates and whose plan
 * has no factorisation returns 107 with an explanatory message.
 * out6: [0] iterations of the last fit over all its solves, [1] solves, [2] iterations of the last solve, [3] its final
 * preconditioned residual (relative), [4] [5] the preconditioner's two moments (density of w^2, mean squared constraint weight);
 * zeros for a plan without the iteration. */
void splpak_plan_pcg_stats(const splpak_plan *plan, double *out6);

/* Per-kernel accounting of the last splpak_plan_fit_dev call, measured with HIP
 * events on the stream the kernels ran on (bench.py's roofline object).  Every BULK
 * trailing-update launch carries a start/stop event pair in its dispatch (hipExtLaunchKernelGGL).
 * "Bulk" is, for the nested-dissection factorisation, every Schur-buffer pass of the fronts
 * (nd_syrk_kernel<4,2,true,..>, f64 MFMA, K up to 1024 per pass: ~73 % of the factorisation's flops at
 * 64^3; the panel updates of the chain are the instantiation <..,false,..>); for the band factorisation
 * the one launch per block step that updates the block columns beyond the next one (syrk64_kernel,
 * ~96 % of the flops):
 *   out[0] = number of timed bulk launches
 *   out[1] = total milliseconds in them
 *   out[2] = floating-point operations they performed (algorithmic: 2*64*64*K per 64x64 item)
 *   out[3] = milliseconds in the whole factorisation
 *   out[4] = floating-point operations of ALL trailing-update launches
 *   out[5] = number of bulk launches, out[6] = their floating-point operations
 * Timing is only collected when enabled. */
void    splpak_plan_enable_kernel_timing(splpak_plan *plan, int32_t on);
void    splpak_plan_kernel_timing(const splpak_plan *plan, double *out7);
/* milliseconds of the stages of the last fit around the factorisation (HIP events on the fit's stream, collected
 * with kernel timing enabled): out[0] binning (keys, scan, scatter, in-cell ordering), [1] Gram blocks + gather,
 * [2] constraint rows, [3] band memset + expansion, [4] one refinement-residual pass, [5] one solve (two sweeps) */
void    splpak_plan_stage_timing(const splpak_plan *plan, double *out6);

/* ---------------------------------------------------------------------------
 * Several GPUs of one node, driven from ONE process: what a Fortran caller reaches
 * through `splpak_type%set_gpus(n)`.  The points are sharded over the GPUs and the factorisation of the
 * normal equations is DISTRIBUTED WITH ITS MEMORY PARTITIONED:
 *   - grids that take the nested-dissection factorisation (2-D / 3-D grids of >= 4 096 columns, 4-D grids of
 *     >= 20 000; since round 4): the subtrees below tree depth ceil(log2 ngpus) are dealt to the GPUs, the
 *     fronts above them are cut into 256-column blocks dealt to the GPUs in chunks of `chunk` blocks; a solved
 *     panel is copied GPU-to-GPU (hipMemcpyPeerAsync over xGMI) by the owners of the columns it updates, a
 *     child's Schur complement is pulled by the owners of the parent's columns through the peer mapping
 *     (csrc/ndtop.inc).  64^3: 4.4-4.6 GB of factorisation per GPU on eight instead of 30 GB; the 4-D 32^4 grid
 *     of BASELINE config 5: 159-190 GB per GPU on eight (476 GB of panels do not fit one GPU).  This form needs
 *     peer access between every pair of distinct devices (kernels read the other GPUs' memory): without it the
 *     plan falls back to the distributed band below, or returns SPLPAK_E_UNSUPPORTED when that cannot hold the grid;
 *   - smaller grids (or SPLPAK_MPLAN_BAND=1): the BAND of the normal equations dealt by block columns (round 2),
 *     every GPU stores and updates only its own block columns, the solved panel of every block step travels
 *     GPU-to-GPU, the triangular sweeps hand the active window from owner to owner (copies only: no peer mapping needed).
 * The sums over the ranks (histogram, normal equations, residuals) are reduce-scatter + all-gather over point-to-point
 * copies in rank order (bitwise reproducible), or -- option mplan_rccl, round 5 -- one ncclAllReduce per rank thread (no group call) of a
 * communicator the plan makes over its devices (ncclCommInitAll; distinct devices only).
 * Results are those of the single-GPU fit (same kernels per tile: with all points on rank 0 the coefficients are
 * bit-identical to it).
 * `chunk` < 1 chooses it automatically (nested dissection: 1; band: about one chunk per GPU inside the band window, at most 8).
 * `devices`: NULL = devices 0..ngpus-1; entries may repeat -- with SPLPAK_VIRTUAL_GPUS=1 in the
 * environment every rank is placed on the current device, which runs the whole protocol on one GPU
 * (the 1-GPU test tier does that).
 * ------------------------------------------------------------------------- */
typedef struct splpak_mplan splpak_mplan;
int32_t splpak_mplan_create(int32_t ngpus, const int32_t *devices, int32_t chunk, int32_t ndim,
                            const int32_t *nodes, const double *xmin, const double *xmax, double xtrap,
                            int64_t max_ndata_per_gpu, splpak_mplan **mplan);
void    splpak_mplan_destroy(splpak_mplan *mplan);
/* device of a rank (for placing its shard) */
int32_t splpak_mplan_device(const splpak_mplan *mplan, int32_t rank);
/* which factorisation the plan's ranks use (codes of splpak_plan_factorisation: 3 distributed band, 5 distributed nested dissection) */
int32_t splpak_mplan_factorisation(const splpak_mplan *mplan, char *buf, int32_t buflen);
/* device memory a rank of the plan holds (bytes): its plan (binning scratch, normal equations, its part of the factor)
 * and its panel / staging buffers */
int64_t splpak_mplan_rank_bytes(const splpak_mplan *mplan, int32_t rank);
/* xdata_dev[r] / ydata_dev[r] / wdata_dev[r] (wdata_dev may be NULL = unweighted) are device pointers
 * on rank r's GPU holding ndata[r] points (0 allowed); coef_dev (ncol doubles) is on rank 0's GPU.
 * Blocks until the fit is complete.  Status and `info` as splpak_plan_fit_dev. */
int32_t splpak_mplan_fit_dev(splpak_mplan *mplan, const double *const *xdata_dev, int32_t l1xdat,
                             const double *const *ydata_dev, const double *const *wdata_dev,
                             const int64_t *ndata, double *coef_dev, double *info);
/* one-shot host entry: as splpak_fit_f64 on `ngpus` GPUs (contiguous shards of the points); ngpus <= 1
 * is splpak_fit_f64 itself.  SPLPAK_DIST_CHUNK sets the chunk (default 0 = automatic). */
int32_t splpak_fit_multi_f64(int32_t ngpus, int32_t ndim, const double *xdata, int32_t l1xdat,
                             const double *ydata, const double *wdata, int64_t ndata,
                             const double *xmin, const double *xmax, const int32_t *nodes,
                             double xtrap, double *coef, int64_t ncf, int64_t nwrk,
                             double *hist_out, double *info);

/* Batched evaluation on resident data (asynchronous on `stream`; no validation
 * beyond the reference's 101..104, which is done on the host from the small
 * arguments). */
int32_t splpak_eval_dev_f64(int32_t ndim, int64_t nq, const double *xq_dev, int32_t ldxq,
                            const int32_t *nderiv /* host, may be NULL */,
                            const double *coef_dev, const double *xmin, const double *xmax,
                            const int32_t *nodes, double *out_dev, void *stream);

/* The same for REAL32 storage (queries, coefficients and results in single precision, arithmetic in double:
 * the batched counterpart of the -DREAL32 build of splfe / splde, src/splpak.F90:33-41). */
int32_t splpak_eval_dev_f32(int32_t ndim, int64_t nq, const float *xq_dev, int32_t ldxq,
                            const int32_t *nderiv /* host, may be NULL */,
                            const float *coef_dev, const float *xmin, const float *xmax,
                            const int32_t *nodes, float *out_dev, void *stream);

/* Value, gradient and (order 2) Hessian of the spline at a batch of points in one pass over the
 * window -- SURVEY 8f: what a caller otherwise obtains from 1 + ndim (+ ndim(ndim+1)/2) splde calls
 * (:1089-1240) per point.  order = 1 or 2.  Row i of `out` (ldout apart, ldout >= number of entries):
 *   [ f, df/dx_1 .. df/dx_ndim, (order 2:) d2f/dx_1dx_1, d2f/dx_1dx_2, .., d2f/dx_1dx_ndim, d2f/dx_2dx_2, .. ]
 * i.e. the upper triangle of the Hessian row by row.  Each entry equals splde with the matching nderiv.
 * Status: 0, 101/102/103 as splde (`out` is zeroed), or a negative library status. */
int32_t splpak_eval_derivs_f64(int32_t ndim, int64_t nq, const double *xq, int32_t ldxq, int32_t order,
                               const double *coef, const double *xmin, const double *xmax,
                               const int32_t *nodes, double *out, int32_t ldout);
int32_t splpak_eval_derivs_f32(int32_t ndim, int64_t nq, const float *xq, int32_t ldxq, int32_t order,
                               const float *coef, const float *xmin, const float *xmax,
                               const int32_t *nodes, float *out, int32_t ldout);
/* the same on resident data (asynchronous on `stream`) */
int32_t splpak_eval_derivs_dev_f64(int32_t ndim, int64_t nq, const double *xq_dev, int32_t ldxq, int32_t order,
                                   const double *coef_dev, const double *xmin, const double *xmax,
                                   const int32_t *nodes, double *out_dev, int32_t ldout, void *stream);

/* Device-side synthetic inputs of SURVEY 8d (Park-Miller stream, seed 42):
 * points first_point .. first_point+ndata-1; any of the outputs may be NULL.
 * xdata_dev is written with leading dimension ndim.  Queries continue the stream
 * after `ndata_before` data points. */
int32_t splpak_synth_points_f64(int32_t ndim, int64_t first_point, int64_t ndata,
                                double *xdata_dev, double *ydata_dev, double *wdata_dev,
                                void *stream);
int32_t splpak_synth_queries_f64(int32_t ndim, int64_t ndata_before, int64_t first_query,
                                 int64_t nq, double *xq_dev, void *stream);

/* Diagnostics: solve A x = b for a symmetric positive definite band matrix with
 * the library's blocked band Cholesky (no refinement).  `a_lower` is the dense
 * column-major n x n matrix on the HOST, of which only the lower triangle within
 * `halfbw` of the diagonal is read.  Returns 0, or 107 if a pivot is not
 * positive.  Exists so the factorisation kernels (f64 MFMA trailing update, panel
 * solve, band sweeps) can be tested in isolation against LAPACK. */
int32_t splpak_debug_spd_band_solve_f64(int32_t n, int32_t halfbw, const double *a_lower,
                                        const double *b, double *x);

/* Diagnostics (host only, no device needed): builds the nested-dissection elimination tree the fit uses for
 * large 3-D / 4-D grids (csrc/ndtree.hpp; separators 3 nodes thick because the normal equations of the
 * window rule src/splpak.F90:821-827 couple nodes up to 3 apart) and reports its size.  split_min <= 0: the
 * library default; check != 0: verify the tree's invariants (slow on big grids).  out16:
 *   [0] fronts  [1] depth  [2] bytes of factor panels  [3] bytes of the two Schur arenas  [4] bytes of all Schur
 *   buffers of one fit  [5] flop of the blocked factorisation (padded)  [6] flop without padding  [7] largest
 *   separator  [8] largest border  [9] 256x256 diagonal blocks  [10] local vector length  [11] own rows (padded)
 *   [12] border rows (padded)  [13] bytes of the diagonal-block inverses.
 * Returns 0, 101/102/103 (grid checks) or a negative SPLPAK_E_* code. */
int32_t splpak_debug_nd_tree(int32_t ndim, const int32_t *nodes, int32_t split_min, int32_t check, double *out16);

/* Diagnostics (host only): the elimination schedule of the nested-dissection factorisation (csrc/ndtree.hpp NdSchedule, round 5)
 * and the Schur-buffer arena it needs.  cut = 0: one stage per tree depth (rounds 3-4); cut > 0: the fronts above depth `cut` one
 * by one in postorder, the subtrees below it one after the other.  packed != 0: Schur buffers as packed lower triangles.  The
 * schedule's invariants are verified (children before parents, live buffers disjoint).  out8: [0] stages, [1] bytes of the
 * arena (peak of the allocation), [2] bytes of all Schur buffers of one fit, [3] bytes of the factor panels, [4] cut, [5] depth. */
int32_t splpak_debug_nd_schedule(int32_t ndim, const int32_t *nodes, int32_t split_min, int32_t cut, int32_t packed, double *out8);

/* Diagnostics (host only): how the nested-dissection factorisation of a grid is distributed over `ngpus` GPUs by the
 * one-process multi-GPU fit (splpak_mplan_*, splpak_fit_multi_f64; csrc/ndtree.hpp NdPartition): the subtrees below tree
 * depth ceil(log2 ngpus) belong to one rank each, the fronts above are distributed by block columns (chunk < 1: 1).
 * out_per_rank8 (8 doubles per rank): [0] bytes of everything the factorisation keeps on that GPU = [1] panels of its
 * subtrees + [2] its Schur arenas + [3] its block columns of the top fronts + [4] block inverses + [5] receive buffers,
 * solve vectors and index tables; [6] padded flop of its subtrees, [7] of its share of the top fronts.
 * out8 (optional): [0] depth of the cut, [1] top fronts, [2] their block steps, [3] bytes of the largest panel that
 * travels, [4] bytes of the all-reduced normal equations every rank also holds, [5] fronts, [6] depth, [7] total flop. */
int32_t splpak_debug_nd_partition(int32_t ndim, const int32_t *nodes, int32_t split_min, int32_t ngpus, int32_t chunk,
                                  double *out_per_rank8, double *out8);

/* Diagnostics (host only): the four basis values of the window of x (1-D grid of `nodes` nodes on [xmin, xmax];
 * src/splpak.F90:206-389, window rule :1201-1209) for n points, (a) as the evaluation kernels compute them -- the closed
 * form of an interior window, the closed form with the end functions put in next to an end of the grid, the general
 * form elsewhere (csrc/basis.hpp; form_out: 0 / 1 / 2) -- and (b) in the general form throughout.  ws_out: first node of
 * the window.  used4 / general4: 4 values per point.  Returns 0, 102/103 (grid checks) or a negative SPLPAK_E_* code. */
int32_t splpak_debug_window_values(int32_t nodes, double xmin, double xmax, int64_t n, const double *x, int32_t *ws_out,
                                   double *used4, double *general4, int32_t *form_out);

/* Releases the calling thread's internal HIP streams, events, queues and evaluation scratch
 * (created lazily and kept for reuse).  Optional; plans stay valid. */
void splpak_shutdown(void);

/* Evaluation strategy of the calling thread (results are bit-identical either way):
 *   mode 0  automatic: batches of >= 2^20 queries on 3-D and 4-D grids of more than 32768 nodes
 *           take the binned path, everything else the direct one
 *   mode 1  direct: one thread per query, coefficients gathered from global memory
 *   mode 2  binned: queries are sorted by grid region, `chunk` queries at a time (0 = 2^24), and
 *           each region is evaluated from a copy of its coefficients in LDS; needs
 *           chunk*(8*ndim+4) bytes of device scratch, kept until splpak_shutdown
 * There is no counterpart in the reference (splde evaluates one point per call, :1089-1240). */
int32_t splpak_set_eval_mode(int32_t mode, int64_t chunk);

/* Human-readable text for the last negative status on this thread. */
int32_t splpak_last_error_message(char *buf, int32_t buflen);

/* Library / device identification: writes e.g. "gfx950:sramecc+:xnack-" */
int32_t splpak_device_name(char *buf, int32_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* SPLPAK_HIP_H */

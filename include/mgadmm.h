/*
 * mgadmm.h -- C ABI of the MI355X-native mixed-graph ADMM hot path (libmgadmm.so).
 *
 * The reference (JiQi-da/Mixed-Graph-ADMM) has no FFI: its boundary is the Python class
 * ADMM_algorithm (ADMM.py:11-648).  Every entry point below names the reference method it
 * replaces.  All signal tensors cross the ABI as plain device pointers in the reference's own
 * layout: contiguous (B, T, N) arrays of `dtype` (C channels are folded into N by the host, see
 * INTEGRATION.md), on the device the graph was created on.  No torch types appear here.
 *
 * Conventions
 *   - every function returns MGADMM_OK (0) or a negative mgadmm_status; no exceptions cross the
 *     ABI; mgadmm_last_error() returns a thread-local message for the last failure.
 *   - handles are opaque; one solver handle per GPU; a handle is not thread-safe, distinct
 *     handles are independent.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Fine-grained entry
 *     points are asynchronous on that stream unless stated otherwise; mgadmm_cg and mgadmm_solve
 *     return after their work has completed (they hand results back in host memory).
 *   - inputs are never modified; outputs are caller-owned buffers.
 */
#ifndef MGADMM_H
#define MGADMM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    MGADMM_OK = 0,
    MGADMM_ERR_INVALID = -1,      /* bad argument / shape / enum                      */
    MGADMM_ERR_HIP = -2,          /* a HIP runtime call failed                        */
    MGADMM_ERR_NONFINITE = -3,    /* NaN/Inf met in the iterates (ADMM.py:534-606 asserts) */
    MGADMM_ERR_UNSUPPORTED = -4,  /* configuration outside what this build implements */
    MGADMM_ERR_NOMEM = -5
} mgadmm_status;

typedef enum { MGADMM_F32 = 0, MGADMM_F64 = 1 } mgadmm_dtype_t;

/* temporal (directed) graph kind: ADMM.py:37-52 */
typedef enum {
    MGADMM_TEMPORAL_SPATIAL = 0,  /* kNN / physical: (t-1, j) -> (t, i) edges with weights W_d      */
    MGADMM_TEMPORAL_BAND = 1      /* use_line_graph: (t-1-s, i) -> (t, i), s < skip_connection       */
} mgadmm_temporal_mode_t;

/* apply_op_Lu / apply_op_Ldr / apply_op_Ldr_T / apply_op_cLdr : ADMM.py:138-228 */
/* MGADMM_OP_LN: apply_op_Ln, the undirected temporal Laplacian (ADMM.py:248-288; unreachable from the reference's solver) */
typedef enum { MGADMM_OP_LU = 0, MGADMM_OP_LDR = 1, MGADMM_OP_LDRT = 2, MGADMM_OP_CLDR = 3, MGADMM_OP_LN = 4 } mgadmm_op_t;
/* LHS_x / LHS_zu / LHS_zd : ADMM.py:371-399 */
typedef enum { MGADMM_LHS_X = 0, MGADMM_LHS_ZU = 1, MGADMM_LHS_ZD = 2 } mgadmm_lhs_t;
/* ablation strings of ADMM.py:31-32, in this order: 'None', 'DGTV', 'DGLR', 'UT' */
typedef enum { MGADMM_ABL_NONE = 0, MGADMM_ABL_DGTV = 1, MGADMM_ABL_DGLR = 2, MGADMM_ABL_UT = 3 } mgadmm_ablation_t;
typedef enum { MGADMM_PATH_AUTO = 0, MGADMM_PATH_STREAM = 1, MGADMM_PATH_LDS = 2 } mgadmm_path_t;
/* When a batched CG solve stops.
 *   PER_SAMPLE  every sample stops on its own residual: B samples = B independent B=1 reference runs (default; the only
 *               batched semantics the reference can be run for, since it crashes for B > 1, ADMM.py:362)
 *   BATCH_MAX   the reference's literal test sqrt(rr).max() < CG_tol (ADMM.py:360): every sample keeps iterating until the
 *               largest residual of the batch is below the tolerance; one iteration count per solve.  Streaming path only. */
typedef enum { MGADMM_CG_PER_SAMPLE = 0, MGADMM_CG_BATCH_MAX = 1 } mgadmm_cg_convergence_t;

/* Number of per-iteration scalar diagnostics (order below) : ADMM.py:609-637 */
#define MGADMM_NMETRIC 11
enum {
    MGADMM_M_XSHIFT = 0,   /* ||x - x_old||           x_shift_list                 */
    MGADMM_M_PRI_ZU = 1,   /* ||x - zu||              p_res_list[.][0]             */
    MGADMM_M_DUAL_ZU = 2,  /* ||zu - zu_old||         d_res_list[.][0]             */
    MGADMM_M_PRI_PHI = 3,  /* ||phi - Ldr x||         p_res_list 'phi'             */
    MGADMM_M_DUAL_PHI = 4, /* ||phi - phi_old||       d_res_list 'phi'             */
    MGADMM_M_PRI_ZD = 5,   /* ||x - zd||                                           */
    MGADMM_M_DUAL_ZD = 6,  /* ||zd - zd_old||                                      */
    MGADMM_M_GLR = 7,      /* mean_b sum x.Lu x       GLR_list   (ADMM.py:245-246) */
    MGADMM_M_DGTV = 8,     /* mean_b ||Ldr x||_1      DGTV_list  (ADMM.py:238-243) */
    MGADMM_M_DGLR = 9,     /* mean_b sum (Ldr x)^2    DGLR_list  (ADMM.py:230-235) */
    MGADMM_M_RECOVER = 10  /* ||Hx - y||              recover_list                 */
};

/* Graph description: host pointers, copied by mgadmm_graph_create.
 * CSR mapping of the reference's padded tables:
 *   W_u row i: columns connect_list[i,1..k] (!= -1) with u_ew[i,:]   (ADMM.py:143-148)
 *   W_d row i: columns connect_list[i,0..k] (!= -1) with d_ew[i,:]   (ADMM.py:166-177)
 * The exact transpose W_d^T (replacing the scatter_add of ADMM.py:200-209) is built inside. */
typedef struct {
    int32_t n_nodes;
    int32_t T;
    int32_t temporal_mode;        /* mgadmm_temporal_mode_t */
    const int32_t* u_rowptr;      /* n_nodes+1 */
    const int32_t* u_col;
    const float* u_val;
    const int32_t* d_rowptr;      /* SPATIAL mode only, else NULL */
    const int32_t* d_col;
    const float* d_val;
    int32_t transpose_by_gather;  /* 1: use_kNN=False branch, Ldr_T gathers with W_d itself (ADMM.py:210-215) */
    int32_t q1_identity_t0;       /* 1: reproduce ADMM.py:221-222 (Ldr_T keeps +I on the t=0 block) */
    int32_t skip;                 /* BAND mode: skip_connection */
    const float* band_w;          /* BAND mode: T*skip weights, row t = weights of x[t-1-s] (ADMM.py:43-49) */
    int32_t reorder;              /* internal node order: 0 keep, 1 RCM (bandwidth), 2 greedy cluster growth (locality for LDS tiles) */
    int32_t device;               /* HIP device ordinal */
} mgadmm_graph_desc;

/* ADMM_info + the mutable attributes of ADMM.py:59-80 */
typedef struct {
    double rho, rho_u, rho_d, mu_u, mu_d1, mu_d2;
    int32_t t_in;
    int32_t ablation;            /* mgadmm_ablation_t */
    double cg_tol;               /* CG_tol        = 1e-8 */
    int32_t max_cg_iter;         /* max_CG_iter   = 100  */
    double admm_tol;             /* ADMM_tol      = 1e-6 */
    int32_t max_admm_iter;       /* max_ADMM_iter = 150  */
    int32_t dtype;               /* mgadmm_dtype_t: arithmetic type of the kernels */
    int32_t check_stop;          /* 1: test ADMM_tol every iteration (ADMM.py:645-646); 0: run max_admm_iter */
    int32_t path;                /* mgadmm_path_t */
    int32_t record_cg_coeffs;    /* 1: keep alpha/beta of every CG iteration (alpha_x ... beta_zd lists) */
    int32_t cg_convergence;      /* mgadmm_cg_convergence_t */
    int32_t max_inner_iter;      /* max_inner_iter = 100: inner iterations of mgadmm_two_loops (ADMM.py:77, 448) */
} mgadmm_params;

/* Host buffers for the residual history; any pointer may be NULL.  Filled by mgadmm_solve. */
typedef struct {
    int32_t n_iters;             /* out: ADMM iterations executed                                   */
    double* metrics;             /* [max_admm_iter][MGADMM_NMETRIC] whole-batch values (norms, not squares) */
    double* delta_x_per_step;    /* [max_admm_iter][T]   ||mean_b(x-x_old)|| over nodes, per time step */
    int32_t* cg_iters;           /* [max_admm_iter][3][B] CG iterations of x, zu, zd per sample (-1 = hit max) */
    double* metrics_per_sample;  /* [max_admm_iter][MGADMM_NMETRIC][B] per-sample sums (squares for norms) */
    double* cg_alpha;            /* [max_admm_iter][3][max_cg_iter][B] (needs record_cg_coeffs), NaN past the end */
    double* cg_beta;             /* same shape */
} mgadmm_history;

/* Device tensors of the ADMM state (API layout (B,T,N), solver dtype).  As the `state_out` of a solve: optional outputs,
 * NULL = skip.  As the `state_in` of mgadmm_solve_from: every field the ablation uses is required (phi / gamma unless
 * 'DGTV'/'UT', zd / gamma_d unless 'DGLR'). */
typedef struct {
    void* zu; void* zd; void* phi; void* gamma; void* gamma_u; void* gamma_d;
} mgadmm_state;

typedef struct mgadmm_graph mgadmm_graph;
typedef struct mgadmm_solver mgadmm_solver;

/* "mgadmm <major>.<minor>.<patch> (gfx950)".  The minor number changes whenever a struct of this header grows (0.2:
 * mgadmm_params gained cg_convergence and max_inner_iter): a caller built against an older header must be rebuilt --
 * compare the string before passing structs. */
const char* mgadmm_version(void);
const char* mgadmm_last_error(void);

/* replaces the table set-up of ADMM_algorithm.__init__ (ADMM.py:25-52) on the device side */
int mgadmm_graph_create(const mgadmm_graph_desc* desc, mgadmm_graph** out);
int mgadmm_graph_destroy(mgadmm_graph* g);
/* test hook: copies the internally built transpose CSR back (row pointers n+1, then nnz entries) */
int mgadmm_graph_transpose_nnz(const mgadmm_graph* g, int32_t* nnz);
int mgadmm_graph_get_transpose(const mgadmm_graph* g, int32_t* rowptr, int32_t* col, float* val);
/* test hook: internal node order (perm[i] = API node stored at internal row i) */
int mgadmm_graph_get_perm(const mgadmm_graph* g, int32_t* perm);

/* allocates every workspace once for batches up to max_batch */
int mgadmm_solver_create(mgadmm_graph* g, const mgadmm_params* p, int32_t max_batch, mgadmm_solver** out);
int mgadmm_solver_destroy(mgadmm_solver* s);
/* attribute assignment after construction (blk.max_ADMM_iter = ..., blk.rho = ...) */
int mgadmm_solver_set_params(mgadmm_solver* s, const mgadmm_params* p);
/* bytes of device workspace held by the solver */
int64_t mgadmm_solver_workspace_bytes(const mgadmm_solver* s);
/* which path (MGADMM_PATH_STREAM / MGADMM_PATH_LDS) a batch of size B would take */
int mgadmm_solver_path(const mgadmm_solver* s, int32_t B);

/* Introspection of the execution plan a solver chose at create time (bench.py derives the LDS-traffic roofline of the
 * fused kernel from it; tests assert the production geometry).  Unknown `what` -> MGADMM_ERR_INVALID. */
typedef enum {
    MGADMM_Q_LDS_OK = 0,        /* 1 when the LDS-resident fused path is available for this graph / dtype          */
    MGADMM_Q_LDS_TPG = 1,       /* time steps per thread of k_admm_lds                                              */
    MGADMM_Q_LDS_THREADS = 2,   /* active threads per workgroup (= N * T / TPG)                                     */
    MGADMM_Q_LDS_BYTES = 3,     /* dynamic LDS bytes per workgroup                                                  */
    MGADMM_Q_LDS_ROW_STRIDE = 4,/* floats between two nodes' time rows in LDS                                       */
    MGADMM_Q_NNZ_U = 5,         /* stored entries of W_u, W_d, W_d^T                                                */
    MGADMM_Q_NNZ_D = 6,
    MGADMM_Q_NNZ_DT = 7,
    MGADMM_Q_TILE_ROWS = 8,     /* node rows per LDS tile of the streaming SpMM kernel (0: plain row kernel)        */
    MGADMM_Q_LDS_UNIFORM = 9,   /* 1: k_admm_lds instance with fixed-width table rows held in registers             */
    MGADMM_Q_LDS_TAIL_PAIRS = 10,/* pairs of W_d^T entries per row in the padded tail table of k_admm_lds            */
    MGADMM_Q_LDS_LEAD = 11,     /* leading W_d^T entries per row (held in registers / read first)                  */
    MGADMM_Q_LDS_SLOTS = 12,    /* 1: two LDS vectors park per-thread operands across the solves                    */
    MGADMM_Q_LDS_CHUNK = 13,    /* ADMM iterations per k_admm_lds launch when the iteration count is fixed          */
    MGADMM_Q_LDS_ROWS = 14,     /* LDS rows of an image: nodes + ghost rows                                         */
    MGADMM_Q_CLDR_SLOTS = 15    /* W_d^T entry slots per row of the fused Ldr^T Ldr kernel in use (12 / 16 / 24); 0: two-pass */
} mgadmm_query_t;
int mgadmm_solver_query(const mgadmm_solver* s, int32_t what, int64_t* out);

/* apply_op_Lu / apply_op_Ldr / apply_op_Ldr_T / apply_op_cLdr (ADMM.py:138-228): y = op(x) */
int mgadmm_apply(mgadmm_solver* s, int32_t op, const void* x, void* y, int32_t B, void* stream);
/* LHS_x(x, mask) / LHS_zu / LHS_zd (ADMM.py:371-399); mask may be NULL */
int mgadmm_lhs(mgadmm_solver* s, int32_t which, const void* x, const void* mask, void* y, int32_t B, void* stream);
/* phi_direct (ADMM.py:401-408) */
int mgadmm_phi_direct(mgadmm_solver* s, const void* x, const void* gamma, void* phi, int32_t B, void* stream);
/* initial_guess(y, t_in, T) (ADMM.py:766-781): y is (B, t_in, N), x is (B, T, N) */
int mgadmm_initial_guess(mgadmm_solver* s, const void* y, void* x, int32_t B, void* stream);
/* initial_interpolation(y, mask) (ADMM.py:783-811): y, mask, x are (B, T, N);
 * mask_is_f32 != 0 reproduces the float32 time moments the reference gets from a float32 mask */
int mgadmm_initial_interpolation(mgadmm_solver* s, const void* y, const void* mask, int32_t mask_is_f32,
                                 void* x, int32_t B, void* stream);
/* CG_solver(LHS_func, RHS, x0, mask=) (ADMM.py:329-368) with per-sample convergence.
 * x0 may be NULL (zero start).  iters: host int32[B] (-1 = not converged in max_cg_iter);
 * alpha/beta: host double[max_cg_iter*B] or NULL, NaN past a sample's last iteration.  Synchronous. */
int mgadmm_cg(mgadmm_solver* s, int32_t which, const void* rhs, const void* x0, const void* mask, void* x,
              int32_t* iters, double* alpha, double* beta, int32_t B, void* stream);
/* combined_loop(y, mask) (ADMM.py:511-648).  Prediction: mask NULL, y is (B, t_in, N).
 * Interpolation: y and mask are (B, T, N).  x_out is (B, T, N).  Synchronous.
 * x_out and the state_out vectors may be written at any time during the call (the LDS-resident path iterates in place in
 * them): they must not overlap y, mask or each other; state_in vectors may be the state_out vectors (resume in place). */
int mgadmm_solve(mgadmm_solver* s, const void* y, const void* mask, int32_t mask_is_f32, int32_t B, void* x_out,
                 const mgadmm_state* state_out, mgadmm_history* hist, void* stream);

/* Warm start / resume (checkpoint row): the loop of mgadmm_solve started from a saved state
 * (x0, state_in) instead of the initial guess and constants of ADMM.py:529-544.  k1 iterations, state_out -> state_in,
 * k2 more iterations reproduce k1 + k2 iterations of one solve bit for bit.  Synchronous. */
int mgadmm_solve_from(mgadmm_solver* s, const void* y, const void* mask, int32_t mask_is_f32, int32_t B, const void* x0,
                      const mgadmm_state* state_in, void* x_out, const mgadmm_state* state_out, mgadmm_history* hist,
                      void* stream);
/* two_loops(y, mask) (ADMM.py:410-508): max_admm_iter outer phi / gamma updates around max_inner_iter inner
 * (x, zu, zd, gamma_u, gamma_d) updates restarted from zu = zd = x, gamma_u = gamma_d = 0.1.  The reference returns
 * None and records only CG counts; here x_out / state_out receive the final state.  hist: n_iters = outer iterations,
 * cg_iters = [max_admm_iter * max_inner_iter][3][B]; the other history fields are not written.  Streaming path.
 * Synchronous. */
int mgadmm_two_loops(mgadmm_solver* s, const void* y, const void* mask, int32_t mask_is_f32, int32_t B, void* x_out,
                     const mgadmm_state* state_out, mgadmm_history* hist, void* stream);

/* ---- graph construction on the GPU; host buffers in and out, synchronous ------------------
 * k_nearest_neighbors(n_nodes, edges, dists, k) (utils.py:183-204): the k+1 nearest nodes of every node by
 * shortest-path distance over the directed edge list (self first), one truncated Dijkstra per node in networkx's
 * own visiting order (ties in distance resolve as in the reference).
 *   edges  int64 (n_edges, 2) row-major (from, to) as in graph_info['u_edges']; a repeated edge overwrites the length
 *   dists  float64 (n_edges)  non-negative edge lengths
 *   nn_out int32 (n_nodes, k+1) with -1 pads;  nd_out float32 (n_nodes, k+1) with +inf pads */
int mgadmm_knn_graph(int32_t n_nodes, int64_t n_edges, const int64_t* edges, const double* dists, int32_t k,
                     int32_t* nn_out, float* nd_out, int32_t device);
/* undirected_graph_from_distance / directed_graph_from_distance (utils.py:206-258) from the (n_nodes, k1) tables
 * connect_list (int64, column 0 = the node itself, -1 pads) and dist_list (float32, +inf pads).
 *   u_sigma, d_sigma  <= 0 selects the reference default max(dmax/50, dmin*50); the values used are returned
 *                     in sigmas_out[2] (may be NULL)
 *   u_ew  float32 (n_nodes, k1-1) or NULL;  d_ew  float32 (n_nodes, k1) or NULL;  regularized as in the reference */
int mgadmm_weight_tables(int32_t n_nodes, int32_t k1, const int64_t* connect_list, const float* dist_list,
                         double u_sigma, double d_sigma, int32_t regularized, float* u_ew, float* d_ew,
                         double* sigmas_out, int32_t device);

/* ---- data front end on the GPU: TrafficDataset (utils.py:54-134) ---------------------------
 * `series` is the (n_steps, n_cols) row-major device array data[t, node*C + c] of dtype MGADMM_F32/F64.
 * Per-column statistics over time (device outputs of n_cols elements in the series dtype, any may be NULL):
 * min / max (utils.py:87-88), mean and the unbiased standard deviation of torch.std (utils.py:83-84).
 * Fixed-order reductions (bitwise repeatable).  Asynchronous on `stream`. */
int mgadmm_series_stats(const void* series, int64_t n_steps, int32_t n_cols, int32_t dtype, void* o_min, void* o_max,
                        void* o_mean, void* o_std, void* stream);
/* In place.  inverse == 0: series = (series - shift[col]) / s[col]; inverse != 0 (recover_data, utils.py:110-118):
 * series = series * s[col] + shift[col]; s = scale - scale_lo when scale_lo != NULL ('normalize': shift = min,
 * scale = max, scale_lo = min), else s = scale ('standardize': shift = mean, scale = std).  Asynchronous. */
int mgadmm_series_affine(void* series, int64_t n_steps, int32_t n_cols, int32_t dtype, const void* shift, const void* scale,
                         const void* scale_lo, int32_t inverse, void* stream);
/* Sliding windows for a batch of start indices (get_predict_data / get_interpolated_data, utils.py:121-134):
 * out[b, t, col] = series[starts[b] + t, col] * (mask ? mask[t, col] : 1), t < win.  starts: device int64[B];
 * mask: device float32 (win, n_cols) or NULL; out: device (B, win, n_cols) of the series dtype.
 * Synchronous (reports a start index outside [0, n_steps - win]). */
int mgadmm_gather_windows(const void* series, int64_t n_steps, int32_t n_cols, int32_t dtype, const int64_t* starts, int32_t B,
                          int32_t win, const float* mask, void* out, void* stream);

/* Per-kernel HIP-event timing on the launch stream (used by bench.py for the roofline figure).
 * tag 0 = sparse-Laplacian SpMM inside CG (the dominant kernel), 1 = CG vector update,
 * 2 = other SpMM launches, 3 = everything else.  Speculative CG launches that found their solve converged and
 * returned at the guard (a few microseconds, no operand traffic) are excluded from counts, time and bytes. */
#define MGADMM_NPROF 4
int mgadmm_prof_begin(mgadmm_solver* s);
int mgadmm_prof_end(mgadmm_solver* s, int64_t* counts /*[MGADMM_NPROF]*/, double* total_ms /*[MGADMM_NPROF]*/,
                    double* bytes /*[MGADMM_NPROF] algorithmic bytes*/);

#ifdef __cplusplus
}
#endif
#endif /* MGADMM_H */

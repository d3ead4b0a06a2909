// Launch interface of the LDS-resident fused path (kernels in lds_kernels.h, compiled in lds_launch.hip).
#pragma once
#include <hip/hip_runtime.h>

// entries of a W_d^T row that k_admm_lds keeps in registers during a CG solve (the rest of the row: the padded tail table)
constexpr int LDS_NLEAD = 5;
// most ADMM iterations one k_admm_lds launch runs (LdsArgs::J; the x pointer table has one entry more)
constexpr int LDS_MAXJ = 16;

// scalar part of the launch arguments: a trip of the loop inside k_admm_lds reads it from the kernarg segment
struct LdsArgsCore {
    int T, N, TN, TS, t_in, G, B, Bp;   // TS: LDS row stride (floats) of a node's time row, >= T
    int nthreads;          // N * G threads own elements; the other threads of the workgroup are GHOSTS: they own an LDS row
                           // of zeros (rows N .. NR-1) and table rows of zero weights, and run the same instruction stream
    int NR;                // LDS rows of an image: N + ghosts
    int has_phi, has_zd, first;
    int J;                 // ADMM iterations this launch runs on every sample (trips of the loop inside k_admm_lds), 1 .. LDS_MAXJ
    int lhsx_kind;         // 1: LHS_x contains cLdr, 0: diagonal ('DGTV'/'UT')
    int band, skip, q1;
    int max_cg;
    int record;            // alpha/beta history
    int stagger_wgs, stagger_ticks;   // the first `stagger_wgs` workgroups start up to `stagger_ticks` (10 ns) late, see k_admm_lds
    int tail_pairs;        // W_d^T rows: entries beyond the LDS_NLEAD leading ones, padded to 2 * tail_pairs per row
    float rho, rho_u, rho_d, mu_u, mu_d1, mu_d2;
    float cx1, cx2;        // LHS_x = HtH + cx1*I + cx2*cLdr
    double cg_tol2;        // CG_tol squared: a solve stops when r.r < CG_tol^2 (ADMM.py:360 without the square root)
    // graph image (global), ints: [rp_u NR+1][rp_d NR+1][pad][ent_u][ent_d][lead_t NR*LDS_NLEAD][tail_t NR*2*tail_pairs][diag 2 NR][pad];
    // entries are {LDS float offset of the neighbour's row, weight}.  The part [lds_img0, lds_img0 + lds_img_ints) is copied
    // to LDS by every workgroup (all of it, or -- instances that read the fixed-length rows from the global image once per
    // solve -- the tail table alone); off_* are offsets into the global image
    const int* csr;
    int lds_img0, lds_img_ints;
    int off_rp_u, off_rp_d, off_en_u, off_en_d, off_lead_t, off_tail_t;
    int off_diag;          // [NR] diagonal of W_d (uniform instances: their W_d rows hold the other entries; else 0) and [NR] of W_d^T, floats
    const float* band_w;   // [T*skip] (band mode)
    // state, sample-major (B, TN)
    float *zu, *zd, *phi, *gam, *gu, *gd;
    const float* y;        // (B, t_in, N) prediction / (B, T, N) mask mode
    const float* mask;     // (B, T, N) or nullptr
    // outputs
    double* ps;            // [J][NMETRIC][Bp] per-sample metric sums
    int* cg_iters;         // [J][3][Bp]
    float* alpha_hist;     // [3][max_cg][Bp] or nullptr
    float* beta_hist;
    int* nonfinite;
    const int* stop;       // device stop word of the ADMM outer loop (nullptr: none): a launch enqueued speculatively after the
                           // stop test of an earlier iteration passed returns at its first instruction
};
struct LdsArgs : LdsArgsCore {
    float* xs[LDS_MAXJ + 1];   // trip k reads the iterate xs[k] and writes xs[k + 1] (every iterate is kept: delta_x_per_step);
                               // indexed by the trip number straight from the kernarg segment
};

// Execution plan of k_admm_lds chosen by Engine::plan_lds
struct LdsLaunch {
    int tpg;        // time steps per thread
    int maxt;       // workgroup-size class the kernel was compiled for (640 / 1024)
    int sb;         // single LDS vector (two workgroups per CU with the 640-thread class)
    int uniform45;  // every W_u row has 4 and every W_d row 5 entries (k = 4 table without pads): unrolled gathers, entries
                    // read from the global image once per solve
    int slots;      // uniform45 only: two more LDS vectors hold per-thread operands across the solves of an iteration
    int block;      // threads per workgroup
    size_t lds_bytes;
};

// J ADMM iterations for B samples (one workgroup per sample); returns a mgadmm_status
int mg_lds_iteration(const LdsLaunch& L, const LdsArgs& a, int B, hipStream_t st);
// initial state (ADMM.py:528-544): x in the reference's sample-major layout, zu / zd / gamma* thread-major for time groups of TPG steps
int mg_lds_init(bool masked, int T, int t_in, int N, int TPG, int B, float tm, float den, const float* y, const float* mask, float* x,
                float* zu, float* zd, float* gam, float* gu, float* gd, int* nonfinite, hipStream_t st);
// one state vector (B, T, N) <-> thread-major layout of the LDS path (lds_kernels.h, lds_state_index); src != dst
int mg_lds_state_layout(bool to_thread_major, int T, int N, int TPG, int B, const float* src, float* dst, hipStream_t st);
// delta_x_per_step on the sample-major layout (ADMM.py:614): scratch = double[TN * (1 + ceil(B/64))], out = double[T];
// stop: device stop word (the kernels return at once when it is set) or nullptr
int mg_lds_dxps(int T, int N, int B, const float* x, const float* xo, double* scratch, double* out, const int* stop, hipStream_t st);
// stop test of one ADMM iteration on the device (ADMM.py:645-646 + the NaN asserts): *stop = it + 1 when both residual maxima
// are below tol, -(it + 1) when a metric or the iterate is not finite; leaves a stop word that is already set alone
int mg_lds_stop_test(const double* metrics_row, const int* nonfinite, int has_phi, int has_zd, double tol, int it, int* stop, hipStream_t st);

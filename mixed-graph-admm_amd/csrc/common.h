// Internal declarations shared by the HIP translation units of libmgadmm.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "mgadmm.h"

void mg_set_error(const char* fmt, ...);

#define MG_HIP(x)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            mg_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_));    \
            return MGADMM_ERR_HIP;                                                             \
        }                                                                                      \
    } while (0)

#define MG_REQUIRE(cond, ...)                                                                  \
    do {                                                                                       \
        if (!(cond)) {                                                                         \
            mg_set_error(__VA_ARGS__);                                                         \
            return MGADMM_ERR_INVALID;                                                         \
        }                                                                                      \
    } while (0)

#define MG_TRY(x)                                                                              \
    do {                                                                                       \
        int rc_ = (x);                                                                         \
        if (rc_ != MGADMM_OK) return rc_;                                                      \
    } while (0)

#include "host_csr.h"

struct DevCsr {
    int n = 0, nnz = 0, max_row = 0, avg_row_ceil = 0;
    int* rowptr = nullptr;
    int* col = nullptr;
    float* val = nullptr;
};

// How a row-operator is evaluated by the row kernels (see k_rows in stream_kernels.h).
enum OpKind { OPK_NONE = 0, OPK_SPATIAL = 1, OPK_BAND = 2 };
enum SelfMode { SELF_ONE = 0, SELF_LDR = 1, SELF_LDRT = 2, SELF_LN_FATHER = 3 };   // LN_FATHER: 1 for t <= T-2, else 0

struct OpDesc {
    int kind;             // OpKind
    const int* rowptr;    // SPATIAL: CSR of the neighbour weights (internal node order)
    const int* col;
    const float* val;
    int shift;            // SPATIAL: time offset of the gathered rows (0, -1, +1)
    const float* band_w;  // BAND: [T*skip]
    int skip;
    int band_dir;         // BAND: -1 = Ldr (looks back), +1 = Ldr_T (looks forward)
    int self_mode;        // SelfMode
    int q1;               // SELF_LDRT: keep the identity on the t=0 block (quirk Q1)
    const double* self_w; // optional per-row factor of the self coefficient (apply_op_Ln: row sums of d_ew); plain row kernel only
};

struct mgadmm_graph {
    int N = 0, T = 0, mode = 0, device = 0;
    int transpose_by_gather = 0, q1 = 0, skip = 0;
    bool has_perm = false;
    int reorder = 0;              // 0 none, 1 RCM, 2 greedy cluster order (enables the LDS-tiled row kernel)
    HostCsr hWu, hWd, hWdT;       // API node order (hWdT: exact transpose, or hWd when transpose_by_gather)
    std::vector<int> perm, iperm;  // perm[i] = API node at internal row i; iperm = inverse
    std::vector<int> cluster_starts;  // reorder = 2: first internal row of every cluster of the node order
    DevCsr Wu, Wd, WdT;            // internal node order, device
    float* band_w = nullptr;       // device
    double* ln_rowsum = nullptr;   // device, internal order: sum_j d_ew[i,j] (self coefficient of apply_op_Ln), built on first use
    int* d_perm = nullptr;         // device (nullptr when identity)
    int max_row_all = 0;

    OpDesc op_lu() const;
    OpDesc op_ldr() const;
    OpDesc op_ldrt() const;
};

// graph.hip
int mg_transpose_csr(const HostCsr& A, HostCsr& At);
int mg_rcm_order(const HostCsr& A, const HostCsr* B, std::vector<int>& perm);
int mg_cluster_order(const HostCsr& A, const HostCsr* B, int csize, std::vector<int>& perm, std::vector<int>* starts = nullptr);
int mg_permute_csr(const HostCsr& A, const std::vector<int>& perm, const std::vector<int>& iperm, HostCsr& out);

struct EngineBase;
struct mgadmm_solver {
    mgadmm_graph* g = nullptr;
    mgadmm_params p;
    int Bmax = 0;
    EngineBase* eng = nullptr;
};

struct EngineBase {
    virtual ~EngineBase() {}
    virtual int init() = 0;
    virtual int set_params(const mgadmm_params& p) = 0;
    virtual int64_t workspace_bytes() const = 0;
    virtual int path_for(int B) const = 0;
    virtual int query(int what, int64_t* out) const = 0;
    virtual int apply(int op, const void* x, void* y, int B, hipStream_t st) = 0;
    virtual int lhs(int which, const void* x, const void* mask, void* y, int B, hipStream_t st) = 0;
    virtual int phi_direct(const void* x, const void* gamma, void* phi, int B, hipStream_t st) = 0;
    virtual int initial_guess(const void* y, void* x, int B, hipStream_t st) = 0;
    virtual int initial_interpolation(const void* y, const void* mask, int mask_f32, void* x, int B,
                                      hipStream_t st) = 0;
    virtual int cg(int which, const void* rhs, const void* x0, const void* mask, void* x, int32_t* iters,
                   double* alpha, double* beta, int B, hipStream_t st) = 0;
    virtual int solve(const void* y, const void* mask, int mask_f32, int B, const void* x0, const mgadmm_state* state_in,
                      void* x_out, const mgadmm_state* state_out, mgadmm_history* hist, hipStream_t st) = 0;
    virtual int two_loops(const void* y, const void* mask, int mask_f32, int B, void* x_out, const mgadmm_state* state_out,
                          mgadmm_history* hist, hipStream_t st) = 0;
    virtual int prof_begin() = 0;
    virtual int prof_end(int64_t* counts, double* total_ms, double* bytes) = 0;
};

EngineBase* mg_make_engine_f32(mgadmm_solver* s);
EngineBase* mg_make_engine_f64(mgadmm_solver* s);

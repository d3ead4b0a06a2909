// Graph handle: CSR ingestion, exact transpose of W_d (replaces the scatter_add of
// reference ADMM.py:200-209), optional bandwidth-reducing node order, upload to HBM.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <numeric>
#include <queue>

#include "common.h"

static thread_local char g_err[1024] = "";

void mg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mgadmm_last_error(void) { return g_err; }
// 0.2: mgadmm_params gained cg_convergence, max_inner_iter; 0.3: round-3 LDS path, more mgadmm_query_t codes (no struct change)
extern "C" const char* mgadmm_version(void) { return "mgadmm 0.3.0 (gfx950)"; }

int mg_transpose_csr(const HostCsr& A, HostCsr& At) {
    const int n = A.n, nnz = A.nnz();
    At.n = n;
    At.rowptr.assign(n + 1, 0);
    At.col.resize(nnz);
    At.val.resize(nnz);
    for (int e = 0; e < nnz; ++e) At.rowptr[A.col[e] + 1]++;
    for (int i = 0; i < n; ++i) At.rowptr[i + 1] += At.rowptr[i];
    std::vector<int> fill(At.rowptr.begin(), At.rowptr.end() - 1);
    // rows visited in increasing order -> entries of every transposed row are sorted by source row:
    // a fixed summation order, so results are bitwise repeatable (no atomics anywhere).
    for (int i = 0; i < n; ++i)
        for (int e = A.rowptr[i]; e < A.rowptr[i + 1]; ++e) {
            int d = fill[A.col[e]]++;
            At.col[d] = i;
            At.val[d] = A.val[e];
        }
    return MGADMM_OK;
}

// Reverse Cuthill-McKee on the symmetrised pattern of A (and B when given).
int mg_rcm_order(const HostCsr& A, const HostCsr* B, std::vector<int>& perm) {
    const int n = A.n;
    std::vector<std::vector<int>> adj(n);
    auto add = [&](const HostCsr& M) {
        for (int i = 0; i < n; ++i)
            for (int e = M.rowptr[i]; e < M.rowptr[i + 1]; ++e) {
                int j = M.col[e];
                if (j != i) {
                    adj[i].push_back(j);
                    adj[j].push_back(i);
                }
            }
    };
    add(A);
    if (B) add(*B);
    for (auto& a : adj) {
        std::sort(a.begin(), a.end());
        a.erase(std::unique(a.begin(), a.end()), a.end());
    }
    std::vector<int> deg(n);
    for (int i = 0; i < n; ++i) deg[i] = (int)adj[i].size();
    std::vector<int> order_by_deg(n);
    std::iota(order_by_deg.begin(), order_by_deg.end(), 0);
    std::stable_sort(order_by_deg.begin(), order_by_deg.end(), [&](int a, int b) { return deg[a] < deg[b]; });
    std::vector<char> seen(n, 0);
    perm.clear();
    perm.reserve(n);
    for (int s : order_by_deg) {
        if (seen[s]) continue;
        // pseudo-peripheral start: two BFS sweeps from the lowest-degree unseen node
        int start = s;
        for (int sweep = 0; sweep < 2; ++sweep) {
            std::vector<int> q{start}, dist_nodes;
            std::vector<char> loc(n, 0);
            loc[start] = 1;
            size_t h = 0;
            int last = start;
            while (h < q.size()) {
                int v = q[h++];
                last = v;
                for (int u : adj[v])
                    if (!loc[u] && !seen[u]) {
                        loc[u] = 1;
                        q.push_back(u);
                    }
            }
            // last level node with the lowest degree
            start = last;
        }
        size_t h = perm.size();
        perm.push_back(start);
        seen[start] = 1;
        while (h < perm.size()) {
            int v = perm[h++];
            std::vector<int> nb;
            for (int u : adj[v])
                if (!seen[u]) {
                    seen[u] = 1;
                    nb.push_back(u);
                }
            std::stable_sort(nb.begin(), nb.end(), [&](int a, int b) { return deg[a] < deg[b]; });
            for (int u : nb) perm.push_back(u);
        }
    }
    std::reverse(perm.begin(), perm.end());
    return MGADMM_OK;
}

// Greedy graph-growing cluster order on the symmetrised pattern of A (and B): clusters of `csize` nodes are
// grown one at a time, always absorbing the frontier node with the most edges into the current cluster; the
// next cluster is seeded on the boundary of the previous ones.  Consecutive rows of the resulting order are
// graph neighbours far more often than under RCM (kNN graph, N=10k: 90 % of the edges within +-16 rows vs
// 38 %), which is what the LDS-tiled row kernel needs.
int mg_cluster_order(const HostCsr& A, const HostCsr* B, int csize, std::vector<int>& perm, std::vector<int>* starts) {
    const int n = A.n;
    std::vector<std::vector<int>> adj(n);
    auto add = [&](const HostCsr& M) {
        for (int i = 0; i < n; ++i)
            for (int e = M.rowptr[i]; e < M.rowptr[i + 1]; ++e) {
                int j = M.col[e];
                if (j != i) {
                    adj[i].push_back(j);
                    adj[j].push_back(i);
                }
            }
    };
    add(A);
    if (B) add(*B);
    for (auto& a : adj) {
        std::sort(a.begin(), a.end());
        a.erase(std::unique(a.begin(), a.end()), a.end());
    }
    std::vector<char> done(n, 0);
    std::vector<int> gain(n, 0), seeds;
    perm.clear();
    perm.reserve(n);
    int next_unassigned = 0;
    int start = 0;
    for (int i = 1; i < n; ++i)
        if (adj[i].size() < adj[start].size()) start = i;
    seeds.push_back(start);
    typedef std::pair<int, int> PI;   // (edges into the cluster, -node) : max-heap, ties -> lowest node id
    while ((int)perm.size() < n) {
        int seed = -1;
        while (!seeds.empty()) {
            int c = seeds.back();
            seeds.pop_back();
            if (!done[c]) { seed = c; break; }
        }
        if (seed < 0) {
            while (done[next_unassigned]) ++next_unassigned;
            seed = next_unassigned;
        }
        std::priority_queue<PI> heap;
        std::vector<int> touched;
        if (starts) starts->push_back((int)perm.size());      // first internal row of this cluster
        heap.push(PI(0, -seed));
        int size = 0;
        while (!heap.empty() && size < csize) {
            PI top = heap.top();
            heap.pop();
            int v = -top.second;
            if (done[v] || top.first != gain[v]) continue;   // stale entry
            done[v] = 1;
            perm.push_back(v);
            ++size;
            for (int u : adj[v])
                if (!done[u]) {
                    if (gain[u] == 0) touched.push_back(u);
                    ++gain[u];
                    heap.push(PI(gain[u], -u));
                }
        }
        // the frontier that was not absorbed seeds the following clusters (closest-first order)
        std::vector<int> bnd;
        for (int u : touched)
            if (!done[u]) bnd.push_back(u);
        std::stable_sort(bnd.begin(), bnd.end(), [&](int a, int b) { return gain[a] < gain[b]; });
        for (int u : bnd) seeds.push_back(u);
        for (int u : touched) gain[u] = 0;
    }
    return MGADMM_OK;
}

int mg_permute_csr(const HostCsr& A, const std::vector<int>& perm, const std::vector<int>& iperm, HostCsr& out) {
    const int n = A.n;
    out.n = n;
    out.rowptr.assign(n + 1, 0);
    out.col.clear();
    out.val.clear();
    out.col.reserve(A.nnz());
    out.val.reserve(A.nnz());
    for (int i = 0; i < n; ++i) {
        int src = perm[i];
        for (int e = A.rowptr[src]; e < A.rowptr[src + 1]; ++e) {
            out.col.push_back(iperm[A.col[e]]);
            out.val.push_back(A.val[e]);
        }
        out.rowptr[i + 1] = (int)out.col.size();
    }
    return MGADMM_OK;
}

static int ingest_csr(int n, const int32_t* rp, const int32_t* col, const float* val, HostCsr& out, const char* nm) {
    MG_REQUIRE(rp && rp[0] == 0, "%s: rowptr missing or rowptr[0] != 0", nm);
    for (int i = 0; i < n; ++i) MG_REQUIRE(rp[i + 1] >= rp[i], "%s: rowptr not monotone at row %d", nm, i);
    int nnz = rp[n];
    MG_REQUIRE(nnz == 0 || (col && val), "%s: col/val missing", nm);
    for (int e = 0; e < nnz; ++e)
        MG_REQUIRE(col[e] >= 0 && col[e] < n, "%s: column index %d out of bounds at entry %d (n=%d)", nm, col[e], e, n);
    out.n = n;
    out.rowptr.assign(rp, rp + n + 1);
    out.col.assign(col, col + nnz);
    out.val.assign(val, val + nnz);
    return MGADMM_OK;
}

static int upload_csr(const HostCsr& h, DevCsr& d) {
    d.n = h.n;
    d.nnz = h.nnz();
    d.max_row = 0;
    for (int i = 0; i < h.n; ++i) d.max_row = std::max(d.max_row, h.rowptr[i + 1] - h.rowptr[i]);
    d.avg_row_ceil = h.n ? (d.nnz + h.n - 1) / h.n : 0;
    MG_HIP(hipMalloc(&d.rowptr, sizeof(int) * (h.n + 1)));
    // CSR_PAD (8) zeroed extra entries: the row kernels read a fixed window of entries per row
    MG_HIP(hipMalloc(&d.col, sizeof(int) * (d.nnz + 8)));
    MG_HIP(hipMalloc(&d.val, sizeof(float) * (d.nnz + 8)));
    MG_HIP(hipMemset(d.col, 0, sizeof(int) * (d.nnz + 8)));
    MG_HIP(hipMemset(d.val, 0, sizeof(float) * (d.nnz + 8)));
    MG_HIP(hipMemcpy(d.rowptr, h.rowptr.data(), sizeof(int) * (h.n + 1), hipMemcpyHostToDevice));
    if (d.nnz) {
        MG_HIP(hipMemcpy(d.col, h.col.data(), sizeof(int) * d.nnz, hipMemcpyHostToDevice));
        MG_HIP(hipMemcpy(d.val, h.val.data(), sizeof(float) * d.nnz, hipMemcpyHostToDevice));
    }
    return MGADMM_OK;
}

static void free_csr(DevCsr& d) {
    if (d.rowptr) (void)hipFree(d.rowptr);
    if (d.col) (void)hipFree(d.col);
    if (d.val) (void)hipFree(d.val);
    d = DevCsr();
}

OpDesc mgadmm_graph::op_lu() const {
    OpDesc o{};
    o.kind = OPK_SPATIAL;
    o.rowptr = Wu.rowptr; o.col = Wu.col; o.val = Wu.val;
    o.shift = 0;
    o.self_mode = SELF_ONE;
    return o;
}

OpDesc mgadmm_graph::op_ldr() const {
    OpDesc o{};
    o.self_mode = SELF_LDR;
    if (mode == MGADMM_TEMPORAL_SPATIAL) {
        o.kind = OPK_SPATIAL;
        o.rowptr = Wd.rowptr; o.col = Wd.col; o.val = Wd.val;
        o.shift = -1;
    } else {
        o.kind = OPK_BAND;
        o.band_w = band_w; o.skip = skip; o.band_dir = -1;
    }
    return o;
}

OpDesc mgadmm_graph::op_ldrt() const {
    OpDesc o{};
    o.self_mode = SELF_LDRT;
    if (mode == MGADMM_TEMPORAL_SPATIAL) {
        o.kind = OPK_SPATIAL;
        o.rowptr = WdT.rowptr; o.col = WdT.col; o.val = WdT.val;
        o.shift = +1;
        o.q1 = q1;
    } else {
        o.kind = OPK_BAND;
        o.band_w = band_w; o.skip = skip; o.band_dir = +1;
        o.q1 = 0;  // line-graph branches are exact transposes (ADMM.py:181-194)
    }
    return o;
}

extern "C" int mgadmm_graph_create(const mgadmm_graph_desc* d, mgadmm_graph** out) {
    MG_REQUIRE(d && out, "graph_create: null argument");
    MG_REQUIRE(d->n_nodes > 0 && d->T > 1, "graph_create: need n_nodes > 0 and T > 1 (got %d, %d)", d->n_nodes, d->T);
    MG_REQUIRE(d->temporal_mode == MGADMM_TEMPORAL_SPATIAL || d->temporal_mode == MGADMM_TEMPORAL_BAND,
               "graph_create: bad temporal_mode %d", d->temporal_mode);
    MG_HIP(hipSetDevice(d->device));
    mgadmm_graph* g = new mgadmm_graph();
    g->N = d->n_nodes; g->T = d->T; g->mode = d->temporal_mode; g->device = d->device;
    g->transpose_by_gather = d->transpose_by_gather; g->q1 = d->q1_identity_t0 ? 1 : 0;
    int rc = ingest_csr(g->N, d->u_rowptr, d->u_col, d->u_val, g->hWu, "W_u");
    if (rc == MGADMM_OK && g->mode == MGADMM_TEMPORAL_SPATIAL) {
        rc = ingest_csr(g->N, d->d_rowptr, d->d_col, d->d_val, g->hWd, "W_d");
        if (rc == MGADMM_OK) {
            if (g->transpose_by_gather) g->hWdT = g->hWd;
            else rc = mg_transpose_csr(g->hWd, g->hWdT);
        }
    } else if (rc == MGADMM_OK) {
        if (d->skip < 1 || d->skip >= d->T || !d->band_w) {
            mg_set_error("graph_create: BAND mode needs 1 <= skip < T and band_w");
            rc = MGADMM_ERR_INVALID;
        } else {
            g->skip = d->skip;
        }
    }
    if (rc != MGADMM_OK) { delete g; return rc; }

    g->perm.resize(g->N);
    std::iota(g->perm.begin(), g->perm.end(), 0);
    if (d->reorder == 1) {
        mg_rcm_order(g->hWu, g->mode == MGADMM_TEMPORAL_SPATIAL ? &g->hWd : nullptr, g->perm);
        g->has_perm = true;
    } else if (d->reorder >= 2) {
        // clusters of 64 rows: one cluster = one tile of the fused cLdr kernel (8 tiles of the 8-row SpMM kernel)
        mg_cluster_order(g->hWu, g->mode == MGADMM_TEMPORAL_SPATIAL ? &g->hWd : nullptr, 64, g->perm, &g->cluster_starts);
        g->has_perm = true;
    }
    g->reorder = d->reorder;
    g->iperm.resize(g->N);
    for (int i = 0; i < g->N; ++i) g->iperm[g->perm[i]] = i;

    auto up = [&](const HostCsr& h, DevCsr& dv) -> int {
        if (!g->has_perm) return upload_csr(h, dv);
        HostCsr p;
        mg_permute_csr(h, g->perm, g->iperm, p);
        return upload_csr(p, dv);
    };
    rc = up(g->hWu, g->Wu);
    if (rc == MGADMM_OK && g->mode == MGADMM_TEMPORAL_SPATIAL) {
        rc = up(g->hWd, g->Wd);
        if (rc == MGADMM_OK) rc = up(g->hWdT, g->WdT);
    }
    if (rc == MGADMM_OK && g->mode == MGADMM_TEMPORAL_BAND) {
        if (hipMalloc(&g->band_w, sizeof(float) * g->T * g->skip) != hipSuccess ||
            hipMemcpy(g->band_w, d->band_w, sizeof(float) * g->T * g->skip, hipMemcpyHostToDevice) != hipSuccess) {
            mg_set_error("graph_create: band_w upload failed");
            rc = MGADMM_ERR_HIP;
        }
    }
    if (rc == MGADMM_OK && g->has_perm) {
        if (hipMalloc(&g->d_perm, sizeof(int) * g->N) != hipSuccess ||
            hipMemcpy(g->d_perm, g->perm.data(), sizeof(int) * g->N, hipMemcpyHostToDevice) != hipSuccess) {
            mg_set_error("graph_create: perm upload failed");
            rc = MGADMM_ERR_HIP;
        }
    }
    if (rc != MGADMM_OK) { mgadmm_graph_destroy(g); return rc; }
    g->max_row_all = std::max(g->Wu.max_row, std::max(g->Wd.max_row, g->WdT.max_row));
    *out = g;
    return MGADMM_OK;
}

extern "C" int mgadmm_graph_destroy(mgadmm_graph* g) {
    if (!g) return MGADMM_OK;
    (void)hipSetDevice(g->device);
    free_csr(g->Wu); free_csr(g->Wd); free_csr(g->WdT);
    if (g->band_w) (void)hipFree(g->band_w);
    if (g->d_perm) (void)hipFree(g->d_perm);
    if (g->ln_rowsum) (void)hipFree(g->ln_rowsum);
    delete g;
    return MGADMM_OK;
}

extern "C" int mgadmm_graph_transpose_nnz(const mgadmm_graph* g, int32_t* nnz) {
    MG_REQUIRE(g && nnz, "transpose_nnz: null argument");
    *nnz = g->hWdT.nnz();
    return MGADMM_OK;
}

extern "C" int mgadmm_graph_get_transpose(const mgadmm_graph* g, int32_t* rowptr, int32_t* col, float* val) {
    MG_REQUIRE(g && rowptr && col && val, "get_transpose: null argument");
    MG_REQUIRE(g->mode == MGADMM_TEMPORAL_SPATIAL, "get_transpose: graph has no spatial temporal weights");
    memcpy(rowptr, g->hWdT.rowptr.data(), sizeof(int) * (g->N + 1));
    memcpy(col, g->hWdT.col.data(), sizeof(int) * g->hWdT.nnz());
    memcpy(val, g->hWdT.val.data(), sizeof(float) * g->hWdT.nnz());
    return MGADMM_OK;
}

extern "C" int mgadmm_graph_get_perm(const mgadmm_graph* g, int32_t* perm) {
    MG_REQUIRE(g && perm, "get_perm: null argument");
    memcpy(perm, g->perm.data(), sizeof(int) * g->N);
    return MGADMM_OK;
}

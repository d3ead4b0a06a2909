// Graph construction on the GPU: the tables the ADMM operators are built from.
//   mgadmm_knn_graph      k_nearest_neighbors   utils.py:183-204  (graph-geodesic kNN, one truncated Dijkstra per node)
//   mgadmm_weight_tables  undirected_graph_from_distance / directed_graph_from_distance   utils.py:206-258
// The reference runs a full single-source Dijkstra (networkx) from every node and keeps the k+1 closest:
// O(N^2 log N), ~3 min at 10k nodes.  A search only ever needs its first k+1 settled nodes, so every source
// is an independent tiny search: one GPU thread per source, candidate list in private memory.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "common.h"

namespace {

constexpr int KNN_MAXC = 128;   // live candidates of one search (pushes not yet popped)
constexpr int KNN_MAXK = 32;    // k + 1 <= 32

// One thread = one source.  Reproduces networkx's _dijkstra_multisource order exactly:
//   * the fringe is a heap of (dist, push counter, node): ties in dist pop in push order;
//   * a neighbour is pushed only when the new distance improves on the best one seen for it;
//   * neighbours are visited in adjacency insertion order (adjacency rows are built that way on the host);
//   * distances accumulate in float64 (Python floats) and are rounded to float32 on output (utils.py:195).
// The first k+1 settled nodes are exactly heapq.nsmallest(k+1, ..., key=dist) of the full search (stable:
// equal distances keep settle order).
__global__ __launch_bounds__(64) void k_knn_dijkstra(int n, int k1, const int* __restrict__ rowptr, const int* __restrict__ adj,
                                                     const double* __restrict__ w, int* __restrict__ nn, float* __restrict__ nd,
                                                     int* __restrict__ overflow) {
    const int src = blockIdx.x * 64 + threadIdx.x;
    if (src >= n) return;
    int cn[KNN_MAXC];
    double cd[KNN_MAXC];
    int ct[KNN_MAXC];
    int dn[KNN_MAXK];
    int cnt = 1, tick = 0, ndone = 0;
    cn[0] = src; cd[0] = 0.0; ct[0] = 0;
    while (cnt > 0 && ndone < k1) {
        int best = 0;
        for (int c = 1; c < cnt; ++c)
            if (cd[c] < cd[best] || (cd[c] == cd[best] && ct[c] < ct[best])) best = c;
        const int v = cn[best];
        const double dist = cd[best];
        --cnt;
        cn[best] = cn[cnt]; cd[best] = cd[cnt]; ct[best] = ct[cnt];
        bool settled = false;
        for (int j = 0; j < ndone; ++j) settled |= dn[j] == v;
        if (settled) continue;
        dn[ndone] = v;
        nn[(size_t)src * k1 + ndone] = v;
        nd[(size_t)src * k1 + ndone] = (float)dist;
        ++ndone;
        if (ndone == k1) break;
        for (int e = rowptr[v]; e < rowptr[v + 1]; ++e) {
            const int u = adj[e];
            bool done_u = false;
            for (int j = 0; j < ndone; ++j) done_u |= dn[j] == u;
            if (done_u) continue;
            const double cand = dist + w[e];
            bool improves = true;           // best[u] = smallest live candidate of u (its minimum cannot have been popped: u is not settled)
            for (int c = 0; c < cnt; ++c)
                if (cn[c] == u && cd[c] <= cand) improves = false;
            if (!improves) continue;
            if (cnt == KNN_MAXC) {
                *overflow = 1;
                return;
            }
            ++tick;
            cn[cnt] = u; cd[cnt] = cand; ct[cnt] = tick;
            ++cnt;
        }
    }
    for (int j = ndone; j < k1; ++j) {      // fewer than k+1 reachable nodes: pads -1 / inf (utils.py:192-193)
        nn[(size_t)src * k1 + j] = -1;
        nd[(size_t)src * k1 + j] = INFINITY;
    }
}

// min / max of the distances that define the default sigma: entries with connect_list != -1 and dist != 0
// (utils.py:219-223).  Positive floats order like their bit patterns.
__global__ __launch_bounds__(256) void k_dist_range(long total, const long long* __restrict__ cl, const float* __restrict__ dl,
                                                    unsigned* __restrict__ mn, unsigned* __restrict__ mx) {
    unsigned lo = 0xffffffffu, hi = 0u;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const float d = dl[i];
        if (cl[i] != -1 && d != 0.f && d > 0.f) {
            const unsigned b = __float_as_uint(d);
            lo = min(lo, b);
            hi = max(hi, b);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, (unsigned)__shfl_xor((int)lo, o));
        hi = max(hi, (unsigned)__shfl_xor((int)hi, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(mn, lo);
        atomicMax(mx, hi);
    }
}

// raw weights exp(-d / sigma) with pads zeroed, and the row sums ("degree")   utils.py:225-232, 248-255
//   undirected: columns 1..k of the (N, k+1) tables -> w (N, k);  directed: all k+1 columns -> w (N, k+1)
__global__ __launch_bounds__(256) void k_raw_weights(int n, int k1, int first, float sigma, const long long* __restrict__ cl,
                                                     const float* __restrict__ dl, float* __restrict__ w, float* __restrict__ deg) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int kc = k1 - first;
    float s = 0.f;
    for (int j = 0; j < kc; ++j) {
        const size_t t = (size_t)i * k1 + first + j;
        float v = (float)exp(-(double)(dl[t] / sigma));    // float32 quotient like the reference; exp rounded once, denormals kept
        if (cl[t] == -1) v = 0.f;
        w[(size_t)i * kc + j] = v;
        s += v;
    }
    deg[i] = s;
}

// undirected: w_ij /= sqrt(deg_i * deg_j), j = connect_list[i, 1 + c]; a pad index (-1) wraps to the last node
// like the reference's fancy indexing (quirk Q5) -- its weight is 0 anyway.   utils.py:233-237
__global__ __launch_bounds__(256) void k_norm_undirected(int n, int k1, const long long* __restrict__ cl, const float* __restrict__ deg,
                                                         float* __restrict__ w) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int kc = k1 - 1;
    const float di = deg[i];
    for (int j = 0; j < kc; ++j) {
        long long c = cl[(size_t)i * k1 + 1 + j];
        if (c < 0) c += n;
        const float prod = di * deg[c];
        const float inv = prod > 0.f ? 1.f / sqrtf(prod) : 0.f;
        w[(size_t)i * kc + j] *= inv;
    }
}

// directed: rows normalised to sum 1   utils.py:252-257
__global__ __launch_bounds__(256) void k_norm_directed(int n, int k1, const float* __restrict__ deg, float* __restrict__ w) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float d = deg[i];
    const float inv = d > 0.f ? 1.f / d : 0.f;
    for (int j = 0; j < k1; ++j) w[(size_t)i * k1 + j] *= inv;
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    template <class T> T* as() { return static_cast<T*>(p); }
};

}  // namespace

extern "C" int mgadmm_knn_graph(int32_t n_nodes, int64_t n_edges, const int64_t* edges, const double* dists, int32_t k,
                                int32_t* nn_out, float* nd_out, int32_t device) {
    MG_REQUIRE(n_nodes > 0 && n_edges >= 0 && k >= 0, "knn_graph: bad sizes (n_nodes %d, n_edges %lld, k %d)", n_nodes,
               (long long)n_edges, k);
    MG_REQUIRE(k + 1 <= KNN_MAXK, "knn_graph: k + 1 = %d exceeds %d", k + 1, KNN_MAXK);
    MG_REQUIRE((n_edges == 0 || (edges && dists)) && nn_out && nd_out, "knn_graph: null pointer");
    // adjacency rows in networkx's order: a node's neighbours in order of first insertion, a repeated edge
    // overwrites the weight but keeps its place (DiGraph.add_edge, utils.py:190-191)
    std::vector<std::vector<std::pair<int, double>>> rows(n_nodes);
    for (int64_t e = 0; e < n_edges; ++e) {
        const int64_t s = edges[2 * e], t = edges[2 * e + 1];
        MG_REQUIRE(s >= 0 && s < n_nodes && t >= 0 && t < n_nodes, "knn_graph: edge %lld = (%lld, %lld) out of range",
                   (long long)e, (long long)s, (long long)t);
        MG_REQUIRE(dists[e] >= 0.0, "knn_graph: negative edge length %g (Dijkstra needs non-negative weights)", dists[e]);
        auto& r = rows[s];
        auto it = std::find_if(r.begin(), r.end(), [&](const std::pair<int, double>& q) { return q.first == (int)t; });
        if (it == r.end()) r.emplace_back((int)t, dists[e]);
        else it->second = dists[e];
    }
    std::vector<int> rowptr(n_nodes + 1, 0), adj;
    std::vector<double> w;
    for (int i = 0; i < n_nodes; ++i) {
        rowptr[i + 1] = rowptr[i] + (int)rows[i].size();
        for (auto& q : rows[i]) { adj.push_back(q.first); w.push_back(q.second); }
    }
    MG_HIP(hipSetDevice(device));
    const int k1 = k + 1;
    DevBuf d_rp, d_adj, d_w, d_nn, d_nd, d_of;
    MG_HIP(hipMalloc(&d_rp.p, sizeof(int) * (n_nodes + 1)));
    MG_HIP(hipMalloc(&d_adj.p, sizeof(int) * std::max<size_t>(adj.size(), 1)));
    MG_HIP(hipMalloc(&d_w.p, sizeof(double) * std::max<size_t>(w.size(), 1)));
    MG_HIP(hipMalloc(&d_nn.p, sizeof(int) * (size_t)n_nodes * k1));
    MG_HIP(hipMalloc(&d_nd.p, sizeof(float) * (size_t)n_nodes * k1));
    MG_HIP(hipMalloc(&d_of.p, sizeof(int)));
    MG_HIP(hipMemcpy(d_rp.p, rowptr.data(), sizeof(int) * (n_nodes + 1), hipMemcpyHostToDevice));
    if (!adj.empty()) {
        MG_HIP(hipMemcpy(d_adj.p, adj.data(), sizeof(int) * adj.size(), hipMemcpyHostToDevice));
        MG_HIP(hipMemcpy(d_w.p, w.data(), sizeof(double) * w.size(), hipMemcpyHostToDevice));
    }
    MG_HIP(hipMemset(d_of.p, 0, sizeof(int)));
    hipLaunchKernelGGL(k_knn_dijkstra, dim3((n_nodes + 63) / 64), dim3(64), 0, 0, n_nodes, k1, d_rp.as<int>(), d_adj.as<int>(),
                       d_w.as<double>(), d_nn.as<int>(), d_nd.as<float>(), d_of.as<int>());
    MG_HIP(hipGetLastError());
    int of = 0;
    MG_HIP(hipMemcpy(&of, d_of.p, sizeof(int), hipMemcpyDeviceToHost));
    if (of) {
        mg_set_error("knn_graph: a search needed more than %d live candidates (dense graph); not supported", KNN_MAXC);
        return MGADMM_ERR_UNSUPPORTED;
    }
    MG_HIP(hipMemcpy(nn_out, d_nn.p, sizeof(int) * (size_t)n_nodes * k1, hipMemcpyDeviceToHost));
    MG_HIP(hipMemcpy(nd_out, d_nd.p, sizeof(float) * (size_t)n_nodes * k1, hipMemcpyDeviceToHost));
    return MGADMM_OK;
}

extern "C" int mgadmm_weight_tables(int32_t n_nodes, int32_t k1, const int64_t* connect_list, const float* dist_list,
                                    double u_sigma, double d_sigma, int32_t regularized, float* u_ew, float* d_ew,
                                    double* sigmas_out, int32_t device) {
    MG_REQUIRE(n_nodes > 0 && k1 >= 1, "weight_tables: bad sizes (n_nodes %d, columns %d)", n_nodes, k1);
    MG_REQUIRE(connect_list && dist_list && (u_ew || d_ew), "weight_tables: null pointer");
    MG_HIP(hipSetDevice(device));
    const size_t total = (size_t)n_nodes * k1;
    for (size_t i = 0; i < total; ++i)
        MG_REQUIRE(connect_list[i] >= -1 && connect_list[i] < n_nodes, "weight_tables: connect_list[%zu] = %lld out of range", i,
                   (long long)connect_list[i]);
    DevBuf d_cl, d_dl, d_w, d_deg, d_mm;
    MG_HIP(hipMalloc(&d_cl.p, sizeof(long long) * total));
    MG_HIP(hipMalloc(&d_dl.p, sizeof(float) * total));
    MG_HIP(hipMalloc(&d_w.p, sizeof(float) * total));
    MG_HIP(hipMalloc(&d_deg.p, sizeof(float) * n_nodes));
    MG_HIP(hipMalloc(&d_mm.p, sizeof(unsigned) * 2));
    MG_HIP(hipMemcpy(d_cl.p, connect_list, sizeof(long long) * total, hipMemcpyHostToDevice));
    MG_HIP(hipMemcpy(d_dl.p, dist_list, sizeof(float) * total, hipMemcpyHostToDevice));
    if (!(u_sigma > 0.0) || !(d_sigma > 0.0)) {
        const unsigned init[2] = {0xffffffffu, 0u};
        MG_HIP(hipMemcpy(d_mm.p, init, sizeof(init), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_dist_range, dim3(std::min<long>(1024, (long)((total + 255) / 256))), dim3(256), 0, 0, (long)total,
                           d_cl.as<long long>(), d_dl.as<float>(), d_mm.as<unsigned>(), d_mm.as<unsigned>() + 1);
        unsigned mm[2];
        MG_HIP(hipMemcpy(mm, d_mm.p, sizeof(mm), hipMemcpyDeviceToHost));
        MG_REQUIRE(mm[0] != 0xffffffffu, "weight_tables: no positive finite distance to derive the default sigma from");
        float fmin, fmax;
        memcpy(&fmin, &mm[0], 4);
        memcpy(&fmax, &mm[1], 4);
        const double def = std::max((double)fmax / 50.0, (double)fmin * 50.0);       // utils.py:222-223, 245-246
        if (!(u_sigma > 0.0)) u_sigma = def;
        if (!(d_sigma > 0.0)) d_sigma = def;
    }
    if (sigmas_out) { sigmas_out[0] = u_sigma; sigmas_out[1] = d_sigma; }
    const dim3 grid((n_nodes + 255) / 256), block(256);
    if (u_ew) {
        MG_REQUIRE(k1 >= 2, "weight_tables: undirected weights need at least one neighbour column");
        hipLaunchKernelGGL(k_raw_weights, grid, block, 0, 0, n_nodes, k1, 1, (float)u_sigma, d_cl.as<long long>(), d_dl.as<float>(),
                           d_w.as<float>(), d_deg.as<float>());
        if (regularized)
            hipLaunchKernelGGL(k_norm_undirected, grid, block, 0, 0, n_nodes, k1, d_cl.as<long long>(), d_deg.as<float>(), d_w.as<float>());
        MG_HIP(hipGetLastError());
        MG_HIP(hipMemcpy(u_ew, d_w.p, sizeof(float) * (size_t)n_nodes * (k1 - 1), hipMemcpyDeviceToHost));
    }
    if (d_ew) {
        hipLaunchKernelGGL(k_raw_weights, grid, block, 0, 0, n_nodes, k1, 0, (float)d_sigma, d_cl.as<long long>(), d_dl.as<float>(),
                           d_w.as<float>(), d_deg.as<float>());
        if (regularized) hipLaunchKernelGGL(k_norm_directed, grid, block, 0, 0, n_nodes, k1, d_deg.as<float>(), d_w.as<float>());
        MG_HIP(hipGetLastError());
        MG_HIP(hipMemcpy(d_ew, d_w.p, sizeof(float) * total, hipMemcpyDeviceToHost));
    }
    return MGADMM_OK;
}

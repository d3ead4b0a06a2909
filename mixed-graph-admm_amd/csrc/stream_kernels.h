// Streaming kernels of the ADMM hot path for gfx950 (wave64).
//
// Device layout ("batch-innermost"): a signal (B,T,N) is held as V[t][i][b], b contiguous, padded to
// Bp columns (a multiple of 64*VEC).  One wavefront owns one graph node row (t,i) over 64*VEC
// consecutive batch columns, so
//   * the node's own row and every neighbour row of the sparse Laplacian are read as one fully
//     coalesced 256 B .. 1 KiB segment per wave instruction (VEC*4 B per lane);
//   * the CSR row (rowptr/col/val) is wave-uniform and is fetched through the scalar cache;
//   * the per-sample CG dot products are per-LANE running sums -- no cross-lane traffic until the
//     fixed-order cross-wave / cross-workgroup reduction (bitwise repeatable, no atomics).
// blockIdx -> work mapping is XCD-aware: consecutive blockIdx values are dealt round-robin over the 8
// XCDs, so block b works on node-block range (b & 7): each XCD sweeps its own eighth of the node
// range of time slice t and then the same eighth of slice t+1, which keeps the rows a time-shifted
// operator (Ldr / Ldr^T) re-reads in that XCD's L2 / the Infinity Cache.
#pragma once
#include <type_traits>

#include "common.h"

template <typename S, int V>
struct alignas(sizeof(S) * V) Vec {
    S v[V];
};

template <typename S, int V>
__device__ __forceinline__ Vec<S, V> ldv(const S* p) {
    return *reinterpret_cast<const Vec<S, V>*>(p);
}
// plain vector store (LDS staging in k_tile)
template <typename S, int V>
__device__ __forceinline__ void stl(S* p, const Vec<S, V>& x) {
    *reinterpret_cast<Vec<S, V>*>(p) = x;
}
// Every vector the streaming kernels write is consumed by a LATER kernel, long after it has left the 4 MiB
// L2 of its XCD: non-temporal stores keep the output from evicting the input slices the time-shifted
// operators and the tile halos re-read (measured on cfg3: SpMM in CG and the vector updates each +1 %).
// Non-temporal LOADS were measured too and rejected: the tile kernel loses 8 %.
template <typename S, int V>
__device__ __forceinline__ void stv(S* p, const Vec<S, V>& x) {
    if constexpr (V == 1) {
        __builtin_nontemporal_store(x.v[0], p);
    } else {
        typedef S vt __attribute__((ext_vector_type(V)));
        vt t;
#pragma unroll
        for (int k = 0; k < V; ++k) t[k] = x.v[k];
        __builtin_nontemporal_store(t, reinterpret_cast<vt*>(p));
    }
}

// the same through a WAVE-UNIFORM base pointer + a 32-bit per-lane byte offset: the scalar-base form of global_load /
// global_store (base in an SGPR pair, one VGPR of offset shared by every row a lane touches) instead of a 64-bit address
// in a VGPR pair per access
template <typename S, int V>
__device__ __forceinline__ Vec<S, V> ldv_u(const S* ubase, unsigned lane_bytes) {
    typedef S vt __attribute__((ext_vector_type(V)));
    const __attribute__((address_space(1))) char* b = (const __attribute__((address_space(1))) char*)ubase;
    Vec<S, V> r;
    if constexpr (V == 1) {
        r.v[0] = *reinterpret_cast<const __attribute__((address_space(1))) S*>(b + lane_bytes);
    } else {
        const vt t = *reinterpret_cast<const __attribute__((address_space(1))) vt*>(b + lane_bytes);
#pragma unroll
        for (int k = 0; k < V; ++k) r.v[k] = t[k];
    }
    return r;
}

template <typename S, int V>
__device__ __forceinline__ void stv_u(S* ubase, unsigned lane_bytes, const Vec<S, V>& x) {
    typedef S vt __attribute__((ext_vector_type(V)));
    __attribute__((address_space(1))) char* b = (__attribute__((address_space(1))) char*)ubase;
    if constexpr (V == 1) {
        __builtin_nontemporal_store(x.v[0], reinterpret_cast<__attribute__((address_space(1))) S*>(b + lane_bytes));
    } else {
        vt t;
#pragma unroll
        for (int k = 0; k < V; ++k) t[k] = x.v[k];
        __builtin_nontemporal_store(t, reinterpret_cast<__attribute__((address_space(1))) vt*>(b + lane_bytes));
    }
}
// where an epilogue finds its element of a row: a flat element offset (k_rows / k_tile), or a wave-uniform element offset
// + the lane's byte offset inside the row (k_cldr: scalar-base accesses)
struct OffSplit {
    size_t u;
    unsigned lb;
};
template <typename S, int V>
__device__ __forceinline__ Vec<S, V> ldo(const S* p, size_t off) { return ldv<S, V>(p + off); }
template <typename S, int V>
__device__ __forceinline__ Vec<S, V> ldo(const S* p, const OffSplit& o) { return ldv_u<S, V>(p + o.u, o.lb); }
template <typename S, int V>
__device__ __forceinline__ void sto(S* p, size_t off, const Vec<S, V>& x) { stv<S, V>(p + off, x); }
template <typename S, int V>
__device__ __forceinline__ void sto(S* p, const OffSplit& o, const Vec<S, V>& x) { stv_u<S, V>(p + o.u, o.lb, x); }

struct Geom {
    int T, N, B, Bp;
    int VEC, CH;         // columns per lane; column chunks of 64*VEC
    int RI;              // rows per work item
    int NX, NBL;         // node rows owned by one XCD (multiple of RI); work items per time slice and XCD
    int n_items;         // T * NBL work items per (XCD, column chunk), enumerated t-major
    int G8;              // workgroups per (XCD, column chunk)
    int P;               // partial-sum rows per column = 8 * G8 (one per workgroup of a chunk)
    int grid;            // 8 * G8 * CH workgroups
    int rev;             // 1: sweep the time slices from T-1 down to 0 (see Engine::cg_internal: the slices a kernel touches
                         //    LAST are the ones still in the 256 MiB Infinity Cache when the next kernel starts)
};

constexpr int CSR_PAD = 8;     // col/val arrays carry this many extra entries (reads past a row end are safe)

// ---------------------------------------------------------------------------------------------
// Band (line-graph) operators and the no-gather case:
//   BAND : l = selfc(t) * in[t][i] - sum_s w[.][s] * in[t -/+ (1+s)][i]      (ADMM.py:153-164, 181-194)
//   NONE : l = in[t][i]
// ---------------------------------------------------------------------------------------------
template <typename S, int VEC>
__device__ __forceinline__ Vec<S, VEC> op_apply_band(const Geom& g, const OpDesc& op, const float* __restrict__ band_w,
                                                     const S* __restrict__ in, int t, int i, int col0,
                                                     const Vec<S, VEC>& self) {
    Vec<S, VEC> sum;
#pragma unroll
    for (int v = 0; v < VEC; ++v) sum.v[v] = S(0);
    S selfc = S(1);
    if (op.kind == OPK_BAND) {
        if (op.band_dir < 0) {
            selfc = (t >= 1) ? S(1) : S(0);
            for (int s = 0; s < op.skip; ++s) {
                const int ts = t - 1 - s;
                if (ts < 0) break;
                const S w = (S)band_w[t * op.skip + s];
                const Vec<S, VEC> nv = ldv<S, VEC>(in + ((size_t)ts * g.N + i) * g.Bp + col0);
#pragma unroll
                for (int v = 0; v < VEC; ++v) sum.v[v] += w * nv.v[v];
            }
        } else {
            selfc = (t > 0) ? S(1) : S(0);
            for (int s = 0; s < op.skip; ++s) {
                const int ts = t + 1 + s;
                if (ts >= g.T) break;
                const S w = (S)band_w[ts * op.skip + s];
                const Vec<S, VEC> nv = ldv<S, VEC>(in + ((size_t)ts * g.N + i) * g.Bp + col0);
#pragma unroll
                for (int v = 0; v < VEC; ++v) sum.v[v] += w * nv.v[v];
            }
        }
    }
    Vec<S, VEC> l;
#pragma unroll
    for (int v = 0; v < VEC; ++v) l.v[v] = selfc * self.v[v] - sum.v[v];
    return l;
}

// wave-uniform description of one CSR row: bounds + the first GW (column, weight) pairs;
// slots past the row end are redirected to the row's own node with weight 0
template <int GW>
struct RowMeta {
    int e0, e1;
    int c[GW];
    float w[GW];
};

__device__ __forceinline__ void row_bounds(const int* __restrict__ rowptr, int i, int& e0, int& e1) {
    e0 = rowptr[i];
    e1 = rowptr[i + 1];
}

template <int GW>
__device__ __forceinline__ void row_entries(const int* __restrict__ colidx, const float* __restrict__ val, int e0, int e1,
                                            int iself, RowMeta<GW>& m) {
    m.e0 = e0;
    m.e1 = e1;
#pragma unroll
    for (int u = 0; u < GW; ++u) {
        const int cc = colidx[e0 + u];     // unconditional: arrays are padded by CSR_PAD entries
        const float ww = val[e0 + u];
        const bool ok = e0 + u < e1;
        m.c[u] = ok ? cc : iself;
        m.w[u] = ok ? ww : 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------
// Row kernel.  Work item = RI consecutive node rows of one time slice.  The grid is persistent-sized
// (8*G8 workgroups per column chunk).  XCD x (= blockIdx & 7, the round-robin dispatch group) owns the
// SAME contiguous node range [x*NX, (x+1)*NX) in every time slice, and its G8 workgroups sweep the items of
// that range t-major (all node blocks of slice 0, then slice 1, ...).  So (a) the neighbour rows of a
// spatial gather are rows this XCD's own L2 has just fetched, and (b) the rows a time-shifted operator
// (Ldr / Ldr^T) gathers from slice t -/+ 1 are rows the same XCD streamed one local slice (NX rows) earlier:
// they are served by its 4 MiB L2 (or the Infinity Cache) instead of being fetched from HBM a second time.
// Each wave owns rows (wave, wave+4, ...) of an item over 64*VEC batch columns; CSR metadata of the
// next row is prefetched through the scalar cache while the gathers of the current row are in flight.
// partials: [NRED][P][Bp] per-workgroup per-column partial sums (summed by k_reduce in fixed order).
// live: optional device flag; a zero value makes the launch a no-op (speculatively enqueued CG
// iterations after every sample has converged).
// ---------------------------------------------------------------------------------------------
template <class E, class = void>
struct has_fetch : std::false_type {};
template <class E>
struct has_fetch<E, std::void_t<typename E::Ops>> : std::true_type {};

template <typename S, int VEC, class Epi, int GW>
__global__ __launch_bounds__(256) void k_rows(Geom g, OpDesc op, const int* __restrict__ rowptr,
                                              const int* __restrict__ colidx, const float* __restrict__ val,
                                              const float* __restrict__ band_w, const S* __restrict__ in, Epi epi_in,
                                              S* __restrict__ partials, const int* __restrict__ live) {
    if (live != nullptr && *live == 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int xcd = blockIdx.x & 7;
    const int q = blockIdx.x >> 3;
    const int chunk = q % g.CH;
    const int r = q / g.CH;
    const int col0 = (chunk * 64 + lane) * VEC;
    const int slot = xcd * g.G8 + r;       // this workgroup's position inside a round / partial row

    Epi epi = epi_in;
    epi.begin(col0);
    constexpr int NR = Epi::NRED > 0 ? Epi::NRED : 1;
    S acc[NR][VEC];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[rr][v] = S(0);

    const bool spatial = op.kind == OPK_SPATIAL;
    const int xlo = xcd * g.NX, xhi = min(g.N, xlo + g.NX);   // node rows this XCD owns, for every time slice
    for (int item = r; item < g.n_items; item += g.G8) {
        const int tq = item / g.NBL;
        const int nb = item - tq * g.NBL;
        const int t = g.rev ? g.T - 1 - tq : tq;
        const int n0 = xlo + nb * g.RI;
        const int n1 = min(xhi, n0 + g.RI);
        int i = n0 + wave;
        if (i >= n1) continue;
        if (spatial) {
            const int ts = t + op.shift;
            const bool tvalid = ts >= 0 && ts < g.T;
            S selfc = S(1);
            if (op.self_mode == SELF_LDR) selfc = (t >= 1) ? S(1) : S(0);
            else if (op.self_mode == SELF_LDRT) selfc = (t > 0 || op.q1) ? S(1) : S(0);
            else if (op.self_mode == SELF_LN_FATHER) selfc = (t + 1 < g.T) ? S(1) : S(0);
            const S selft = selfc;
            const S* gbase = in + (size_t)(tvalid ? ts : t) * g.N * g.Bp + col0;   // slice the gathers read
            const S* obase = in + (size_t)t * g.N * g.Bp + col0;                  // slice of the rows we own
            RowMeta<GW> cur, nxt;
            int be0, be1, ne0, ne1;
            row_bounds(rowptr, i, be0, be1);
            if (!tvalid) be1 = be0;
            row_entries<GW>(colidx, val, be0, be1, i, cur);
            row_bounds(rowptr, min(i + 4, g.N - 1), ne0, ne1);
            if (!tvalid) ne1 = ne0;
            for (; i < n1; i += 4) {
                const size_t roff = (size_t)i * g.Bp;
                if (op.self_w) selfc = selft * (S)op.self_w[i];     // apply_op_Ln: row sum of d_ew as self coefficient
                Vec<S, VEC> sum;
#pragma unroll
                for (int v = 0; v < VEC; ++v) sum.v[v] = S(0);
                if (cur.e1 - cur.e0 > GW) {          // entries past the fixed gather window (rare rows)
                    for (int e = cur.e0 + GW; e < cur.e1; ++e) {
                        const Vec<S, VEC> x = ldv<S, VEC>(gbase + (size_t)colidx[e] * g.Bp);
                        const S w = (S)val[e];
#pragma unroll
                        for (int v = 0; v < VEC; ++v) sum.v[v] += w * x.v[v];
                    }
                }
                // own row + GW neighbour rows in flight together; pad slots re-read the own node (weight 0)
                const Vec<S, VEC> self = ldv<S, VEC>(obase + roff);
                Vec<S, VEC> nv[GW];
#pragma unroll
                for (int u = 0; u < GW; ++u) nv[u] = ldv<S, VEC>(gbase + (size_t)cur.c[u] * g.Bp);
                __builtin_amdgcn_sched_barrier(0);   // keep the loads issued up here
                // scalar prefetch for the rows this wave handles next (one and two steps ahead)
                row_entries<GW>(colidx, val, ne0, ne1, min(i + 4, g.N - 1), nxt);
                int fe0, fe1;
                row_bounds(rowptr, min(i + 8, g.N - 1), fe0, fe1);
                if (!tvalid) fe1 = fe0;
#pragma unroll
                for (int u = 0; u < GW; ++u)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) sum.v[v] += (S)cur.w[u] * nv[u].v[v];
                Vec<S, VEC> l;
#pragma unroll
                for (int v = 0; v < VEC; ++v) l.v[v] = selfc * self.v[v] - sum.v[v];
                epi.row(t, (size_t)t * g.N * g.Bp + roff + col0, self, l, acc);
                cur = nxt;
                ne0 = fe0;
                ne1 = fe1;
            }
        } else {
            // element-wise epilogue with a fetch/row_ops split and no operator: the loads of up to ROWS_IN_FLIGHT
            // rows (input row + operand rows) are all issued before the first store (the stores of row k may
            // alias the loads of row k+1 as far as the compiler knows, so it would not overlap them by itself).
            // Row order and arithmetic are unchanged; worth < 1 % (the kernels are bandwidth-bound).
            bool done = false;
            if constexpr (has_fetch<Epi>::value) {
                if (op.kind == OPK_NONE) {
                    constexpr int ROWS_IN_FLIGHT = 4;
                    for (; i < n1; i += 4 * ROWS_IN_FLIGHT) {
                        Vec<S, VEC> self[ROWS_IN_FLIGHT];
                        typename Epi::Ops ops[ROWS_IN_FLIGHT];
                        size_t off[ROWS_IN_FLIGHT];
#pragma unroll
                        for (int k = 0; k < ROWS_IN_FLIGHT; ++k) {
                            const int ik = min(i + 4 * k, n1 - 1);      // clamped: a short item re-reads its last row
                            off[k] = ((size_t)t * g.N + ik) * g.Bp + col0;
                            self[k] = ldv<S, VEC>(in + off[k]);
                            ops[k] = epi.fetch(off[k]);
                        }
#pragma unroll
                        for (int k = 0; k < ROWS_IN_FLIGHT; ++k)
                            if (i + 4 * k < n1) epi.row_ops(t, off[k], self[k], self[k], acc, ops[k]);
                    }
                    done = true;
                }
            }
            if (!done) {
                for (; i < n1; i += 4) {
                    const size_t off = ((size_t)t * g.N + i) * g.Bp + col0;
                    const Vec<S, VEC> self = ldv<S, VEC>(in + off);
                    const Vec<S, VEC> l = op_apply_band<S, VEC>(g, op, band_w, in, t, i, col0, self);
                    epi.row(t, off, self, l, acc);
                }
            }
        }
    }

    if (Epi::NRED > 0) {
        __shared__ S sm[3][VEC][64];
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            if (wave > 0) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) sm[wave - 1][v][lane] = acc[rr][v];
            }
            __syncthreads();
            if (wave == 0) {
                Vec<S, VEC> o;
#pragma unroll
                for (int v = 0; v < VEC; ++v) o.v[v] = ((acc[rr][v] + sm[0][v][lane]) + sm[1][v][lane]) + sm[2][v][lane];
                stv<S, VEC>(partials + ((size_t)rr * g.P + slot) * g.Bp + col0, o);
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LDS-tiled row kernel for the spatial operators (Lu, Ldr, Ldr^T) on a cluster-ordered graph.
//
// A workgroup owns a TILE of R consecutive node rows (a graph cluster after mg_cluster_order) of one
// column chunk and sweeps the T time slices in order (ascending for shift <= 0, descending for Ldr^T).
// The tile's rows of the slice the gathers read (t + shift: the slice the workgroup streamed in its
// previous step, or the current one for Lu) are kept in LDS, so an in-tile neighbour costs one
// conflict-free ds_read_b128 instead of a trip through L1/L2, and every slice is fetched from HBM once
// instead of twice (once as "own" rows, once as gathered rows).  The rows of out-of-tile neighbours (the
// halo: 7.6 rows per 8-row tile on the 10k-node kNN graph, up to TILE_HMAX) are staged in LDS once per step;
// consecutive tiles run on the same XCD, so those loads often hit that XCD's L2.  The own rows of step s+1
// are requested while step s computes; the epilogue functors are the ones of k_rows.  The kernel is
// latency-bound (VALU ~25 %, LDS ~15-30 % busy, waits > 50 %): small tiles / many resident workgroups win.
// ---------------------------------------------------------------------------------------------
struct TileGeom {
    int T, N, B, Bp;
    int VEC, CH;
    int R;        // node rows per tile (<= 4 * TILE_MAXR)
    int NTILE;    // tiles per column chunk
    int TPX;      // tile slots per XCD = ceil(NTILE / 8)
    int P;        // partial rows per column = 8 * TPX
    int grid;     // 8 * TPX * CH workgroups
    int lds_bytes;
};
constexpr int TILE_MAXR = 8;    // rows per wave and time step (4 waves -> R <= 32)
constexpr int TILE_GW_MAX = 8;  // neighbour slots per row held in VGPR lanes: 4 (W_u), 6 (W_d) or 8 (W_d^T); MAXR*GW <= 64
constexpr int TILE_HMAX = 20;   // halo rows (out-of-tile neighbours) staged in LDS per tile and step
constexpr int TILE_HPW = TILE_HMAX / 4;   // halo rows loaded by one wave

// Per-(matrix, R) metadata built on the host (Engine::tile_meta):
//   tl_col/tl_w [N][GW]      : the row's first GW "local" neighbours and weights.  Local index
//                              0..R-1 = row of the own tile, R..R+H-1 = position in the tile's halo list;
//                              unused slots point at the row itself with weight 0
//   halo [NTILE][TILE_HMAX]  : global row indices of the tile's halo rows (-1 = unused)
//   h_rowptr/h_col/h_val     : CSR of everything that did not fit (more than GW neighbours, more than
//                              TILE_HMAX halo rows): gathered from global memory, normally empty
struct TileMeta {
    const int* tl_col;
    const float* tl_w;
    const int* halo;
    const int* h_rowptr;
    const int* h_col;
    const float* h_val;
};

// MR = rows per wave (tile = 4*MR rows), a compile-time constant: the row loops are straight-line code (the
// LDS reads of one row overlap the FMAs of the previous one, no per-row branches) and the register arrays
// hold exactly MR rows.  Rows past the end of a short last tile are clamped to the tile's last row for their
// loads and gathers and skip the epilogue.
// Where the vector the operator is applied to comes from (as for k_cldr):
//   TileSrcPlain : x = in.
//   TileSrcFold  : Lu only (shift 0).  The CG direction is formed ON LOAD, x = p_new = r + beta p_old (ADMM.py:366; `in` is r),
//                  for the own rows and for the halo rows; the owner of a row also stores p_new and applies the deferred
//                  x += alpha p_old (ADMM.py:352): the vector-update kernel of the previous CG iteration of the zu solve
//                  (EpiPUpdate, 20 B/element) disappears into this launch.  p_new goes to a second buffer: other
//                  workgroups still read p_old for their halos.
template <typename S, int VEC>
struct TileSrcPlain {
    static constexpr bool FOLD = false;
};
template <typename S, int VEC>
struct TileSrcFold {
    static constexpr bool FOLD = true;
    const S* p_old;
    S* p_new;
    S* x;
    const S* alpha;
    const S* beta;
};

template <typename S, int VEC, class Epi, int TILE_GW, int MR, class Src = TileSrcPlain<S, VEC>>
__global__ __launch_bounds__(256) void k_tile(TileGeom g, OpDesc op, const int* __restrict__ tl_col,
                                              const float* __restrict__ tl_w, const int* __restrict__ halo,
                                              const int* __restrict__ h_rowptr, const int* __restrict__ h_col,
                                              const float* __restrict__ h_val, const S* __restrict__ in, Epi epi_in,
                                              S* __restrict__ partials, const int* __restrict__ live, Src src = Src()) {
    static_assert(MR * TILE_GW <= 64 && MR <= TILE_MAXR, "per-row metadata must fit the 64 lanes of a wave");
    extern __shared__ __align__(16) unsigned char tile_raw[];
    if (live != nullptr && *live == 0) return;
    constexpr int W = 64 * VEC;
    constexpr int R = 4 * MR;
    S* tile = reinterpret_cast<S*>(tile_raw);          // [R + TILE_HMAX][W]: own-tile rows, then halo rows
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int xcd = blockIdx.x & 7;
    const int q = blockIdx.x >> 3;
    const int chunk = q % g.CH;
    const int r = q / g.CH;
    const int tidx = xcd * g.TPX + r;                  // consecutive tiles share an XCD (halo rows hit its L2)
    const int col0 = (chunk * 64 + lane) * VEC;
    const bool tile_ok = tidx < g.NTILE;               // the grid is padded to 8 * TPX tiles
    const int n0 = tile_ok ? tidx * R : 0;
    const int nlast = tile_ok ? min(g.N, n0 + R) - 1 : 0;
    int rowi[MR];                                      // global row of local row wave + 4*j (clamped in a short tile)
    bool rowok[MR];
#pragma unroll
    for (int j = 0; j < MR; ++j) {
        rowok[j] = tile_ok && (n0 + wave + 4 * j <= nlast);
        rowi[j] = min(n0 + wave + 4 * j, nlast);
    }

    // wave-resident metadata: lane (j*TILE_GW + u) holds slot u of row j; lane j also holds the overflow
    // bounds of row j; lane k < TILE_HPW holds the global index of halo row (wave + 4*k) of this tile
    int mcol = 0, mw = 0, hs = 0, hc = 0, hrow = -1;
    {
        const int j = lane / TILE_GW, u = lane - j * TILE_GW;
        if (j < MR) {
            const int i = min(n0 + wave + 4 * j, nlast);
            mcol = tl_col[(size_t)i * TILE_GW + u];
            mw = __float_as_int(tl_w[(size_t)i * TILE_GW + u]);
        }
        if (lane < MR) {
            const int i = min(n0 + wave + 4 * lane, nlast);
            hs = h_rowptr[i];
            hc = h_rowptr[i + 1] - hs;
        }
        if (lane < TILE_HPW && tile_ok) hrow = halo[(size_t)tidx * TILE_HMAX + wave + 4 * lane];
    }

    Epi epi = epi_in;
    epi.begin(col0);
    constexpr int NR = Epi::NRED > 0 ? Epi::NRED : 1;
    S acc[NR][VEC];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[rr][v] = S(0);

    if (tile_ok) {
        const int shift = op.shift;
        const S* trow = tile + lane * VEC;
        S* hdst = tile + (size_t)R * W + lane * VEC;
        auto step_t = [&](int sidx) { return shift > 0 ? g.T - 1 - sidx : sidx; };
        S fa[VEC], fb[VEC];             // FOLD: alpha, beta of this lane's columns
#pragma unroll
        for (int v = 0; v < VEC; ++v) fa[v] = fb[v] = S(0);
        if constexpr (Src::FOLD) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                fa[v] = src.alpha[col0 + v];
                fb[v] = src.beta[col0 + v];
            }
        }
        // FOLD: p_new = r + beta p_old of any row (halo rows, overflow neighbours): nothing stored
        auto folded = [&](size_t off) {
            Vec<S, VEC> rv = ldv<S, VEC>(in + off);
            if constexpr (Src::FOLD) {
                const Vec<S, VEC> po = ldv<S, VEC>(src.p_old + off);
#pragma unroll
                for (int v = 0; v < VEC; ++v) rv.v[v] = fma(fb[v], po.v[v], rv.v[v]);
            }
            return rv;
        };
        constexpr int MRF = Src::FOLD ? MR : 1;
        // FOLD: an own row becomes p_new; its owner stores p_new and x += alpha p_old (same roundings as EpiPUpdate)
        auto own_combine = [&](int tt, int j, const Vec<S, VEC>& rv, const Vec<S, VEC>& po, const Vec<S, VEC>& xo) {
            Vec<S, VEC> pv = rv;
            if constexpr (Src::FOLD) {
                Vec<S, VEC> xv;
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    xv.v[v] = fma(fa[v], po.v[v], xo.v[v]);
                    pv.v[v] = fma(fb[v], po.v[v], rv.v[v]);
                }
                if (rowok[j]) {
                    const size_t off = ((size_t)tt * g.N + rowi[j]) * g.Bp + col0;
                    stv<S, VEC>(src.x + off, xv);
                    stv<S, VEC>(src.p_new + off, pv);
                }
            }
            return pv;
        };
        // own rows of the first step; afterwards the own rows of step s+1 are requested while step s computes
        Vec<S, VEC> own[MR];
        {
            const size_t ob = (size_t)step_t(0) * g.N * g.Bp + col0;
#pragma unroll
            for (int j = 0; j < MR; ++j) {
                const size_t off = ob + (size_t)rowi[j] * g.Bp;
                const Vec<S, VEC> rv = ldv<S, VEC>(in + off);
                if constexpr (Src::FOLD) own[j] = own_combine(step_t(0), j, rv, ldv<S, VEC>(src.p_old + off), ldv<S, VEC>(src.x + off));
                else own[j] = rv;
            }
        }
        for (int step = 0; step < g.T; ++step) {
            const int t = step_t(step);
            const int ts = t + shift;
            const bool tvalid = ts >= 0 && ts < g.T;
            S selfc = S(1);
            if (op.self_mode == SELF_LDR) selfc = (t >= 1) ? S(1) : S(0);
            else if (op.self_mode == SELF_LDRT) selfc = (t > 0 || op.q1) ? S(1) : S(0);
            // 1. requests of this step, in the order they are needed: halo rows of the gathered slice, the
            //    epilogue's operand rows, and the own rows of the NEXT step
            // (halo slots in two branch-free batches -- the first two slots of the wave, then the other three if the wave has
            // any: a slot that is not in use reads the tile's first row and is never gathered.  With a test per slot every load
            // sat in its own basic block behind an s_waitcnt vmcnt(0): the halo rows left one after the other, see k_cldr)
            Vec<S, VEC> hv[TILE_HPW];
            if (tvalid) {
                const size_t hb = (size_t)ts * g.N * g.Bp + col0;
#pragma unroll
                for (int k = 0; k < 2 && k < TILE_HPW; ++k) {
                    const int hr = __builtin_amdgcn_readlane(hrow, k);
                    hv[k] = folded(hb + (size_t)(hr >= 0 ? hr : n0) * g.Bp);
                }
                if (__builtin_amdgcn_readlane(hrow, 2 < TILE_HPW ? 2 : 0) >= 0) {
#pragma unroll
                    for (int k = 2; k < TILE_HPW; ++k) {
                        const int hr = __builtin_amdgcn_readlane(hrow, k);
                        hv[k] = folded(hb + (size_t)(hr >= 0 ? hr : n0) * g.Bp);
                    }
                }
            }
            Vec<S, VEC> pre[MR];
            if constexpr (Epi::HAS_PRE) {
#pragma unroll
                for (int j = 0; j < MR; ++j) pre[j] = epi.pre(((size_t)t * g.N + rowi[j]) * g.Bp + col0);
            }
            Vec<S, VEC> ownn[MR], pon[MRF], xon[MRF];      // FOLD: r, p_old, x of the next step's own rows
            {
                const size_t nb = (size_t)step_t(min(step + 1, g.T - 1)) * g.N * g.Bp + col0;
#pragma unroll
                for (int j = 0; j < MR; ++j) {
                    const size_t off = nb + (size_t)rowi[j] * g.Bp;
                    ownn[j] = ldv<S, VEC>(in + off);
                    if constexpr (Src::FOLD) {
                        pon[j] = ldv<S, VEC>(src.p_old + off);
                        xon[j] = ldv<S, VEC>(src.x + off);
                    }
                }
            }
            if (tvalid) {
#pragma unroll
                for (int k = 0; k < 2 && k < TILE_HPW; ++k) stl<S, VEC>(hdst + (size_t)(wave + 4 * k) * W, hv[k]);
                if (__builtin_amdgcn_readlane(hrow, 2 < TILE_HPW ? 2 : 0) >= 0) {
#pragma unroll
                    for (int k = 2; k < TILE_HPW; ++k) stl<S, VEC>(hdst + (size_t)(wave + 4 * k) * W, hv[k]);
                }
            }
            if (shift == 0) {                               // Lu gathers from the slice of this step
#pragma unroll
                for (int j = 0; j < MR; ++j) stl<S, VEC>(tile + (size_t)(wave + 4 * j) * W + lane * VEC, own[j]);
            }
            __syncthreads();                                // halo (and for Lu the own rows) visible
            // 2. gathers + epilogue, row by row: every regular neighbour is a conflict-free LDS row read
#pragma unroll
            for (int j = 0; j < MR; ++j) {
                Vec<S, VEC> sum;
#pragma unroll
                for (int v = 0; v < VEC; ++v) sum.v[v] = S(0);
                if (tvalid) {
                    // neighbour rows in batches of at most 4 LDS reads (keeps the register footprint at 4 rows:
                    // one more resident workgroup per CU than with all TILE_GW reads in flight)
                    constexpr int GB = TILE_GW > 4 ? TILE_GW / 2 : TILE_GW;
#pragma unroll
                    for (int u0 = 0; u0 < TILE_GW; u0 += GB) {
                        Vec<S, VEC> nv[GB];
#pragma unroll
                        for (int u = 0; u < GB; ++u) {
                            const int lc = __builtin_amdgcn_readlane(mcol, j * TILE_GW + u0 + u);
                            nv[u] = ldv<S, VEC>(trow + (size_t)lc * W);
                        }
#pragma unroll
                        for (int u = 0; u < GB; ++u) {
                            const S w = (S)__int_as_float(__builtin_amdgcn_readlane(mw, j * TILE_GW + u0 + u));
#pragma unroll
                            for (int v = 0; v < VEC; ++v) sum.v[v] += w * nv[u].v[v];
                        }
                        if (TILE_GW > GB) __builtin_amdgcn_sched_barrier(0);   // do not merge the batches again
                    }
                    const int hcount = __builtin_amdgcn_readlane(hc, j);   // overflow entries (normally 0)
                    if (hcount > 0) {
                        const int hstart = __builtin_amdgcn_readlane(hs, j);
                        for (int e = hstart; e < hstart + hcount; ++e) {
                            const Vec<S, VEC> x = folded((size_t)(tvalid ? ts : t) * g.N * g.Bp + col0 + (size_t)h_col[e] * g.Bp);
                            const S w = (S)h_val[e];
#pragma unroll
                            for (int v = 0; v < VEC; ++v) sum.v[v] += w * x.v[v];
                        }
                    }
                }
                Vec<S, VEC> l;
#pragma unroll
                for (int v = 0; v < VEC; ++v) l.v[v] = selfc * own[j].v[v] - sum.v[v];
                const size_t off = ((size_t)t * g.N + rowi[j]) * g.Bp + col0;
                // rows past the end of a short last tile were clamped to its last row for the loads; their
                // epilogue must not run (in-place epilogues such as the gamma / phi updates are not idempotent)
                if (rowok[j]) {
                    if constexpr (Epi::HAS_PRE) epi.row_pre(t, off, own[j], l, acc, pre[j]);
                    else epi.row(t, off, own[j], l, acc);
                }
            }
            __syncthreads();                                // every gather of this step is done
            // 3. this step's rows become the gathered slice of the next step
            if (shift != 0) {
#pragma unroll
                for (int j = 0; j < MR; ++j) stl<S, VEC>(tile + (size_t)(wave + 4 * j) * W + lane * VEC, own[j]);
            }
            if constexpr (Src::FOLD) {
                if (step + 1 < g.T) {
#pragma unroll
                    for (int j = 0; j < MR; ++j) own[j] = own_combine(step_t(step + 1), j, ownn[j], pon[j], xon[j]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < MR; ++j) own[j] = ownn[j];
            }
        }
    }

    if (Epi::NRED > 0) {
        __syncthreads();                                            // last step's tile stores are done
        S(*sm)[VEC][64] = reinterpret_cast<S(*)[VEC][64]>(tile);   // the tile is dead now: reuse it
        const int slot = tidx;
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            if (wave > 0) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) sm[wave - 1][v][lane] = acc[rr][v];
            }
            __syncthreads();
            if (wave == 0) {
                Vec<S, VEC> o;
#pragma unroll
                for (int v = 0; v < VEC; ++v) o.v[v] = ((acc[rr][v] + sm[0][v][lane]) + sm[1][v][lane]) + sm[2][v][lane];
                stv<S, VEC>(partials + ((size_t)rr * g.P + slot) * g.Bp + col0, o);
            }
            __syncthreads();
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Fused cLdr kernel: l = Ldr^T(Ldr x) in ONE pass over x (reference ADMM.py:225-228; the operator inside LHS_x and
// LHS_zd, i.e. inside two of the three CG solves of every ADMM iteration).  The two-pass form writes q = Ldr x to
// HBM and reads it back (8 B/element) and fetches the halo of both operators; here q never leaves the CU.
//
// A workgroup (NW waves) owns a tile of up to NW*MA consecutive node rows (one cluster of the node order) of one
// 64*VECT-column chunk and sweeps t = 0 .. T-1.  For the step t it holds in LDS
//     P = x_t      on C2 = C1 u {W_d neighbours of C1}      (2-hop set, <= NW*MP rows)
//     Q = q_{t+1}  on C1 = tile u {W_d^T neighbours of it}  (1-hop set, <= NW*MQ rows), q = Ldr x recomputed on the halo
// and in registers x_{t+1} of its C2 rows (requested one step ahead; two steps ahead was measured: 24 more VGPRs, 5 %
// slower), x_t and q_t of its own rows:
//     phase A   q_{t+1}[j] = x_{t+1}[j] - sum_e W_d[j,e] P[col_e]            for its rows j of C1   -> Q
//     barrier
//     phase C   l_t[i] = [t>0 or q1] q_t[i] - sum_e W_d^T[i,e] Q[col_e]      for its own rows i     -> epilogue
//               P <- x_{t+1};  request x_{t+2}
//     barrier
// Local row l of a tile belongs to wave l % NW for all three roles, so the self terms are already in that wave's
// registers.  Per-row (local column, weight) slots live in VGPR lanes and are broadcast with v_readlane exactly like
// in k_tile (tables built on the host by build_cldr_tiles, cldr_tiles.h, which has a CPU replay test).  Sums run in
// CSR entry order and q is rounded to S like the stored q of the two-pass form: the result is BITWISE the same.
// ---------------------------------------------------------------------------------------------
struct CldrGeom {
    int T, N, B, Bp;
    int CH;           // column chunks of 64*VECT
    int NT;           // tiles
    int TPX;          // tile slots per XCD = ceil(NT / 8)
    int P;            // partial rows per column = 8 * TPX
    int grid;         // 8 * TPX * CH workgroups
    int tile_major;   // 1: consecutive workgroups of an XCD take consecutive TILES of one chunk (shared halos stay hot in
                      //    its L2); 0: consecutive workgroups take the CHUNKS of one tile (whole rows are consumed together)
    int q1;           // quirk Q1 self coefficient at t = 0 (immaterial: q_0 = 0)
    int lds_bytes;
};
struct CldrMeta {
    const int* n0;     // [NT+1]
    const int* nC;     // [NT][2]  |C1|, |C2|
    const int* rows;   // [NT][C2cap]
    const int* dcol;   // [NT][C1cap][GD]
    const float* dw;
    const int* dcnt;   // [NT][C1cap]
    const int* tcol;   // [NT][Rcap][GT]
    const float* tw;
    const int* tcnt;   // [NT][Rcap]
};

// sum_e w_e * IMG[col_e] for one row whose slots sit in lanes [slot0, slot0 + G) of the metadata registers.  The first
// GFIX slots are read unconditionally and TOGETHER (pad slots carry weight 0): one LDS round trip for a typical row and
// no branch, so the scheduler can overlap the rows of a phase; pairs of slots past GFIX are skipped when the row has
// no entry there (wave-uniform).  The sum runs in slot (= CSR entry) order.
template <typename S, int VECT, int GFIX, int G, int K>
__device__ __forceinline__ Vec<S, VECT> cldr_gather(const S* __restrict__ img_lane, const int (&mc)[K], const int (&mw)[K], int slot0, int cnt) {
    Vec<S, VECT> sum;
#pragma unroll
    for (int v = 0; v < VECT; ++v) sum.v[v] = S(0);
    {
        Vec<S, VECT> nv[GFIX];
#pragma unroll
        for (int u = 0; u < GFIX; ++u) {
            const int s = slot0 + u;
            nv[u] = ldv<S, VECT>(img_lane + __builtin_amdgcn_readlane(mc[s >> 6], s & 63));
        }
#pragma unroll
        for (int u = 0; u < GFIX; ++u) {
            const int s = slot0 + u;
            const S w = (S)__int_as_float(__builtin_amdgcn_readlane(mw[s >> 6], s & 63));
#pragma unroll
            for (int v = 0; v < VECT; ++v) sum.v[v] = fma(w, nv[u].v[v], sum.v[v]);   // explicit: one rounding, like k_tile
        }
    }
#pragma unroll
    for (int u0 = GFIX; u0 < G; u0 += 2) {
        if (cnt > u0) {
            Vec<S, VECT> nv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u0 + u < G) {
                    const int s = slot0 + u0 + u;
                    nv[u] = ldv<S, VECT>(img_lane + __builtin_amdgcn_readlane(mc[s >> 6], s & 63));
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u0 + u < G) {
                    const int s = slot0 + u0 + u;
                    const S w = (S)__int_as_float(__builtin_amdgcn_readlane(mw[s >> 6], s & 63));
#pragma unroll
                    for (int v = 0; v < VECT; ++v) sum.v[v] = fma(w, nv[u].v[v], sum.v[v]);
                }
            }
        }
    }
    return sum;
}

// Where the vector the operator is applied to comes from.
//   CldrSrcPlain : x = in.
//   CldrSrcFold  : the CG direction is formed ON LOAD, x = p_new = r + beta p_old (ADMM.py:366), for every row of C2 (the
//                  halo rows are recomputed like q); the owner of a row also stores p_new and applies the deferred
//                  x += alpha p_old (ADMM.py:352) -- the whole vector-update kernel of the previous CG iteration
//                  (EpiPUpdate, 20 B/element) disappears into this launch.  p_new goes to a SECOND buffer: other
//                  workgroups still read p_old for their halos.
template <typename S, int VECT>
struct CldrSrcPlain {
    static constexpr bool FOLD = false;
    const S* in;
};
template <typename S, int VECT>
struct CldrSrcFold {
    static constexpr bool FOLD = true;
    const S* in;        // r
    const S* p_old;
    S* p_new;
    S* x;
    const S* alpha;
    const S* beta;
};

// Request schedule of k_cldr, compile-time switches of the measurements in DESIGN.md section 3b ("Round 3"); the defaults are
// the measured best (make EXTRA=-DMG_CLDR_BATCH=2 builds a variant):
//   MG_CLDR_BATCH  0: all rows of a step at once   1: own rows, wait, halo rows (default)   2: pair by pair, a wait after each
//                  3: own rows, then the halo rows pair by pair
//   MG_CLDR_EARLY  1: P image stored and the next rows requested at the START of phase C instead of its end
#ifndef MG_CLDR_EARLY
#define MG_CLDR_EARLY 0
#endif
#ifndef MG_CLDR_BATCH
#define MG_CLDR_BATCH 1
#endif
// MINW: waves per SIMD the register allocation must leave room for (= resident workgroups per CU * NW / 4)
template <typename S, int VECT, class Epi, class Src, int NW, int MA, int MQ, int MP, int GD, int GT, int MINW>
__global__ __launch_bounds__(NW * 64, MINW) void k_cldr(CldrGeom g, CldrMeta m, Src src, Epi epi_in,
                                                  S* __restrict__ partials, const int* __restrict__ live) {
    static_assert(MA <= MQ && MQ <= MP && MP <= 64, "own rows are a prefix of C1, C1 a prefix of C2");
    extern __shared__ __align__(16) unsigned char cldr_raw[];
    if (live != nullptr && *live == 0) return;
    constexpr int W = 64 * VECT;
    constexpr int RCAP = NW * MA, C1CAP = NW * MQ, C2CAP = NW * MP;
    constexpr int KD = (MQ * GD + 63) / 64, KT = (MA * GT + 63) / 64;
    constexpr int GDF = GD < 6 ? GD : 6, GTF = GT < 6 ? GT : (GT > 12 ? 4 : 6);   // slots gathered without a test (k + 1 = 5 entries per W_d row; 78 % of the W_d^T rows have <= 6)
    S* Pimg = reinterpret_cast<S*>(cldr_raw);          // [C2CAP][W]
    S* Qimg = Pimg + (size_t)C2CAP * W;                // [C1CAP][W]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int xcd = blockIdx.x & 7;
    const int qq = blockIdx.x >> 3;
    int chunk, r;
    if (g.tile_major) { r = qq % g.TPX; chunk = qq / g.TPX; }
    else { chunk = qq % g.CH; r = qq / g.CH; }
    const int tidx = xcd * g.TPX + r;
    const bool tile_ok = tidx < g.NT;                  // the grid is padded to 8 * TPX tiles
    const int col0 = (chunk * 64 + lane) * VECT;

    int n0 = 0, R = 0, nC1 = 0;
    int hrow = -1, dcn = 0, tcn = 0;
    int mdc[KD], mdw[KD], mtc[KT], mtw[KT];
#pragma unroll
    for (int k = 0; k < KD; ++k) mdc[k] = mdw[k] = 0;
#pragma unroll
    for (int k = 0; k < KT; ++k) mtc[k] = mtw[k] = 0;
    if (tile_ok) {
        n0 = m.n0[tidx];
        R = m.n0[tidx + 1] - n0;
        nC1 = m.nC[2 * tidx];
        if (lane < MP) hrow = m.rows[(size_t)tidx * C2CAP + wave + NW * lane];       // -1 past |C2|
        if (lane < MQ) dcn = m.dcnt[(size_t)tidx * C1CAP + wave + NW * lane];
        if (lane < MA) tcn = m.tcnt[(size_t)tidx * RCAP + wave + NW * lane];
#pragma unroll
        for (int k = 0; k < KD; ++k) {
            const int s = k * 64 + lane, mm = s / GD, u = s - mm * GD;
            if (mm < MQ) {
                const size_t o = ((size_t)tidx * C1CAP + wave + NW * mm) * GD + u;
                mdc[k] = m.dcol[o] * W;
                mdw[k] = __float_as_int(m.dw[o]);
            }
        }
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const int s = k * 64 + lane, mm = s / GT, u = s - mm * GT;
            if (mm < MA) {
                const size_t o = ((size_t)tidx * RCAP + wave + NW * mm) * GT + u;
                mtc[k] = m.tcol[o] * W;
                mtw[k] = __float_as_int(m.tw[o]);
            }
        }
    }

    Epi epi = epi_in;
    epi.begin(col0);
    constexpr int NR = Epi::NRED > 0 ? Epi::NRED : 1;
    S acc[NR][VECT];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr)
#pragma unroll
        for (int v = 0; v < VECT; ++v) acc[rr][v] = S(0);

    if (tile_ok) {
        const S* pl = Pimg + lane * VECT;
        const S* ql = Qimg + lane * VECT;
        const size_t slice = (size_t)g.N * g.Bp;
        constexpr int MPF = Src::FOLD ? MP : 1, MAF = Src::FOLD ? MA : 1;
        Vec<S, VECT> pn[MP];            // x_{t+1} of this wave's rows of C2 (requested one step ahead; FOLD: r_{t+1})
        Vec<S, VECT> po[MPF], xo[MAF];  // FOLD: p_old of the same rows, x of the own rows
        Vec<S, VECT> pc[MA], qp[MA];    // x_t and q_t of its own rows
        S fa[VECT], fb[VECT];           // FOLD: alpha, beta of this lane's columns (re-reading them with the rows of every step
                                        // frees 8 registers in phase A but costs 2 % of the launch: 702 vs 687 us on cfg3)
#pragma unroll
        for (int v = 0; v < VECT; ++v) fa[v] = fb[v] = S(0);
        if constexpr (Src::FOLD) {
#pragma unroll
            for (int v = 0; v < VECT; ++v) {
                fa[v] = src.alpha[col0 + v];
                fb[v] = src.beta[col0 + v];
            }
        }
#pragma unroll
        for (int k = 0; k < MP; ++k)
#pragma unroll
            for (int v = 0; v < VECT; ++v) pn[k].v[v] = S(0);     // row slots past |C2| are never loaded
        // rows of time slice tt -> registers, in TWO BATCHES: the wave's own rows (k < MA: r, p_old, x), then -- when those have
        // arrived -- its halo rows.  BRANCH-FREE: a row slot past |C2| reads the tile's first row instead (an L1 / L2 hit, never
        // used).  Rounds 1-2 tested every slot: each load sat in its own basic block and the compiler put s_waitcnt vmcnt(0)
        // in front of it -- the (up to) 12 row loads of a step went out one after the other, a full round trip each (7 in
        // a row).  All 12 at once is worse still (measured: 974 us against 755, 3.79 GB fetched against 2.20): the copies of a
        // halo row that several tiles of the XCD request at the same moment all miss in its L2.  Own rows first, halo rows a
        // round trip later keeps the order that makes the halo copies hit (their owners asked for them a moment earlier).
        const unsigned lane_b = (unsigned)(lane * VECT * sizeof(S));          // per-lane byte offset inside a row of the chunk
        auto request_rows = [&](int tt, const int k0, const int k1, bool own) {
            const size_t so = (size_t)tt * slice + (size_t)chunk * 64 * VECT;     // wave-uniform
            // (opaque copy of the lane offset, made next to the loads: the optimiser otherwise adds it to the loop-invariant
            // part of every row address OUTSIDE the time loop and keeps a 64-bit address per row in vector registers)
            unsigned lb = lane_b;
            asm volatile("" : "+v"(lb));
#pragma unroll
            for (int k = 0; k < MP; ++k) {
                if (k >= k0 && k < k1) {
                    const int hr = __builtin_amdgcn_readlane(hrow, k);
                    const size_t ro = so + (size_t)(hr >= 0 ? hr : n0) * g.Bp;
                    pn[k] = ldv_u<S, VECT>(src.in + ro, lb);
                    if constexpr (Src::FOLD) po[k] = ldv_u<S, VECT>(src.p_old + ro, lb);
                }
            }
            if constexpr (Src::FOLD) {
                if (own) {
#pragma unroll
                    for (int k = 0; k < MA; ++k)
                        xo[k] = ldv_u<S, VECT>(src.x + so + (size_t)(n0 + (wave + NW * k < R ? wave + NW * k : 0)) * g.Bp, lb);
                }
            }
        };
        auto request = [&](int tt) {
            if (tt < g.T) {
#if MG_CLDR_BATCH == 0
                request_rows(tt, 0, MP, true);
#elif MG_CLDR_BATCH == 1
                request_rows(tt, 0, MA, true);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                request_rows(tt, MA, MP, false);
#elif MG_CLDR_BATCH == 2
                request_rows(tt, 0, 1, true);
#pragma unroll
                for (int k = 1; k < MP; ++k) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    request_rows(tt, k, k + 1, false);
                }
#else
                request_rows(tt, 0, MA, true);
#pragma unroll
                for (int k = MA; k < MP; ++k) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    request_rows(tt, k, k + 1, false);
                }
#endif
            }
        };
        // FOLD: the requested rows of slice tt become p_new = r + beta p_old; own rows: x += alpha p_old, p_new stored
        auto combine = [&](int tt) {
            if constexpr (Src::FOLD) {
                unsigned lbs = lane_b;
                asm volatile("" : "+v"(lbs));
#pragma unroll
                for (int k = 0; k < MP; ++k) {
                    Vec<S, VECT> pv;
#pragma unroll
                    for (int v = 0; v < VECT; ++v) pv.v[v] = fma(fb[v], po[k].v[v], pn[k].v[v]);
                    if (k < MA) {
                        if (wave + NW * k < R) {
                            const size_t uoff = (size_t)tt * slice + (size_t)(n0 + wave + NW * k) * g.Bp + (size_t)chunk * 64 * VECT;
                            Vec<S, VECT> xv;
#pragma unroll
                            for (int v = 0; v < VECT; ++v) xv.v[v] = fma(fa[v], po[k].v[v], xo[k < MAF ? k : 0].v[v]);
                            stv_u<S, VECT>(src.x + uoff, lbs, xv);
                            stv_u<S, VECT>(src.p_new + uoff, lbs, pv);
                        }
                    }
                    pn[k] = pv;
                }
            }
        };
        // P image <- the requested rows (every slot: the rows past |C2| of the image are never gathered)
        auto store_p = [&]() {
#pragma unroll
            for (int k = 0; k < MP; ++k) stl<S, VECT>(Pimg + (size_t)(wave + NW * k) * W + lane * VECT, pn[k]);
        };
        // prologue: x_0 -> P image, q_0 = 0, request x_1
        request(0);
        combine(0);
        store_p();
#pragma unroll
        for (int k = 0; k < MA; ++k) {
            pc[k] = pn[k];
#pragma unroll
            for (int v = 0; v < VECT; ++v) qp[k].v[v] = S(0);
        }
        request(1);
        __syncthreads();
        for (int t = 0; t < g.T; ++t) {
            const bool nxt = t + 1 < g.T;
            Vec<S, VECT> qn[MA];
            // ---- phase A: q_{t+1} on this wave's rows of C1 (pn holds x_{t+1}, requested during the previous step)
            if (nxt) {
                combine(t + 1);
                // one LDS round trip per row (the first GDF slots are read together); row slots past |C1| are skipped
#pragma unroll
                for (int k = 0; k < MQ; ++k) {
                    const int l = wave + NW * k;
                    if (l < nC1) {
                        const Vec<S, VECT> sum = cldr_gather<S, VECT, GDF, GD, KD>(pl, mdc, mdw, k * GD, __builtin_amdgcn_readlane(dcn, k));
                        Vec<S, VECT> qv;
#pragma unroll
                        for (int v = 0; v < VECT; ++v) qv.v[v] = S(1) * pn[k].v[v] - sum.v[v];
                        stl<S, VECT>(Qimg + (size_t)l * W + lane * VECT, qv);
                        if (k < MA) qn[k] = qv;
                    }
                }
            }
            __syncthreads();                              // Q complete; every gather from P is done
#if MG_CLDR_EARLY
            // ---- phase C, early-request form: P <- x_{t+1} and the request of x_{t+2} first (the loads fly during the gathers and
            // the epilogue), x_{t+1} of the own rows comes back from the P image afterwards
            if (nxt) store_p();
            request(t + 2);
#endif
            // ---- phase C: own rows of l_t through the epilogue, then P <- x_{t+1} and the request of x_{t+2}
            const S selfc = (t > 0 || g.q1) ? S(1) : S(0);
            Vec<S, VECT> lv[MA];
#pragma unroll
            for (int k = 0; k < MA; ++k) {
                Vec<S, VECT> sum;
#pragma unroll
                for (int v = 0; v < VECT; ++v) sum.v[v] = S(0);
                if (nxt) sum = cldr_gather<S, VECT, GTF, GT, KT>(ql, mtc, mtw, k * GT, __builtin_amdgcn_readlane(tcn, k));
#pragma unroll
                for (int v = 0; v < VECT; ++v) lv[k].v[v] = selfc * qp[k].v[v] - sum.v[v];
            }
            {
                unsigned lbe = lane_b;
                asm volatile("" : "+v"(lbe));
#pragma unroll
                for (int k = 0; k < MA; ++k) {
                    const int l = wave + NW * k;
                    if (l < R) epi.row(t, OffSplit{((size_t)t * g.N + n0 + l) * g.Bp + (size_t)chunk * 64 * VECT, lbe}, pc[k], lv[k], acc);
                }
            }
#if MG_CLDR_EARLY
            if (nxt) {
#pragma unroll
                for (int k = 0; k < MA; ++k) {
                    pc[k] = ldv<S, VECT>(Pimg + (size_t)(wave + NW * k) * W + lane * VECT);       // this wave's own store: no barrier
                    qp[k] = qn[k];
                }
            }
#else
            if (nxt) {
                store_p();
#pragma unroll
                for (int k = 0; k < MA; ++k) {
                    pc[k] = pn[k];
                    qp[k] = qn[k];
                }
            }
            request(t + 2);
#endif
            __syncthreads();                              // P = x_{t+1} complete; every gather from Q is done
        }
    }

    if (Epi::NRED > 0) {
        __syncthreads();
        S(*sm)[VECT][64] = reinterpret_cast<S(*)[VECT][64]>(cldr_raw);   // the images are dead now
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            if (wave > 0) {
#pragma unroll
                for (int v = 0; v < VECT; ++v) sm[wave - 1][v][lane] = acc[rr][v];
            }
            __syncthreads();
            if (wave == 0) {
                Vec<S, VECT> o;
#pragma unroll
                for (int v = 0; v < VECT; ++v) {
                    S a = acc[rr][v];
#pragma unroll
                    for (int w2 = 0; w2 < NW - 1; ++w2) a += sm[w2][v][lane];
                    o.v[v] = a;
                }
                stv<S, VECT>(partials + ((size_t)rr * g.P + tidx) * g.Bp + col0, o);
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------ epilogues
template <typename S, int VEC>
struct EpiStore {  // out = l
    static constexpr int NRED = 0;
    static constexpr bool HAS_PRE = false;
    S* out;
    __device__ void begin(int) {}
    template <class O>
    __device__ void row(int, O off, const Vec<S, VEC>&, const Vec<S, VEC>& l, S (*)[VEC]) { sto<S, VEC>(out, off, l); }
};

// Ap = HtH p + c1 p + c2 l ; acc0 += p.Ap       (LHS_x / LHS_zu / LHS_zd applied to the CG direction,
// ADMM.py:349, 371-399).  p == nullptr: the gathered vector is p itself (Lu / diagonal case).
template <typename S, int VEC>
struct EpiLhs {
    static constexpr int NRED = 1;
    static constexpr bool HAS_PRE = true;   // the row of p can be requested ahead of time (time-innermost sweep)
    const S* p;
    const S* mask;  // only for the stand-alone mgadmm_lhs entry point (LHS_x(x, mask))
    S* Ap;
    int hth, t_in;
    S c1, c2;
    __device__ void begin(int) {}
    __device__ Vec<S, VEC> pre(size_t off) const {
        if (p) return ldo<S, VEC>(p, off);
        Vec<S, VEC> z;
#pragma unroll
        for (int v = 0; v < VEC; ++v) z.v[v] = S(0);
        return z;
    }
    template <class O>
    __device__ void row(int t, O off, const Vec<S, VEC>& self, const Vec<S, VEC>& l, S (*acc)[VEC]) {
        row_pre(t, off, self, l, acc, p ? ldo<S, VEC>(p, off) : self);
    }
    template <class O>
    __device__ void row_pre(int t, O off, const Vec<S, VEC>& self, const Vec<S, VEC>& l, S (*acc)[VEC],
                            const Vec<S, VEC>& pin) {
        const Vec<S, VEC> pv = p ? pin : self;
        Vec<S, VEC> d;
        if (mask) d = ldo<S, VEC>(mask, off);
        else {
            const S dd = (hth && t < t_in) ? S(1) : S(0);
#pragma unroll
            for (int v = 0; v < VEC; ++v) d.v[v] = dd;
        }
        Vec<S, VEC> o;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            o.v[v] = d.v[v] * pv.v[v] + c1 * pv.v[v] + c2 * l.v[v];
            acc[0][v] += pv.v[v] * o.v[v];
        }
        sto<S, VEC>(Ap, off, o);
    }
};

// CG start: r = rhs - A x0 ; p = r ; x = x0 ; acc0 += r.r      (ADMM.py:339-347; `mask` only here, quirk Q2)
template <typename S, int VEC>
struct EpiCgInit {
    static constexpr int NRED = 1;
    static constexpr bool HAS_PRE = false;
    const S* x0;    // nullptr: the gathered vector is x0 itself
    const S* rhs;
    const S* mask;  // nullptr: HtH = [t < t_in]
    S *r, *p, *x;
    int hth, t_in;
    S c1, c2;
    __device__ void begin(int) {}
    template <class O>
    __device__ void row(int t, O off, const Vec<S, VEC>& self, const Vec<S, VEC>& l, S (*acc)[VEC]) {
        const Vec<S, VEC> xv = x0 ? ldo<S, VEC>(x0, off) : self;
        const Vec<S, VEC> bv = ldo<S, VEC>(rhs, off);
        Vec<S, VEC> d;
        if (mask) d = ldo<S, VEC>(mask, off);
        else {
            const S dd = (hth && t < t_in) ? S(1) : S(0);
#pragma unroll
            for (int v = 0; v < VEC; ++v) d.v[v] = dd;
        }
        Vec<S, VEC> rv;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const S ax = d.v[v] * xv.v[v] + c1 * xv.v[v] + c2 * l.v[v];
            rv.v[v] = bv.v[v] - ax;
            acc[0][v] += rv.v[v] * rv.v[v];
        }
        sto<S, VEC>(r, off, rv);
        sto<S, VEC>(p, off, rv);
        sto<S, VEC>(x, off, xv);
    }
};

// CG vector updates (ADMM.py:352-366), split so that p is read once per iteration:
//   EpiCgUpdate : r -= alpha Ap ; sum r^2                    (in = Ap; 12 B/element)
//   EpiPUpdate  : x += alpha p ; p = r + beta p               (in = r;  20 B/element)
// x += alpha p is the reference's ADMM.py:352 moved behind the beta reduction -- same operands, same
// arithmetic, bitwise the same x.  Both kernels of iteration k are guarded by n_active[k-1] ("some sample
// was active when iteration k started"), so the last x update of a sample that converges in iteration k
// still happens; frozen samples have alpha = beta = 0.
template <typename S, int VEC>
struct EpiCgUpdate {
    static constexpr bool ELEMENTWISE = true;   // never launched with a spatial operator
    static constexpr int NRED = 1;
    static constexpr bool HAS_PRE = false;
    const S* alpha;
    S* r;
    S a[VEC];
    struct Ops { Vec<S, VEC> r; };          // operand rows, fetched for several rows before any of them is stored
    __device__ void begin(int col0) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) a[v] = alpha[col0 + v];
    }
    __device__ Ops fetch(size_t off) const { return Ops{ldv<S, VEC>(r + off)}; }
    __device__ void row(int t, size_t off, const Vec<S, VEC>& av, const Vec<S, VEC>& l, S (*acc)[VEC]) { row_ops(t, off, av, l, acc, fetch(off)); }
    __device__ void row_ops(int, size_t off, const Vec<S, VEC>& av, const Vec<S, VEC>&, S (*acc)[VEC], const Ops& o) {
        Vec<S, VEC> rv = o.r;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            rv.v[v] = rv.v[v] - a[v] * av.v[v];
            acc[0][v] += rv.v[v] * rv.v[v];
        }
        stv<S, VEC>(r + off, rv);
    }
};

template <typename S, int VEC>
struct EpiPUpdate {
    static constexpr bool ELEMENTWISE = true;   // never launched with a spatial operator
    static constexpr int NRED = 0;
    static constexpr bool HAS_PRE = false;
    const S* alpha;
    const S* beta;
    S *x, *p;
    S a[VEC], b[VEC];
    __device__ void begin(int col0) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            a[v] = alpha[col0 + v];
            b[v] = beta[col0 + v];
        }
    }
    struct Ops { Vec<S, VEC> p, x; };
    __device__ Ops fetch(size_t off) const { return Ops{ldv<S, VEC>(p + off), ldv<S, VEC>(x + off)}; }
    __device__ void row(int t, size_t off, const Vec<S, VEC>& rv, const Vec<S, VEC>& l, S (*acc)[VEC]) { row_ops(t, off, rv, l, acc, fetch(off)); }
    __device__ void row_ops(int, size_t off, const Vec<S, VEC>& rv, const Vec<S, VEC>&, S (*)[VEC], const Ops& o) {
        Vec<S, VEC> pv = o.p, xv = o.x;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            xv.v[v] = fma(a[v], pv.v[v], xv.v[v]);       // explicit fma: the same rounding as the form folded into k_cldr
            pv.v[v] = fma(b[v], pv.v[v], rv.v[v]);
        }
        stv<S, VEC>(x + off, xv);
        stv<S, VEC>(p + off, pv);
    }
};

// x += alpha p (ADMM.py:352) for the LAST iteration of a CG solve whose p-updates are folded into the SpMM kernel
template <typename S, int VEC>
struct EpiXFinal {
    static constexpr bool ELEMENTWISE = true;
    static constexpr int NRED = 0;
    static constexpr bool HAS_PRE = false;
    const S* alpha;
    S* x;
    S a[VEC];
    __device__ void begin(int col0) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) a[v] = alpha[col0 + v];
    }
    struct Ops { Vec<S, VEC> x; };
    __device__ Ops fetch(size_t off) const { return Ops{ldv<S, VEC>(x + off)}; }
    __device__ void row(int t, size_t off, const Vec<S, VEC>& pv, const Vec<S, VEC>& l, S (*acc)[VEC]) { row_ops(t, off, pv, l, acc, fetch(off)); }
    __device__ void row_ops(int, size_t off, const Vec<S, VEC>& pv, const Vec<S, VEC>&, S (*)[VEC], const Ops& o) {
        Vec<S, VEC> xv = o.x;
#pragma unroll
        for (int v = 0; v < VEC; ++v) xv.v[v] = fma(a[v], pv.v[v], xv.v[v]);
        stv<S, VEC>(x + off, xv);
    }
};

// out = a*in + b*w     (gamma + rho*phi ; gamma_u/2 + rho_u/2 x : ADMM.py:559, 579, 587)
template <typename S, int VEC>
struct EpiLin2 {
    static constexpr bool ELEMENTWISE = true;   // never launched with a spatial operator
    static constexpr int NRED = 0;
    static constexpr bool HAS_PRE = false;
    const S* w;
    S* out;
    S a, b;
    __device__ void begin(int) {}
    struct Ops { Vec<S, VEC> w; };
    __device__ Ops fetch(size_t off) const { return Ops{ldv<S, VEC>(w + off)}; }
    __device__ void row(int t, size_t off, const Vec<S, VEC>& u, const Vec<S, VEC>& l, S (*acc)[VEC]) { row_ops(t, off, u, l, acc, fetch(off)); }
    __device__ void row_ops(int, size_t off, const Vec<S, VEC>& u, const Vec<S, VEC>&, S (*)[VEC], const Ops& ops) {
        Vec<S, VEC> o;
#pragma unroll
        for (int v = 0; v < VEC; ++v) o.v[v] = a * u.v[v] + b * ops.w.v[v];
        stv<S, VEC>(out + off, o);
    }
};

// RHS_x = [Ldr^T(gamma + rho phi)]/2 + (rho_u zu + rho_d zd)/2 - (gamma_u + gamma_d)/2 + Hty   (ADMM.py:556-564)
template <typename S, int VEC>
struct EpiRhsX {
    static constexpr int NRED = 0;
    static constexpr bool HAS_PRE = false;
    const S *zu, *zd, *gu, *gd, *y;
    S* out;
    S rho_u, rho_d;
    int use_l, use_zd;
    __device__ void begin(int) {}
    __device__ void row(int, size_t off, const Vec<S, VEC>&, const Vec<S, VEC>& l, S (*)[VEC]) {
        const Vec<S, VEC> zuv = ldv<S, VEC>(zu + off), guv = ldv<S, VEC>(gu + off), yv = ldv<S, VEC>(y + off);
        Vec<S, VEC> o;
        if (use_zd) {
            const Vec<S, VEC> zdv = ldv<S, VEC>(zd + off), gdv = ldv<S, VEC>(gd + off);
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                o.v[v] = (rho_u * zuv.v[v] + rho_d * zdv.v[v]) / S(2) - (guv.v[v] + gdv.v[v]) / S(2) + yv.v[v];
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) o.v[v] = rho_u * zuv.v[v] / S(2) - guv.v[v] / S(2) + yv.v[v];
        }
        if (use_l) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) o.v[v] = l.v[v] / S(2) + o.v[v];
        }
        stv<S, VEC>(out + off, o);
    }
};

// Dual updates of gamma_u, gamma_d (ADMM.py:595-597) fused with the residual norms that do not need a
// Laplacian (ADMM.py:612-636).  Gathered vector = new x.
// acc: 0 ||x-x_old||^2, 1 ||x-zu||^2, 2 ||zu-zu_old||^2, 3 ||x-zd||^2, 4 ||zd-zd_old||^2, 5 ||Hx-y||^2
template <typename S, int VEC>
struct EpiDual {
    static constexpr bool ELEMENTWISE = true;   // never launched with a spatial operator
    static constexpr int NRED = 6;
    static constexpr bool HAS_PRE = false;
    const S *xold, *zu, *zuold, *zd, *zdold, *y, *mask;
    S *gu, *gd;
    S rho_u, rho_d;
    int has_zd, t_in;
    __device__ void begin(int) {}
    __device__ void row(int t, size_t off, const Vec<S, VEC>& x, const Vec<S, VEC>&, S (*acc)[VEC]) {
        const Vec<S, VEC> xo = ldv<S, VEC>(xold + off), z = ldv<S, VEC>(zu + off), zo = ldv<S, VEC>(zuold + off);
        Vec<S, VEC> g = ldv<S, VEC>(gu + off);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const S dx = x.v[v] - xo.v[v], pz = x.v[v] - z.v[v], dz = z.v[v] - zo.v[v];
            acc[0][v] += dx * dx;
            acc[1][v] += pz * pz;
            acc[2][v] += dz * dz;
            g.v[v] = g.v[v] + rho_u * pz;
        }
        stv<S, VEC>(gu + off, g);
        if (has_zd) {
            const Vec<S, VEC> zdv = ldv<S, VEC>(zd + off), zdo = ldv<S, VEC>(zdold + off);
            Vec<S, VEC> gdv = ldv<S, VEC>(gd + off);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const S pz = x.v[v] - zdv.v[v], dz = zdv.v[v] - zdo.v[v];
                acc[3][v] += pz * pz;
                acc[4][v] += dz * dz;
                gdv.v[v] = gdv.v[v] + rho_d * pz;
            }
            stv<S, VEC>(gd + off, gdv);
        }
        if (mask) {
            const Vec<S, VEC> m = ldv<S, VEC>(mask + off), yv = ldv<S, VEC>(y + off);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const S e = x.v[v] * m.v[v] - yv.v[v];
                acc[5][v] += e * e;
            }
        } else if (t < t_in) {
            const Vec<S, VEC> yv = ldv<S, VEC>(y + off);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const S e = x.v[v] - yv.v[v];
                acc[5][v] += e * e;
            }
        }
    }
};

template <typename S>
__device__ __forceinline__ S soft_thr(S s, S thr) {
    const S u = fabs(s) - thr;
    return (u > S(0)) ? (s > S(0) ? u : -u) : S(0);
}

// phi prox + gamma dual update (ADMM.py:401-408, 600-606) with l = Ldr x computed once, fused with
// ||phi - Ldr x||^2, ||phi - phi_old||^2, ||Ldr x||_1 and sum (Ldr x)^2.  update == 0: metrics only.
template <typename S, int VEC>
struct EpiPhi {
    static constexpr int NRED = 4;
    static constexpr bool HAS_PRE = false;
    const S* phi_old;
    S* phi_new;
    S* gamma;
    S rho, thr;
    int update;
    __device__ void begin(int) {}
    __device__ void row(int, size_t off, const Vec<S, VEC>&, const Vec<S, VEC>& l, S (*acc)[VEC]) {
        if (update) {
            const Vec<S, VEC> po = ldv<S, VEC>(phi_old + off);
            Vec<S, VEC> g = ldv<S, VEC>(gamma + off), pn;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                pn.v[v] = soft_thr<S>(l.v[v] - g.v[v] / rho, thr);
                const S d = pn.v[v] - l.v[v], dp = pn.v[v] - po.v[v];
                g.v[v] = g.v[v] + rho * d;
                acc[0][v] += d * d;
                acc[1][v] += dp * dp;
            }
            stv<S, VEC>(phi_new + off, pn);
            stv<S, VEC>(gamma + off, g);
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            acc[2][v] += fabs(l.v[v]);
            acc[3][v] += l.v[v] * l.v[v];
        }
    }
};

// stand-alone phi_direct(x, gamma)  (ADMM.py:401-408)
template <typename S, int VEC>
struct EpiPhiDirect {
    static constexpr int NRED = 0;
    static constexpr bool HAS_PRE = false;
    const S* gamma;
    S* phi;
    S rho, thr;
    __device__ void begin(int) {}
    __device__ void row(int, size_t off, const Vec<S, VEC>&, const Vec<S, VEC>& l, S (*)[VEC]) {
        const Vec<S, VEC> g = ldv<S, VEC>(gamma + off);
        Vec<S, VEC> o;
#pragma unroll
        for (int v = 0; v < VEC; ++v) o.v[v] = soft_thr<S>(l.v[v] - g.v[v] / rho, thr);
        stv<S, VEC>(phi + off, o);
    }
};

// GLR: acc0 += x . Lu x     (ADMM.py:245-246)
template <typename S, int VEC>
struct EpiDot {
    static constexpr int NRED = 1;
    static constexpr bool HAS_PRE = false;
    __device__ void begin(int) {}
    __device__ void row(int, size_t, const Vec<S, VEC>& x, const Vec<S, VEC>& l, S (*acc)[VEC]) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[0][v] += x.v[v] * l.v[v];
    }
};

// ---------------------------------------------------------------------------------------------
// Fixed-order reduction of the per-workgroup partials + the per-sample scalar step that follows it.
// One workgroup = 64 batch columns x 16 waves; wave w sums partial rows w, w+16, ...; wave 0 combines
// the 16 sub-sums in order and runs Fin for its 64 columns.
// ---------------------------------------------------------------------------------------------
template <typename S, int NRED, class Fin>
__global__ __launch_bounds__(1024) void k_reduce(const S* __restrict__ partials, int P, int Bp, Fin fin,
                                                 const int* __restrict__ live) {
    if (live != nullptr && *live == 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    double s[NRED];
#pragma unroll
    for (int r = 0; r < NRED; ++r) s[r] = 0.0;
    // RB partial rows per trip: the loads are issued together (the kernel is pure latency: Bp/64 workgroups, up
    // to ~80 rows per wave on the 10k-node graph, ~800 on the 100k-node one), the additions stay in row order
    // -- same sums, bit for bit
    constexpr int RB = NRED <= 2 ? 16 : (NRED * sizeof(S) > 32 ? 4 : 8);     // at most 48 registers of loads in flight
    int p = wave;
    for (; p + (RB - 1) * 16 < P; p += RB * 16) {
        S v[RB][NRED];
#pragma unroll
        for (int k = 0; k < RB; ++k)
#pragma unroll
            for (int r = 0; r < NRED; ++r) v[k][r] = partials[((size_t)r * P + p + 16 * k) * Bp + c];
#pragma unroll
        for (int k = 0; k < RB; ++k)
#pragma unroll
            for (int r = 0; r < NRED; ++r) s[r] += (double)v[k][r];
    }
    // the remaining rows (< RB per wave) in ONE more trip: rows past P are read at the last row and enter as 0
    if (p < P) {
        S v[RB][NRED];
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            const int row = p + 16 * k;
#pragma unroll
            for (int r = 0; r < NRED; ++r) v[k][r] = partials[((size_t)r * P + (row < P ? row : P - 1)) * Bp + c];
        }
#pragma unroll
        for (int k = 0; k < RB; ++k)
#pragma unroll
            for (int r = 0; r < NRED; ++r) s[r] += (p + 16 * k < P) ? (double)v[k][r] : 0.0;
    }
    __shared__ double sm[15][NRED][64];
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < NRED; ++r) sm[wave - 1][r][lane] = s[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll 3      // fully unrolled the compiler reads all 15 * NRED doubles first (NRED = 6: 44 VGPRs spilled)
        for (int w = 0; w < 15; ++w)
#pragma unroll
            for (int r = 0; r < NRED; ++r) s[r] += sm[w][r][lane];
        fin(c, s);
    }
}

// CG scalars, one entry per batch column.
template <typename S>
struct CgScalars {
    S* rr;         // r.r
    S* alpha;
    S* beta;
    int* active;   // 1 while the sample still iterates
    int* iters;    // iteration count at convergence, -1 otherwise
    int* n_active; // [max_cg_iter] number of active samples after iteration k
    S* alpha_hist; // [max_cg_iter][Bp] or nullptr
    S* beta_hist;
    int* nonfinite;
};

template <typename S>
struct FinCgInit {
    CgScalars<S> c;
    int B;
    __device__ void operator()(int col, const double* s) const {
        c.rr[col] = (S)s[0];
        c.active[col] = col < B ? 1 : 0;
        c.iters[col] = -1;
        c.alpha[col] = S(0);
        c.beta[col] = S(0);
    }
};

template <typename S>
struct FinCgAlpha {  // alpha = r.r / p.Ap   (ADMM.py:350)
    CgScalars<S> c;
    int k, Bp;
    __device__ void operator()(int col, const double* s) const {
        const bool act = c.active[col] != 0;
        const S a = act ? c.rr[col] / (S)s[0] : S(0);
        c.alpha[col] = a;
        if (c.alpha_hist) c.alpha_hist[(size_t)k * Bp + col] = act ? a : (S)NAN;
    }
};

template <typename S>
struct FinCgBeta {  // beta = rr'/rr ; convergence sqrt(rr') < tol ; (ADMM.py:355-362), per sample (Q6)
    CgScalars<S> c;
    int k, Bp;
    double tol;
    int batch_max;   // 1: the reference's batch-global stop (ADMM.py:360) -- a sample below the tolerance keeps iterating
                     // until all are; n_active[k] then counts the samples still ABOVE it (0 = the solve has stopped)
    __device__ void operator()(int col, const double* s) const {
        bool act = c.active[col] != 0;
        bool above = false;
        S b = S(0);
        if (act) {
            const S rrn = (S)s[0];
            b = rrn / c.rr[col];
            c.rr[col] = rrn;
            if (c.beta_hist) c.beta_hist[(size_t)k * Bp + col] = b;
            if (!(fabs((double)rrn) <= 1.79e308)) {  // NaN / Inf: stop this sample, report
                *c.nonfinite = 1;
                c.active[col] = 0;
                act = false;
            } else if ((double)sqrt(rrn) < tol) {
                if (!batch_max) {
                    c.iters[col] = k + 1;
                    c.active[col] = 0;
                    act = false;
                }
            } else {
                above = true;
            }
        } else if (c.beta_hist) {
            c.beta_hist[(size_t)k * Bp + col] = (S)NAN;
        }
        c.beta[col] = act ? b : S(0);
        const unsigned long long m = __ballot(batch_max ? above : act);
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(&c.n_active[k], __popcll(m));
    }
};

// batch_max: the iteration count of the whole solve = first k with n_active[k] == 0 (+1), -1 when never reached
static __global__ void k_cg_batchmax_iters(const int* __restrict__ n_active, int K, int* __restrict__ iters, int B) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= B) return;
    int it = -1;
    for (int k = 0; k < K; ++k)
        if (n_active[k] == 0) { it = k + 1; break; }
    iters[col] = it;
}

// apply_op_Ln on the line graph (ADMM.py:253-261): y[t] = x[t] - x[t+1]/sqrt(2) for t < T-1, y[T-1] = x[T-1] - x[T-2]/sqrt(2)
template <typename S, int VEC>
struct EpiLnLine {
    static constexpr bool ELEMENTWISE = true;
    static constexpr int NRED = 0;
    static constexpr bool HAS_PRE = false;
    const S* x;
    S* out;
    int T;
    size_t slice;     // N * Bp
    __device__ void begin(int) {}
    __device__ void row(int t, size_t off, const Vec<S, VEC>& self, const Vec<S, VEC>&, S (*)[VEC]) {
        const Vec<S, VEC> nb = ldv<S, VEC>(t + 1 < T ? x + off + slice : x + off - slice);
        Vec<S, VEC> o;
#pragma unroll
        for (int v = 0; v < VEC; ++v) o.v[v] = self.v[v] - nb.v[v] / (S)1.4142135623730951;
        stv<S, VEC>(out + off, o);
    }
};

// out = add + l
template <typename S, int VEC>
struct EpiAddTo {
    static constexpr int NRED = 0;
    static constexpr bool HAS_PRE = false;
    const S* add;
    S* out;
    __device__ void begin(int) {}
    __device__ void row(int, size_t off, const Vec<S, VEC>&, const Vec<S, VEC>& l, S (*)[VEC]) {
        const Vec<S, VEC> a = ldv<S, VEC>(add + off);
        Vec<S, VEC> o;
#pragma unroll
        for (int v = 0; v < VEC; ++v) o.v[v] = a.v[v] + l.v[v];
        stv<S, VEC>(out + off, o);
    }
};

// per-sample metric sums -> ps[slot[r]][col]
template <int NRED>
struct FinMetrics {
    double* ps;  // [MGADMM_NMETRIC][Bp]
    int Bp;
    int slot[NRED];
    __device__ void operator()(int col, const double* s) const {
#pragma unroll
        for (int r = 0; r < NRED; ++r)
            if (slot[r] >= 0) ps[(size_t)slot[r] * Bp + col] = s[r];
    }
};

// whole-batch value of every metric for this ADMM iteration (fixed summation order over samples):
// norms -> sqrt(sum_b), regularisers -> mean_b      (ADMM.py:612-637, 230-246)
static __global__ void k_batch_metrics(const double* __restrict__ ps, int Bp, int B, double* __restrict__ out,
                                double* __restrict__ out_ps /* [NMETRIC][B] or nullptr */) {
    const int m = blockIdx.x;
    __shared__ double sm[256];
    double s = 0.0;
    for (int c = threadIdx.x; c < B; c += 256) {
        const double v = ps[(size_t)m * Bp + c];
        s += v;
        if (out_ps) out_ps[(size_t)m * B + c] = v;
    }
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double tot = sm[0];
        const bool is_mean = (m == MGADMM_M_GLR || m == MGADMM_M_DGTV || m == MGADMM_M_DGLR);
        out[m] = is_mean ? tot / (double)B : sqrt(tot);
    }
}

// delta_x_per_step: || mean_b (x - x_old) ||_2 over nodes, per time step (ADMM.py:614).
// part[t][nbk]: sum over the 64 nodes of block nbk of (mean_b dx)^2.
template <typename S>
__global__ __launch_bounds__(256) void k_dxps(int T, int N, int Bp, int B, int NBK, const S* __restrict__ x,
                                              const S* __restrict__ xold, double* __restrict__ part) {
    const int t = blockIdx.x / NBK, nbk = blockIdx.x % NBK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i1 = min(N, (nbk + 1) * 64);
    double a = 0.0;
    for (int i = nbk * 64 + wave; i < i1; i += 4) {
        const size_t off = ((size_t)t * N + i) * Bp;
        double s = 0.0;
        for (int c = lane; c < Bp; c += 64) s += (double)x[off + c] - (double)xold[off + c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const double m = s / (double)B;
        a += m * m;
    }
    __shared__ double sm[4];
    if (lane == 0) sm[wave] = a;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ((sm[0] + sm[1]) + sm[2]) + sm[3];
}

static __global__ void k_dxps_final(int T, int NBK, const double* __restrict__ part, double* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    double s = 0.0;
    for (int k = 0; k < NBK; ++k) s += part[(size_t)t * NBK + k];
    out[t] = sqrt(s);
}

// ---------------------------------------------------------------------------------------------
// Layout conversion between the reference's (B, Ts, N) tensors and the internal [T][N][Bp] layout,
// through a 64x64 LDS tile so both sides are coalesced.  perm (optional) maps internal row i -> API
// node.  pack zero-fills t >= Ts, b >= B.
// ---------------------------------------------------------------------------------------------
template <typename S>
__global__ __launch_bounds__(256) void k_pack(int T, int Ts, int N, int B, int Bp, const int* __restrict__ perm,
                                              const S* __restrict__ src, S* __restrict__ dst) {
    __shared__ S tile[64][65];
    const int i0 = blockIdx.x * 64, t = blockIdx.y, b0 = blockIdx.z * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int i = i0 + tx;
    int node = -1;
    if (i < N && t < Ts) node = perm ? perm[i] : i;
    for (int bb = ty; bb < 64; bb += 4) {
        const int b = b0 + bb;
        S v = S(0);
        if (node >= 0 && b < B) v = src[((size_t)b * Ts + t) * N + node];
        tile[bb][tx] = v;
    }
    __syncthreads();
    for (int ii = ty; ii < 64; ii += 4) {
        if (i0 + ii < N) dst[((size_t)t * N + i0 + ii) * Bp + b0 + tx] = tile[tx][ii];
    }
}

template <typename S>
__global__ __launch_bounds__(256) void k_unpack(int T, int N, int B, int Bp, const int* __restrict__ perm,
                                                const S* __restrict__ src, S* __restrict__ dst) {
    __shared__ S tile[64][65];
    const int i0 = blockIdx.x * 64, t = blockIdx.y, b0 = blockIdx.z * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int ii = ty; ii < 64; ii += 4) {
        S v = S(0);
        if (i0 + ii < N) v = src[((size_t)t * N + i0 + ii) * Bp + b0 + tx];
        tile[ii][tx] = v;
    }
    __syncthreads();
    const int i = i0 + tx;
    if (i < N) {
        const int node = perm ? perm[i] : i;
        for (int bb = ty; bb < 64; bb += 4) {
            const int b = b0 + bb;
            if (b < B) dst[((size_t)b * T + t) * N + node] = tile[tx][bb];
        }
    }
}

template <typename S>
__global__ void k_fill(S* __restrict__ p, size_t n, S v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// initial_guess (ADMM.py:766-781): least-squares line per (node, sample) through t < t_in, in S
// arithmetic with the reference's float32 time moments (tm, den passed from the host).
template <typename S>
__global__ __launch_bounds__(256) void k_initial_guess(int T, int t_in, int N, int Bp, S tm, S den,
                                                       const S* __restrict__ y, S* __restrict__ x) {
#pragma clang fp contract(off)  // the reference evaluates these scalars with separate roundings
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    const int c = blockIdx.y * 64 + lane;
    if (i >= N) return;
    S sy = S(0), sty = S(0);
    for (int t = 0; t < t_in; ++t) {
        const S v = y[((size_t)t * N + i) * Bp + c];
        x[((size_t)t * N + i) * Bp + c] = v;
        sy += v;
        sty += (S)t * v;
    }
    const S ym = sy / (S)t_in;
    const S w = (sty / (S)t_in - tm * ym) / den;
    const S b = ym - w * tm;
    for (int t = t_in; t < T; ++t) x[((size_t)t * N + i) * Bp + c] = w * (S)t + b;
}

// initial_interpolation (ADMM.py:783-811).  F32MOM: time moments (n, t_mean, t2_mean) in float32 as the
// reference gets them from a float32 mask, whatever the signal dtype.
template <typename S, bool F32MOM>
__global__ __launch_bounds__(256) void k_initial_interp(int T, int N, int Bp, int B, const S* __restrict__ y,
                                                        const S* __restrict__ mask, S* __restrict__ x,
                                                        int* __restrict__ nonfinite) {
#pragma clang fp contract(off)  // t2_mean - t_mean^2 cancels: keep the reference's separate roundings
    typedef typename std::conditional<F32MOM, float, S>::type M;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    const int c = blockIdx.y * 64 + lane;
    if (i >= N) return;
    if (c >= B) {
        for (int t = 0; t < T; ++t) x[((size_t)t * N + i) * Bp + c] = S(0);
        return;
    }
    M n = 0, ts = 0, t2s = 0;
    S ys = 0, tys = 0;
    for (int t = 0; t < T; ++t) {
        const size_t o = ((size_t)t * N + i) * Bp + c;
        const S m = mask[o], v = y[o];
        n += (M)m;
        ts += (M)t * (M)m;
        t2s += (M)(t * t) * (M)m;
        ys += v * m;
        tys += (S)t * v * m;
    }
    const M tmean = ts / n, t2mean = t2s / n;
    const S ymean = ys / (S)n, tymean = tys / (S)n;
    M tm2 = tmean * tmean;
    asm volatile("" : "+v"(tm2));   // keep the product rounded on its own: t2_mean - t_mean^2 cancels
    const M den = t2mean - tm2;
    const S w = (tymean - (S)tmean * ymean) / (S)den;
    const S b = ymean - w * (S)tmean;
    if (!(fabs((double)w) <= 1.79e308) || !(fabs((double)b) <= 1.79e308)) *nonfinite = 1;
    for (int t = 0; t < T; ++t) {
        const size_t o = ((size_t)t * N + i) * Bp + c;
        x[o] = (w * (S)t + b) * (S(1) - mask[o]) + y[o];
    }
}

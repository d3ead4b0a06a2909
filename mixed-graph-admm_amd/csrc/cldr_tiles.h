// Host-side tile builder of the fused cLdr kernel (k_cldr in stream_kernels.h).  Plain C++, no HIP: the same code is
// compiled into libmgadmm.so and into the CPU check tests/cpu/cldr_tiles_check.cpp, which replays the kernel's
// dataflow on the host from this metadata and compares it with Ldr^T(Ldr x) taken straight from the CSR matrices.
//
// cLdr = Ldr^T Ldr (reference ADMM.py:225-228) in ONE pass over x needs, for a tile of node rows R,
//     q_{t+1} on C1 = R u {transposed neighbours of R}          (q = Ldr x, recomputed on the 1-hop halo)
//     x_t     on C2 = C1 u {W_d neighbours of C1}               (2-hop halo)
// Both sets live in LDS for one time step; local indices: [0, |R|) own rows (consecutive global rows n0 ...),
// [|R|, |C1|) the rest of C1 in order of first use, [|C1|, |C2|) the rest of C2 in order of first use.
#pragma once
#include <algorithm>
#include <vector>

#include "host_csr.h"

struct CldrCaps {
    int Rcap, C1cap, C2cap;   // rows of a tile / of the Q image / of the P image in LDS
    int GD, GT;               // entry slots per W_d row and per W_d^T row
};

struct CldrTiles {
    CldrCaps caps{};
    int NT = 0;
    std::vector<int> n0;      // [NT+1] first global row of every tile
    std::vector<int> nC1, nC2;// [NT]
    std::vector<int> rows;    // [NT][C2cap] global row of local index l (-1 past nC2)
    std::vector<int> dcol;    // [NT][C1cap][GD] local P row of entry u of W_d row (local) j; pad slots: j itself, weight 0
    std::vector<float> dw;
    std::vector<int> dcnt;    // [NT][C1cap] entries in use
    std::vector<int> tcol;    // [NT][Rcap][GT] local Q row of entry u of W_d^T row (local) i; pad slots: i itself, weight 0
    std::vector<float> tw;
    std::vector<int> tcnt;    // [NT][Rcap]
    long sumC1 = 0, sumC2 = 0;
};

// Wd, WdT in the internal node order.  `cuts`: sorted row indices where a tile may start (cluster boundaries of the
// node order; 0 first); a cluster that does not fit the caps is cut into the largest tiles that do.  Returns false when some row has
// more entries than the slot counts allow (the caller then keeps the two-pass path).
// `row_limit` (0: Rcap): tiles take at most that many rows -- the engine uses it to fill the last round of workgroups.
inline bool build_cldr_tiles(const HostCsr& Wd, const HostCsr& WdT, const std::vector<int>& cuts, const CldrCaps& caps,
                             CldrTiles& out, int row_limit = 0) {
    const int N = Wd.n;
    for (int i = 0; i < N; ++i) {
        if (Wd.rowptr[i + 1] - Wd.rowptr[i] > caps.GD) return false;
        if (WdT.rowptr[i + 1] - WdT.rowptr[i] > caps.GT) return false;
    }
    out = CldrTiles();
    out.caps = caps;
    std::vector<int> local(N, -1), list;       // local index of a global row inside the tile under construction
    list.reserve(caps.C2cap + 64);
    auto collect = [&](int lo, int hi, int& c1, int& c2) {
        for (int r : list) local[r] = -1;
        list.clear();
        for (int i = lo; i < hi; ++i) { local[i] = i - lo; list.push_back(i); }
        for (int i = lo; i < hi; ++i)
            for (int e = WdT.rowptr[i]; e < WdT.rowptr[i + 1]; ++e) {
                const int c = WdT.col[e];
                if (local[c] < 0) { local[c] = (int)list.size(); list.push_back(c); }
            }
        c1 = (int)list.size();
        for (int l = 0; l < c1; ++l) {
            const int j = list[l];
            for (int e = Wd.rowptr[j]; e < Wd.rowptr[j + 1]; ++e) {
                const int c = Wd.col[e];
                if (local[c] < 0) { local[c] = (int)list.size(); list.push_back(c); }
            }
        }
        c2 = (int)list.size();
    };
    std::vector<int> bounds(cuts);
    if (bounds.empty() || bounds[0] != 0) bounds.insert(bounds.begin(), 0);
    bounds.push_back(N);
    out.n0.push_back(0);
    for (size_t b = 0; b + 1 < bounds.size(); ++b) {
        int lo = bounds[b];
        const int end = bounds[b + 1];
        while (lo < end) {
            int hi = std::min(end, lo + (row_limit > 0 ? std::min(row_limit, caps.Rcap) : caps.Rcap)), c1 = 0, c2 = 0;
            for (;;) {
                collect(lo, hi, c1, c2);
                if ((c1 <= caps.C1cap && c2 <= caps.C2cap) || hi - lo == 1) break;
                --hi;          // the LARGEST tile that fits (round 3; rounds 1-2 shrank by a quarter per try: 11.8 rows per tile of 16
                               // on the 10k-node graph, 846 tiles; one row per try: fuller tiles, fewer of them, a smaller halo share)
            }
            if (c1 > caps.C1cap || c2 > caps.C2cap) return false;      // a single row does not fit: hub node
            const int t = out.NT++;
            out.n0.push_back(hi);
            out.nC1.push_back(c1);
            out.nC2.push_back(c2);
            out.sumC1 += c1;
            out.sumC2 += c2;
            out.rows.resize((size_t)(t + 1) * caps.C2cap, -1);
            for (int l = 0; l < c2; ++l) out.rows[(size_t)t * caps.C2cap + l] = list[l];
            out.dcol.resize((size_t)(t + 1) * caps.C1cap * caps.GD, 0);
            out.dw.resize((size_t)(t + 1) * caps.C1cap * caps.GD, 0.f);
            out.dcnt.resize((size_t)(t + 1) * caps.C1cap, 0);
            for (int l = 0; l < caps.C1cap; ++l) {
                const size_t base = ((size_t)t * caps.C1cap + l) * caps.GD;
                const int self = l < c1 ? l : 0;
                int used = 0;
                if (l < c1) {
                    const int j = list[l];
                    for (int e = Wd.rowptr[j]; e < Wd.rowptr[j + 1]; ++e, ++used) {
                        out.dcol[base + used] = local[Wd.col[e]];
                        out.dw[base + used] = Wd.val[e];
                    }
                    out.dcnt[(size_t)t * caps.C1cap + l] = used;
                }
                for (; used < caps.GD; ++used) out.dcol[base + used] = self;
            }
            out.tcol.resize((size_t)(t + 1) * caps.Rcap * caps.GT, 0);
            out.tw.resize((size_t)(t + 1) * caps.Rcap * caps.GT, 0.f);
            out.tcnt.resize((size_t)(t + 1) * caps.Rcap, 0);
            for (int l = 0; l < caps.Rcap; ++l) {
                const size_t base = ((size_t)t * caps.Rcap + l) * caps.GT;
                const int self = l < hi - lo ? l : 0;
                int used = 0;
                if (l < hi - lo) {
                    const int i = lo + l;
                    for (int e = WdT.rowptr[i]; e < WdT.rowptr[i + 1]; ++e, ++used) {
                        out.tcol[base + used] = local[WdT.col[e]];
                        out.tw[base + used] = WdT.val[e];
                    }
                    out.tcnt[(size_t)t * caps.Rcap + l] = used;
                }
                for (; used < caps.GT; ++used) out.tcol[base + used] = self;
            }
            lo = hi;
        }
    }
    return true;
}

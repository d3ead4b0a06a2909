// Solver engine: host-side orchestration of the ADMM loop of reference ADMM.py:511-648 over the streaming HIP kernels
// (stream_kernels.h) and the LDS-resident fused kernel (lds_launch.hip).  Included by solver_f32.hip and solver_f64.hip,
// which instantiate it for one scalar type each (separate translation units: the library builds in parallel).
#pragma once
#include <math.h>
#include <cmath>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>

#include "common.h"
#include "cldr_tiles.h"
#include "stream_kernels.h"
#include "lds_args.h"
#include "lds_banks.h"

namespace {

struct LhsDef {
    int kind;  // 0 diagonal only, 1 cLdr = Ldr^T Ldr, 2 Lu
    int hth;   // include the observation operator H^T H
    double c1, c2;
};

enum VecId {
    V_XA, V_XB, V_ZUA, V_ZUB, V_ZDA, V_ZDB, V_PHIA, V_PHIB, V_GAM, V_GU, V_GD, V_Y, V_MASK,
    V_R, V_P, V_Q, V_AP, V_RHS, V_TMP, V_IO0, V_IO1, V_COUNT
};

constexpr int LAG = 2;          // CG iterations enqueued ahead of the host's convergence check
constexpr int NRED_MAX = 6;
constexpr int PROF_POOL = 32768;
constexpr int LDS_SETS = 3;             // interior iterate-buffer sets / per-sample metric sets of the chunked schedule (chunks in flight)
constexpr int LDS_NBOUND = 4;           // iterate buffers at the chunk boundaries
constexpr int LDS_MAXJ_POOL = LDS_MAXJ;   // longest chunk of the LDS path's chunked schedule: 2 (J - 1) + 3 iterate buffers -- 15 out of the
                                          // workspace (enough for J = 7), the rest allocated on the first solve that asks for a longer chunk
constexpr int NACT_LOG = 1 << 16;   // pinned log of the per-iteration active-sample counts (one int per CG iteration enqueued)

template <typename S>
struct Engine : EngineBase {
    mgadmm_solver* sv;
    mgadmm_graph* g;
    mgadmm_params p;
    int T, N, Bmax, Bp_max;
    hipStream_t st = nullptr;

    S* vec[V_COUNT] = {nullptr};
    S* vec_pool = nullptr;
    int dev_cus = 0;               // compute units of the device (staggered start of k_admm_lds)
    size_t vec_elems = 0;
    S* partials = nullptr;
    size_t partials_elems = 0;
    // CG scalars
    S *d_rr = nullptr, *d_alpha = nullptr, *d_beta = nullptr, *d_alpha_hist = nullptr, *d_beta_hist = nullptr;
    int *d_active = nullptr, *d_iters_tmp = nullptr, *d_nact = nullptr, *d_nonfinite = nullptr;
    // history
    double *d_ps = nullptr, *d_hist = nullptr, *d_dxps = nullptr, *d_dxpart = nullptr, *d_hist_ps = nullptr;
    size_t hist_ps_elems = 0;
    int* d_cg_iters = nullptr;
    int max_admm_alloc = 0, max_cg_alloc = 0;
    // pinned host
    int* h_nact = nullptr;
    double* h_row = nullptr;
    int* h_flag = nullptr;
    hipEvent_t ev_ring[LAG + 1] = {nullptr};
    int want_blocks = 2048;
    struct TileMetaDev { int R = 0; int GW = 0; int* tl_col = nullptr; float* tl_w = nullptr; int* halo = nullptr; int* h_rowptr = nullptr; int* h_col = nullptr; float* h_val = nullptr; };
    TileMetaDev tmeta[3];         // W_u, W_d, W_d^T metadata for the tile size in use
    // fused cLdr kernel (k_cldr): tile tables for the geometry in use; state 0 = not built yet, 1 = usable, -1 = the graph
    // does not fit its slot counts (hub rows): the two-pass form stays
    struct CldrDev {
        int state = 0, NT = 0;
        int *n0 = nullptr, *nC = nullptr, *rows = nullptr, *dcol = nullptr, *dcnt = nullptr, *tcol = nullptr, *tcnt = nullptr;
        float *dw = nullptr, *tw = nullptr;
    } cldr_dev;
    int use_fused = 1;            // MGADMM_FUSED=0 keeps the two-pass cLdr (tests compare the two)
    int use_fold = 1;             // MGADMM_FOLD=0: p = r + beta p and x += alpha p stay in their own kernel (EpiPUpdate)
    int use_fold_lu = 1;          // MGADMM_FOLD_LU=0: the same for the Lu kernel of the zu solve only
    int sweep_rev = 1;            // MGADMM_SWEEP_REV=0: r -= alpha Ap sweeps the time slices upwards like every other kernel
    int cldr_tile_major = 1;      // MGADMM_CLDR_ORDER=0: chunks of a tile adjacent in dispatch order
    int cur_P = 0;                // partial rows written by the last row-kernel launch (k_rows or k_tile)
    int use_tile = 1;             // LDS-tiled spatial kernel on cluster-ordered graphs (reorder = 2); MGADMM_TILE=0 disables
    int64_t ws_bytes = 0;
    // LDS-resident fused path (float32, small graphs)
    struct LdsPlan {
        bool ok = false;
        int G = 0, TPG = 0, TS = 0, nthreads = 0, block = 0, NR = 0, csr_ints = 0, maxt = 1024, sb = 0, uniform45 = 0, slots = 0;
        int tail_pairs = 0, lds_img0 = 0, lds_img_ints = 0;
        int off_rp_u = 0, off_rp_d = 0, off_en_u = 0, off_en_d = 0, off_lead_t = 0, off_tail_t = 0, off_diag = 0;
        size_t lds_bytes = 0;
    } lds;
    int* d_lds_csr = nullptr;
    double* d_m2 = nullptr;
    // ADMM outer loop of the LDS path without host round trips (solve_lds): device stop word, second metric buffer, helper
    // stream for the whole-batch metric kernels and the events that order the two streams
    int lds_async = 1;            // MGADMM_LDS_ASYNC=0: one stream, the host tests the stop criterion after every iteration
    int* d_stop = nullptr;
    double* d_ps_ring = nullptr;  // [LDS_SETS][J][NMETRIC][Bp]: per-sample metric sums of the chunked schedule
    std::vector<float*> lds_ring_extra;   // iterate buffers beyond the 15 workspace vectors (chunks longer than 7 iterations)
    int lds_chunk = LDS_MAXJ_POOL;   // MGADMM_LDS_CHUNK: ADMM iterations per k_admm_lds launch when the iteration count is fixed (1 .. LDS_MAXJ_POOL)
    hipStream_t st_side = nullptr;
    hipEvent_t ev_main[LDS_NBOUND] = {nullptr}, ev_side[LDS_NBOUND] = {nullptr};
    // profiling
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;
    std::vector<int> prof_tag;
    std::vector<double> prof_lb;   // algorithmic bytes of each timed launch
    std::vector<int> prof_ref;     // h_nact index that tells whether the launch did work (-1: unconditional launch)
    int prof_cur_ref = -1;         // set by cg_internal around the launches of one CG iteration
    int nact_cur = 0;              // next free slot of the h_nact log
    bool nact_locked = false;      // profiling session has used the whole log: later solves use the scratch tail
    int64_t prof_noop = 0;         // timed launches that returned at the converged-CG guard
    size_t prof_used = 0;
    int64_t prof_count[MGADMM_NPROF] = {0};
    double prof_bytes[MGADMM_NPROF] = {0};

    explicit Engine(mgadmm_solver* s) : sv(s), g(s->g), p(s->p), T(s->g->T), N(s->g->N), Bmax(s->Bmax) {}

    ~Engine() override {
        (void)hipSetDevice(g->device);
        auto fr = [](void* q) { if (q) (void)hipFree(q); };
        fr(vec_pool); fr(partials); fr(d_rr); fr(d_alpha); fr(d_beta); fr(d_alpha_hist); fr(d_beta_hist);
        fr(d_active); fr(d_iters_tmp); fr(d_nact); fr(d_nonfinite); fr(d_ps); fr(d_hist); fr(d_dxps);
        fr(d_dxpart); fr(d_hist_ps); fr(d_cg_iters); fr(d_lds_csr); fr(d_m2); fr(d_stop); fr(d_ps_ring);
        for (float* b : lds_ring_extra) if (b) (void)hipFree(b);
        if (st_side) (void)hipStreamDestroy(st_side);
        for (auto& e : ev_main) if (e) (void)hipEventDestroy(e);
        for (auto& e : ev_side) if (e) (void)hipEventDestroy(e);
        for (auto& tmd : tmeta) { fr(tmd.tl_col); fr(tmd.tl_w); fr(tmd.halo); fr(tmd.h_rowptr); fr(tmd.h_col); fr(tmd.h_val); }
        fr(cldr_dev.n0); fr(cldr_dev.nC); fr(cldr_dev.rows); fr(cldr_dev.dcol); fr(cldr_dev.dcnt); fr(cldr_dev.tcol); fr(cldr_dev.tcnt);
        fr(cldr_dev.dw); fr(cldr_dev.tw);
        if (h_nact) (void)hipHostFree(h_nact);
        if (h_row) (void)hipHostFree(h_row);
        if (h_flag) (void)hipHostFree(h_flag);
        for (auto& e : ev_ring) if (e) (void)hipEventDestroy(e);
        for (auto& e : prof_ev) (void)hipEventDestroy(e);
    }

    // ---------------------------------------------------------------- geometry
    Geom make_geom(int B) const {
        Geom q;
        q.T = T; q.N = N; q.B = B;
        int vecw;
        if (sizeof(S) == 4) vecw = B >= 192 ? 4 : (B >= 96 ? 2 : 1);
        else vecw = B >= 96 ? 2 : 1;
        const int cw = 64 * vecw;
        q.VEC = vecw;
        q.Bp = (B + cw - 1) / cw * cw;
        q.CH = q.Bp / cw;
        // persistent-sized grid: ~want_blocks workgroups in total, a multiple of 8 per column chunk
        int g8 = want_blocks / (8 * q.CH);
        if (g8 < 1) g8 = 1;
        q.G8 = g8;
        q.P = 8 * g8;
        // rows per work item: 16 (4 per wave) unless the problem is too small to give every workgroup two items
        long rows = (long)T * N;
        int ri = 16;
        while (ri > 4 && rows / ri < 2L * q.P) ri -= 4;
        q.RI = ri;
        const int per_xcd = (N + 7) / 8;
        q.NBL = (per_xcd + ri - 1) / ri;
        q.NX = q.NBL * ri;
        q.n_items = T * q.NBL;
        q.grid = q.P * q.CH;
        q.rev = 0;
        return q;
    }

    // geometry of the LDS-tiled spatial kernel for the same column layout as `q`
    bool make_tile_geom(const Geom& q, TileGeom& tg) const {
        if (!use_tile || g->reorder < 2 || g->mode != MGADMM_TEMPORAL_SPATIAL) return false;
        tg.T = T; tg.N = N; tg.B = q.B; tg.Bp = q.Bp; tg.VEC = q.VEC; tg.CH = q.CH;
        // tile size: 8 rows (2 per wave).  With the rows per wave a template constant the register footprint
        // follows the tile size (EpiLhs: 100 VGPRs at 2 rows, 148 at 5), and the kernel is latency-bound, so
        // more resident workgroups win: SpMM inside CG on the 10k-node graph 4.30 TB/s at 20 rows, 4.60 at 12,
        // 4.72 at 8, 4.71 at 4 (more halo reads); the 100k-node graph is flat (4.86 -> 4.91).  MGADMM_TILE_R=20
        // selects the large tile for experiments.
        int best = 8;
        if (const char* e = getenv("MGADMM_TILE_R")) { int rr = atoi(e); if (rr == 8 || rr == 20) best = rr; }
        tg.R = best;
        tg.NTILE = (N + best - 1) / best;
        tg.TPX = (tg.NTILE + 7) / 8;
        tg.P = 8 * tg.TPX;
        tg.grid = tg.P * q.CH;
        const size_t row_bytes = (size_t)64 * q.VEC * sizeof(S);
        tg.lds_bytes = (int)std::max(row_bytes * (best + TILE_HMAX), (size_t)3 * q.VEC * 64 * sizeof(S));
        return true;
    }

    // host-side preprocessing of one CSR matrix for tile size R (see TileMeta in stream_kernels.h)
    int tile_meta(int which, int R, int TILE_GW, TileMeta& out) {
        TileMetaDev& d = tmeta[which];
        if (d.R != R || d.GW != TILE_GW) {
            const HostCsr& src = which == 0 ? g->hWu : (which == 1 ? g->hWd : g->hWdT);
            HostCsr A;
            if (g->has_perm) mg_permute_csr(src, g->perm, g->iperm, A);
            else A = src;
            const int ntile = (N + R - 1) / R;
            std::vector<int> tc((size_t)N * TILE_GW), hr(N + 1, 0), hcol, halo((size_t)ntile * TILE_HMAX, -1);
            std::vector<float> tw((size_t)N * TILE_GW, 0.f), hval;
            for (int tl = 0; tl < ntile; ++tl) {
                const int lo = tl * R, hi = std::min(N, lo + R);
                // halo list of the tile: out-of-tile columns in order of first use, at most TILE_HMAX
                std::vector<int> hl;
                auto halo_pos = [&](int c) -> int {
                    for (size_t k = 0; k < hl.size(); ++k)
                        if (hl[k] == c) return (int)k;
                    if ((int)hl.size() < TILE_HMAX) { hl.push_back(c); return (int)hl.size() - 1; }
                    return -1;
                };
                for (int i = lo; i < hi; ++i) {
                    int used = 0;
                    for (int e = A.rowptr[i]; e < A.rowptr[i + 1]; ++e) {
                        const int c = A.col[e];
                        int local = -1;
                        if (used < TILE_GW) {
                            if (c >= lo && c < hi) local = c - lo;
                            else {
                                const int hp = halo_pos(c);
                                if (hp >= 0) local = R + hp;
                            }
                        }
                        if (local >= 0) {
                            tc[(size_t)i * TILE_GW + used] = local;
                            tw[(size_t)i * TILE_GW + used] = A.val[e];
                            ++used;
                        } else {
                            hcol.push_back(c);
                            hval.push_back(A.val[e]);
                        }
                    }
                    for (; used < TILE_GW; ++used) tc[(size_t)i * TILE_GW + used] = i - lo;
                    hr[i + 1] = (int)hcol.size();
                }
                for (size_t k = 0; k < hl.size(); ++k) halo[(size_t)tl * TILE_HMAX + k] = hl[k];
            }
            if (getenv("MGADMM_TILE_STATS")) {           // diagnostics: how well the tile geometry fits the graph
                long hsum = 0; int hfull = 0;
                for (int tl = 0; tl < ntile; ++tl) {
                    int c = 0;
                    for (int k = 0; k < TILE_HMAX; ++k) c += halo[(size_t)tl * TILE_HMAX + k] >= 0;
                    hsum += c; hfull += c == TILE_HMAX;
                }
                fprintf(stderr, "[mgadmm] tile_meta matrix %d: R=%d GW=%d tiles=%d  halo rows/tile mean %.2f (capacity %d, full tiles %d)  "
                                "entries %d, overflow entries %zu (%.2f %%)\n", which, R, TILE_GW, ntile, (double)hsum / ntile, TILE_HMAX,
                        hfull, A.nnz(), hcol.size(), 100.0 * hcol.size() / std::max(1, A.nnz()));
            }
            MG_HIP(hipStreamSynchronize(st));
            auto fr = [](void* q) { if (q) (void)hipFree(q); };
            fr(d.tl_col); fr(d.tl_w); fr(d.halo); fr(d.h_rowptr); fr(d.h_col); fr(d.h_val);
            d = TileMetaDev();
            const size_t nh = hcol.size() + 8;
            hcol.resize(nh, 0);
            hval.resize(nh, 0.f);
            MG_HIP(hipMalloc(&d.tl_col, tc.size() * sizeof(int)));
            MG_HIP(hipMalloc(&d.tl_w, tw.size() * sizeof(float)));
            MG_HIP(hipMalloc(&d.halo, halo.size() * sizeof(int)));
            MG_HIP(hipMemcpy(d.halo, halo.data(), halo.size() * sizeof(int), hipMemcpyHostToDevice));
            MG_HIP(hipMalloc(&d.h_rowptr, hr.size() * sizeof(int)));
            MG_HIP(hipMalloc(&d.h_col, nh * sizeof(int)));
            MG_HIP(hipMalloc(&d.h_val, nh * sizeof(float)));
            MG_HIP(hipMemcpy(d.tl_col, tc.data(), tc.size() * sizeof(int), hipMemcpyHostToDevice));
            MG_HIP(hipMemcpy(d.tl_w, tw.data(), tw.size() * sizeof(float), hipMemcpyHostToDevice));
            MG_HIP(hipMemcpy(d.h_rowptr, hr.data(), hr.size() * sizeof(int), hipMemcpyHostToDevice));
            MG_HIP(hipMemcpy(d.h_col, hcol.data(), nh * sizeof(int), hipMemcpyHostToDevice));
            MG_HIP(hipMemcpy(d.h_val, hval.data(), nh * sizeof(float), hipMemcpyHostToDevice));
            d.R = R;
            d.GW = TILE_GW;
        }
        out.tl_col = d.tl_col; out.tl_w = d.tl_w; out.halo = d.halo; out.h_rowptr = d.h_rowptr; out.h_col = d.h_col; out.h_val = d.h_val;
        return MGADMM_OK;
    }

    // ---------------------------------------------------------------- fused cLdr kernel (k_cldr)
    // Tile geometries (VECT columns per lane, NW waves, rows per wave of the tile / of C1 / of C2).  LDS per workgroup =
    // NW * (MQ + MP) * 64 * VECT * sizeof(S).  MGADMM_CLDR_GEOM selects one (experiments); default per scalar type below.
    template <int VECT_, int NW_, int MA_, int MQ_, int MP_, int MINW_>
    struct ClG { static constexpr int VECT = VECT_, NW = NW_, MA = MA_, MQ = MQ_, MP = MP_, MINW = MINW_; };
    // measured on cfg3 (N = 10 000, B = 512), SpMM + LHS launch / with the folded vector update:
    //   geometry 1  380 us / 644 us      geometry 2  700 us / -      (a 16 / 32 / 48-row geometry 0 of 80 KiB ran 365 us / 711 us and
    //   spilled 52 VGPRs in the folded form: removed in round 3)
    typedef ClG<4, 8, 2, 4, 5, 4> ClG1;     // 16 / 32 / 40 rows of 256 columns: 72 KiB (float), two workgroups per CU   (default)
    typedef ClG<1, 8, 8, 11, 15, sizeof(S) == 4 ? 4 : 2> ClG2;   // 64 / 88 / 120 rows of 64 columns: 52 KiB (float) / 104 KiB (double); float: 128 VGPRs (6 waves per SIMD = 80 VGPRs spilled 87-100 of them in the folded form)
    typedef ClG<2, 8, 4, 8, 10, 4> ClG3;    // 32 / 64 / 80 rows of 128 columns: 72 KiB (float): twice the rows per tile, a smaller halo share
    typedef ClG<4, 16, 2, 3, 4, 4> ClG4;    // 32 / 48 / 64 rows of 256 columns, 16 waves: 112 KiB (float), one workgroup per CU
    // W_d^T slots per row: 12, or 16 / 24 when a row is longer (round 3: the PEMS-like graphs of 600 ... 2000 nodes have rows of 13 and
    // 14 entries and fell back to the two-pass path; the 16-slot instance is 5 % slower on graphs that do not need it).
    // W_d slots: 6 (no test at all) when no row is longer, else 8
    int cl_gt = 12;
    int cl_gd = 8;
    int cl_geom = sizeof(S) == 4 ? 1 : 2;   // float64: the narrow geometry (104 KiB)
    void cl_dims(int& vect, int& nw, int& ma, int& mq, int& mp) const {
        switch (cl_geom) {
            case 1: vect = ClG1::VECT; nw = ClG1::NW; ma = ClG1::MA; mq = ClG1::MQ; mp = ClG1::MP; break;
            case 3: vect = ClG3::VECT; nw = ClG3::NW; ma = ClG3::MA; mq = ClG3::MQ; mp = ClG3::MP; break;
            case 4: vect = ClG4::VECT; nw = ClG4::NW; ma = ClG4::MA; mq = ClG4::MQ; mp = ClG4::MP; break;
            default: vect = ClG2::VECT; nw = ClG2::NW; ma = ClG2::MA; mq = ClG2::MQ; mp = ClG2::MP; break;
        }
    }
    bool cldr_usable() {
        if (!use_fused || !use_tile || g->reorder < 2 || g->mode != MGADMM_TEMPORAL_SPATIAL) return false;
        if (cldr_dev.state == 0) cldr_prepare();
        return cldr_dev.state == 1;
    }
    void cldr_prepare() {
        cldr_dev.state = -1;
        if (const char* e = getenv("MGADMM_CLDR_GEOM")) { const int v = atoi(e); if (v >= 1 && v <= 4 && (v <= 2 || sizeof(S) == 4)) cl_geom = v; }
        int vect, nw, ma, mq, mp;
        cl_dims(vect, nw, ma, mq, mp);
        if ((size_t)nw * (mq + mp) * 64 * vect * sizeof(S) > 150 * 1024) { cl_geom = 2; cl_dims(vect, nw, ma, mq, mp); }
        HostCsr A, At;
        if (g->has_perm) { mg_permute_csr(g->hWd, g->perm, g->iperm, A); mg_permute_csr(g->hWdT, g->perm, g->iperm, At); }
        else { A = g->hWd; At = g->hWdT; }
        cl_gd = 6;
        cl_gt = 12;
        for (int i = 0; i < N; ++i) {
            if (A.rowptr[i + 1] - A.rowptr[i] > 6) cl_gd = 8;
            const int tl = At.rowptr[i + 1] - At.rowptr[i];
            if (tl > 12) cl_gt = std::max(cl_gt, tl > 16 ? 24 : 16);
        }
        CldrCaps caps{nw * ma, nw * mq, nw * mp, cl_gd, cl_gt};
        CldrTiles tl;
        int row_limit = 0;
        if (const char* e = getenv("MGADMM_CLDR_ROWS")) row_limit = atoi(e);
        if (!build_cldr_tiles(A, At, g->cluster_starts, caps, tl, row_limit)) return;
        if (getenv("MGADMM_TILE_STATS"))
            fprintf(stderr, "[mgadmm] cldr tiles (geometry %d): %d tiles, rows/tile %.1f, |C1| %.1f, |C2| %.1f (caps %d/%d/%d): reads %.2fx, q recomputed %.2fx\n",
                    cl_geom, tl.NT, (double)N / tl.NT, (double)tl.sumC1 / tl.NT, (double)tl.sumC2 / tl.NT, caps.Rcap, caps.C1cap, caps.C2cap,
                    (double)tl.sumC2 / N, (double)tl.sumC1 / N);
        std::vector<int> nC((size_t)2 * tl.NT);
        for (int t = 0; t < tl.NT; ++t) { nC[2 * t] = tl.nC1[t]; nC[2 * t + 1] = tl.nC2[t]; }
        auto up = [&](auto*& dst, const auto& v) -> bool {
            typedef typename std::remove_reference<decltype(*dst)>::type E;
            if (hipMalloc(&dst, std::max<size_t>(1, v.size()) * sizeof(E)) != hipSuccess) return false;
            return v.empty() || hipMemcpy(dst, v.data(), v.size() * sizeof(E), hipMemcpyHostToDevice) == hipSuccess;
        };
        if (!(up(cldr_dev.n0, tl.n0) && up(cldr_dev.nC, nC) && up(cldr_dev.rows, tl.rows) && up(cldr_dev.dcol, tl.dcol) &&
              up(cldr_dev.dw, tl.dw) && up(cldr_dev.dcnt, tl.dcnt) && up(cldr_dev.tcol, tl.tcol) && up(cldr_dev.tw, tl.tw) &&
              up(cldr_dev.tcnt, tl.tcnt))) {
            (void)hipGetLastError();
            return;
        }
        cldr_dev.NT = tl.NT;
        cldr_dev.state = 1;
    }
    CldrGeom make_cldr_geom(const Geom& q) const {
        int vect, nw, ma, mq, mp;
        cl_dims(vect, nw, ma, mq, mp);
        CldrGeom cg;
        cg.T = T; cg.N = N; cg.B = q.B; cg.Bp = q.Bp;
        while (q.Bp % (64 * vect)) vect /= 2;              // never happens: Bp is padded to the solver's own vector width
        cg.CH = q.Bp / (64 * vect);
        cg.NT = cldr_dev.NT;
        cg.TPX = (cg.NT + 7) / 8;
        cg.P = 8 * cg.TPX;
        cg.grid = cg.P * cg.CH;
        cg.tile_major = cldr_tile_major;
        cg.q1 = g->q1;
        cg.lds_bytes = (int)((size_t)nw * (mp + mq) * 64 * vect * sizeof(S));
        return cg;
    }
    // the fused kernel needs Bp to be a multiple of its own chunk width
    bool cldr_fits(const Geom& q) {
        if (!cldr_usable()) return false;
        int vect, nw, ma, mq, mp;
        cl_dims(vect, nw, ma, mq, mp);
        return q.Bp % (64 * vect) == 0 && (sizeof(S) == 4 || vect <= 2);
    }
    template <class G, template <typename, int> class E, template <typename, int> class SRC, class... A>
    int rows_cldr_g(const Geom& q, const SRC<S, G::VECT>& src, const int* live, A... a) {
        if (cl_gt == 12) return cl_gd == 6 ? rows_cldr_gd<G, 6, 12, E, SRC>(q, src, live, a...) : rows_cldr_gd<G, 8, 12, E, SRC>(q, src, live, a...);
        if (cl_gt == 16) return cl_gd == 6 ? rows_cldr_gd<G, 6, 16, E, SRC>(q, src, live, a...) : rows_cldr_gd<G, 8, 16, E, SRC>(q, src, live, a...);
        return cl_gd == 6 ? rows_cldr_gd<G, 6, 24, E, SRC>(q, src, live, a...) : rows_cldr_gd<G, 8, 24, E, SRC>(q, src, live, a...);
    }
    template <class G, int GD, int GT, template <typename, int> class E, template <typename, int> class SRC, class... A>
    int rows_cldr_gd(const Geom& q, const SRC<S, G::VECT>& src, const int* live, A... a) {
        const CldrGeom cg = make_cldr_geom(q);
        CldrMeta mm{cldr_dev.n0, cldr_dev.nC, cldr_dev.rows, cldr_dev.dcol, cldr_dev.dw, cldr_dev.dcnt, cldr_dev.tcol, cldr_dev.tw, cldr_dev.tcnt};
        typedef E<S, G::VECT> Epi;
        auto fn = k_cldr<S, G::VECT, Epi, SRC<S, G::VECT>, G::NW, G::MA, G::MQ, G::MP, GD, GT, G::MINW>;
        MG_TRY(allow_dynamic_lds((const void*)fn, 150 * 1024));
        hipLaunchKernelGGL(fn, dim3(cg.grid), dim3(G::NW * 64), cg.lds_bytes, st, cg, mm, src, Epi{a...}, partials, live);
        cur_P = cg.P;
        return MGADMM_OK;
    }
    template <template <typename, int> class E, template <typename, int> class SRC, class MK, class... A>
    int rows_cldr_any(const Geom& q, MK mk_src, const int* live, int tag, double bytes, A... a) {
        const bool timed = prof_open(tag, bytes);
        int rc;
        if constexpr (sizeof(S) == 4) {
            switch (cl_geom) {
                case 1: rc = rows_cldr_g<ClG1, E, SRC>(q, mk_src(ClG1()), live, a...); break;
                case 3: rc = rows_cldr_g<ClG3, E, SRC>(q, mk_src(ClG3()), live, a...); break;
                case 4: rc = rows_cldr_g<ClG4, E, SRC>(q, mk_src(ClG4()), live, a...); break;
                default: rc = rows_cldr_g<ClG2, E, SRC>(q, mk_src(ClG2()), live, a...); break;
            }
        } else {
            rc = rows_cldr_g<ClG2, E, SRC>(q, mk_src(ClG2()), live, a...);
        }
        if (timed) prof_close();
        MG_TRY(rc);
        MG_HIP(hipGetLastError());
        return MGADMM_OK;
    }
    // in -> epilogue(l = Ldr^T Ldr in); `passes`: algorithmic vector passes of the launch for the roofline accounting
    template <template <typename, int> class E, class... A>
    int rows_cldr(const Geom& q, const S* in, const int* live, int tag, int passes, A... a) {
        const double bytes = pass_bytes(q, passes) + csr_bytes(g->op_ldr()) + csr_bytes(g->op_ldrt());
        return rows_cldr_any<E, CldrSrcPlain>(q, [&](auto gg) { return CldrSrcPlain<S, decltype(gg)::VECT>{in}; }, live, tag, bytes, a...);
    }
    // the same with the CG direction formed on load: in = p_new = r + beta p_old; owners store p_new and x += alpha p_old
    template <template <typename, int> class E, class... A>
    int rows_cldr_fold(const Geom& q, const S* r, const S* p_old, S* p_new, S* x, const int* live, int tag, int passes, A... a) {
        const double bytes = pass_bytes(q, passes) + csr_bytes(g->op_ldr()) + csr_bytes(g->op_ldrt());
        return rows_cldr_any<E, CldrSrcFold>(q, [&](auto gg) { return CldrSrcFold<S, decltype(gg)::VECT>{r, p_old, p_new, x, d_alpha, d_beta}; },
                                             live, tag, bytes, a...);
    }

    int ensure_partials(const Geom& q) {
        TileGeom tg;
        int pmax = make_tile_geom(q, tg) ? std::max(q.P, tg.P) : q.P;
        if (cldr_fits(q)) pmax = std::max(pmax, make_cldr_geom(q).P);
        size_t need = (size_t)NRED_MAX * pmax * q.Bp;
        if (need <= partials_elems) return MGADMM_OK;
        MG_HIP(hipStreamSynchronize(st));
        if (partials) MG_HIP(hipFree(partials));
        partials = nullptr;
        MG_HIP(hipMalloc(&partials, need * sizeof(S)));
        ws_bytes += (int64_t)(need - partials_elems) * sizeof(S);
        partials_elems = need;
        return MGADMM_OK;
    }

    int init() override {
        MG_HIP(hipSetDevice(g->device));
        if (const char* e = getenv("MGADMM_WANT_BLOCKS")) want_blocks = std::max(64, atoi(e));
        if (const char* e = getenv("MGADMM_TILE")) use_tile = atoi(e);
        if (const char* e = getenv("MGADMM_FUSED")) use_fused = atoi(e);
        if (const char* e = getenv("MGADMM_FOLD")) use_fold = atoi(e);
        if (const char* e = getenv("MGADMM_FOLD_LU")) use_fold_lu = atoi(e);
        if (const char* e = getenv("MGADMM_SWEEP_REV")) sweep_rev = atoi(e);
        if (const char* e = getenv("MGADMM_CLDR_ORDER")) cldr_tile_major = atoi(e);
        if (const char* e = getenv("MGADMM_LDS_ASYNC")) lds_async = atoi(e);
        if (const char* e = getenv("MGADMM_LDS_CHUNK")) lds_chunk = std::max(1, std::min(atoi(e), LDS_MAXJ_POOL));
        Geom q = make_geom(Bmax);
        Bp_max = q.Bp;
        // Bp for smaller batches never exceeds Bp_max rounded to 256
        Bp_max = std::max(Bp_max, ((Bmax + 63) / 64) * 64);
        vec_elems = (size_t)T * N * Bp_max;
        MG_HIP(hipMalloc(&vec_pool, vec_elems * V_COUNT * sizeof(S)));
        ws_bytes += (int64_t)vec_elems * V_COUNT * sizeof(S);
        for (int i = 0; i < V_COUNT; ++i) vec[i] = vec_pool + vec_elems * i;
        MG_HIP(hipMalloc(&d_rr, Bp_max * sizeof(S)));
        MG_HIP(hipMalloc(&d_alpha, Bp_max * sizeof(S)));
        MG_HIP(hipMalloc(&d_beta, Bp_max * sizeof(S)));
        MG_HIP(hipMalloc(&d_active, Bp_max * sizeof(int)));
        MG_HIP(hipMalloc(&d_iters_tmp, Bp_max * sizeof(int)));
        MG_HIP(hipMalloc(&d_nonfinite, sizeof(int)));
        MG_HIP(hipMemset(d_nonfinite, 0, sizeof(int)));
        MG_HIP(hipMalloc(&d_ps, sizeof(double) * MGADMM_NMETRIC * Bp_max));
        const int nbk = (N + 63) / 64;
        MG_HIP(hipMalloc(&d_dxpart, sizeof(double) * T * nbk));
        MG_HIP(hipHostMalloc(&h_row, sizeof(double) * (MGADMM_NMETRIC + 1)));
        MG_HIP(hipHostMalloc(&h_flag, sizeof(int) * 4));
        for (auto& e : ev_ring) MG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        MG_TRY(alloc_iter_dependent());
        MG_TRY(plan_lds());
        return MGADMM_OK;
    }

    int alloc_iter_dependent() {
        auto fr = [](void* q) { if (q) (void)hipFree(q); };
        if (p.max_cg_iter > max_cg_alloc) {
            fr(d_nact); fr(d_alpha_hist); fr(d_beta_hist);
            if (h_nact) (void)hipHostFree(h_nact);
            d_nact = nullptr; d_alpha_hist = d_beta_hist = nullptr; h_nact = nullptr;
            max_cg_alloc = p.max_cg_iter;
            MG_HIP(hipMalloc(&d_nact, sizeof(int) * max_cg_alloc));
            MG_HIP(hipMalloc(&d_alpha_hist, sizeof(S) * 3 * (size_t)max_cg_alloc * Bp_max));
            MG_HIP(hipMalloc(&d_beta_hist, sizeof(S) * 3 * (size_t)max_cg_alloc * Bp_max));
            MG_HIP(hipHostMalloc(&h_nact, sizeof(int) * (NACT_LOG + (size_t)max_cg_alloc)));
            nact_cur = 0;
        }
        if (p.max_admm_iter > max_admm_alloc) {
            fr(d_hist); fr(d_dxps); fr(d_cg_iters);
            d_hist = d_dxps = nullptr; d_cg_iters = nullptr;
            max_admm_alloc = p.max_admm_iter;
            MG_HIP(hipMalloc(&d_hist, sizeof(double) * (size_t)max_admm_alloc * MGADMM_NMETRIC));
            MG_HIP(hipMalloc(&d_dxps, sizeof(double) * (size_t)max_admm_alloc * T));
            MG_HIP(hipMalloc(&d_cg_iters, sizeof(int) * (size_t)max_admm_alloc * 3 * Bp_max));
        }
        return MGADMM_OK;
    }

    int set_params(const mgadmm_params& np) override {
        MG_REQUIRE(np.dtype == p.dtype, "set_params: dtype cannot change after solver_create");
        MG_REQUIRE(np.t_in >= 1 && np.t_in <= T, "set_params: t_in out of range");
        MG_REQUIRE(np.max_cg_iter >= 1 && np.max_admm_iter >= 1, "set_params: iteration limits must be >= 1");
        MG_REQUIRE(np.ablation >= 0 && np.ablation <= 3, "set_params: bad ablation");
        MG_REQUIRE(np.cg_convergence == MGADMM_CG_PER_SAMPLE || np.cg_convergence == MGADMM_CG_BATCH_MAX,
                   "set_params: cg_convergence should be per_sample (0) or batch_max (1), got %d", np.cg_convergence);
        MG_REQUIRE(np.max_inner_iter >= 0, "set_params: max_inner_iter must be >= 0 (got %d)", np.max_inner_iter);
        if (np.path == MGADMM_PATH_LDS && np.cg_convergence == MGADMM_CG_BATCH_MAX) {
            mg_set_error("set_params: the LDS-resident path implements per-sample CG convergence only (batch_max: streaming path)");
            return MGADMM_ERR_UNSUPPORTED;
        }
        p = np;
        sv->p = np;
        MG_HIP(hipSetDevice(g->device));
        return alloc_iter_dependent();
    }

    int64_t workspace_bytes() const override { return ws_bytes; }
    int path_for(int) const override { return use_lds() ? MGADMM_PATH_LDS : MGADMM_PATH_STREAM; }
    // the fused LDS kernel runs one sample per workgroup with its CG loops inside: a batch-global stop cannot be expressed there
    bool use_lds() const { return lds.ok && p.path != MGADMM_PATH_STREAM && p.cg_convergence == MGADMM_CG_PER_SAMPLE; }
    int query(int what, int64_t* out) const override {
        switch (what) {
            case MGADMM_Q_LDS_OK: *out = lds.ok ? 1 : 0; break;
            case MGADMM_Q_LDS_TPG: *out = lds.TPG; break;
            case MGADMM_Q_LDS_THREADS: *out = lds.nthreads; break;
            case MGADMM_Q_LDS_BYTES: *out = (int64_t)lds.lds_bytes; break;
            case MGADMM_Q_LDS_ROW_STRIDE: *out = lds.TS; break;
            case MGADMM_Q_LDS_UNIFORM: *out = lds.uniform45; break;
            case MGADMM_Q_LDS_TAIL_PAIRS: *out = lds.tail_pairs; break;
            case MGADMM_Q_LDS_LEAD: *out = LDS_NLEAD; break;
            case MGADMM_Q_LDS_SLOTS: *out = lds.slots; break;
            case MGADMM_Q_LDS_CHUNK: *out = std::max(1, std::min(lds_chunk, LDS_MAXJ_POOL)); break;
            case MGADMM_Q_LDS_ROWS: *out = lds.NR; break;
            case MGADMM_Q_CLDR_SLOTS: *out = cldr_dev.state == 1 ? cl_gt : 0; break;      // (prepared by the first operator application)
            case MGADMM_Q_NNZ_U: *out = g->hWu.nnz(); break;
            case MGADMM_Q_NNZ_D: *out = g->hWd.nnz(); break;
            case MGADMM_Q_NNZ_DT: *out = g->hWdT.nnz(); break;
            case MGADMM_Q_TILE_ROWS: {
                TileGeom tg;
                *out = make_tile_geom(make_geom(Bmax), tg) ? tg.R : 0;
                break;
            }
            default: mg_set_error("solver_query: unknown item %d", what); return MGADMM_ERR_INVALID;
        }
        return MGADMM_OK;
    }

    // ---------------------------------------------------------------- profiling
    int prof_begin() override {
        MG_HIP(hipSetDevice(g->device));
        if (prof_ev.empty()) {
            prof_ev.resize(2 * PROF_POOL);
            for (auto& e : prof_ev) MG_HIP(hipEventCreateWithFlags(&e, hipEventDisableSystemFence));   // timing only: no system-scope cache flush per record
            prof_tag.resize(PROF_POOL);
            prof_lb.resize(PROF_POOL);
            prof_ref.resize(PROF_POOL);
        }
        prof_used = 0;
        nact_cur = 0;
        nact_locked = false;
        for (int i = 0; i < MGADMM_NPROF; ++i) { prof_count[i] = 0; prof_bytes[i] = 0; }
        prof_on = true;
        return MGADMM_OK;
    }
    int prof_end(int64_t* counts, double* total_ms, double* bytes) override {
        prof_on = false;
        MG_HIP(hipDeviceSynchronize());
        double ms[MGADMM_NPROF] = {0}, by[MGADMM_NPROF] = {0};
        int64_t timed[MGADMM_NPROF] = {0};
        prof_noop = 0;
        for (size_t i = 0; i < prof_used; ++i) {
            float f = 0;
            MG_HIP(hipEventElapsedTime(&f, prof_ev[2 * i], prof_ev[2 * i + 1]));
            // A speculative CG launch of iteration k+1 that found every sample converged after iteration k returns
            // at its `live` guard without touching its operands.  Whether that happened is read from the log of
            // active-sample counts the host copies back for its own convergence check (h_nact), not guessed from
            // the duration: such a launch is left out of BOTH the bytes and the time, so the reported rates are
            // those of launches that did the work.
            if (prof_ref[i] >= 0 && h_nact[prof_ref[i]] == 0) {
                prof_noop++;
                continue;
            }
            ms[prof_tag[i]] += f;
            by[prof_tag[i]] += prof_lb[i];
            timed[prof_tag[i]]++;
        }
        for (int i = 0; i < MGADMM_NPROF; ++i) {
            counts[i] = timed[i];
            total_ms[i] = ms[i];
            bytes[i] = by[i];
        }
        return MGADMM_OK;
    }
    inline bool prof_open(int tag, double bytes) {
        if (!prof_on) return false;
        prof_count[tag]++;
        prof_bytes[tag] += bytes;
        if (tag == 3 || prof_used >= (size_t)PROF_POOL) return false;
        prof_tag[prof_used] = tag;
        prof_lb[prof_used] = bytes;
        prof_ref[prof_used] = prof_cur_ref;
        (void)hipEventRecord(prof_ev[2 * prof_used], st);
        return true;
    }
    inline void prof_close() {
        (void)hipEventRecord(prof_ev[2 * prof_used + 1], st);
        prof_used++;
    }

    double pass_bytes(const Geom& q, int passes) const { return (double)passes * q.B * (double)T * N * sizeof(S); }
    double csr_bytes(const OpDesc& op) const {
        if (op.kind != OPK_SPATIAL) return 0.0;
        int nnz = 0;
        if (op.rowptr == g->Wu.rowptr) nnz = g->Wu.nnz;
        else if (op.rowptr == g->Wd.rowptr) nnz = g->Wd.nnz;
        else nnz = g->WdT.nnz;
        return (double)nnz * 8 + (double)(N + 1) * 4;
    }

    // ---------------------------------------------------------------- launch helpers
    // hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: remember which (kernel, device)
    // pairs have been raised -- one solver per GPU may live in the same process.
    std::vector<const void*> lds_attr_done;
    int allow_dynamic_lds(const void* fn, int bytes) {
        for (const void* f : lds_attr_done)
            if (f == fn) return MGADMM_OK;
        MG_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        lds_attr_done.push_back(fn);
        return MGADMM_OK;
    }
    template <class E, class = void>
    struct is_elementwise : std::false_type {};
    template <class E>
    struct is_elementwise<E, std::void_t<decltype(E::ELEMENTWISE)>> : std::true_type {};

    template <int VEC, class Epi, int TGW>
    int launch_tile(const TileGeom& tg, const OpDesc& op, const TileMeta& tmv, const S* in, const Epi& epi, const int* live) {
        return tg.R == 20 ? launch_tile2<VEC, Epi, TGW, 5>(tg, op, tmv, in, epi, live)
                          : launch_tile2<VEC, Epi, TGW, 2>(tg, op, tmv, in, epi, live);
    }
    template <int VEC, class Epi, int TGW, int MR>
    int launch_tile2(const TileGeom& tg, const OpDesc& op, const TileMeta& tmv, const S* in, const Epi& epi, const int* live) {
        auto fn = k_tile<S, VEC, Epi, TGW, MR>;
        MG_TRY(allow_dynamic_lds((const void*)fn, 80 * 1024));
        hipLaunchKernelGGL(fn, dim3(tg.grid), dim3(256), tg.lds_bytes, st, tg, op, tmv.tl_col, tmv.tl_w, tmv.halo, tmv.h_rowptr,
                           tmv.h_col, tmv.h_val, in, epi, partials, live, TileSrcPlain<S, VEC>());
        return MGADMM_OK;
    }

    // ---- Lu with the CG vector update folded into its loads (k_tile with TileSrcFold): zu solve.  Built for the
    // production shape only (k = 4 table: 4 slots per row; 8-row tiles; the solver's widest vector): other shapes keep
    // the separate EpiPUpdate launch.
    static constexpr int LU_FOLD_VEC = sizeof(S) == 4 ? 4 : 2;
    bool lu_fold_fits(const Geom& q) {
        TileGeom tg;
        if (!use_fold || !use_fold_lu || !use_tile || g->reorder < 2 || q.VEC != LU_FOLD_VEC || g->Wu.max_row > 4) return false;
        return make_tile_geom(q, tg) && tg.R != 20;
    }
    template <template <typename, int> class E, class... A>
    int rows_lu_fold(const Geom& q, const S* r, const S* p_old, S* p_new, S* x, const int* live, int tag, int passes, A... a) {
        constexpr int VEC = LU_FOLD_VEC;
        typedef E<S, VEC> Epi;
        typedef TileSrcFold<S, VEC> Src;
        const OpDesc op = g->op_lu();
        const bool timed = prof_open(tag, pass_bytes(q, passes) + csr_bytes(op));
        TileGeom tg;
        if (!make_tile_geom(q, tg)) { mg_set_error("rows_lu_fold: no tile geometry"); return MGADMM_ERR_INVALID; }
        TileMeta tmv;
        MG_TRY(tile_meta(0, tg.R, 4, tmv));
        auto fn = k_tile<S, VEC, Epi, 4, 2, Src>;
        MG_TRY(allow_dynamic_lds((const void*)fn, 80 * 1024));
        hipLaunchKernelGGL(fn, dim3(tg.grid), dim3(256), tg.lds_bytes, st, tg, op, tmv.tl_col, tmv.tl_w, tmv.halo, tmv.h_rowptr,
                           tmv.h_col, tmv.h_val, r, Epi{a...}, partials, live, Src{p_old, p_new, x, d_alpha, d_beta});
        cur_P = tg.P;
        if (timed) prof_close();
        MG_HIP(hipGetLastError());
        return MGADMM_OK;
    }

    template <int VEC, int GW, class Epi>
    int rows_v(const Geom& q, const OpDesc& op, const S* in, const Epi& epi, const int* live, int tag, double bytes) {
        const bool timed = prof_open(tag, bytes);
        TileGeom tg;
        if (op.kind == OPK_SPATIAL && op.self_w == nullptr && make_tile_geom(q, tg)) {
            if constexpr (is_elementwise<Epi>::value) {
                mg_set_error("rows: element-wise epilogue launched with a spatial operator");
                return MGADMM_ERR_INVALID;
            } else {
                const int which = op.rowptr == g->Wu.rowptr ? 0 : (op.rowptr == g->Wd.rowptr ? 1 : 2);
                // neighbour slots kept in VGPR lanes: 4 for W_u (k = 4), 6 for W_d, 8 for the ragged W_d^T rows
                const DevCsr& dc = which == 0 ? g->Wu : (which == 1 ? g->Wd : g->WdT);
                const int tgw = dc.max_row <= 4 ? 4 : (which == 2 || dc.max_row > 6 ? 8 : 6);
                TileMeta tmv;
                MG_TRY(tile_meta(which, tg.R, tgw, tmv));
                switch (tgw) {
                    case 4: MG_TRY((launch_tile<VEC, Epi, 4>(tg, op, tmv, in, epi, live))); break;
                    case 6: MG_TRY((launch_tile<VEC, Epi, 6>(tg, op, tmv, in, epi, live))); break;
                    default: MG_TRY((launch_tile<VEC, Epi, 8>(tg, op, tmv, in, epi, live))); break;
                }
                cur_P = tg.P;
            }
        } else {
            hipLaunchKernelGGL((k_rows<S, VEC, Epi, GW>), dim3(q.grid), dim3(256), 0, st, q, op, op.rowptr, op.col, op.val,
                               op.band_w, in, epi, partials, live);
            cur_P = q.P;
        }
        if (timed) prof_close();
        MG_HIP(hipGetLastError());
        return MGADMM_OK;
    }

    // gather window of the row kernel: 4 in-flight neighbour rows for the k=4 spatial Laplacian, 6 otherwise
    int gather_width(const OpDesc& op) const {
        if (op.kind != OPK_SPATIAL) return 4;
        int mr = op.rowptr == g->Wu.rowptr ? g->Wu.max_row : (op.rowptr == g->Wd.rowptr ? g->Wd.max_row : g->WdT.avg_row_ceil);
        return mr <= 4 ? 4 : 6;
    }

    template <template <typename, int> class E, class... A>
    int rows(const Geom& q, const OpDesc& op, const S* in, const int* live, int tag, int passes, A... a) {
        const double bytes = pass_bytes(q, passes) + csr_bytes(op);
        const bool wide = gather_width(op) > 4;
        switch (q.VEC) {
            case 1:
                return wide ? rows_v<1, 6>(q, op, in, E<S, 1>{a...}, live, tag, bytes)
                            : rows_v<1, 4>(q, op, in, E<S, 1>{a...}, live, tag, bytes);
            case 2:
                return wide ? rows_v<2, 6>(q, op, in, E<S, 2>{a...}, live, tag, bytes)
                            : rows_v<2, 4>(q, op, in, E<S, 2>{a...}, live, tag, bytes);
            case 4:
                if constexpr (sizeof(S) == 4)
                    return wide ? rows_v<4, 6>(q, op, in, E<S, 4>{a...}, live, tag, bytes)
                                : rows_v<4, 4>(q, op, in, E<S, 4>{a...}, live, tag, bytes);
        }
        mg_set_error("rows: unsupported VEC %d", q.VEC);
        return MGADMM_ERR_UNSUPPORTED;
    }

    template <int NRED, class Fin>
    int reduce(const Geom& q, const Fin& fin, const int* live) {
        prof_open(3, 0);
        hipLaunchKernelGGL((k_reduce<S, NRED, Fin>), dim3(q.Bp / 64), dim3(1024), 0, st, partials, cur_P, q.Bp, fin, live);
        MG_HIP(hipGetLastError());
        return MGADMM_OK;
    }

    static OpDesc op_none() {
        OpDesc o{};
        o.kind = OPK_NONE;
        return o;
    }

    int pack(const Geom& q, const void* src, int Ts, S* dst) {
        dim3 grid((N + 63) / 64, T, q.Bp / 64);
        prof_open(3, 0);
        hipLaunchKernelGGL((k_pack<S>), grid, dim3(256), 0, st, T, Ts, N, q.B, q.Bp, g->d_perm, (const S*)src, dst);
        MG_HIP(hipGetLastError());
        return MGADMM_OK;
    }
    int unpack(const Geom& q, const S* src, void* dst) {
        dim3 grid((N + 63) / 64, T, q.Bp / 64);
        prof_open(3, 0);
        hipLaunchKernelGGL((k_unpack<S>), grid, dim3(256), 0, st, T, N, q.B, q.Bp, g->d_perm, src, (S*)dst);
        MG_HIP(hipGetLastError());
        return MGADMM_OK;
    }
    int fill(S* dst, size_t n, S v) {
        hipLaunchKernelGGL((k_fill<S>), dim3(1024), dim3(256), 0, st, dst, n, v);
        MG_HIP(hipGetLastError());
        return MGADMM_OK;
    }
    size_t velems(const Geom& q) const { return (size_t)T * N * q.Bp; }

    int check_B(int B, const char* who) {
        MG_REQUIRE(B >= 1 && B <= Bmax, "%s: batch %d outside [1, max_batch=%d]", who, B, Bmax);
        MG_HIP(hipSetDevice(g->device));
        return MGADMM_OK;
    }

    LhsDef lhs_def(int which) const {
        switch (which) {
            case MGADMM_LHS_X:
                if (p.ablation == MGADMM_ABL_NONE) return {1, 1, (p.rho_u + p.rho_d) / 2, p.rho / 2};
                if (p.ablation == MGADMM_ABL_DGLR) return {1, 1, p.rho_u / 2, p.rho / 2};
                return {0, 1, (p.rho_u + p.rho_d) / 2, 0.0};
            case MGADMM_LHS_ZU: return {2, 0, p.rho_u / 2, p.mu_u};
            default: return {1, 0, p.rho_d / 2, p.mu_d2};
        }
    }

    // ---------------------------------------------------------------- operators (internal layout)
    int op_store(const Geom& q, const OpDesc& op, const S* in, S* out, int tag = 2) {
        return rows<EpiStore>(q, op, in, nullptr, tag, 2, out);
    }
    int cldr(const Geom& q, const S* in, S* out) {
        if (cldr_fits(q)) return rows_cldr<EpiStore>(q, in, nullptr, 2, 4, out);
        MG_TRY(op_store(q, g->op_ldr(), in, vec[V_Q]));
        return op_store(q, g->op_ldrt(), vec[V_Q], out);
    }

    // apply_op_Ln (ADMM.py:248-288): y[t] = [t>=1](s x[t] - W_d x[t-1]) + [t<=T-2](s x[t] - M x[t+1]), s_i = sum_j d_ew[i,j],
    // M = W_d^T (kNN: scatter_add) or W_d (physical: gather); line graph: x[t] - x[t+-1]/sqrt(2)
    int op_ln(const Geom& q, const S* in, S* out) {
        if (g->mode == MGADMM_TEMPORAL_BAND)
            return rows<EpiLnLine>(q, op_none(), in, nullptr, 2, 2, in, out, T, (size_t)N * q.Bp);
        if (!g->ln_rowsum) {
            std::vector<double> rs(N);
            for (int r = 0; r < N; ++r) {
                const int i = g->has_perm ? g->perm[r] : r;
                double a = 0.0;
                for (int e = g->hWd.rowptr[i]; e < g->hWd.rowptr[i + 1]; ++e) a += (double)g->hWd.val[e];
                rs[r] = a;
            }
            MG_HIP(hipMalloc(&g->ln_rowsum, sizeof(double) * N));
            MG_HIP(hipMemcpy(g->ln_rowsum, rs.data(), sizeof(double) * N, hipMemcpyHostToDevice));
        }
        OpDesc child = g->op_ldr();
        child.self_w = g->ln_rowsum;
        OpDesc father = g->op_ldrt();
        father.self_mode = SELF_LN_FATHER;
        father.q1 = 0;
        father.self_w = g->ln_rowsum;
        MG_TRY(rows<EpiStore>(q, child, in, nullptr, 2, 2, vec[V_Q]));
        return rows<EpiAddTo>(q, father, in, nullptr, 2, 3, (const S*)vec[V_Q], out);
    }

    // Ap = A p for an LhsDef; dot partial lands in partials[0]
    int lhs_apply(const Geom& q, const LhsDef& d, const S* pin, const S* mask, S* Ap, const int* live) {
        if (d.kind == 1) {
            // one launch = two SpMM applications (Ldr, Ldr^T): 2 x 8 B/element algorithmic
            if (cldr_fits(q))
                return rows_cldr<EpiLhs>(q, pin, live, 0, 4, (const S*)nullptr, mask, Ap, d.hth, p.t_in, (S)d.c1, (S)d.c2);
            MG_TRY(rows<EpiStore>(q, g->op_ldr(), pin, live, 0, 2, vec[V_Q]));
            return rows<EpiLhs>(q, g->op_ldrt(), vec[V_Q], live, 0, 3, pin, mask, Ap, d.hth, p.t_in, (S)d.c1, (S)d.c2);
        }
        if (d.kind == 2)
            return rows<EpiLhs>(q, g->op_lu(), pin, live, 0, 2, (const S*)nullptr, mask, Ap, d.hth, p.t_in, (S)d.c1, (S)d.c2);
        return rows<EpiLhs>(q, op_none(), pin, live, 1, 2, (const S*)nullptr, mask, Ap, d.hth, p.t_in, (S)d.c1, (S)0);
    }

    // ---------------------------------------------------------------- CG (ADMM.py:329-368)
    // rhs, x0, xout: internal layout.  iters_dev: int[Bp] destination for the per-sample counts.
    int cg_internal(const Geom& q, const LhsDef& d, const S* rhs, const S* x0, const S* mask, S* xout, int* iters_dev,
                    bool record, int* n_iter_launched = nullptr) {
        CgScalars<S> c;
        c.rr = d_rr; c.alpha = d_alpha; c.beta = d_beta; c.active = d_active; c.iters = iters_dev; c.n_active = d_nact;
        c.alpha_hist = record ? d_alpha_hist : nullptr;
        c.beta_hist = record ? d_beta_hist : nullptr;
        c.nonfinite = d_nonfinite;
        const int K = p.max_cg_iter;
        const int batch_max = p.cg_convergence == MGADMM_CG_BATCH_MAX ? 1 : 0;
        MG_HIP(hipMemsetAsync(d_nact, 0, sizeof(int) * K, st));
        if (record) {
            MG_TRY(fill(d_alpha_hist, (size_t)K * q.Bp, (S)NAN));
            MG_TRY(fill(d_beta_hist, (size_t)K * q.Bp, (S)NAN));
        }
        S *r = vec[V_R], *pp = vec[V_P], *Ap = vec[V_AP];
        // r = rhs - A x0 ; p = r ; x = x0
        if (d.kind == 1 && cldr_fits(q)) {
            MG_TRY(rows_cldr<EpiCgInit>(q, x0, nullptr, 2, 7, (const S*)nullptr, rhs, mask, r, pp, xout, d.hth, p.t_in, (S)d.c1, (S)d.c2));
        } else if (d.kind == 1) {
            MG_TRY(rows<EpiStore>(q, g->op_ldr(), x0, nullptr, 2, 2, vec[V_Q]));
            MG_TRY(rows<EpiCgInit>(q, g->op_ldrt(), vec[V_Q], nullptr, 2, 6, x0, rhs, mask, r, pp, xout, d.hth, p.t_in,
                                   (S)d.c1, (S)d.c2));
        } else if (d.kind == 2) {
            MG_TRY(rows<EpiCgInit>(q, g->op_lu(), x0, nullptr, 2, 5, (const S*)nullptr, rhs, mask, r, pp, xout, d.hth,
                                   p.t_in, (S)d.c1, (S)d.c2));
        } else {
            MG_TRY(rows<EpiCgInit>(q, op_none(), x0, nullptr, 1, 5, (const S*)nullptr, rhs, mask, r, pp, xout, d.hth,
                                   p.t_in, (S)d.c1, (S)0));
        }
        MG_TRY((reduce<1>(q, FinCgInit<S>{c, q.B}, nullptr)));
        // slice of the pinned h_nact log used by this solve (a profiling session keeps every slice until prof_end)
        if (nact_cur + K > NACT_LOG) {
            if (prof_on) nact_locked = true;
            else nact_cur = 0;
        }
        const int base = nact_locked ? NACT_LOG : nact_cur;
        // kind 1 on the fused kernel: the vector update of iteration k (p = r + beta p, x += alpha p) is folded into the
        // SpMM launch of iteration k+1 (k_cldr with CldrSrcFold); p alternates between two buffers because other tiles
        // still gather the old direction for their halos.  The x update of the last iteration is applied after the loop.
        const bool fold1 = d.kind == 1 && use_fold && cldr_fits(q);
        const bool fold2 = d.kind == 2 && lu_fold_fits(q);      // zu solve: the same fold in the LDS-tiled Lu kernel
        const bool fold = fold1 || fold2;
        S* pbuf[2] = {pp, vec[V_Q]};          // V_Q is free: the fused kernel keeps q = Ldr p on chip
        int k = 0;
        for (; k < K; ++k) {
            const int* live = k == 0 ? nullptr : d_nact + (k - 1);
            prof_cur_ref = (k == 0 || nact_locked) ? -1 : base + k - 1;
            if (fold2) {
                // 1 SpMM application (8 B/element) + the absorbed vector update (20 B/element) per launch
                MG_TRY(rows_lu_fold<EpiLhs>(q, r, pbuf[k & 1], pbuf[(k + 1) & 1], xout, live, 0, 7, (const S*)nullptr, (const S*)nullptr, Ap,
                                            d.hth, p.t_in, (S)d.c1, (S)d.c2));
            } else if (fold1) {
                // 2 SpMM applications (16 B/element) + the absorbed vector update (20 B/element) per launch
                MG_TRY(rows_cldr_fold<EpiLhs>(q, r, pbuf[k & 1], pbuf[(k + 1) & 1], xout, live, 0, 9, (const S*)nullptr, (const S*)nullptr, Ap,
                                              d.hth, p.t_in, (S)d.c1, (S)d.c2));
            } else {
                MG_TRY(lhs_apply(q, d, pp, nullptr, Ap, live));                  // Ap = A p (no mask: quirk Q2)
            }
            MG_TRY((reduce<1>(q, FinCgAlpha<S>{c, k, q.Bp}, live)));
            {
                // r -= alpha Ap, r.r -- swept from the LAST time slice down: the SpMM kernel before it finished with the
                // last slices of Ap (still in the Infinity Cache), and the kernel after it starts with the first slices of r
                Geom qr = q;
                qr.rev = sweep_rev;
                MG_TRY(rows<EpiCgUpdate>(qr, op_none(), Ap, live, 1, 3, (const S*)d_alpha, r));
            }
            MG_TRY((reduce<1>(q, FinCgBeta<S>{c, k, q.Bp, p.cg_tol, batch_max}, live)));
            if (!fold)
                MG_TRY(rows<EpiPUpdate>(q, op_none(), r, live, 1, 5, (const S*)d_alpha, (const S*)d_beta, xout, pp));   // x += alpha p, p = r + beta p
            prof_cur_ref = -1;
            MG_HIP(hipMemcpyAsync(h_nact + base + k, d_nact + k, sizeof(int), hipMemcpyDeviceToHost, st));
            MG_HIP(hipEventRecord(ev_ring[k % (LAG + 1)], st));
            if (k >= LAG) {
                MG_HIP(hipEventSynchronize(ev_ring[(k - LAG) % (LAG + 1)]));
                if (h_nact[base + k - LAG] == 0) { ++k; break; }
            }
        }
        if (fold) {
            // the last iteration that did work: the first whose active count is 0 (later launches returned at the guard),
            // else the last one launched; its direction sits in pbuf[(kl + 1) & 1]
            MG_HIP(hipStreamSynchronize(st));
            const int launched = std::min(k, K);
            int kl = launched - 1;
            for (int j = 0; j < launched; ++j)
                if (h_nact[base + j] == 0) { kl = j; break; }
            MG_TRY(rows<EpiXFinal>(q, op_none(), pbuf[(kl + 1) & 1], nullptr, 1, 3, (const S*)d_alpha, xout));
        }
        if (!nact_locked) nact_cur += std::min(k, K);
        if (batch_max) {          // one iteration count for the whole batch: the first iteration after which no sample was above the tolerance
            hipLaunchKernelGGL(k_cg_batchmax_iters, dim3((q.B + 255) / 256), dim3(256), 0, st, (const int*)d_nact, std::min(k, K), iters_dev, q.B);
            MG_HIP(hipGetLastError());
        }
        if (n_iter_launched) *n_iter_launched = k;
        return MGADMM_OK;
    }

    // ---------------------------------------------------------------- fine-grained ABI entry points
    int apply(int op, const void* x, void* y, int B, hipStream_t s) override {
        MG_TRY(check_B(B, "apply"));
        MG_REQUIRE(x && y, "apply: null pointer");
        st = s;
        Geom q = make_geom(B);
        MG_TRY(ensure_partials(q));
        MG_TRY(pack(q, x, T, vec[V_IO0]));
        switch (op) {
            case MGADMM_OP_LU: MG_TRY(op_store(q, g->op_lu(), vec[V_IO0], vec[V_IO1])); break;
            case MGADMM_OP_LDR: MG_TRY(op_store(q, g->op_ldr(), vec[V_IO0], vec[V_IO1])); break;
            case MGADMM_OP_LDRT: MG_TRY(op_store(q, g->op_ldrt(), vec[V_IO0], vec[V_IO1])); break;
            case MGADMM_OP_CLDR: MG_TRY(cldr(q, vec[V_IO0], vec[V_IO1])); break;
            case MGADMM_OP_LN: MG_TRY(op_ln(q, vec[V_IO0], vec[V_IO1])); break;
            default: mg_set_error("apply: bad op %d", op); return MGADMM_ERR_INVALID;
        }
        return unpack(q, vec[V_IO1], y);
    }

    int lhs(int which, const void* x, const void* mask, void* y, int B, hipStream_t s) override {
        MG_TRY(check_B(B, "lhs"));
        MG_REQUIRE(x && y, "lhs: null pointer");
        MG_REQUIRE(which >= 0 && which <= 2, "lhs: bad operator id %d", which);
        MG_REQUIRE(!(which == MGADMM_LHS_ZD && p.ablation == MGADMM_ABL_DGLR), "lhs: LHS_zd is undefined for ablation 'DGLR' (ADMM.py:392-399)");
        st = s;
        Geom q = make_geom(B);
        MG_TRY(ensure_partials(q));
        MG_TRY(pack(q, x, T, vec[V_IO0]));
        const S* m = nullptr;
        if (mask && which == MGADMM_LHS_X) {
            MG_TRY(pack(q, mask, T, vec[V_MASK]));
            m = vec[V_MASK];
        }
        MG_TRY(lhs_apply(q, lhs_def(which), vec[V_IO0], m, vec[V_IO1], nullptr));
        return unpack(q, vec[V_IO1], y);
    }

    int phi_direct(const void* x, const void* gamma, void* phi, int B, hipStream_t s) override {
        MG_TRY(check_B(B, "phi_direct"));
        MG_REQUIRE(x && gamma && phi, "phi_direct: null pointer");
        st = s;
        Geom q = make_geom(B);
        MG_TRY(ensure_partials(q));
        MG_TRY(pack(q, x, T, vec[V_IO0]));
        MG_TRY(pack(q, gamma, T, vec[V_TMP]));
        MG_TRY(rows<EpiPhiDirect>(q, g->op_ldr(), vec[V_IO0], nullptr, 2, 4, (const S*)vec[V_TMP], vec[V_IO1], (S)p.rho,
                                  (S)(p.mu_d1 / p.rho)));
        return unpack(q, vec[V_IO1], phi);
    }

    int guess_internal(const Geom& q, const S* ypad, S* x) {
        // float32 time moments of ADMM.py:772-775
        float tm = 0, t2m = 0;
        for (int t = 0; t < p.t_in; ++t) { tm += (float)t; t2m += (float)t * (float)t; }
        tm /= (float)p.t_in;
        t2m /= (float)p.t_in;
        const float den = t2m - tm * tm;
        dim3 grid((N + 3) / 4, q.Bp / 64);
        hipLaunchKernelGGL((k_initial_guess<S>), grid, dim3(256), 0, st, T, p.t_in, N, q.Bp, (S)tm, (S)den, ypad, x);
        MG_HIP(hipGetLastError());
        return MGADMM_OK;
    }
    int interp_internal(const Geom& q, const S* y, const S* mask, int mask_f32, S* x) {
        dim3 grid((N + 3) / 4, q.Bp / 64);
        if (mask_f32 && sizeof(S) == 8)
            hipLaunchKernelGGL((k_initial_interp<S, true>), grid, dim3(256), 0, st, T, N, q.Bp, q.B, y, mask, x, d_nonfinite);
        else
            hipLaunchKernelGGL((k_initial_interp<S, false>), grid, dim3(256), 0, st, T, N, q.Bp, q.B, y, mask, x, d_nonfinite);
        MG_HIP(hipGetLastError());
        return MGADMM_OK;
    }

    int initial_guess(const void* y, void* x, int B, hipStream_t s) override {
        MG_TRY(check_B(B, "initial_guess"));
        MG_REQUIRE(x && y, "initial_guess: null pointer");
        st = s;
        Geom q = make_geom(B);
        MG_TRY(pack(q, y, p.t_in, vec[V_Y]));
        MG_TRY(guess_internal(q, vec[V_Y], vec[V_IO1]));
        return unpack(q, vec[V_IO1], x);
    }

    int initial_interpolation(const void* y, const void* mask, int mask_f32, void* x, int B, hipStream_t s) override {
        MG_TRY(check_B(B, "initial_interpolation"));
        MG_REQUIRE(x && y && mask, "initial_interpolation: null pointer");
        st = s;
        Geom q = make_geom(B);
        MG_TRY(pack(q, y, T, vec[V_Y]));
        MG_TRY(pack(q, mask, T, vec[V_MASK]));
        MG_TRY(interp_internal(q, vec[V_Y], vec[V_MASK], mask_f32, vec[V_IO1]));
        return unpack(q, vec[V_IO1], x);
    }

    int fetch_hist(const S* dev, size_t n, double* out, size_t B, size_t Bp, size_t K) {
        // dev: [K][Bp] -> out: [K][B] doubles
        std::vector<S> tmp(n);
        MG_HIP(hipMemcpyAsync(tmp.data(), dev, n * sizeof(S), hipMemcpyDeviceToHost, st));
        MG_HIP(hipStreamSynchronize(st));
        for (size_t k = 0; k < K; ++k)
            for (size_t b = 0; b < B; ++b) out[k * B + b] = (double)tmp[k * Bp + b];
        return MGADMM_OK;
    }

    int cg(int which, const void* rhs, const void* x0, const void* mask, void* x, int32_t* iters, double* alpha,
           double* beta, int B, hipStream_t s) override {
        MG_TRY(check_B(B, "cg"));
        MG_REQUIRE(rhs && x && iters, "cg: null pointer");
        MG_REQUIRE(which >= 0 && which <= 2, "cg: bad operator id %d", which);
        MG_REQUIRE(!(which == MGADMM_LHS_ZD && p.ablation == MGADMM_ABL_DGLR), "cg: LHS_zd is undefined for ablation 'DGLR'");
        st = s;
        Geom q = make_geom(B);
        MG_TRY(ensure_partials(q));
        MG_HIP(hipMemsetAsync(d_nonfinite, 0, sizeof(int), st));
        MG_TRY(pack(q, rhs, T, vec[V_RHS]));
        if (x0) MG_TRY(pack(q, x0, T, vec[V_IO0]));
        else MG_HIP(hipMemsetAsync(vec[V_IO0], 0, velems(q) * sizeof(S), st));
        const S* m = nullptr;
        if (mask && which == MGADMM_LHS_X) {
            MG_TRY(pack(q, mask, T, vec[V_MASK]));
            m = vec[V_MASK];
        }
        MG_TRY(cg_internal(q, lhs_def(which), vec[V_RHS], vec[V_IO0], m, vec[V_IO1], d_iters_tmp, true));
        MG_TRY(unpack(q, vec[V_IO1], x));
        MG_HIP(hipMemcpyAsync(iters, d_iters_tmp, sizeof(int) * B, hipMemcpyDeviceToHost, st));
        MG_HIP(hipMemcpyAsync(h_flag, d_nonfinite, sizeof(int), hipMemcpyDeviceToHost, st));
        MG_HIP(hipStreamSynchronize(st));
        const size_t K = p.max_cg_iter;
        if (alpha) MG_TRY(fetch_hist(d_alpha_hist, K * q.Bp, alpha, B, q.Bp, K));
        if (beta) MG_TRY(fetch_hist(d_beta_hist, K * q.Bp, beta, B, q.Bp, K));
        if (h_flag[0]) {
            mg_set_error("cg: non-finite residual met");
            return MGADMM_ERR_NONFINITE;
        }
        return MGADMM_OK;
    }

    // ---------------------------------------------------------------- combined_loop (ADMM.py:511-648)
    // every state tensor the ablation uses must be present in a warm-start state
    int check_state_in(const void* x0, const mgadmm_state* si) const {
        const int abl = p.ablation;
        const bool has_phi = (abl == MGADMM_ABL_NONE || abl == MGADMM_ABL_DGLR), has_zd = (abl != MGADMM_ABL_DGLR);
        MG_REQUIRE(x0 && si && si->zu && si->gamma_u, "solve_from: x0, state_in.zu and state_in.gamma_u are required");
        MG_REQUIRE(!has_phi || (si->phi && si->gamma), "solve_from: state_in.phi and state_in.gamma are required for this ablation");
        MG_REQUIRE(!has_zd || (si->zd && si->gamma_d), "solve_from: state_in.zd and state_in.gamma_d are required for this ablation");
        return MGADMM_OK;
    }

    int solve(const void* y, const void* mask, int mask_f32, int B, const void* x0, const mgadmm_state* state_in, void* x_out,
              const mgadmm_state* state_out, mgadmm_history* hist, hipStream_t s) override {
        MG_TRY(check_B(B, "solve"));
        MG_REQUIRE(y && x_out, "solve: null pointer");
        if (state_in) MG_TRY(check_state_in(x0, state_in));
        st = s;
        if (p.path == MGADMM_PATH_LDS && !lds.ok) {
            mg_set_error("solve: the LDS-resident path needs float32, T*N*8 B + CSR <= 160 KiB and N*G <= 1024 (N=%d, T=%d)", N, T);
            return MGADMM_ERR_UNSUPPORTED;
        }
        if (use_lds()) return solve_lds(y, mask, B, x0, state_in, x_out, state_out, hist);
        const Geom q = make_geom(B);
        MG_TRY(ensure_partials(q));
        const size_t ne = velems(q);
        const int abl = p.ablation;
        const bool has_phi = (abl == MGADMM_ABL_NONE || abl == MGADMM_ABL_DGLR);
        const bool has_zd = (abl != MGADMM_ABL_DGLR);
        const int max_it = p.max_admm_iter;
        const bool record = p.record_cg_coeffs && hist && hist->cg_alpha && hist->cg_beta;

        if (hist && hist->metrics_per_sample) {
            size_t need = (size_t)max_it * MGADMM_NMETRIC * B;
            if (need > hist_ps_elems) {
                if (d_hist_ps) MG_HIP(hipFree(d_hist_ps));
                d_hist_ps = nullptr;
                MG_HIP(hipMalloc(&d_hist_ps, need * sizeof(double)));
                hist_ps_elems = need;
            }
        }
        MG_HIP(hipMemsetAsync(d_nonfinite, 0, sizeof(int), st));
        MG_HIP(hipMemsetAsync(d_ps, 0, sizeof(double) * MGADMM_NMETRIC * q.Bp, st));
        MG_HIP(hipMemsetAsync(d_cg_iters, 0, sizeof(int) * (size_t)max_it * 3 * q.Bp, st));

        int xc = V_XA, xn = V_XB, zuc = V_ZUA, zun = V_ZUB, zdc = V_ZDA, zdn = V_ZDB, phc = V_PHIA, phn = V_PHIB;
        const S* m = nullptr;
        if (mask) {
            MG_TRY(pack(q, y, T, vec[V_Y]));
            MG_TRY(pack(q, mask, T, vec[V_MASK]));
            m = vec[V_MASK];
            if (!state_in) MG_TRY(interp_internal(q, vec[V_Y], m, mask_f32, vec[xc]));
        } else {
            MG_TRY(pack(q, y, p.t_in, vec[V_Y]));
            if (!state_in) MG_TRY(guess_internal(q, vec[V_Y], vec[xc]));
        }
        if (state_in) {                       // warm start: the saved state replaces ADMM.py:529-544
            MG_TRY(pack(q, x0, T, vec[xc]));
            MG_TRY(pack(q, state_in->zu, T, vec[zuc]));
            MG_TRY(pack(q, state_in->gamma_u, T, vec[V_GU]));
            if (has_zd) {
                MG_TRY(pack(q, state_in->zd, T, vec[zdc]));
                MG_TRY(pack(q, state_in->gamma_d, T, vec[V_GD]));
            }
            if (has_phi) {
                MG_TRY(pack(q, state_in->phi, T, vec[phc]));
                MG_TRY(pack(q, state_in->gamma, T, vec[V_GAM]));
            }
        } else {
            MG_TRY(fill(vec[V_GU], ne, (S)0.1));
            MG_TRY(fill(vec[V_GD], ne, (S)0.1));
            if (has_phi) {
                MG_TRY(fill(vec[V_GAM], ne, (S)0.1));
                MG_TRY(op_store(q, g->op_ldr(), vec[xc], vec[phc]));
            }
            MG_HIP(hipMemcpyAsync(vec[zuc], vec[xc], ne * sizeof(S), hipMemcpyDeviceToDevice, st));
            MG_HIP(hipMemcpyAsync(vec[zdc], vec[xc], ne * sizeof(S), hipMemcpyDeviceToDevice, st));
        }

        const S rho = (S)p.rho, rho_u = (S)p.rho_u, rho_d = (S)p.rho_d;
        int n_done = 0;
        int rc_final = MGADMM_OK;
        for (int it = 0; it < max_it; ++it) {
            // ---- RHS_x (ADMM.py:556-564)
            if (has_phi) {
                MG_TRY(rows<EpiLin2>(q, op_none(), vec[V_GAM], nullptr, 3, 3, (const S*)vec[phc], vec[V_TMP], (S)1, rho));
                MG_TRY(rows<EpiRhsX>(q, g->op_ldrt(), vec[V_TMP], nullptr, 2, has_zd ? 7 : 5, (const S*)vec[zuc],
                                     (const S*)vec[zdc], (const S*)vec[V_GU], (const S*)vec[V_GD], (const S*)vec[V_Y],
                                     vec[V_RHS], rho_u, rho_d, 1, has_zd ? 1 : 0));
            } else {
                MG_TRY(rows<EpiRhsX>(q, op_none(), vec[zuc], nullptr, 3, 6, (const S*)vec[zuc], (const S*)vec[zdc],
                                     (const S*)vec[V_GU], (const S*)vec[V_GD], (const S*)vec[V_Y], vec[V_RHS], rho_u,
                                     rho_d, 0, 1));
            }
            // ---- x, zu, zd CG solves (ADMM.py:571-592)
            int* it_base = d_cg_iters + (size_t)it * 3 * q.Bp;
            MG_TRY(cg_internal(q, lhs_def(MGADMM_LHS_X), vec[V_RHS], vec[xc], m, vec[xn], it_base, record));
            if (record) MG_TRY(save_coeffs(q, hist, it, 0, B));
            MG_TRY(rows<EpiLin2>(q, op_none(), vec[V_GU], nullptr, 3, 3, (const S*)vec[xn], vec[V_RHS], (S)0.5, (S)(p.rho_u / 2)));
            MG_TRY(cg_internal(q, lhs_def(MGADMM_LHS_ZU), vec[V_RHS], vec[zuc], nullptr, vec[zun], it_base + q.Bp, record));
            if (record) MG_TRY(save_coeffs(q, hist, it, 1, B));
            if (has_zd) {
                MG_TRY(rows<EpiLin2>(q, op_none(), vec[V_GD], nullptr, 3, 3, (const S*)vec[xn], vec[V_RHS], (S)0.5, (S)(p.rho_d / 2)));
                MG_TRY(cg_internal(q, lhs_def(MGADMM_LHS_ZD), vec[V_RHS], vec[zdc], nullptr, vec[zdn], it_base + 2 * q.Bp, record));
                if (record) MG_TRY(save_coeffs(q, hist, it, 2, B));
            }
            const S* zd_now = has_zd ? vec[zdn] : vec[zdc];
            // ---- dual updates + Laplacian-free residuals (ADMM.py:595-597, 612-636)
            MG_TRY(rows<EpiDual>(q, op_none(), vec[xn], nullptr, 3, has_zd ? 11 : 6, (const S*)vec[xc], (const S*)vec[zun],
                                 (const S*)vec[zuc], zd_now, (const S*)vec[zdc], (const S*)vec[V_Y], m, vec[V_GU],
                                 vec[V_GD], rho_u, rho_d, has_zd ? 1 : 0, p.t_in));
            {
                FinMetrics<6> f{d_ps, q.Bp, {MGADMM_M_XSHIFT, MGADMM_M_PRI_ZU, MGADMM_M_DUAL_ZU, has_zd ? MGADMM_M_PRI_ZD : -1,
                                            has_zd ? MGADMM_M_DUAL_ZD : -1, MGADMM_M_RECOVER}};
                MG_TRY((reduce<6>(q, f, nullptr)));
            }
            // ---- phi prox, gamma update, Ldr-based residuals and regularisers (ADMM.py:600-606, 627-637)
            MG_TRY(rows<EpiPhi>(q, g->op_ldr(), vec[xn], nullptr, 2, has_phi ? 5 : 1, (const S*)vec[phc], vec[phn], vec[V_GAM],
                                rho, (S)(p.mu_d1 / p.rho), has_phi ? 1 : 0));
            {
                FinMetrics<4> f{d_ps, q.Bp, {has_phi ? MGADMM_M_PRI_PHI : -1, has_phi ? MGADMM_M_DUAL_PHI : -1,
                                            has_phi ? MGADMM_M_DGTV : -1, has_zd ? MGADMM_M_DGLR : -1}};
                MG_TRY((reduce<4>(q, f, nullptr)));
            }
            MG_TRY(rows<EpiDot>(q, g->op_lu(), vec[xn], nullptr, 2, 1));
            {
                FinMetrics<1> f{d_ps, q.Bp, {MGADMM_M_GLR}};
                MG_TRY((reduce<1>(q, f, nullptr)));
            }
            const int nbk = (N + 63) / 64;
            hipLaunchKernelGGL((k_dxps<S>), dim3(T * nbk), dim3(256), 0, st, T, N, q.Bp, B, nbk, (const S*)vec[xn],
                               (const S*)vec[xc], d_dxpart);
            hipLaunchKernelGGL(k_dxps_final, dim3((T + 63) / 64), dim3(64), 0, st, T, nbk, (const double*)d_dxpart,
                               d_dxps + (size_t)it * T);
            hipLaunchKernelGGL(k_batch_metrics, dim3(MGADMM_NMETRIC), dim3(256), 0, st, (const double*)d_ps, q.Bp, B,
                               d_hist + (size_t)it * MGADMM_NMETRIC,
                               (hist && hist->metrics_per_sample) ? d_hist_ps + (size_t)it * MGADMM_NMETRIC * B : nullptr);
            MG_HIP(hipGetLastError());
            std::swap(xc, xn);
            std::swap(zuc, zun);
            if (has_zd) std::swap(zdc, zdn);
            if (has_phi) std::swap(phc, phn);
            n_done = it + 1;
            if (p.check_stop) {
                MG_HIP(hipMemcpyAsync(h_row, d_hist + (size_t)it * MGADMM_NMETRIC, sizeof(double) * MGADMM_NMETRIC,
                                      hipMemcpyDeviceToHost, st));
                MG_HIP(hipMemcpyAsync(h_flag, d_nonfinite, sizeof(int), hipMemcpyDeviceToHost, st));
                MG_HIP(hipStreamSynchronize(st));
                bool finite = h_flag[0] == 0;
                for (int k = 0; k < MGADMM_NMETRIC; ++k) finite = finite && std::isfinite(h_row[k]);
                if (!finite) { rc_final = MGADMM_ERR_NONFINITE; break; }
                double pri = h_row[MGADMM_M_PRI_ZU], dual = h_row[MGADMM_M_DUAL_ZU];
                if (has_phi) { pri = std::max(pri, h_row[MGADMM_M_PRI_PHI]); dual = std::max(dual, h_row[MGADMM_M_DUAL_PHI]); }
                if (has_zd) { pri = std::max(pri, h_row[MGADMM_M_PRI_ZD]); dual = std::max(dual, h_row[MGADMM_M_DUAL_ZD]); }
                if (pri < p.admm_tol && dual < p.admm_tol) break;      // ADMM.py:645-646
            }
        }
        // ---- results
        MG_TRY(unpack(q, vec[xc], x_out));
        if (state_out) {
            if (state_out->zu) MG_TRY(unpack(q, vec[zuc], state_out->zu));
            if (state_out->zd) MG_TRY(unpack(q, vec[zdc], state_out->zd));
            if (state_out->phi && has_phi) MG_TRY(unpack(q, vec[phc], state_out->phi));
            if (state_out->gamma && has_phi) MG_TRY(unpack(q, vec[V_GAM], state_out->gamma));
            if (state_out->gamma_u) MG_TRY(unpack(q, vec[V_GU], state_out->gamma_u));
            if (state_out->gamma_d) MG_TRY(unpack(q, vec[V_GD], state_out->gamma_d));
        }
        return finish_history(hist, n_done, B, q.Bp, rc_final);
    }

    // ---------------------------------------------------------------- two_loops (ADMM.py:410-508)
    // Same kernels as `solve`, other schedule: the phi / gamma update moves to an outer loop and every outer iteration
    // restarts the inner ADMM on (x, zu, zd, gamma_u, gamma_d) from zu = zd = x, gamma_u = gamma_d = 0.1.  No residual
    // history (the reference records none there: "TODO: residual", ADMM.py:494, 506); CG counts per inner iteration.
    int two_loops(const void* y, const void* mask, int mask_f32, int B, void* x_out, const mgadmm_state* state_out,
                  mgadmm_history* hist, hipStream_t s) override {
        MG_TRY(check_B(B, "two_loops"));
        MG_REQUIRE(y && x_out, "two_loops: null pointer");
        MG_REQUIRE(p.max_inner_iter >= 1, "two_loops: max_inner_iter must be >= 1 (got %d)", p.max_inner_iter);
        st = s;
        const Geom q = make_geom(B);
        MG_TRY(ensure_partials(q));
        const size_t ne = velems(q);
        const int abl = p.ablation;
        const bool has_phi = (abl == MGADMM_ABL_NONE || abl == MGADMM_ABL_DGLR);
        const bool has_zd = (abl != MGADMM_ABL_DGLR);
        const int n_outer = p.max_admm_iter, n_inner = p.max_inner_iter;
        const size_t rows_needed = (size_t)n_outer * n_inner;
        if (rows_needed > (size_t)max_admm_alloc) {          // d_cg_iters holds one row of 3 x Bp counts per inner iteration
            // the three history buffers grow together (max_admm_alloc is their common row count); the new ones are all
            // allocated before an old one is released, so a failed allocation leaves the solver as it was
            int* nc = nullptr;
            double *nh = nullptr, *nd = nullptr;
            const hipError_t e1 = hipMalloc(&nc, sizeof(int) * rows_needed * 3 * Bp_max);
            const hipError_t e2 = e1 == hipSuccess ? hipMalloc(&nh, sizeof(double) * rows_needed * MGADMM_NMETRIC) : e1;
            const hipError_t e3 = e2 == hipSuccess ? hipMalloc(&nd, sizeof(double) * rows_needed * T) : e2;
            if (e3 != hipSuccess) {
                if (nc) (void)hipFree(nc);
                if (nh) (void)hipFree(nh);
                if (nd) (void)hipFree(nd);
                (void)hipGetLastError();
                mg_set_error("two_loops: cannot allocate the CG-count history of %zu inner iterations x %d samples (%s)", rows_needed,
                             Bp_max, hipGetErrorString(e3));
                return MGADMM_ERR_NOMEM;
            }
            if (d_cg_iters) (void)hipFree(d_cg_iters);
            if (d_hist) (void)hipFree(d_hist);
            if (d_dxps) (void)hipFree(d_dxps);
            d_cg_iters = nc; d_hist = nh; d_dxps = nd;
            max_admm_alloc = (int)rows_needed;
        }
        MG_HIP(hipMemsetAsync(d_nonfinite, 0, sizeof(int), st));
        MG_HIP(hipMemsetAsync(d_ps, 0, sizeof(double) * MGADMM_NMETRIC * q.Bp, st));
        MG_HIP(hipMemsetAsync(d_cg_iters, 0, sizeof(int) * rows_needed * 3 * q.Bp, st));
        int xc = V_XA, xn = V_XB, zuc = V_ZUA, zun = V_ZUB, zdc = V_ZDA, zdn = V_ZDB, phc = V_PHIA, phn = V_PHIB;
        const S* m = nullptr;
        if (mask) {
            MG_TRY(pack(q, y, T, vec[V_Y]));
            MG_TRY(pack(q, mask, T, vec[V_MASK]));
            m = vec[V_MASK];
            MG_TRY(interp_internal(q, vec[V_Y], m, mask_f32, vec[xc]));
        } else {
            MG_TRY(pack(q, y, p.t_in, vec[V_Y]));
            MG_TRY(guess_internal(q, vec[V_Y], vec[xc]));
        }
        if (has_phi) {
            MG_TRY(fill(vec[V_GAM], ne, (S)0.1));
            MG_TRY(op_store(q, g->op_ldr(), vec[xc], vec[phc]));
        }
        const S rho = (S)p.rho, rho_u = (S)p.rho_u, rho_d = (S)p.rho_d;
        int n_done = 0;
        for (int o = 0; o < n_outer; ++o) {
            MG_TRY(fill(vec[V_GU], ne, (S)0.1));
            MG_TRY(fill(vec[V_GD], ne, (S)0.1));
            MG_HIP(hipMemcpyAsync(vec[zuc], vec[xc], ne * sizeof(S), hipMemcpyDeviceToDevice, st));
            MG_HIP(hipMemcpyAsync(vec[zdc], vec[xc], ne * sizeof(S), hipMemcpyDeviceToDevice, st));
            for (int in = 0; in < n_inner; ++in) {
                if (has_phi) {
                    MG_TRY(rows<EpiLin2>(q, op_none(), vec[V_GAM], nullptr, 3, 3, (const S*)vec[phc], vec[V_TMP], (S)1, rho));
                    MG_TRY(rows<EpiRhsX>(q, g->op_ldrt(), vec[V_TMP], nullptr, 2, has_zd ? 7 : 5, (const S*)vec[zuc],
                                         (const S*)vec[zdc], (const S*)vec[V_GU], (const S*)vec[V_GD], (const S*)vec[V_Y],
                                         vec[V_RHS], rho_u, rho_d, 1, has_zd ? 1 : 0));
                } else {
                    MG_TRY(rows<EpiRhsX>(q, op_none(), vec[zuc], nullptr, 3, 6, (const S*)vec[zuc], (const S*)vec[zdc],
                                         (const S*)vec[V_GU], (const S*)vec[V_GD], (const S*)vec[V_Y], vec[V_RHS], rho_u,
                                         rho_d, 0, 1));
                }
                int* it_base = d_cg_iters + ((size_t)o * n_inner + in) * 3 * q.Bp;
                MG_TRY(cg_internal(q, lhs_def(MGADMM_LHS_X), vec[V_RHS], vec[xc], m, vec[xn], it_base, false));
                MG_TRY(rows<EpiLin2>(q, op_none(), vec[V_GU], nullptr, 3, 3, (const S*)vec[xn], vec[V_RHS], (S)0.5, (S)(p.rho_u / 2)));
                MG_TRY(cg_internal(q, lhs_def(MGADMM_LHS_ZU), vec[V_RHS], vec[zuc], nullptr, vec[zun], it_base + q.Bp, false));
                if (has_zd) {
                    MG_TRY(rows<EpiLin2>(q, op_none(), vec[V_GD], nullptr, 3, 3, (const S*)vec[xn], vec[V_RHS], (S)0.5, (S)(p.rho_d / 2)));
                    MG_TRY(cg_internal(q, lhs_def(MGADMM_LHS_ZD), vec[V_RHS], vec[zdc], nullptr, vec[zdn], it_base + 2 * q.Bp, false));
                }
                const S* zd_now = has_zd ? vec[zdn] : vec[zdc];
                // gamma_u += rho_u (x - zu), gamma_d += rho_d (x - zd)   (ADMM.py:488-490); the fused norms are not kept
                MG_TRY(rows<EpiDual>(q, op_none(), vec[xn], nullptr, 3, has_zd ? 11 : 6, (const S*)vec[xc], (const S*)vec[zun],
                                     (const S*)vec[zuc], zd_now, (const S*)vec[zdc], (const S*)vec[V_Y], m, vec[V_GU],
                                     vec[V_GD], rho_u, rho_d, has_zd ? 1 : 0, p.t_in));
                std::swap(xc, xn);
                std::swap(zuc, zun);
                if (has_zd) std::swap(zdc, zdn);
            }
            if (has_phi) {          // phi = phi_direct(x, gamma); gamma += rho (phi - Ldr x)   (ADMM.py:497-504)
                MG_TRY(rows<EpiPhi>(q, g->op_ldr(), vec[xc], nullptr, 2, 5, (const S*)vec[phc], vec[phn], vec[V_GAM], rho,
                                    (S)(p.mu_d1 / p.rho), 1));
                std::swap(phc, phn);
            }
            n_done = o + 1;
        }
        MG_TRY(unpack(q, vec[xc], x_out));
        if (state_out) {
            if (state_out->zu) MG_TRY(unpack(q, vec[zuc], state_out->zu));
            if (state_out->zd) MG_TRY(unpack(q, vec[zdc], state_out->zd));
            if (state_out->phi && has_phi) MG_TRY(unpack(q, vec[phc], state_out->phi));
            if (state_out->gamma && has_phi) MG_TRY(unpack(q, vec[V_GAM], state_out->gamma));
            if (state_out->gamma_u) MG_TRY(unpack(q, vec[V_GU], state_out->gamma_u));
            if (state_out->gamma_d) MG_TRY(unpack(q, vec[V_GD], state_out->gamma_d));
        }
        if (hist) {
            hist->n_iters = n_done;
            if (hist->cg_iters) {
                std::vector<int> tmp(rows_needed * 3 * q.Bp);
                MG_HIP(hipMemcpyAsync(tmp.data(), d_cg_iters, sizeof(int) * tmp.size(), hipMemcpyDeviceToHost, st));
                MG_HIP(hipStreamSynchronize(st));
                for (size_t r = 0; r < rows_needed * 3; ++r) memcpy(hist->cg_iters + r * B, tmp.data() + r * q.Bp, sizeof(int) * B);
            }
        }
        MG_HIP(hipMemcpyAsync(h_flag, d_nonfinite, sizeof(int), hipMemcpyDeviceToHost, st));
        MG_HIP(hipStreamSynchronize(st));
        if (h_flag[0]) {
            mg_set_error("two_loops: NaN/Inf met in the iterates (cf. the asserts of ADMM.py:432-504)");
            return MGADMM_ERR_NONFINITE;
        }
        return MGADMM_OK;
    }

    // ---------------------------------------------------------------- LDS-resident fused path
    int plan_lds() {
        lds = LdsPlan();
        if (hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, g->device) != hipSuccess) { (void)hipGetLastError(); dev_cus = 0; }
        if (!std::is_same<S, float>::value) return MGADMM_OK;
        const int tpgs[] = {1, 2, 3, 4, 6, 8, 12};
        int best = 0;
        const char* force = getenv("MGADMM_LDS_TPG");      // tests / experiments: force one time-group width
        // does the graph qualify for the uniform-row instances (TPG 8 and 12: table rows in registers, branch-free solves)?
        bool uni_graph = g->mode != MGADMM_TEMPORAL_BAND && !getenv("MGADMM_LDS_SB") && !getenv("MGADMM_LDS_RAGGED");
        for (int i = 0; i < N && uni_graph; ++i) {
            int ndiag = 0;
            for (int e = g->hWd.rowptr[i]; e < g->hWd.rowptr[i + 1]; ++e) ndiag += g->hWd.col[e] == i;
            uni_graph = g->hWu.rowptr[i + 1] - g->hWu.rowptr[i] == 4 && g->hWd.rowptr[i + 1] - g->hWd.rowptr[i] == 5 && ndiag == 1;
        }
        for (int tpg : tpgs) {
            if (T % tpg) continue;
            const int G = T / tpg;
            if ((long)N * G > 1024) continue;
            if (force && atoi(force) != tpg) continue;
            if (!best) best = tpg;        // smallest TPG = most threads ...
            if (uni_graph && !force && tpg == 8) best = 8;      // ... unless the uniform-row instance of width 8 applies: it is the
                                                                 // fastest form also for graphs small enough for narrower groups
        }
        if (!best) return MGADMM_OK;
        const bool band = g->mode == MGADMM_TEMPORAL_BAND;
        // (Twelve time steps per thread in a 640-thread workgroup -- 10 waves, 168 registers per thread, MGADMM_LDS_TPG=12 -- were
        // 5 % ahead of eight in a 960-thread one while the eight-step instance still spilled around its solves, and are 4 %
        // behind since it does not: cfg2 2.08 M against 2.17 M sample-iterations/s.  The smallest width stays the default.)
        const int sb_env = getenv("MGADMM_LDS_SB") ? atoi(getenv("MGADMM_LDS_SB")) : 0;   // 1: one LDS vector for p and q (experiments)
        lds.TPG = best;
        lds.G = T / best;
        lds.nthreads = N * lds.G;
        lds.block = (lds.nthreads + 63) / 64 * 64;
        // register budget: the kernel is compiled for the smallest workgroup-size class that holds the block
        lds.maxt = (best == 12 && lds.block <= 640) ? 640 : 1024;
        lds.sb = (sb_env && ((best == 12 && lds.maxt == 640) || best == 8)) ? 1 : 0;
        // the threads of the last wave that own no element are GHOSTS (lds_kernels.h): each gets an LDS row of zeros and table
        // rows of zero weights
        const int NR = N + (lds.block - lds.nthreads);
        lds.NR = NR;
        // kNN tables with k = 4 and no pads (the reference's setting): every W_u row has 4, every W_d row 5 entries -> the
        // instance with unrolled gathers that reads its rows from the global image
        lds.uniform45 = (!band && (best == 8 || best == 12) && !lds.sb && !getenv("MGADMM_LDS_RAGGED")) ? 1 : 0;
        for (int i = 0; i < N && lds.uniform45; ++i) {
            int ndiag = 0;
            for (int e = g->hWd.rowptr[i]; e < g->hWd.rowptr[i + 1]; ++e) ndiag += g->hWd.col[e] == i;
            if (g->hWu.rowptr[i + 1] - g->hWu.rowptr[i] != 4 || g->hWd.rowptr[i + 1] - g->hWd.rowptr[i] != 5 || ndiag != 1) lds.uniform45 = 0;
        }
        // diagonal entries: W_d^T[i][i] (every instance) and W_d[i][i] (uniform instances) are kept out of the tables -- inside a
        // CG solve their operand is the thread's own vector (registers): no LDS read
        std::vector<float> diag_d(NR, 0.f), diag_t(NR, 0.f);
        auto strip_diag = [&](const HostCsr& h, std::vector<float>& dg) {
            HostCsr o;
            o.rowptr.push_back(0);
            for (int i = 0; i < N; ++i) {
                for (int e = h.rowptr[i]; e < h.rowptr[i + 1]; ++e) {
                    if (h.col[e] == i) dg[i] += h.val[e];
                    else { o.col.push_back(h.col[e]); o.val.push_back(h.val[e]); }
                }
                o.rowptr.push_back((int)o.col.size());
            }
            return o;
        };
        const HostCsr hWdT_off = band ? HostCsr() : strip_diag(g->hWdT, diag_t);
        // W_d^T: LDS_NLEAD leading entries per row + a tail table of 2 * tail_pairs entries per row (rows padded with
        // {own row, weight 0}): one trip count for every lane of the workgroup
        int maxlen_t = 0;
        if (!band)
            for (int i = 0; i < N; ++i) maxlen_t = std::max(maxlen_t, hWdT_off.rowptr[i + 1] - hWdT_off.rowptr[i]);
        const int tp = band ? 0 : (std::max(0, maxlen_t - LDS_NLEAD) + 1) / 2;
        const int WT = LDS_NLEAD + 2 * tp;
        lds.tail_pairs = tp;
        const HostCsr hWd_tab = band ? HostCsr() : (lds.uniform45 ? strip_diag(g->hWd, diag_d) : g->hWd);
        // host tables with the ghosts' rows appended
        auto with_ghosts = [&](const HostCsr& h, int fixed_len) {
            HostCsr o;
            o.rowptr.assign(h.rowptr.begin(), h.rowptr.begin() + N + 1);
            o.col = h.col; o.val = h.val;
            for (int r = N; r < NR; ++r) {
                for (int e = 0; e < fixed_len; ++e) { o.col.push_back(r); o.val.push_back(0.f); }
                o.rowptr.push_back((int)o.col.size());
            }
            return o;
        };
        const HostCsr hu = with_ghosts(g->hWu, lds.uniform45 ? 4 : 0);
        const HostCsr hd = band ? HostCsr() : with_ghosts(hWd_tab, lds.uniform45 ? 4 : 0);
        const int nu = hu.nnz(), nd = band ? 0 : hd.nnz();
        auto al4 = [](int v) { return (v + 3) & ~3; };
        int off = 0;
        lds.off_rp_u = off; off += NR + 1;
        lds.off_rp_d = off; off += NR + 1;
        off = al4(off);
        lds.off_en_u = off; off += 2 * nu;
        lds.off_en_d = off; off += 2 * nd;
        lds.off_lead_t = off; off += 2 * NR * LDS_NLEAD;
        off = al4(off);
        lds.off_tail_t = off; off += 2 * NR * 2 * tp + 4;     // + one pair: gather_tail requests the next pair ahead
        const int tail_ints = off - lds.off_tail_t;
        lds.off_diag = off; off += 2 * NR;
        off += 8;                                            // the paired loops of the ragged gathers read three entries ahead
        lds.csr_ints = off;
        lds.lds_img0 = lds.uniform45 ? lds.off_tail_t : 0;
        lds.lds_img_ints = lds.uniform45 ? tail_ints : lds.off_diag;
        // LDS row stride: T padded to an odd number of 16-byte slots (rows then start on every bank group);
        // fall back to the unpadded stride when the padded vectors do not fit
        int ts = (T + 3) / 4 * 4;
        if (((ts / 4) & 1) == 0) ts += 4;
        auto bytes_for = [&](int stride, int slots) {
            const size_t LN = (size_t)NR * stride;
            return sizeof(float) * ((lds.sb ? 1 : 2) * LN + ((4 - (LN & 3)) & 3) + 32 + 16 * 12 + (size_t)slots * 2 * lds.block * best)
                   + sizeof(int) * (size_t)lds.lds_img_ints;
        };
        if (bytes_for(ts, 0) > 160 * 1024) ts = T;
        lds.TS = ts;
        if (bytes_for(ts, 0) > 160 * 1024) { lds = LdsPlan(); return MGADMM_OK; }
        // two more LDS vectors for per-thread operands (uniform instance): when they fit beside the padded images
        lds.slots = (lds.uniform45 && !getenv("MGADMM_LDS_NOSLOTS") && bytes_for(ts, 1) <= 160 * 1024) ? 1 : 0;
        lds.lds_bytes = bytes_for(ts, lds.slots);
        std::vector<int> img(off, 0);
        // Bank-aware entry order (lds_banks.h).  A gather instruction reads entry e of 64 consecutive threads' rows (64
        // consecutive nodes, mostly); ds_read_b128 serves it in four groups of 16 lanes, and two lanes of a group collide
        // when their neighbour rows start in the same 16-byte slot of the 256-byte bank line.  WHICH neighbour sits in entry
        // e of a row is free.  Round 1 kept the table order (36 % of the LDS cycles of k_admm_lds were bank conflicts); round 2
        // a greedy order, rows in node order, as the start of a min-conflicts search against an exact replay of the kernel's
        // read stream (ldsbank::improve_targeted): cfg2 goes from 7 800 to 1 200 weighted conflict cycles in 0.15 s of host
        // time per solver.  The sum of a row runs in the chosen order (fixed per graph: repeatable).
        // MGADMM_LDS_TABLE_ORDER=1 keeps the table order, MGADMM_LDS_BANK_SEARCH=<steps> sets the search length (0: greedy only).
        const bool bank_order = !band && best % 4 == 0 && ((lds.TS / 4) & 1) && !getenv("MGADMM_LDS_TABLE_ORDER");
        long search_steps = 4000;
        if (const char* e = getenv("MGADMM_LDS_BANK_SEARCH")) search_steps = atol(e);
        // order[e] = index into h.col / h.val of the entry the kernel reads at position e (real rows only; ghosts' rows follow as they are)
        auto slot_order = [&](const HostCsr& h, ldsbank::Stream stream) {
            std::vector<int> order(h.nnz());
            for (int e = 0; e < h.nnz(); ++e) order[e] = e;
            if (!bank_order || h.nnz() == 0) return order;
            ldsbank::Geometry q;
            q.N = N; q.G = T / best; q.TPG = best; q.TS = lds.TS; q.nlead = LDS_NLEAD;
            ldsbank::Mat m;
            m.rowptr.assign(h.rowptr.begin(), h.rowptr.begin() + N + 1);
            m.col.assign(h.col.begin(), h.col.begin() + h.rowptr[N]);
            m.src.assign(order.begin(), order.begin() + h.rowptr[N]);
            m.stream = stream;
            ldsbank::greedy_order(q, m);
            if (search_steps > 0) {
                std::vector<int> pos(N);
                for (int i = 0; i < N; ++i) pos[i] = i;
                const ldsbank::Result r = ldsbank::improve_targeted(q, m, pos, search_steps);
                if (getenv("MGADMM_LDS_BANK_STATS"))
                    fprintf(stderr, "[mgadmm] lds bank search: stream %d, %d entries: %.0f -> %.0f conflict cycles per application (%ld steps)\n",
                            (int)stream, h.rowptr[N], r.before, r.after, r.moves);
            }
            for (int e = 0; e < h.rowptr[N]; ++e) order[e] = m.src[e];
            return order;
        };
        auto put_entry = [&](int at, int col, float w) {
            img[at] = col * lds.TS;                           // LDS float offset of the neighbour's time row
            memcpy(&img[at + 1], &w, 4);
        };
        auto put_csr = [&](const HostCsr& h, int off_rp, int off_en, ldsbank::Stream stream) {
            const std::vector<int> order = slot_order(h, stream);
            for (int i = 0; i <= NR; ++i) img[off_rp + i] = h.rowptr[i];
            for (int e = 0; e < h.nnz(); ++e) put_entry(off_en + 2 * e, h.col[order[e]], h.val[order[e]]);
        };
        const ldsbank::Stream fixed_or_pairs = lds.uniform45 ? ldsbank::FIXED : ldsbank::PAIRS;     // gather_regs / gather
        put_csr(hu, lds.off_rp_u, lds.off_en_u, fixed_or_pairs);
        if (!band) {
            put_csr(hd, lds.off_rp_d, lds.off_en_d, fixed_or_pairs);
            // W_d^T as a table of WT entries per row: the row's entries, then {own row, 0}; every lane reads every position
            // (FIXED stream of the bank model), the first LDS_NLEAD positions from registers, the others from the tail table
            HostCsr ht;
            ht.rowptr.push_back(0);
            std::vector<int> from;                            // index into hWdT, -1: padding
            for (int i = 0; i < N; ++i) {
                const int e0 = hWdT_off.rowptr[i], len = hWdT_off.rowptr[i + 1] - e0;
                for (int e = 0; e < WT; ++e) {
                    ht.col.push_back(e < len ? hWdT_off.col[e0 + e] : i);
                    ht.val.push_back(e < len ? hWdT_off.val[e0 + e] : 0.f);
                    from.push_back(e < len ? e0 + e : -1);
                }
                ht.rowptr.push_back((int)ht.col.size());
            }
            const std::vector<int> order = slot_order(ht, ldsbank::FIXED);
            for (int r = 0; r < NR; ++r)
                for (int e = 0; e < WT; ++e) {
                    const int at = e < LDS_NLEAD ? lds.off_lead_t + 2 * (r * LDS_NLEAD + e)
                                                 : lds.off_tail_t + 2 * (r * 2 * tp + (e - LDS_NLEAD));
                    if (r < N) put_entry(at, ht.col[order[r * WT + e]], ht.val[order[r * WT + e]]);
                    else put_entry(at, r, 0.f);
                }
            put_entry(lds.off_tail_t + 2 * NR * 2 * tp, 0, 0.f);
            put_entry(lds.off_tail_t + 2 * NR * 2 * tp + 2, 0, 0.f);
            memcpy(&img[lds.off_diag], diag_d.data(), sizeof(float) * NR);
            memcpy(&img[lds.off_diag + NR], diag_t.data(), sizeof(float) * NR);
        }
        MG_HIP(hipMalloc(&d_lds_csr, sizeof(int) * off));
        MG_HIP(hipMemcpy(d_lds_csr, img.data(), sizeof(int) * off, hipMemcpyHostToDevice));
        MG_HIP(hipMalloc(&d_m2, sizeof(double) * T * N * (1 + (size_t)(Bmax + 63) / 64)));
        MG_HIP(hipMalloc(&d_stop, sizeof(int)));
        MG_HIP(hipMemset(d_stop, 0, sizeof(int)));
        MG_HIP(hipMalloc(&d_ps_ring, sizeof(double) * LDS_SETS * LDS_MAXJ_POOL * MGADMM_NMETRIC * Bp_max));
        {
            // helper stream of the overlapped outer loop: non-blocking (the caller's stream may be the legacy default stream,
            // which would serialise a blocking stream with itself), lowest priority (the k_admm_lds workgroups of the next
            // iteration take the CUs first; the one-wave workgroups of the metric kernels fill the room they leave)
            int lo = 0, hi = 0;
            MG_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
            if (const char* e = getenv("MGADMM_LDS_SIDE_PRIO")) lo = atoi(e) > 0 ? hi : lo;
            MG_HIP(hipStreamCreateWithPriority(&st_side, hipStreamNonBlocking, lo));
            for (auto& e : ev_main) MG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            for (auto& e : ev_side) MG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        lds.ok = true;
        return MGADMM_OK;
    }

    int launch_lds(const LdsArgs& a, int B) {
        const bool timed = prof_open(0, 0.0);
        LdsLaunch L{lds.TPG, lds.maxt, lds.sb, lds.uniform45, lds.slots, lds.block, lds.lds_bytes};
        const int rc = mg_lds_iteration(L, a, B, st);
        if (timed) prof_close();
        return rc;
    }

    int solve_lds(const void* y, const void* mask, int B, const void* x0, const mgadmm_state* state_in, void* x_out,
                  const mgadmm_state* state_out, mgadmm_history* hist) {
        if constexpr (!std::is_same<S, float>::value) {
            return MGADMM_ERR_UNSUPPORTED;
        } else {
            const int abl = p.ablation;
            const bool has_phi = (abl == MGADMM_ABL_NONE || abl == MGADMM_ABL_DGLR);
            const bool has_zd = (abl != MGADMM_ABL_DGLR);
            const int max_it = p.max_admm_iter;
            const bool record = p.record_cg_coeffs && hist && hist->cg_alpha && hist->cg_beta;
            const int Bp = (B + 63) / 64 * 64;
            const size_t TN = (size_t)T * N;
            if (hist && hist->metrics_per_sample) {
                size_t need = (size_t)max_it * MGADMM_NMETRIC * B;
                if (need > hist_ps_elems) {
                    if (d_hist_ps) MG_HIP(hipFree(d_hist_ps));
                    d_hist_ps = nullptr;
                    MG_HIP(hipMalloc(&d_hist_ps, need * sizeof(double)));
                    hist_ps_elems = need;
                }
            }
            MG_HIP(hipMemsetAsync(d_nonfinite, 0, sizeof(int), st));
            MG_HIP(hipMemsetAsync(d_ps, 0, sizeof(double) * MGADMM_NMETRIC * Bp, st));
            MG_HIP(hipMemsetAsync(d_cg_iters, 0, sizeof(int) * (size_t)max_it * 3 * Bp, st));
            // Schedule of the outer loop (ADMM.py:546-646 is one loop with the stop test at its end):
            //   SYNC     (MGADMM_LDS_ASYNC=0, or the CG coefficients are recorded: the host copies them out per iteration)
            //            one stream, the host reads the metrics of an iteration and tests the stop criterion before it enqueues the next;
            //   DEVSTOP  (check_stop) one stream, the stop test runs on the device (k_lds_stop_test sets the stop word, every
            //            later launch returns at its guard), the host enqueues iterations ahead and looks at the word LAG
            //            iterations late: same iterates and history, no host round trip per iteration;
            //   CHUNKS   (fixed iteration count) one k_admm_lds launch runs a CHUNK of J iterations on every sample (the workgroup
            //            keeps its sample: see the kernel); every iterate x_k goes to a buffer of its own, and the whole-batch
            //            metric kernels of a chunk (delta_x_per_step re-reads x_k and x_{k+1} of the batch, 241 MB per iteration
            //            at cfg2) run on a helper stream beside the launch of the next chunk, which leaves HBM idle.  Iterates
            //            at chunk boundaries rotate through three buffers, the iterates inside a chunk and the per-sample metric
            //            sums through two sets: launch c+2 overwrites what the metric kernels of chunk c read and waits for them.
            //            J = 1 is the overlapped one-iteration-per-launch schedule of round 2.
            enum { SYNC, DEVSTOP, CHUNKS };
            const int sched = (!lds_async || record) ? SYNC : (p.check_stop ? DEVSTOP : CHUNKS);
            // The caller's output buffers ARE the working state of this path (same (B, T*N) layout): no copy-out at the end
            // (round 2 until here: seven 120 MB device copies per solve at cfg2 = 0.7 ms of a 59 ms solve).  The iterates are
            // assigned to buffers so that the one of the LAST iteration lands in x_out (an early stop on another buffer
            // costs one copy).  Outputs must not alias y / mask (mgadmm.h).
            float* const xo_ = static_cast<float*>(x_out);
            // buffers for the iterates: the workspace vectors this path does not use otherwise
            static const int ring_ids[] = {V_XA, V_XB, V_ZUB, V_ZDB, V_PHIB, V_Y, V_MASK, V_R, V_P, V_Q, V_AP, V_RHS, V_TMP, V_IO0, V_IO1};
            constexpr int NRING = (int)(sizeof(ring_ids) / sizeof(ring_ids[0]));
            const int J = sched == CHUNKS ? std::max(1, std::min(std::min(lds_chunk, LDS_MAXJ_POOL), max_it)) : 1;
            {   // buffers beyond the workspace vectors for chunks longer than 7 iterations (kept for the solver's lifetime)
                const int need = LDS_SETS * (J - 1) + LDS_NBOUND - NRING;
                while ((int)lds_ring_extra.size() < need) {
                    float* b = nullptr;
                    MG_HIP(hipMalloc(&b, vec_elems * sizeof(S)));
                    lds_ring_extra.push_back(b);
                    ws_bytes += (int64_t)(vec_elems * sizeof(S));
                }
            }
            auto ring = [&](int j) -> float* { return j < NRING ? (float*)vec[ring_ids[j]] : lds_ring_extra[j - NRING]; };
            // iterate k (k = 0: the initial guess) lives in xbuf(k)
            auto xbuf = [&](int k) -> float* {
                if (k == max_it) return xo_;
                if (sched != CHUNKS) return ring(k & 1);
                if (k % J == 0) return ring((k / J) % LDS_NBOUND);                                        // chunk boundary
                return ring(LDS_NBOUND + ((k / J) % LDS_SETS) * (J - 1) + (k % J - 1));                    // inside chunk k / J
            };
            // zu, zd, phi and the dual variables: workspace vectors in the kernel's thread-major layout (lds_kernels.h,
            // lds_state_index); a warm start is converted in, the exported state is converted out at the end
            float *zu = vec[V_ZUA], *zd = vec[V_ZDA], *phi = vec[V_PHIA], *gam = vec[V_GAM], *gu = vec[V_GU], *gd = vec[V_GD];
            if (state_in) {
                const size_t nb = (size_t)B * TN * sizeof(float);
                if (x0 != nullptr && xbuf(0) != x0) MG_HIP(hipMemcpyAsync(xbuf(0), x0, nb, hipMemcpyDeviceToDevice, st));
                auto in = [&](float* dst, const void* src) {
                    return src == nullptr ? (int)MGADMM_OK : mg_lds_state_layout(true, T, N, lds.TPG, B, (const float*)src, dst, st);
                };
                MG_TRY(in(zu, state_in->zu));
                MG_TRY(in(gu, state_in->gamma_u));
                MG_TRY(in(zd, state_in->zd));            // the vectors an ablation does not iterate on are carried through
                MG_TRY(in(gd, state_in->gamma_d));
                MG_TRY(in(phi, state_in->phi));
                MG_TRY(in(gam, state_in->gamma));
            } else {
                float tm = 0, t2m = 0;
                for (int t = 0; t < p.t_in; ++t) { tm += (float)t; t2m += (float)t * (float)t; }
                tm /= (float)p.t_in;
                t2m /= (float)p.t_in;
                const float den = t2m - tm * tm;
                MG_TRY(mg_lds_init(mask != nullptr, T, p.t_in, N, lds.TPG, B, tm, den, (const float*)y, (const float*)mask, xbuf(0), zu, zd, gam, gu, gd,
                                   d_nonfinite, st));
            }
            LdsArgs a{};
            a.T = T; a.N = N; a.TN = (int)TN; a.TS = lds.TS; a.t_in = p.t_in; a.G = lds.G; a.B = B; a.Bp = Bp;
            a.nthreads = lds.nthreads; a.NR = lds.NR; a.tail_pairs = lds.tail_pairs;
            a.has_phi = has_phi; a.has_zd = has_zd;
            const LhsDef dx = lhs_def(MGADMM_LHS_X);
            a.lhsx_kind = dx.kind == 1 ? 1 : 0;
            a.cx1 = (float)dx.c1; a.cx2 = (float)dx.c2;
            a.band = g->mode == MGADMM_TEMPORAL_BAND; a.skip = g->skip;
            a.q1 = a.band ? 0 : g->q1;
            a.max_cg = p.max_cg_iter; a.record = record ? 1 : 0;
            a.rho = (float)p.rho; a.rho_u = (float)p.rho_u; a.rho_d = (float)p.rho_d;
            a.mu_u = (float)p.mu_u; a.mu_d1 = (float)p.mu_d1; a.mu_d2 = (float)p.mu_d2;
            a.cg_tol2 = p.cg_tol * p.cg_tol;
            a.csr = d_lds_csr; a.lds_img0 = lds.lds_img0; a.lds_img_ints = lds.lds_img_ints;
            a.off_rp_u = lds.off_rp_u; a.off_rp_d = lds.off_rp_d;
            a.off_en_u = lds.off_en_u; a.off_en_d = lds.off_en_d; a.off_lead_t = lds.off_lead_t; a.off_tail_t = lds.off_tail_t; a.off_diag = lds.off_diag;
            a.band_w = g->band_w;
            a.zu = zu; a.zd = zd; a.phi = phi; a.gam = gam; a.gu = gu; a.gd = gd;
            // staggered start (k_admm_lds): only when the launch runs several rounds of workgroups per CU
            {
                const int ncu = dev_cus > 0 ? dev_cus : 256;
                int us = 48;
                if (const char* e = getenv("MGADMM_LDS_STAGGER_US")) us = atoi(e);
                a.stagger_wgs = ncu;
                a.stagger_ticks = (B >= 2 * ncu && us > 0) ? us * 100 : 0;
            }
            a.y = (const float*)y; a.mask = (const float*)mask;
            a.alpha_hist = record ? (float*)d_alpha_hist : nullptr; a.beta_hist = record ? (float*)d_beta_hist : nullptr;
            a.nonfinite = d_nonfinite;
            a.stop = sched == DEVSTOP ? d_stop : nullptr;
            if (sched == DEVSTOP) MG_HIP(hipMemsetAsync(d_stop, 0, sizeof(int), st));
            if (sched == CHUNKS) MG_HIP(hipMemsetAsync(d_ps_ring, 0, sizeof(double) * LDS_SETS * J * MGADMM_NMETRIC * Bp, st));
            const size_t K = p.max_cg_iter;
            int n_done = 0, rc_final = MGADMM_OK;
            // the whole-batch metrics of iteration `it` (delta_x_per_step, norms / means over the samples) on stream `s`
            auto batch_metrics = [&](int it, const double* ps, hipStream_t s) -> int {
                MG_TRY(mg_lds_dxps(T, N, B, (const float*)xbuf(it + 1), (const float*)xbuf(it), d_m2, d_dxps + (size_t)it * T, a.stop, s));
                hipLaunchKernelGGL(k_batch_metrics, dim3(MGADMM_NMETRIC), dim3(256), 0, s, ps, Bp, B,
                                   d_hist + (size_t)it * MGADMM_NMETRIC,
                                   (hist && hist->metrics_per_sample) ? d_hist_ps + (size_t)it * MGADMM_NMETRIC * B : nullptr);
                MG_HIP(hipGetLastError());
                return MGADMM_OK;
            };
            if (sched == CHUNKS) {
                int c = 0;
                for (int it0 = 0; it0 < max_it; it0 += J, ++c) {
                    const int Jc = std::min(J, max_it - it0);
                    a.first = it0 == 0 && !state_in;      // phi = Ldr x0 is formed by the first iteration of a cold start
                    a.J = Jc;
                    for (int k = 0; k <= Jc; ++k) a.xs[k] = xbuf(it0 + k);
                    a.cg_iters = d_cg_iters + (size_t)it0 * 3 * Bp;
                    a.ps = d_ps_ring + (size_t)(c % LDS_SETS) * J * MGADMM_NMETRIC * Bp;
                    // launch c overwrites the iterate buffers and the metric sums that the metric kernels of chunk c-3 read (interior
                    // set and metric set c % 3; boundary buffer (c + 1) % 4 = the start of chunk c-3).  Two sets / three boundary
                    // buffers (until the end of round 3) made launch c wait for the metrics of chunk c-2, which run BESIDE launch c-1
                    // and get CU slots only when it drains: 0.45 ms between two 26 ms launches (profiles/r03/cfg2_iteration_timeline.txt)
                    if (c >= LDS_SETS) MG_HIP(hipStreamWaitEvent(st, ev_side[(c - LDS_SETS) % LDS_NBOUND], 0));
                    MG_TRY(launch_lds(a, B));
                    MG_HIP(hipEventRecord(ev_main[c % LDS_NBOUND], st));
                    MG_HIP(hipStreamWaitEvent(st_side, ev_main[c % LDS_NBOUND], 0));
                    for (int k = 0; k < Jc; ++k) MG_TRY(batch_metrics(it0 + k, a.ps + (size_t)k * MGADMM_NMETRIC * Bp, st_side));
                    MG_HIP(hipEventRecord(ev_side[c % LDS_NBOUND], st_side));
                    n_done = it0 + Jc;
                }
                if (c > 0)            // join the helper stream (its kernels run in order: the last event covers all)
                    MG_HIP(hipStreamWaitEvent(st, ev_side[(c - 1) % LDS_NBOUND], 0));
            }
            for (int it = 0; sched != CHUNKS && it < max_it; ++it) {
                a.first = it == 0 && !state_in;      // phi = Ldr x0 is formed by the first launch of a cold start
                a.J = 1;
                a.xs[0] = xbuf(it); a.xs[1] = xbuf(it + 1);
                a.cg_iters = d_cg_iters + (size_t)it * 3 * Bp;
                a.ps = d_ps;
                if (record) {
                    MG_TRY(fill((S*)d_alpha_hist, 3 * K * Bp, (S)NAN));
                    MG_TRY(fill((S*)d_beta_hist, 3 * K * Bp, (S)NAN));
                }
                MG_TRY(launch_lds(a, B));
                MG_TRY(batch_metrics(it, a.ps, st));
                if (record) {
                    for (int w = 0; w < 3; ++w) {
                        if (w == 2 && !has_zd) continue;
                        double* ao = hist->cg_alpha + ((size_t)it * 3 + w) * K * B;
                        double* bo = hist->cg_beta + ((size_t)it * 3 + w) * K * B;
                        MG_TRY(fetch_hist(d_alpha_hist + (size_t)w * K * Bp, K * Bp, ao, B, Bp, K));
                        MG_TRY(fetch_hist(d_beta_hist + (size_t)w * K * Bp, K * Bp, bo, B, Bp, K));
                    }
                }
                n_done = it + 1;
                if (sched == DEVSTOP) {
                    MG_TRY(mg_lds_stop_test(d_hist + (size_t)it * MGADMM_NMETRIC, d_nonfinite, has_phi, has_zd, p.admm_tol, it, d_stop, st));
                    MG_HIP(hipMemcpyAsync(h_flag + 1 + it % (LAG + 1), d_stop, sizeof(int), hipMemcpyDeviceToHost, st));
                    MG_HIP(hipEventRecord(ev_ring[it % (LAG + 1)], st));
                    if (it >= LAG) {
                        MG_HIP(hipEventSynchronize(ev_ring[(it - LAG) % (LAG + 1)]));
                        if (h_flag[1 + (it - LAG) % (LAG + 1)] != 0) break;      // the launches enqueued since returned at their guards
                    }
                } else if (p.check_stop) {
                    MG_HIP(hipMemcpyAsync(h_row, d_hist + (size_t)it * MGADMM_NMETRIC, sizeof(double) * MGADMM_NMETRIC,
                                          hipMemcpyDeviceToHost, st));
                    MG_HIP(hipMemcpyAsync(h_flag, d_nonfinite, sizeof(int), hipMemcpyDeviceToHost, st));
                    MG_HIP(hipStreamSynchronize(st));
                    bool finite = h_flag[0] == 0;
                    for (int k = 0; k < MGADMM_NMETRIC; ++k) finite = finite && std::isfinite(h_row[k]);
                    if (!finite) { rc_final = MGADMM_ERR_NONFINITE; break; }
                    double pri = h_row[MGADMM_M_PRI_ZU], dual = h_row[MGADMM_M_DUAL_ZU];
                    if (has_phi) { pri = std::max(pri, h_row[MGADMM_M_PRI_PHI]); dual = std::max(dual, h_row[MGADMM_M_DUAL_PHI]); }
                    if (has_zd) { pri = std::max(pri, h_row[MGADMM_M_PRI_ZD]); dual = std::max(dual, h_row[MGADMM_M_DUAL_ZD]); }
                    if (pri < p.admm_tol && dual < p.admm_tol) break;
                }
            }
            if (sched == DEVSTOP) {
                // the stop word after everything enqueued has run: 0 = no stop (all enqueued iterations ran), k > 0 = the stop
                // test passed at the end of iteration k - 1, k < 0 = iteration -k - 1 met a NaN / Inf
                MG_HIP(hipMemcpyAsync(h_flag + 1, d_stop, sizeof(int), hipMemcpyDeviceToHost, st));
                MG_HIP(hipStreamSynchronize(st));
                const int sw = h_flag[1];
                if (sw > 0) n_done = sw;
                else if (sw < 0) { n_done = -sw; rc_final = MGADMM_ERR_NONFINITE; }
            }
            float* const xc = xbuf(n_done);
            if (xc != xo_)        // early stop on the other parity
                MG_HIP(hipMemcpyAsync(x_out, xc, (size_t)B * TN * sizeof(float), hipMemcpyDeviceToDevice, st));
            if (state_out) {      // the state the caller asked for, in the reference's layout
                auto out = [&](void* dst, const float* src) {
                    return dst == nullptr ? (int)MGADMM_OK : mg_lds_state_layout(false, T, N, lds.TPG, B, src, (float*)dst, st);
                };
                MG_TRY(out(state_out->zu, zu));
                MG_TRY(out(state_out->zd, zd));
                if (has_phi) {
                    MG_TRY(out(state_out->phi, phi));
                    MG_TRY(out(state_out->gamma, gam));
                }
                MG_TRY(out(state_out->gamma_u, gu));
                MG_TRY(out(state_out->gamma_d, gd));
            }
            return finish_history(hist, n_done, B, Bp, rc_final);
        }
    }

    // copies the device-side history of a finished solve into the caller's host buffers
    int finish_history(mgadmm_history* hist, int n_done, int B, int Bp, int rc_final) {
        if (hist) {
            hist->n_iters = n_done;
            if (hist->metrics)
                MG_HIP(hipMemcpyAsync(hist->metrics, d_hist, sizeof(double) * (size_t)n_done * MGADMM_NMETRIC, hipMemcpyDeviceToHost, st));
            if (hist->delta_x_per_step)
                MG_HIP(hipMemcpyAsync(hist->delta_x_per_step, d_dxps, sizeof(double) * (size_t)n_done * T, hipMemcpyDeviceToHost, st));
            if (hist->metrics_per_sample)
                MG_HIP(hipMemcpyAsync(hist->metrics_per_sample, d_hist_ps, sizeof(double) * (size_t)n_done * MGADMM_NMETRIC * B,
                                      hipMemcpyDeviceToHost, st));
            if (hist->cg_iters) {
                std::vector<int> tmp((size_t)n_done * 3 * Bp);
                MG_HIP(hipMemcpyAsync(tmp.data(), d_cg_iters, sizeof(int) * tmp.size(), hipMemcpyDeviceToHost, st));
                MG_HIP(hipStreamSynchronize(st));
                for (size_t r = 0; r < (size_t)n_done * 3; ++r)
                    memcpy(hist->cg_iters + r * B, tmp.data() + r * Bp, sizeof(int) * B);
            }
        }
        MG_HIP(hipMemcpyAsync(h_flag, d_nonfinite, sizeof(int), hipMemcpyDeviceToHost, st));
        MG_HIP(hipStreamSynchronize(st));
        if (rc_final == MGADMM_OK && h_flag[0]) rc_final = MGADMM_ERR_NONFINITE;
        if (rc_final == MGADMM_OK && hist && hist->metrics) {
            for (size_t k = 0; k < (size_t)n_done * MGADMM_NMETRIC; ++k)
                if (!std::isfinite(hist->metrics[k])) rc_final = MGADMM_ERR_NONFINITE;
        }
        if (rc_final == MGADMM_ERR_NONFINITE) mg_set_error("solve: NaN/Inf met in the iterates (cf. the asserts of ADMM.py:534-606)");
        return rc_final;
    }

    int save_coeffs(const Geom& q, mgadmm_history* hist, int it, int which, int B) {
        const size_t K = p.max_cg_iter;
        double* a = hist->cg_alpha + ((size_t)it * 3 + which) * K * B;
        double* b = hist->cg_beta + ((size_t)it * 3 + which) * K * B;
        MG_TRY(fetch_hist(d_alpha_hist, K * q.Bp, a, B, q.Bp, K));
        return fetch_hist(d_beta_hist, K * q.Bp, b, B, q.Bp, K);
    }
};

}  // namespace

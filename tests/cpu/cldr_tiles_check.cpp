// CPU check of the fused cLdr tile metadata (mixed-graph-admm_amd/csrc/cldr_tiles.h): replays the dataflow of the HIP
// kernel k_cldr on the host -- per tile an image P of x_t on the 2-hop set C2 and an image Q of q_{t+1} = (Ldr x)_{t+1}
// on the 1-hop set C1, swept over the time steps -- and compares the result with Ldr^T(Ldr x) evaluated directly from
// the CSR matrices with the operator definitions of reference ADMM.py:150-228.  Test infrastructure (g++ only).
//   usage: cldr_tiles_check <n> <T> <Rcap> <C1cap> <C2cap> <GD> <GT> <cluster>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>

#include "cldr_tiles.h"

static HostCsr transpose(const HostCsr& A) {
    HostCsr At;
    At.n = A.n;
    At.rowptr.assign(A.n + 1, 0);
    At.col.resize(A.nnz());
    At.val.resize(A.nnz());
    for (int e = 0; e < A.nnz(); ++e) At.rowptr[A.col[e] + 1]++;
    for (int i = 0; i < A.n; ++i) At.rowptr[i + 1] += At.rowptr[i];
    std::vector<int> fill(At.rowptr.begin(), At.rowptr.end() - 1);
    for (int i = 0; i < A.n; ++i)
        for (int e = A.rowptr[i]; e < A.rowptr[i + 1]; ++e) {
            int d = fill[A.col[e]]++;
            At.col[d] = i;
            At.val[d] = A.val[e];
        }
    return At;
}

int main(int argc, char** argv) {
    if (argc < 9) return 2;
    const int n = atoi(argv[1]), T = atoi(argv[2]);
    CldrCaps caps{atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7])};
    const int cluster = atoi(argv[8]);
    std::mt19937 rng(7);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    // points on a jittered grid walked in boustrophedon strips: consecutive rows are spatial neighbours
    const int side = (int)std::ceil(std::sqrt((double)n));
    std::vector<double> px(n), py(n);
    for (int i = 0; i < n; ++i) {
        const int strip = (i / side) / 8, in = i - strip * 8 * side, cx = in / 8, cy = in % 8;
        px[i] = cx + 0.8 * U(rng);
        py[i] = strip * 8 + ((cx & 1) ? 7 - cy : cy) + 0.8 * U(rng);
    }
    const int k = 4;
    HostCsr Wd;
    Wd.n = n;
    Wd.rowptr.push_back(0);
    for (int i = 0; i < n; ++i) {
        std::vector<std::pair<double, int>> d;
        for (int j = 0; j < n; ++j) {
            const double dx = px[i] - px[j], dy = py[i] - py[j];
            if (dx * dx + dy * dy < 36.0) d.push_back({dx * dx + dy * dy, j});
        }
        std::sort(d.begin(), d.end());
        int take = std::min<int>(k + 1, (int)d.size());
        if (i % 97 == 0) take = std::min(take, 2);          // ragged rows
        double s = 0;
        for (int u = 0; u < take; ++u) s += std::exp(-std::sqrt(d[u].first) / 3.0);
        for (int u = 0; u < take; ++u) {
            Wd.col.push_back(d[u].second);
            Wd.val.push_back((float)(std::exp(-std::sqrt(d[u].first) / 3.0) / s));
        }
        Wd.rowptr.push_back((int)Wd.col.size());
    }
    HostCsr WdT = transpose(Wd);
    std::vector<int> cuts;
    for (int c = 0; c < n; c += cluster) cuts.push_back(c);
    CldrTiles tl;
    if (!build_cldr_tiles(Wd, WdT, cuts, caps, tl)) {
        printf("INELIGIBLE\n");
        return 0;
    }
    // tiles partition the rows
    if (tl.n0.front() != 0 || tl.n0.back() != n) { printf("FAIL partition\n"); return 1; }
    std::vector<double> x((size_t)T * n), ref((size_t)T * n), got((size_t)T * n, 1e300);
    for (auto& v : x) v = U(rng) - 0.5;
    // direct: q = Ldr x ; y = Ldr^T q   (kNN branch, quirk Q1 irrelevant because q_0 = 0)
    std::vector<double> q((size_t)T * n, 0.0);
    for (int t = 1; t < T; ++t)
        for (int i = 0; i < n; ++i) {
            double s = 0;
            for (int e = Wd.rowptr[i]; e < Wd.rowptr[i + 1]; ++e) s += (double)Wd.val[e] * x[(size_t)(t - 1) * n + Wd.col[e]];
            q[(size_t)t * n + i] = x[(size_t)t * n + i] - s;
        }
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < n; ++i) {
            double s = 0;
            if (t + 1 < T)
                for (int e = WdT.rowptr[i]; e < WdT.rowptr[i + 1]; ++e) s += (double)WdT.val[e] * q[(size_t)(t + 1) * n + WdT.col[e]];
            ref[(size_t)t * n + i] = q[(size_t)t * n + i] - s;
        }
    // kernel dataflow
    const CldrCaps& c = tl.caps;
    for (int tile = 0; tile < tl.NT; ++tile) {
        const int n0 = tl.n0[tile], R = tl.n0[tile + 1] - n0, c1 = tl.nC1[tile], c2 = tl.nC2[tile];
        const int* rows = &tl.rows[(size_t)tile * c.C2cap];
        for (int l = 0; l < R; ++l)
            if (rows[l] != n0 + l) { printf("FAIL own rows\n"); return 1; }
        std::vector<double> P(c.C2cap, 0.0), Q(c.C1cap, 0.0), pnext(c.C2cap), pcur(c.Rcap), qprev(c.Rcap, 0.0), qnew(c.C1cap);
        for (int l = 0; l < c2; ++l) P[l] = x[rows[l]];
        for (int l = 0; l < R; ++l) pcur[l] = P[l];
        for (int t = 0; t < T; ++t) {
            const bool nxt = t + 1 < T;
            if (nxt) {
                for (int l = 0; l < c2; ++l) pnext[l] = x[(size_t)(t + 1) * n + rows[l]];
                for (int j = 0; j < c1; ++j) {                               // phase A: q_{t+1} on C1
                    double s = 0;
                    for (int u = 0; u < c.GD; ++u) {
                        const size_t o = ((size_t)tile * c.C1cap + j) * c.GD + u;
                        if (tl.dcol[o] < 0 || tl.dcol[o] >= c2) { printf("FAIL dcol range\n"); return 1; }
                        s += (double)tl.dw[o] * P[tl.dcol[o]];
                    }
                    qnew[j] = pnext[j] - s;
                    Q[j] = qnew[j];
                }
            }
            for (int i = 0; i < R; ++i) {                                   // phase C: y_t on the own rows
                double s = 0;
                if (nxt)
                    for (int u = 0; u < c.GT; ++u) {
                        const size_t o = ((size_t)tile * c.Rcap + i) * c.GT + u;
                        if (tl.tcol[o] < 0 || tl.tcol[o] >= c1) { printf("FAIL tcol range\n"); return 1; }
                        s += (double)tl.tw[o] * Q[tl.tcol[o]];
                    }
                got[(size_t)t * n + n0 + i] = qprev[i] - s;
            }
            if (nxt) {
                for (int l = 0; l < c2; ++l) P[l] = pnext[l];
                for (int i = 0; i < R; ++i) { qprev[i] = qnew[i]; pcur[i] = pnext[i]; }
            }
        }
    }
    double err = 0, nrm = 0;
    for (size_t i = 0; i < ref.size(); ++i) { err = std::max(err, std::fabs(ref[i] - got[i])); nrm = std::max(nrm, std::fabs(ref[i])); }
    printf("tiles %d meanR %.1f C1 %.1f C2 %.1f  max err %.3e (max |ref| %.3f)\n", tl.NT, (double)n / tl.NT, (double)tl.sumC1 / tl.NT,
           (double)tl.sumC2 / tl.NT, err, nrm);
    if (!(err < 1e-12)) { printf("FAIL\n"); return 1; }
    printf("OK\n");
    return 0;
}

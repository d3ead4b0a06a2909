// CPU check / experiment driver of the LDS bank model (mixed-graph-admm_amd/csrc/lds_banks.h).
//   lds_banks_check <graph file> <budget> <move_rows 0|1>
// graph file: "N G TPG TS nlead", then per matrix "stream weight nnz", N+1 row pointers, nnz columns.
// Prints the simulated LDS cycles / conflict cycles per matrix before and after the search and checks the invariants:
// every row still holds the same multiset of columns, `src` is the permutation that was applied, positions stay a
// permutation, the conflict count did not grow and equals a fresh replay.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include "lds_banks.h"

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    FILE* f = fopen(argv[1], "r");
    if (!f) return 2;
    ldsbank::Geometry q;
    if (fscanf(f, "%d %d %d %d %d", &q.N, &q.G, &q.TPG, &q.TS, &q.nlead) != 5) return 2;
    std::vector<ldsbank::Mat> mats;
    for (;;) {
        int stream, nnz;
        double w;
        if (fscanf(f, "%d %lf %d", &stream, &w, &nnz) != 3) break;
        ldsbank::Mat m;
        m.stream = (ldsbank::Stream)stream;
        m.weight = w;
        m.rowptr.resize(q.N + 1);
        m.col.resize(nnz);
        for (auto& v : m.rowptr) if (fscanf(f, "%d", &v) != 1) return 2;
        for (auto& v : m.col) if (fscanf(f, "%d", &v) != 1) return 2;
        m.src.resize(nnz);
        for (int e = 0; e < nnz; ++e) m.src[e] = e;
        mats.push_back(m);
    }
    fclose(f);
    const long budget = atol(argv[2]);
    const bool move_rows = atoi(argv[3]) != 0;
    const bool greedy = argc > 4 && atoi(argv[4]) != 0;
    std::vector<int> pos(q.N);
    for (int i = 0; i < q.N; ++i) pos[i] = i;
    const std::vector<ldsbank::Mat> orig = mats;
    if (greedy) for (auto& m : mats) ldsbank::greedy_order(q, m);
    for (size_t m = 0; m < mats.size(); ++m) {
        const auto c = ldsbank::simulate(q, mats[m], pos);
        printf("before  matrix %zu stream %d: cycles %ld conflicts %ld (%.1f %%)\n", m, (int)mats[m].stream, c.cycles, c.conflicts, 100.0 * c.conflicts / c.cycles);
    }
    const bool targeted = argc > 5 && atoi(argv[5]) != 0;
    ldsbank::Result r;
    if (targeted) {
        for (auto& m : mats) {
            const auto rr = ldsbank::improve_targeted(q, m, pos, budget);
            r.moves += rr.moves; r.accepted += rr.accepted;
            r.before += m.weight * rr.before; r.after += m.weight * rr.after;
        }
        double wsum = 0;
        for (auto& m : mats) wsum += m.weight;
        r.before += wsum * ldsbank::store_conflicts(q, pos); r.after += wsum * ldsbank::store_conflicts(q, pos);
    } else {
        r = ldsbank::improve(q, mats, pos, move_rows, budget);
    }
    double check = 0, wsum = 0;
    for (size_t m = 0; m < mats.size(); ++m) {
        const auto c = ldsbank::simulate(q, mats[m], pos);
        printf("after   matrix %zu stream %d: cycles %ld conflicts %ld (%.1f %%)\n", m, (int)mats[m].stream, c.cycles, c.conflicts, 100.0 * c.conflicts / c.cycles);
        check += mats[m].weight * c.conflicts;
        wsum += mats[m].weight;
    }
    check += wsum * ldsbank::store_conflicts(q, pos);
    printf("weighted conflicts %.1f -> %.1f (replay %.1f), store conflicts %ld, %ld moves, %ld accepted\n", r.before, r.after, check,
           ldsbank::store_conflicts(q, pos), r.moves, r.accepted);
    bool ok = r.after <= r.before && std::fabs(check - r.after) < 1e-6 * (1 + check);
    std::vector<int> seen(q.N, 0);
    for (int i = 0; i < q.N; ++i) { if (pos[i] < 0 || pos[i] >= q.N || seen[pos[i]]++) ok = false; }
    for (size_t m = 0; m < mats.size(); ++m)
        for (int i = 0; i < q.N; ++i) {
            std::map<int, int> a, b;
            for (int e = orig[m].rowptr[i]; e < orig[m].rowptr[i + 1]; ++e) {
                ++a[orig[m].col[e]]; ++b[mats[m].col[e]];
                if (mats[m].src[e] < orig[m].rowptr[i] || mats[m].src[e] >= orig[m].rowptr[i + 1] || orig[m].col[mats[m].src[e]] != mats[m].col[e]) ok = false;
            }
            if (a != b) ok = false;
        }
    printf(ok ? "OK\n" : "FAILED\n");
    return ok ? 0 : 1;
}

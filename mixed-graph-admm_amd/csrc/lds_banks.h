// LDS bank model of the gathers of k_admm_lds (lds_kernels.h) and the host-side search that arranges the LDS image for
// it.  Plain C++, no HIP: compiled into libmgadmm.so (Engine::plan_lds) and into the CPU check tests/cpu/lds_banks_check.cpp.
//
// A gather instruction of the kernel is one ds_read_b128 per lane: lane (g, i) -- node i, time group g -- reads 16 bytes of
// the LDS row of the neighbour in entry e of ITS CSR row.  The hardware serves a wave's ds_read_b128 in four groups of 16
// lanes, one LDS cycle per group when the 16 addresses fall into 16 different 16-byte slots of the 256-byte bank line, one
// more cycle for every further distinct address in the busiest slot (MI355X_MICROARCH.md, LDS).  A timing build in which
// every lane reads its own row (no conflict possible, same instruction stream) runs the cfg2 launch 18 % faster than the
// production kernel: the conflicts of the gathers, not their volume, are what the LDS costs.
//
// Two things are free on the host and decide the slots:
//   * WHICH neighbour sits in entry e of a row (the sum of a row then runs in that order: fixed per graph, repeatable);
//   * WHERE the row of node j lies in LDS: row position pos[j] (a permutation of the nodes; HBM layout and the
//     thread -> node map are untouched, the entries carry pos[col] * TS and a thread stores at pos[i] * TS).
// `simulate` replays the exact read stream of one operator application (which lanes take part in which instruction,
// including the leading entries gather_lead reads past the end of a short row and the paired tail loop) and counts LDS
// cycles; `improve` is a seeded hill climb over entry swaps and row-position swaps against that count.
#pragma once
#include <algorithm>
#include <cstdint>
#include <utility>
#include <vector>

namespace ldsbank {

// lane (mod 32) -> which of the two 16-lane groups of its half-wave serves it in a ds_read_b128
static const int kGrpOfLane[32] = {0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1};

enum Stream {
    FIXED = 0,       // gather_fixed<L>: every lane reads the L entries of its row (all rows hold L)
    LEAD_PAIRS = 1,  // gather_lead<NLEAD>: NLEAD entries read by every lane (also past the end of its row), the rest in pairs
    PAIRS = 2        // gather: pairs of entries, then a single one
};

struct Geometry {
    int N = 0, G = 0, TPG = 0, TS = 0;   // nodes, time groups, time steps per thread, LDS row stride (floats)
    int nlead = 4;
};

// one CSR matrix in the order the kernel reads it: col[rowptr[i] + e] = neighbour in entry e of row i; reads past the end
// of the table (gather_lead on the last rows) hit the zero padding of the image: row position 0
struct Mat {
    std::vector<int> rowptr, col;
    std::vector<int> src;    // src[rowptr[i] + e] = index of that entry in the caller's table (filled by the caller, carried by the swaps)
    Stream stream = PAIRS;
    double weight = 1.0;     // applications of the operator per ADMM iteration (relative): what a cycle of it is worth
};

struct WaveCost { long cycles = 0, conflicts = 0; };

// LDS cycles of the ds_read_b128 one wave issues for ONE 16-byte piece of every entry (the TPG/4 pieces of an entry have the
// same pattern, shifted by one slot each)
// culprits (optional): (node, entry index) of every lane that sits in a busiest slot of a group that needed extra cycles
inline WaveCost simulate_wave(const Geometry& q, const Mat& m, const std::vector<int>& pos, int w,
                              std::vector<std::pair<int, int>>* culprits = nullptr) {
    int node[64], t0[64], e0[64], len[64];
    bool act[64];
    int maxlen = 0;
    const int nnz = (int)m.col.size();
    for (int l = 0; l < 64; ++l) {
        const int tid = w * 64 + l;
        act[l] = tid < q.N * q.G;
        const int g = act[l] ? tid / q.N : 0;
        node[l] = act[l] ? tid - g * q.N : 0;
        t0[l] = g * q.TPG;
        e0[l] = m.rowptr[node[l]];
        len[l] = m.rowptr[node[l] + 1] - e0[l];
        if (act[l]) maxlen = std::max(maxlen, len[l]);
    }
    WaveCost out;
    int ent[64];         // entry index read by the lane in the current instruction, -1: lane masked off
    auto issue = [&]() {
        for (int grp = 0; grp < 4; ++grp) {
            int addr[16][16], cnt[16] = {0}, slot_of[64];
            bool any = false;
            for (int l = 0; l < 64; ++l) {
                slot_of[l] = -1;
                if (ent[l] < 0 || (l >> 5) * 2 + kGrpOfLane[l & 31] != grp) continue;
                any = true;
                const int c = ent[l] < nnz ? m.col[ent[l]] : -1;
                const int a = (c < 0 ? 0 : pos[c] * q.TS) + t0[l];           // float index of the 16-byte piece
                const int s = (a >> 2) & 15;
                slot_of[l] = s;
                bool seen = false;
                for (int k = 0; k < cnt[s]; ++k) seen = seen || addr[s][k] == a;
                if (!seen) addr[s][cnt[s]++] = a;
            }
            if (!any) continue;
            int worst = 1;
            for (int s = 0; s < 16; ++s) worst = std::max(worst, cnt[s]);
            out.cycles += worst;
            out.conflicts += worst - 1;
            if (culprits && worst > 1)
                for (int l = 0; l < 64; ++l)
                    if (slot_of[l] >= 0 && cnt[slot_of[l]] == worst && ent[l] < nnz) {
                        // the entry may belong to a LATER row (gather_lead past the end of a short row): find its row
                        int r = node[l];
                        while (r + 1 < q.N && m.rowptr[r + 1] <= ent[l]) ++r;
                        culprits->push_back({r, ent[l] - m.rowptr[r]});
                    }
        }
    };
    int first_pair = 0;
    if (m.stream == FIXED) {
        for (int u = 0; u < maxlen; ++u) {
            for (int l = 0; l < 64; ++l) ent[l] = (act[l] && u < len[l]) ? e0[l] + u : -1;
            issue();
        }
        return out;
    }
    if (m.stream == LEAD_PAIRS) {
        for (int u = 0; u < q.nlead; ++u) {
            for (int l = 0; l < 64; ++l) ent[l] = act[l] ? e0[l] + u : -1;          // read whether or not the row is that long
            issue();
        }
        first_pair = q.nlead;
    }
    // paired loop: trip k is run by the lanes whose row still holds two entries; then the lanes with one entry left
    for (int k = 0;; ++k) {
        bool any = false;
        for (int l = 0; l < 64; ++l) any = any || (act[l] && len[l] > first_pair && first_pair + 2 * k + 1 < len[l]);
        if (!any) break;
        for (int half = 0; half < 2; ++half) {
            for (int l = 0; l < 64; ++l)
                ent[l] = (act[l] && len[l] > first_pair && first_pair + 2 * k + 1 < len[l]) ? e0[l] + first_pair + 2 * k + half : -1;
            issue();
        }
    }
    {
        bool any = false;
        for (int l = 0; l < 64; ++l) {
            const int rest = len[l] - first_pair;
            ent[l] = (act[l] && rest > 0 && (rest & 1)) ? e0[l] + len[l] - 1 : -1;
            any = any || ent[l] >= 0;
        }
        if (any) issue();
    }
    return out;
}

inline WaveCost simulate(const Geometry& q, const Mat& m, const std::vector<int>& pos) {
    WaveCost t;
    const int nw = (q.N * q.G + 63) / 64;
    for (int w = 0; w < nw; ++w) {
        const WaveCost c = simulate_wave(q, m, pos, w);
        t.cycles += c.cycles;
        t.conflicts += c.conflicts;
    }
    return t;
}

// LDS cycles of the thread's own 16-byte stores (ds_write_b128: eight groups of eight consecutive lanes, 32 banks of the
// 128-byte half line: two rows collide when their positions agree mod 8 for an odd slot stride)
inline long store_conflicts(const Geometry& q, const std::vector<int>& pos) {
    long c = 0;
    const int nw = (q.N * q.G + 63) / 64;
    for (int w = 0; w < nw; ++w)
        for (int g8 = 0; g8 < 8; ++g8) {
            int cnt[8] = {0};
            for (int l = g8 * 8; l < g8 * 8 + 8; ++l) {
                const int tid = w * 64 + l;
                if (tid >= q.N * q.G) continue;
                const int gi = tid / q.N, i = tid - gi * q.N;
                ++cnt[((pos[i] * q.TS + gi * q.TPG) >> 2) & 7];
            }
            int worst = 1;
            for (int s = 0; s < 8; ++s) worst = std::max(worst, cnt[s]);
            c += worst - 1;
        }
    return c;
}

// Starting point of the search (round-2 heuristic): rows are taken in node order and every row picks the permutation of its
// entries that collides least (slot colour = column mod 16 under the identity row positions) with the rows already placed
// in its lane groups -- all permutations up to 5 entries, the offset-sorted order and random shuffles beyond.
inline void greedy_order(const Geometry& q, Mat& m) {
    const int N = q.N, G = q.G;
    int maxlen = 0;
    for (int i = 0; i < N; ++i) maxlen = std::max(maxlen, m.rowptr[i + 1] - m.rowptr[i]);
    const int nwaves = (N * G + 63) / 64;
    std::vector<int> used((size_t)nwaves * 4 * std::max(maxlen, 1) * 16, -1);   // [(wave*4 + group)*maxlen + e][colour] = column (-1 free, -2 several)
    unsigned rng = 12345u;
    auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
    const std::vector<int> col0 = m.col, src0 = m.src;
    for (int i = 0; i < N; ++i) {
        const int e0 = m.rowptr[i], len = m.rowptr[i + 1] - e0;
        if (len <= 1) continue;
        int keys[32];
        for (int gq = 0; gq < G && gq < 32; ++gq) {
            const int tid = i + N * gq, lane = tid & 63;
            keys[gq] = (tid >> 6) * 4 + (lane >> 5) * 2 + kGrpOfLane[lane & 31];
        }
        auto cost = [&](const std::vector<int>& perm) {
            int c = 0;
            for (int e = 0; e < len; ++e) {
                const int col = col0[e0 + perm[e]], colour = col & 15;
                for (int gq = 0; gq < G && gq < 32; ++gq) {
                    const int u = used[((size_t)keys[gq] * maxlen + e) * 16 + colour];
                    if (u != -1 && u != col) ++c;
                }
            }
            return c;
        };
        std::vector<int> perm(len), bestp;
        for (int e = 0; e < len; ++e) perm[e] = e;
        int bc = 1 << 30;
        if (len <= 5) {
            do {
                const int c = cost(perm);
                if (c < bc) { bc = c; bestp = perm; }
            } while (bc > 0 && std::next_permutation(perm.begin(), perm.end()));
        } else {
            std::vector<int> p2 = perm;
            std::stable_sort(p2.begin(), p2.end(), [&](int a, int b) { return col0[e0 + a] - i < col0[e0 + b] - i; });
            for (int trial = 0; trial < 300 && bc > 0; ++trial) {
                const int c = cost(p2);
                if (c < bc) { bc = c; bestp = p2; }
                for (int e = len - 1; e > 0; --e) std::swap(p2[e], p2[rnd() % (e + 1)]);
            }
        }
        for (int e = 0; e < len; ++e) {
            m.col[e0 + e] = col0[e0 + bestp[e]];
            if (!m.src.empty()) m.src[e0 + e] = src0[e0 + bestp[e]];
            const int col = m.col[e0 + e], colour = col & 15;
            for (int gq = 0; gq < G && gq < 32; ++gq) {
                int& u = used[((size_t)keys[gq] * maxlen + e) * 16 + colour];
                u = (u == -1 || u == col) ? col : -2;
            }
        }
    }
}

struct Result {
    double before = 0, after = 0;     // weighted conflict cycles of one ADMM iteration's gathers (+ stores)
    long moves = 0, accepted = 0;
};

// Seeded hill climb.  mats: the operators' tables (entry order is rewritten in place), pos: row positions (rewritten when
// `move_rows`).  A move swaps two entries of one row or the positions of two rows; it is kept when the weighted conflict
// count does not grow.  Only the waves that hold threads of the touched rows are replayed for an entry swap.
inline Result improve(const Geometry& q, std::vector<Mat>& mats, std::vector<int>& pos, bool move_rows, long budget, unsigned seed = 12345u) {
    const int nw = (q.N * q.G + 63) / 64;
    const int nm = (int)mats.size();
    std::vector<std::vector<long>> wc(nm, std::vector<long>(nw));
    auto total_of = [&](const std::vector<std::vector<long>>& t, long st) {
        double s = 0;
        for (int m = 0; m < nm; ++m) {
            long c = 0;
            for (int w = 0; w < nw; ++w) c += t[m][w];
            s += mats[m].weight * (double)c;
        }
        double wsum = 0;
        for (int m = 0; m < nm; ++m) wsum += mats[m].weight;
        return s + wsum * (double)st;        // every operator application stores one image
    };
    for (int m = 0; m < nm; ++m)
        for (int w = 0; w < nw; ++w) wc[m][w] = simulate_wave(q, mats[m], pos, w).conflicts;
    long stc = store_conflicts(q, pos);
    Result r;
    r.before = total_of(wc, stc);
    double cur = r.before;
    unsigned rng = seed;
    auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
    std::vector<int> waves_of_node;      // waves holding threads of node i: one per time group
    auto waves_for = [&](int i, std::vector<int>& out) {
        out.clear();
        for (int g = 0; g < q.G; ++g) {
            const int w = (g * q.N + i) / 64;
            if (std::find(out.begin(), out.end(), w) == out.end()) out.push_back(w);
        }
    };
    for (long it = 0; it < budget && cur > 0; ++it) {
        ++r.moves;
        if (move_rows && (rnd() % 4) == 0) {
            const int a = rnd() % q.N, b = rnd() % q.N;
            if (a == b) continue;
            std::swap(pos[a], pos[b]);
            std::vector<std::vector<long>> t(nm, std::vector<long>(nw));
            for (int m = 0; m < nm; ++m)
                for (int w = 0; w < nw; ++w) t[m][w] = simulate_wave(q, mats[m], pos, w).conflicts;
            const long st2 = store_conflicts(q, pos);
            const double c2 = total_of(t, st2);
            if (c2 <= cur) { cur = c2; wc.swap(t); stc = st2; ++r.accepted; }
            else std::swap(pos[a], pos[b]);
            continue;
        }
        const int m = rnd() % nm;
        Mat& M = mats[m];
        const int i = rnd() % q.N;
        const int e0 = M.rowptr[i], len = M.rowptr[i + 1] - e0;
        if (len < 2) continue;
        const int a = rnd() % len, b = rnd() % len;
        if (a == b) continue;
        // gather_lead reads entries of the NEXT rows past the end of a short row: the waves of the preceding rows see this row too
        std::vector<int> ws;
        waves_for(i, ws);
        if (M.stream == LEAD_PAIRS)
            for (int back = 1; back <= q.nlead && i - back >= 0; ++back) {
                std::vector<int> w2;
                waves_for(i - back, w2);
                for (int w : w2) if (std::find(ws.begin(), ws.end(), w) == ws.end()) ws.push_back(w);
            }
        auto swap_entries = [&]() {
            std::swap(M.col[e0 + a], M.col[e0 + b]);
            if (!M.src.empty()) std::swap(M.src[e0 + a], M.src[e0 + b]);
        };
        swap_entries();
        double delta = 0;
        std::vector<long> nv(ws.size());
        for (size_t k = 0; k < ws.size(); ++k) {
            nv[k] = simulate_wave(q, M, pos, ws[k]).conflicts;
            delta += M.weight * (double)(nv[k] - wc[m][ws[k]]);
        }
        if (delta <= 0) {
            for (size_t k = 0; k < ws.size(); ++k) wc[m][ws[k]] = nv[k];
            cur += delta;
            ++r.accepted;
        } else {
            swap_entries();
        }
    }
    r.after = cur;
    return r;
}

// Min-conflicts search over the entry order of ONE matrix (row positions fixed): take an entry that sits in a busiest slot of
// a conflicting lane group, try every other place of its row for it, keep the swap that lowers the conflict count most
// (ties included: plateau moves); stop when nothing conflicts or after `steps` tries.  Far fewer replays than the random
// climb above: cfg2 (N = 307, 4 300 entries) goes from 7 800 weighted conflict cycles (greedy order) to a few hundred in
// well under a second of host time.
inline Result improve_targeted(const Geometry& q, Mat& M, const std::vector<int>& pos, long steps, unsigned seed = 777u) {
    const int nw = (q.N * q.G + 63) / 64;
    std::vector<long> wc(nw);
    std::vector<std::pair<int, int>> cul;
    auto rescan = [&]() {
        cul.clear();
        long t = 0;
        for (int w = 0; w < nw; ++w) { wc[w] = simulate_wave(q, M, pos, w, &cul).conflicts; t += wc[w]; }
        return t;
    };
    Result r;
    long cur = rescan();
    r.before = (double)cur;
    unsigned rng = seed;
    auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
    auto waves_for = [&](int i, std::vector<int>& out) {
        for (int g = 0; g < q.G; ++g) {
            const int w = (g * q.N + i) / 64;
            if (std::find(out.begin(), out.end(), w) == out.end()) out.push_back(w);
        }
    };
    std::vector<int> ws;
    std::vector<long> nv, bestv;
    for (long it = 0; it < steps && cur > 0 && !cul.empty(); ++it) {
        ++r.moves;
        const auto pick = cul[rnd() % cul.size()];
        const int i = pick.first, a = pick.second;
        const int e0 = M.rowptr[i], len = M.rowptr[i + 1] - e0;
        if (len < 2 || a >= len) { if ((it & 63) == 63) cur = rescan(); continue; }
        ws.clear();
        waves_for(i, ws);
        if (M.stream == LEAD_PAIRS)
            for (int back = 1; back <= q.nlead && i - back >= 0; ++back) waves_for(i - back, ws);
        long base = 0;
        for (int w : ws) base += wc[w];
        long best = 1L << 60;
        int bestb = -1, nbest = 0;
        for (int b = 0; b < len; ++b) {
            if (b == a) continue;
            std::swap(M.col[e0 + a], M.col[e0 + b]);
            long c = 0;
            nv.resize(ws.size());
            for (size_t k = 0; k < ws.size(); ++k) { nv[k] = simulate_wave(q, M, pos, ws[k]).conflicts; c += nv[k]; }
            std::swap(M.col[e0 + a], M.col[e0 + b]);
            if (c < best) { best = c; bestb = b; bestv = nv; nbest = 1; }
            else if (c == best && (rnd() % ++nbest) == 0) { bestb = b; bestv = nv; }
        }
        if (bestb >= 0 && best <= base) {
            std::swap(M.col[e0 + a], M.col[e0 + bestb]);
            if (!M.src.empty()) std::swap(M.src[e0 + a], M.src[e0 + bestb]);
            for (size_t k = 0; k < ws.size(); ++k) wc[ws[k]] = bestv[k];
            cur += best - base;
            ++r.accepted;
        }
        if ((it & 15) == 15 || best < base) cur = rescan();        // refresh the culprit list
    }
    r.after = (double)rescan();
    return r;
}

}  // namespace ldsbank

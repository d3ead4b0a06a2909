// LDS-resident fused ADMM iteration for small graphs (PEMS-size: T*N*8 B + CSR fits the 160 KiB LDS).
//
// One workgroup = one sample.  One launch = one full ADMM iteration of reference ADMM.py:546-646:
// RHS_x, the three CG solves (x, zu, zd) with all their iterations, the dual updates, the phi prox and
// every residual / regulariser of the history.  Inside a CG solve nothing touches HBM:
//   * thread (g, i) owns node i at TPG consecutive time steps t0 = g*TPG ...; x, r, p, Ap of those
//     elements live in registers for the whole solve;
//   * a copy of the CG direction p (and of q = Ldr p) lives in LDS so that neighbours can be gathered, stored SHIFTED by
//     the time offset of the operator that reads it (LdsCtx::put), so that every gather is an aligned ds_read_b128 run --
//     the only LDS traffic of an iteration is those gathers and one store of p and q;
//   * the three CSR matrices (W_u, W_d, W_d^T as packed {col, weight} pairs) live in LDS too; a thread
//     reads each entry of ITS node's row once per operator application and applies it to its TPG time steps;
//   * p.Ap and r.r are reduced with wave shuffles + a 16-entry LDS exchange in fixed order (repeatable);
//     alpha, beta and the convergence test are computed redundantly by every thread (workgroup-uniform).
// State (x, zu, zd, phi, gamma*) is kept in HBM in the reference's own sample-major (B, T*N) layout, so the
// ABI tensors need no layout conversion on this path; per ADMM iteration a sample moves ~35 vectors of
// T*N*4 B through HBM instead of ~650.
#pragma once
#include "common.h"
#include "lds_args.h"



// wave64 sum with DPP row shifts / row broadcasts (no LDS traffic); the total ends up in lane 63 and is
// broadcast with readlane.  Fixed association order -> bitwise repeatable.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_step(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_step<0x111, 0xf>(v);   // row_shr:1
    v = dpp_step<0x112, 0xf>(v);   // row_shr:2
    v = dpp_step<0x114, 0xf>(v);   // row_shr:4
    v = dpp_step<0x118, 0xf>(v);   // row_shr:8   -> lane 15 of every row holds the row total
    v = dpp_step<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_step<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Fixed-order workgroup sum: per-wave float totals (DPP) -> 16 LDS slots -> every thread adds the 16 slots
// in a fixed float tree.  `red` holds 2 x 16 floats used alternately, so one barrier per call is enough.
struct BlockRed {
    float* red;
    int par;
    int lane, wave, nwaves;
    __device__ __forceinline__ double sum(float v) {
        v = wave_sum(v);
        float* buf = red + par * 16;
        if (lane == 0) buf[wave] = v;
        __syncthreads();
        const float4 a = reinterpret_cast<const float4*>(buf)[0], b = reinterpret_cast<const float4*>(buf)[1],
                     c = reinterpret_cast<const float4*>(buf)[2], d = reinterpret_cast<const float4*>(buf)[3];
        // fixed association: ((a+b)+(c+d)) component-wise, then (x+y)+(z+w)
        const float sx = (a.x + b.x) + (c.x + d.x), sy = (a.y + b.y) + (c.y + d.y);
        const float sz = (a.z + b.z) + (c.z + d.z), sw = (a.w + b.w) + (c.w + d.w);
        const double s = (double)((sx + sy) + (sz + sw));
        par ^= 1;
        return s;
    }
    // Same exchange for the CG dot products, float in / float out: after the barrier every lane reads ONE of the
    // 16 wave totals (slot lane & 15) and the 16 values are summed inside each 16-lane row with DPP row shifts
    // (all four rows compute the same sum in the same order); lane 15's value is broadcast.  8 VALU + 1 ds_read
    // instead of 15 VALU + 4 ds_read_b128 + two float<->double conversions per reduction.
    __device__ __forceinline__ float sumf(float v) {
#ifdef MGADMM_KO_NORED           // knock-out timing build: no workgroup reduction (and none of its barriers)
        return v + 1.0f;
#endif
        v = wave_sum(v);
        float* buf = red + par * 16;
        if (lane == 0) buf[wave] = v;      // (all 64 lanes storing the same word instead: +1.2 % per launch)
        __syncthreads();
        float t = buf[lane & 15];
        t = dpp_step<0x111, 0xf>(t);   // row_shr:1
        t = dpp_step<0x112, 0xf>(t);   // row_shr:2
        t = dpp_step<0x114, 0xf>(t);   // row_shr:4
        t = dpp_step<0x118, 0xf>(t);   // row_shr:8   -> lane 15 of every row: sum of the 16 slots
        par ^= 1;
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 15));
    }
};

// vector access to TPG consecutive floats in LDS (alignment: TPG*4 B when TPG is a multiple of 4, 8 B when
// even -- guaranteed by the node-major [N][T] LDS layout with TPG | T)
typedef float lds_f4 __attribute__((ext_vector_type(4)));
template <int TPG>
__device__ __forceinline__ void lds_load(const float* p, float (&v)[TPG]) {
    if constexpr (TPG % 4 == 0) {
#pragma unroll
        for (int j = 0; j < TPG / 4; ++j) {
            // native vector type + assume_aligned: a struct float4 load gets split into align-4 scalars and
            // re-fused as ds_read_b96/read2_b32 (32-bank forms); this stays one ds_read_b128
            const lds_f4 q = static_cast<const lds_f4*>(__builtin_assume_aligned(p, 16))[j];
            v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
        }
    } else if constexpr (TPG % 2 == 0) {
#pragma unroll
        for (int j = 0; j < TPG / 2; ++j) {
            const float2 q = reinterpret_cast<const float2*>(p)[j];
            v[2 * j] = q.x; v[2 * j + 1] = q.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < TPG; ++j) v[j] = p[j];
    }
}
template <int TPG>
__device__ __forceinline__ void lds_store(float* p, const float (&v)[TPG]) {
    if constexpr (TPG % 4 == 0) {
#pragma unroll
        for (int j = 0; j < TPG / 4; ++j) {
            // native vector type + assume_aligned, as in lds_load: a struct float4 store is split into scalars and
            // re-fused as ds_write2_b32 pairs
            lds_f4 q;
            q.x = v[4 * j]; q.y = v[4 * j + 1]; q.z = v[4 * j + 2]; q.w = v[4 * j + 3];
            static_cast<lds_f4*>(__builtin_assume_aligned(p, 16))[j] = q;
        }
    } else if constexpr (TPG % 2 == 0) {
#pragma unroll
        for (int j = 0; j < TPG / 2; ++j) reinterpret_cast<float2*>(p)[j] = make_float2(v[2 * j], v[2 * j + 1]);
    } else {
#pragma unroll
        for (int j = 0; j < TPG; ++j) p[j] = v[j];
    }
}

// Global load as uniform base + 32-bit unsigned byte offset: the SGPR-base form of global_load (one offset VGPR shared by
// every vector of a request) instead of a 64-bit address pair per access.
__device__ __forceinline__ float ldg(const float* base, unsigned boff) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + boff);
}
// knock-out timing build MGADMM_KO_OWNROW: every gather reads the thread's OWN row instead of the neighbour's (consecutive
// nodes -> distinct 16-byte slots: no bank conflict) -- same instruction stream, wrong results, tells what the conflicts cost
// MGADMM_KO_CIRC: entry number u of every row reads node (i + u) mod N -- distinct rows per entry, consecutive nodes per
// lane group: conflict-free under the bank model of lds_banks.h without the identical addresses of KO_OWNROW
#if defined(MGADMM_KO_OWNROW)
#define MG_ENX(x, u) (i * TS)
#elif defined(MGADMM_KO_CIRC)
#define MG_ENX(x, u) (((i + (u)) % N) * TS)
#else
#define MG_ENX(x, u) (x)
#endif
// REQUEST fence: the loads before it are issued together and waited for together, nothing crosses it.
#define MG_REQ_FENCE() asm volatile("" ::: "memory")

// Per-thread view: node i, time steps t0 .. t0+TPG-1.
//   LDS vectors are node-major, time innermost: A[node*TS + t] (row stride TS >= T, an odd number of 16-B
//   slots so that rows start on all banks) -> the TPG time steps of any node are one contiguous, aligned
//   run (vector ds_read), also for a gathered neighbour;
//   HBM state is the reference's (T, N) order per sample: element (t, i) at t*N + i (coalesced over i).
template <int TPG, bool BAND, int NU = 0, int ND = 0>
struct LdsCtx {
    int T, TS, N, t0, i;
    bool active;
    float* P;
    float* Q;
    const int2* en_u; const int2* en_d; const int2* en_t;
    int u0, u1, d0, d1, t0e, t1e;   // this node's CSR row bounds
    int skip, q1;
    const float* band_w;

    __device__ __forceinline__ int gl(int k) const { return (t0 + k) * N + i; }   // HBM index
    __device__ __forceinline__ unsigned glb(int k) const { return 4u * (unsigned)((t0 + k) * N + i); }   // ... as a byte offset
    __device__ __forceinline__ int own() const { return i * TS + t0; }            // LDS index of element k = 0

    // SHIFTED LDS IMAGES.  A vector that is only gathered through a time-shifted operator is stored shifted: the image
    // read by Ldr (gathers x[t-1]) holds time t-1 at index t, the image read by Ldr^T (gathers x[t+1]) holds time t+1 at
    // index t, 0 outside [0,T).  Every gather of a neighbour's TPG-step window is then the ALIGNED run [t0, t0+TPG-1]:
    // conflict-free ds_read_b128 only, no edge group, no rotation of the accumulators (round 1 read an aligned run plus
    // one more 16-byte group holding the edge element: 3 reads per entry at TPG = 8 instead of 2).  The price is the
    // store of the own elements, unaligned by one float (scalar ds_write_b32; stores are 1/15 of the LDS traffic).
    //
    // acc[k] = sum_e w_e * IMG[col_e][t0+k].  Entries carry the LDS float offset of the neighbour's row (col*TS,
    // precomputed on the host).  Two entries are in flight per trip (their LDS reads are issued together, the next pair
    // of entries is fetched meanwhile); the sum runs in entry order.
    __device__ __forceinline__ void gather(const float* SRC, const int2* EN, int e0, int e1, float (&acc)[TPG]) const {
#pragma unroll
        for (int k = 0; k < TPG; ++k) acc[k] = 0.f;
#ifdef MGADMM_KO_NOGATHER        // knock-out timing build (make EXTRA=-DMGADMM_KO_...): everything but the LDS gathers
        return;
#endif
        const float* base = SRC + t0;
        int2 na = EN[e0], nb = EN[e0 + 1];          // the arrays are padded by 3 entries: reads past e1 are safe
        int e = e0;
        for (; e + 1 < e1; e += 2) {
            const int2 ea = na, eb = nb;
            float va[TPG], vb[TPG];
            lds_load<TPG>(base + MG_ENX(ea.x, e - e0), va);
            lds_load<TPG>(base + MG_ENX(eb.x, e - e0 + 1), vb);
            na = EN[e + 2];
            nb = EN[e + 3];
            const float wa = __int_as_float(ea.y), wb = __int_as_float(eb.y);
#pragma unroll
            for (int k = 0; k < TPG; ++k) acc[k] += wa * va[k];
#pragma unroll
            for (int k = 0; k < TPG; ++k) acc[k] += wb * vb[k];
        }
        if (e < e1) {
            float va[TPG];
            lds_load<TPG>(base + MG_ENX(na.x, e - e0), va);
            const float wa = __int_as_float(na.y);
#pragma unroll
            for (int k = 0; k < TPG; ++k) acc[k] += wa * va[k];
        }
    }
    // The same for a matrix whose rows all hold exactly NFIX entries (W_u: k, W_d: k + 1 of a kNN table without pads):
    // straight-line code, every entry of the row is read up front, no per-lane trip count -- the divergent loop above
    // costs an exec-mask region and a register copy per accumulator and trip.
    // PRE != nullptr: the row's entries were read once before the CG loop and live in registers (they do not change
    // during a solve): no entry reads, and the neighbour reads no longer wait for them, inside the loop
    template <int NFIX>
    __device__ __forceinline__ void gather_fixed(const float* SRC, const int2* EN, int e0, float (&acc)[TPG], const int2* PRE = nullptr) const {
#pragma unroll
        for (int k = 0; k < TPG; ++k) acc[k] = 0.f;
#ifdef MGADMM_KO_NOGATHER
        return;
#endif
        const float* base = SRC + t0;
        int2 en[NFIX];
#pragma unroll
        for (int u = 0; u < NFIX; ++u) en[u] = PRE ? PRE[u] : EN[e0 + u];
#pragma unroll
        for (int u = 0; u + 1 < NFIX; u += 2) {
            float va[TPG], vb[TPG];
            lds_load<TPG>(base + MG_ENX(en[u].x, u), va);
            lds_load<TPG>(base + MG_ENX(en[u + 1].x, u + 1), vb);
            const float wa = __int_as_float(en[u].y), wb = __int_as_float(en[u + 1].y);
#pragma unroll
            for (int k = 0; k < TPG; ++k) acc[k] += wa * va[k];
#pragma unroll
            for (int k = 0; k < TPG; ++k) acc[k] += wb * vb[k];
        }
        if (NFIX & 1) {
            float va[TPG];
            lds_load<TPG>(base + MG_ENX(en[NFIX - 1].x, NFIX - 1), va);
            const float wa = __int_as_float(en[NFIX - 1].y);
#pragma unroll
            for (int k = 0; k < TPG; ++k) acc[k] += wa * va[k];
        }
    }
    // Ragged rows (W_d^T: the in-degree of a kNN graph varies): the first NLEAD entries are read up front and gathered
    // together like a fixed row -- entries past the end of the row belong to the next row (or to the padding of the
    // table: NLEAD entries), they are read but enter with weight 0 --, the rest of a long row in the paired loop of
    // `gather`.  The sum runs in entry order (a weight-0 term adds exactly 0).  One LDS latency chain for a typical row
    // instead of one per pair of entries.
    template <int NLEAD>
    __device__ __forceinline__ void gather_lead(const float* SRC, const int2* EN, int e0, int e1, float (&acc)[TPG], const int2* PRE = nullptr) const {
#pragma unroll
        for (int k = 0; k < TPG; ++k) acc[k] = 0.f;
#ifdef MGADMM_KO_NOGATHER
        return;
#endif
        const float* base = SRC + t0;
        const int len = e1 - e0;
        int2 en[NLEAD];
#pragma unroll
        for (int u = 0; u < NLEAD; ++u) en[u] = PRE ? PRE[u] : EN[e0 + u];
#pragma unroll
        for (int u = 0; u < NLEAD; u += 2) {
            float va[TPG], vb[TPG];
            lds_load<TPG>(base + MG_ENX(en[u].x, u), va);
            if (u + 1 < NLEAD) lds_load<TPG>(base + MG_ENX(en[u + 1].x, u + 1), vb);
            const float wa = u < len ? __int_as_float(en[u].y) : 0.f;
#pragma unroll
            for (int k = 0; k < TPG; ++k) acc[k] += wa * va[k];
            if (u + 1 < NLEAD) {
                const float wb = u + 1 < len ? __int_as_float(en[u + 1].y) : 0.f;
#pragma unroll
                for (int k = 0; k < TPG; ++k) acc[k] += wb * vb[k];
            }
        }
        // (knock-out timing build MGADMM_KO_NOTAIL: rows cut after the leading entries -- this tail, a tenth of the gathers, costs a
        // fifth of the cfg2 launch: every wave holds a 9- or 10-entry row and runs three dependent trips.  Measured and not
        // kept: a second batch of four entries gathered like the leading ones -- 220 B of scratch, +18 %; the first pair of
        // entries requested ahead of the leading rows -- no change)
#ifdef MGADMM_KO_NOTAIL
        if (false) {
#else
        if (len > NLEAD) {
#endif
            int e = e0 + NLEAD;
            int2 na = EN[e], nb = EN[e + 1];
            for (; e + 1 < e1; e += 2) {
                const int2 ea = na, eb = nb;
                float va[TPG], vb[TPG];
                lds_load<TPG>(base + MG_ENX(ea.x, e - e0), va);
                lds_load<TPG>(base + MG_ENX(eb.x, e - e0 + 1), vb);
                na = EN[e + 2];
                nb = EN[e + 3];
                const float wa = __int_as_float(ea.y), wb = __int_as_float(eb.y);
#pragma unroll
                for (int k = 0; k < TPG; ++k) acc[k] += wa * va[k];
#pragma unroll
                for (int k = 0; k < TPG; ++k) acc[k] += wb * vb[k];
            }
            if (e < e1) {
                float va[TPG];
                lds_load<TPG>(base + MG_ENX(na.x, e - e0), va);
                const float wa = __int_as_float(na.y);
#pragma unroll
                for (int k = 0; k < TPG; ++k) acc[k] += wa * va[k];
            }
        }
    }
    // band (line-graph) stencils on the node's own time row
    __device__ __forceinline__ void band_back(const float* SRC, float (&acc)[TPG]) const {
        const float* row = SRC + i * TS;
#pragma unroll
        for (int k = 0; k < TPG; ++k) {
            acc[k] = 0.f;
            const int t = t0 + k;
            for (int s = 0; s < skip; ++s) {
                const int tt = t - 1 - s;
                if (tt < 0) break;
                acc[k] += band_w[t * skip + s] * row[tt];
            }
        }
    }
    __device__ __forceinline__ void band_fwd(const float* SRC, float (&acc)[TPG]) const {
        const float* row = SRC + i * TS;
#pragma unroll
        for (int k = 0; k < TPG; ++k) {
            acc[k] = 0.f;
            const int t = t0 + k;
            for (int s = 0; s < skip; ++s) {
                const int tt = t + 1 + s;
                if (tt >= T) break;
                acc[k] += band_w[tt * skip + s] * row[tt];
            }
        }
    }
    // l = Lu(src): neighbours from SRC (LDS), the thread's own elements of src from registers (self)     ADMM.py:138-148
    __device__ __forceinline__ void op_lu(const float* SRC, const float (&self)[TPG], float (&l)[TPG], const int2* PRE = nullptr) const {
        float acc[TPG];
        if constexpr (NU > 0) gather_fixed<NU>(SRC, en_u, u0, acc, PRE);
        else gather(SRC, en_u, u0, u1, acc);
#pragma unroll
        for (int k = 0; k < TPG; ++k) l[k] = self[k] - acc[k];
    }
    // l = Ldr(src)      ADMM.py:150-177
    __device__ __forceinline__ void op_ldr(const float* SRC, const float (&self)[TPG], float (&l)[TPG], const int2* PRE = nullptr) const {
        float acc[TPG];
        if constexpr (BAND) band_back(SRC, acc);
        else if constexpr (ND > 0) gather_fixed<ND>(SRC, en_d, d0, acc, PRE);       // SRC: image stored with put<-1>
        else gather(SRC, en_d, d0, d1, acc);
#pragma unroll
        for (int k = 0; k < TPG; ++k) l[k] = ((k >= 1 || t0 >= 1) ? self[k] : 0.f) - acc[k];
    }
    // l = Ldr_T(src)    ADMM.py:179-223 (q1: identity kept on the t=0 block)
    __device__ __forceinline__ void op_ldrt(const float* SRC, const float (&self)[TPG], float (&l)[TPG], const int2* PRE = nullptr) const {
        float acc[TPG];
        if constexpr (!BAND) gather_lead<LDS_NLEAD>(SRC, en_t, t0e, t1e, acc, PRE);     // SRC: image stored with put<+1>
        else band_fwd(SRC, acc);
#pragma unroll
        for (int k = 0; k < TPG; ++k) l[k] = ((k > 0 || t0 > 0 || q1) ? self[k] : 0.f) - acc[k];
    }
    // own elements -> LDS image read by an operator that gathers time t + SH (SH = 0: Lu and the band stencils, -1: Ldr,
    // +1: Ldr^T): index j of a node's row holds time j + SH, 0 outside [0,T)
    // EDGE = false: the slot that holds time -1 resp. T (always 0) is left alone -- within one CG solve it is written
    // once, with the first image, and no other store touches it
    template <int SH, bool EDGE = true>
    __device__ __forceinline__ void put(float* DST, const float (&v)[TPG]) const {
        if (!active) return;
        if constexpr (SH == 0 || BAND) {
            lds_store<TPG>(DST + own(), v);
        } else {
            float* row = DST + i * TS + t0 - SH;            // index of element k = 0
            // the 16-byte groups of the image that lie entirely inside this thread's elements go out as one aligned
            // ds_write_b128 each (SH = -1: v[3..6], v[7..10], ...; SH = +1: v[1..4], v[5..8], ...); the elements at the two
            // ends are scalar stores (rows are 16-byte aligned, so scalar stores of 32 consecutive nodes hit 8 banks: 4-way)
            constexpr int K0 = SH < 0 ? 3 : 1;              // first element of the first whole group
            constexpr int NG = TPG % 4 == 0 ? (TPG - K0) / 4 : 0;
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                lds_f4 q;
                q.x = v[K0 + 4 * j]; q.y = v[K0 + 4 * j + 1]; q.z = v[K0 + 4 * j + 2]; q.w = v[K0 + 4 * j + 3];
                *static_cast<lds_f4*>(__builtin_assume_aligned(row + K0 + 4 * j, 16)) = q;
            }
            // (the aligned pair among the remaining elements as one ds_write_b64 instead of the ds_write2_b32 the compiler
            // forms: +2.3 % per launch, not kept)
#pragma unroll
            for (int k = 0; k < TPG; ++k) {
                if (k >= K0 && k < K0 + 4 * NG) continue;
                const bool ok = SH < 0 ? (k < TPG - 1 || t0 + TPG < T) : (k > 0 || t0 > 0);   // index T resp. -1 does not exist
                if (ok) row[k] = v[k];
            }
            if (EDGE && SH < 0 && t0 == 0) DST[i * TS] = 0.f;                      // time -1
            if (EDGE && SH > 0 && t0 + TPG == T) DST[i * TS + T - 1] = 0.f;        // time T
        }
    }
    // own elements -> HBM state vector (sample base already applied)
    __device__ __forceinline__ void putg(float* DST, const float (&v)[TPG]) const {
        if (active) {
#pragma unroll
            for (int k = 0; k < TPG; ++k) DST[gl(k)] = v[k];
        }
    }
};

// av = (A v) on the thread's own elements, for the vector v held in ctx.P (LDS; own elements also in v):
//   KIND 1: dc*v + c2*Ldr_T(Ldr v) ; KIND 2: dc*v + c2*Lu v ; KIND 0: dc*v
// dc = diagonal coefficient of the own elements (H^T H or mask value, plus the rho/2 terms), see lds_diag.
// Returns sum_k v_k * (A v)_k of the own elements.  Uses ctx.Q as scratch; contains a barrier for KIND 1.
// Callers separate successive calls by barriers.
template <int TPG, bool BAND, int KIND, bool SB, int NU, int ND, bool EDGE = true>
__device__ __forceinline__ float lds_apply(const LdsCtx<TPG, BAND, NU, ND>& c, const float (&v)[TPG], float (&av)[TPG], const float (&dc)[TPG],
                                           float c2, const int2* PRE = nullptr, const int2* PRET = nullptr) {
    float l[TPG];
#pragma unroll
    for (int k = 0; k < TPG; ++k) l[k] = 0.f;
    if (KIND == 1) {
        if constexpr (BAND) {
            float q[TPG];
#pragma unroll
            for (int k = 0; k < TPG; ++k) q[k] = 0.f;
            if (c.active) c.op_ldr(c.P, v, q, PRE);
            if (SB) __syncthreads();           // single LDS vector (Q aliases P): every gather of p is done before q replaces it
            c.template put<+1, EDGE || SB>(c.Q, q);      // SB: p and q alternate in one vector, their zero slots included
            __syncthreads();
            if (c.active) c.op_ldrt(c.Q, q, l, PRET);
        } else {
            // SHIFTED OWNERSHIP of q = Ldr v.  Ldr reads time t-1 and Ldr^T time t+1, so with one partition of the time axis for
            // both vectors either the store or the gather of an image is off by one float against the 16-byte groups (until
            // mid round 2 the images were stored shifted: one ds_write_b128 + four scalar ds_write_b32 per store, the scalar
            // stores of 32 consecutive nodes on 8 banks).  Here the thread that owns v at times t0 .. t0+TPG-1 computes and owns
            // q at times t0+1 .. t0+TPG: its gather for q[t0+1+k] is the aligned run v[col][t0+k] of the UNSHIFTED image of v,
            // it stores its q values as the aligned run at positions t0 .. (position j of the q image holds q[j+1]; time T
            // does not exist: 0), and the Ldr^T gather for time t0+k reads q[col][t0+k+1] = that aligned run again.  Both
            // images go out as whole ds_write_b128, every gather is an aligned run; the price is one scalar read per
            // operator for the self term that sits in the neighbouring time group (v[t0+TPG] resp. q[t0]).  Same products in
            // the same order: bitwise the former results.
            float q[TPG];        // q[k] = (Ldr v)[t0 + k + 1]
#pragma unroll
            for (int k = 0; k < TPG; ++k) q[k] = 0.f;
            if (c.active) {
                float acc[TPG];
                if constexpr (ND > 0) c.template gather_fixed<ND>(c.P, c.en_d, c.d0, acc, PRE);
                else c.gather(c.P, c.en_d, c.d0, c.d1, acc);
                const bool has_next = c.t0 + TPG < c.T;
                // v at the first time of the next group, own node: taken from its aligned 16-byte group (a scalar read of the
                // same column of 32 consecutive rows is a 4-way bank conflict, the ds_read_b128 of consecutive rows none)
                float vnext = 0.f;
                if (has_next) {
                    if constexpr (TPG % 4 == 0) vnext = static_cast<const lds_f4*>(__builtin_assume_aligned(c.P + c.own() + TPG, 16))[0].x;
                    else vnext = c.P[c.own() + TPG];
                }
#pragma unroll
                for (int k = 0; k + 1 < TPG; ++k) q[k] = v[k + 1] - acc[k];
                q[TPG - 1] = has_next ? vnext - acc[TPG - 1] : 0.f;
            }
            if (SB) __syncthreads();           // single LDS vector (Q aliases P): every gather of p is done before q replaces it
            if (c.active) lds_store<TPG>(c.Q + c.own(), q);
            __syncthreads();
            if (c.active) {
                float acc[TPG];
                c.template gather_lead<LDS_NLEAD>(c.Q, c.en_t, c.t0e, c.t1e, acc, PRET);
                float qprev = 0.f;          // q[t0]: the last value of the previous group (q[0] = 0), from its aligned 16-byte group
                if (c.t0 > 0) {
                    if constexpr (TPG % 4 == 0) qprev = static_cast<const lds_f4*>(__builtin_assume_aligned(c.Q + c.own() - 4, 16))[0].w;
                    else qprev = c.Q[c.own() - 1];
                }
                l[0] = qprev - acc[0];
#pragma unroll
                for (int k = 1; k < TPG; ++k) l[k] = q[k - 1] - acc[k];
            }
        }
    } else if (KIND == 2) {
        if (c.active) c.op_lu(c.P, v, l, PRE);
    }
    float part = 0.f;
#pragma unroll
    for (int k = 0; k < TPG; ++k) {
        av[k] = (KIND == 0) ? dc[k] * v[k] : dc[k] * v[k] + c2 * l[k];
        part += v[k] * av[k];
    }
    return part;
}
// diagonal coefficient d + c1 of the own elements: d = dg[el] when dg != nullptr (mask values, global
// memory), else [hth && t < t_in]   (ADMM.py:371-379: H^T H x resp. mask * x; ADMM.py:381-399: none)
template <int TPG, bool BAND, int NU, int ND>
__device__ __forceinline__ void lds_diag(const LdsCtx<TPG, BAND, NU, ND>& c, const float* dg, int hth, int t_in, float c1, float (&dc)[TPG]) {
#pragma unroll
    for (int k = 0; k < TPG; ++k) {
        const float d = dg ? (c.active ? dg[c.gl(k)] : 0.f) : ((hth && c.t0 + k < t_in) ? 1.f : 0.f);
        dc[k] = d + c1;
    }
}

// CG_solver (ADMM.py:329-368) for one sample.  x, r of the own elements live in registers, the direction
// p in registers with a copy in LDS (ctx.P) for the neighbours' gathers, A p in registers.  x holds x0 on entry and the solution on exit.  dmask: diagonal
// of the initial residual when a mask is given (global memory); the iterations always use [t<t_in]
// (quirk Q2).  Returns the iteration count (k+1) or -1.  Entry requirement: no thread still reads P/Q.
template <int TPG, bool BAND, int KIND, bool SB, int NU, int ND>
__device__ __forceinline__ int lds_cg(const LdsCtx<TPG, BAND, NU, ND>& c, BlockRed& br, float (&x)[TPG], const float (&rhs)[TPG], const float* dmask,
                      int hth, int t_in, float c1, float c2, int max_cg, double tol, float* ah, float* bh, int Bp,
                      int* nonfinite) {
    constexpr int SHP = 0;     // the image of the CG direction is stored unshifted for every operator (lds_apply: shifted ownership of q)
    float r[TPG], pv[TPG], av[TPG], dc[TPG];
    c.template put<SHP>(c.P, x);
    __syncthreads();
    lds_diag<TPG, BAND, NU, ND>(c, dmask, hth, t_in, c1, dc);
    (void)lds_apply<TPG, BAND, KIND, SB, NU, ND>(c, x, av, dc, c2);
    if (dmask != nullptr) lds_diag<TPG, BAND, NU, ND>(c, nullptr, hth, t_in, c1, dc);     // quirk Q2: iterations use [t < t_in]
    float part = 0.f;
#pragma unroll
    for (int k = 0; k < TPG; ++k) {
        r[k] = c.active ? rhs[k] - av[k] : 0.f;
        pv[k] = r[k];                    // p = r
        part += r[k] * r[k];
    }
    float rr = br.sumf(part);            // barrier: every read of P (= x0) is done
    c.template put<SHP, SB>(c.P, pv);
    // entries of this thread's fixed-length row (W_d for the cLdr solves, W_u for the zu solve), once per solve
    constexpr int NPRE = (KIND == 1 && !BAND && ND > 0) ? ND : ((KIND == 2 && NU > 0) ? NU : 0);
    int2 pre[NPRE > 0 ? NPRE : 1];
    if constexpr (NPRE > 0) {
        const int2* EN = KIND == 1 ? c.en_d : c.en_u;
        const int e0 = KIND == 1 ? c.d0 : c.u0;
#pragma unroll
        for (int u = 0; u < NPRE; ++u) pre[u] = EN[e0 + u];
    }
    constexpr int NPRET = (KIND == 1 && !BAND) ? LDS_NLEAD : 0;      // ... and the leading entries of its W_d^T row
    int2 pret[NPRET > 0 ? NPRET : 1];
    if constexpr (NPRET > 0) {
#pragma unroll
        for (int u = 0; u < NPRET; ++u) pret[u] = c.en_t[c.t0e + u];
    }
    int iters = -1;
    for (int it = 0; it < max_cg; ++it) {
        __syncthreads();                 // p complete in LDS
        part = lds_apply<TPG, BAND, KIND, SB, NU, ND, false>(c, pv, av, dc, c2, NPRE > 0 ? pre : nullptr, NPRET > 0 ? pret : nullptr);
        const float pAp = br.sumf(part);         // barrier: every gather from P/Q of this iteration is done
#ifdef MGADMM_KO_FIXED           // knock-out timing builds run a fixed number of iterations on made-up coefficients
        const float alpha = 1e-3f + 0.f * pAp;
#else
        // alpha, beta through v_rcp_f32 (1 ulp) instead of the correctly rounded division (a chain of ~10 dependent
        // instructions every thread waits for, twice per iteration: -2.9 % per launch).  The coefficients differ from
        // the IEEE quotient by at most 1.5 ulp -- below the rounding noise of the dot products they are formed from;
        // the streaming path keeps the IEEE division (its coefficients come from a separate tiny kernel).
        const float alpha = rr * __builtin_amdgcn_rcpf(pAp);
#endif
        part = 0.f;
#pragma unroll
        for (int k = 0; k < TPG; ++k) {
            x[k] = x[k] + alpha * pv[k];
            r[k] = r[k] - alpha * av[k];
            part += r[k] * r[k];
        }
        const float rrn = br.sumf(part);
#ifdef MGADMM_KO_FIXED
        const float beta = 0.5f + 0.f * rrn;
#else
        const float beta = rrn * __builtin_amdgcn_rcpf(rr);
#endif
        rr = rrn;
        if (ah != nullptr && threadIdx.x == 0) {
            ah[(size_t)it * Bp] = alpha;
            bh[(size_t)it * Bp] = beta;
        }
#ifdef MGADMM_KO_FIXED
#ifndef MGADMM_KO_ITERS
#define MGADMM_KO_ITERS 0
#endif
        if (it + 1 == (MGADMM_KO_ITERS ? MGADMM_KO_ITERS : (KIND == 2 ? 11 : 17))) { iters = it + 1; break; }
#else
        if (!(fabsf(rrn) <= 3.0e38f)) {       // NaN / Inf: report and stop this sample
            if (threadIdx.x == 0) *nonfinite = 1;
            break;
        }
        if ((double)sqrtf(rrn) < tol) {
            iters = it + 1;
            break;
        }
#endif
#pragma unroll
        for (int k = 0; k < TPG; ++k) pv[k] = r[k] + beta * pv[k];
        c.template put<SHP, SB>(c.P, pv);
    }
    return iters;
}

// MAXT: workgroup-size class the kernel is compiled for (register budget).  SB: single LDS vector (q = Ldr p replaces p
// in place, one more barrier per cLdr application); with the 640-thread class it is compiled for TWO resident
// workgroups per CU (5 waves per SIMD, <= 96 VGPRs): two samples in flight per CU overlap each other's barriers.
// NU / ND > 0: every row of W_u / W_d holds exactly that many entries (a kNN table without pads): unrolled gathers.
template <int TPG, bool BAND, int MAXT, bool SB, int NU = 0, int ND = 0>
__global__ __launch_bounds__(MAXT, (SB && MAXT == 640) ? 5 : 1) void k_admm_lds(LdsArgs a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    float* P = reinterpret_cast<float*>(lds_raw);
    const int LN = a.N * a.TS;                                         // floats per LDS vector (TS % 4 == 0 or TS == T)
    float* Q = SB ? P : P + LN;                                      // SB: one LDS vector serves p and q = Ldr p in turn
    float* red = Q + LN + ((4 - (LN & 3)) & 3);                       // 16-byte aligned, 2 x 16 floats
    int* csr = reinterpret_cast<int*>(red + 32);
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    // ADMM outer loop without host round trips: iterations are enqueued ahead of the host's look at the stop test, a launch
    // that follows the stopping iteration must leave the state alone (workgroup-uniform scalar load)
    if (a.stop != nullptr && *a.stop != 0) return;
    // diagnostic build (make EXTRA=-DMGADMM_PHASE_CLOCK): the per-sample metric slots receive the 100 MHz clock at the
    // phase boundaries instead of the metrics (tools/lds_phase_clock.py)
    auto stamp = [&](int m) {
#ifdef MGADMM_PHASE_CLOCK
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long tck = wall_clock64();
        if (tid == 0) a.ps[(size_t)m * a.Bp + b] = (double)tck;
#endif
    };
    // MGADMM_PHASE_CLOCK=2: slots 8..10 split the "batch 1 + RHS_x" phase instead of timing the last three phases
    auto stamp_late = [&](int m) {
#if defined(MGADMM_PHASE_CLOCK) && MGADMM_PHASE_CLOCK == 1
        stamp(m);
#endif
    };
    auto stamp_rhs = [&](int m) {
#if defined(MGADMM_PHASE_CLOCK) && MGADMM_PHASE_CLOCK == 2
        stamp(m);
#endif
    };
    // STAGGERED START.  Samples of one batch need (nearly) the same CG iteration counts, so the workgroups of a launch
    // run in lock step: all CUs store their results and request the next sample's operands at the same moment, while HBM
    // idles during the CG solves.  The workgroups of the first round (one per CU) start spread over `stagger_ticks`,
    // their successors inherit the offset: the operand requests of different CUs no longer coincide (cfg2: 48 us of
    // spread, -1.3 % per launch although the launch itself gets 48 us longer).
    if (a.stagger_ticks > 0 && b < a.stagger_wgs) {
        const unsigned long long wait = (unsigned long long)a.stagger_ticks * (unsigned)b / (unsigned)a.stagger_wgs;
        const unsigned long long t_in0 = wall_clock64();
        while (wall_clock64() - t_in0 < wait) __builtin_amdgcn_s_sleep(16);
    }
    stamp(0);
    LdsCtx<TPG, BAND, NU, ND> c;
    c.T = a.T; c.TS = a.TS; c.N = a.N;
    c.active = tid < a.nthreads;
    const int g = c.active ? tid / a.N : 0;
    c.i = c.active ? tid - g * a.N : 0;
    c.t0 = g * TPG;
    c.P = P; c.Q = Q;
    c.skip = a.skip; c.q1 = a.q1; c.band_w = a.band_w;
    c.en_u = reinterpret_cast<const int2*>(csr + a.off_en_u);
    c.en_d = reinterpret_cast<const int2*>(csr + a.off_en_d);
    c.en_t = reinterpret_cast<const int2*>(csr + a.off_en_t);

    const size_t sb = (size_t)b * a.TN;
    const float* xo = a.x_old + sb;
    float* xn = a.x_new + sb;
    float *zu = a.zu + sb, *zd = a.zd + sb, *phi = a.phi + sb, *gam = a.gam + sb, *gu = a.gu + sb, *gd = a.gd + sb;
    const float* mk = a.mask ? a.mask + sb : nullptr;
    const int ty = a.mask ? a.T : a.t_in;
    const float* yb = a.y + (size_t)b * ty * a.N;

    // REQUEST of two operand vectors (TPG elements each per thread): 2 * TPG unconditional loads issued together and
    // waited for together; nothing crosses the fence, so the destination registers of one request are all a batch needs.
    auto request2 = [&](const float* A, const float* B, const unsigned (&oa)[TPG], const unsigned (&ob)[TPG], float (&va)[TPG], float (&vb)[TPG]) {
#pragma unroll
        for (int k = 0; k < TPG; ++k) { va[k] = ldg(A, oa[k]); vb[k] = ldg(B, ob[k]); }
        MG_REQ_FENCE();
    };

    // ---- batch 1: x_old and every operand of RHS_x (ADMM.py:556-564), requested FIRST, before the CSR image is copied:
    // nothing else is live yet, so a request keeps its destination registers without spilling.
    // The operands are read in REQUESTS of two vectors whose loads are unconditional and branch-free: an operand the
    // ablation does not use is read through a pointer to a vector that exists, rows of y past t_in through a clamped
    // index, idle threads read node 0, and the values are selected afterwards.  With the loads inside `if (flag)` regions
    // (round 1 .. mid round 2) the compiler waited for every group of one or two loads: 40 dependent memory round trips,
    // 19 us + 2 us for the CSR copy in the instrumented build (tools/lds_phase_clock.py), against 9 us for both now; a
    // workgroup alone pulls the 221 KB in 6 us (tools/probe_batch_load.hip).  Production build: -1.9 % per launch.
    float x[TPG], o[TPG], v[TPG];
    {
        const bool use_v = a.has_phi && !a.first;
        const float* zdp = a.has_zd ? zd : zu;
        const float* gdp = a.has_zd ? gd : gu;
        const float* gamp = use_v ? gam : gu;
        const float* phip = use_v ? phi : zu;
        unsigned off[TPG], offy[TPG];
#pragma unroll
        for (int k = 0; k < TPG; ++k) {
            const int t = c.t0 + k;
            off[k] = c.glb(k);
            offy[k] = 4u * (unsigned)((t < ty ? t : ty - 1) * a.N + c.i);
        }
        float ta[TPG], tb[TPG];
        request2(zu, zdp, off, off, ta, tb);
#pragma unroll
        for (int k = 0; k < TPG; ++k) o[k] = a.has_zd ? (a.rho_u * ta[k] + a.rho_d * tb[k]) / 2.f : a.rho_u * ta[k] / 2.f;
        request2(gu, gdp, off, off, ta, tb);
#pragma unroll
        for (int k = 0; k < TPG; ++k) o[k] = o[k] - (a.has_zd ? (ta[k] + tb[k]) / 2.f : ta[k] / 2.f);
        request2(yb, xo, offy, off, ta, tb);
#pragma unroll
        for (int k = 0; k < TPG; ++k) {
            o[k] = c.active ? o[k] + ((c.t0 + k < ty) ? ta[k] : 0.f) : 0.f;
            x[k] = c.active ? tb[k] : 0.f;
        }
        request2(gamp, phip, off, off, ta, tb);
#pragma unroll
        for (int k = 0; k < TPG; ++k) v[k] = (c.active && use_v) ? ta[k] + a.rho * tb[k] : 0.f;
    }
    stamp_rhs(8);
    // CSR image -> LDS, eight words per thread in flight (one word per trip cost ten dependent L2 round trips, 3.6 us)
    for (int k0 = tid; k0 < a.csr_ints; k0 += 8 * (int)blockDim.x) {
        int wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + u * (int)blockDim.x;
            wv[u] = a.csr[k < a.csr_ints ? k : a.csr_ints - 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + u * (int)blockDim.x;
            if (k < a.csr_ints) csr[k] = wv[u];
        }
    }
    if (tid < 32) red[tid] = 0.f;          // slots of non-existent waves must read as 0
    __syncthreads();
    stamp(1);
    c.u0 = csr[a.off_rp_u + c.i]; c.u1 = csr[a.off_rp_u + c.i + 1];
    c.d0 = c.d1 = c.t0e = c.t1e = 0;
    if (!BAND) {
        c.d0 = csr[a.off_rp_d + c.i]; c.d1 = csr[a.off_rp_d + c.i + 1];
        c.t0e = csr[a.off_rp_t + c.i]; c.t1e = csr[a.off_rp_t + c.i + 1];
    }
    BlockRed br;
    br.red = red; br.par = 0; br.lane = tid & 63; br.wave = tid >> 6; br.nwaves = (blockDim.x + 63) >> 6;
    // per-sample metric sums are reduced and stored as soon as they are known (keeps no accumulator alive
    // across the CG solves); the whole-batch values are formed by k_batch_metrics
    auto emit = [&](int m, double v, bool keep) {
        const double sres = br.sum((float)v);
#ifdef MGADMM_PHASE_CLOCK
        if (tid == 0 && sres == 123.456) a.ps[(size_t)m * a.Bp + b] = sres;
#else
        if (tid == 0) a.ps[(size_t)m * a.Bp + b] = keep ? sres : 0.0;
#endif
    };

    // ---- first iteration only: phi = Ldr x0 (ADMM.py:541); the dual variables were filled by k_init_lds
    if (a.has_phi && a.first) {
        float ph[TPG];
#pragma unroll
        for (int k = 0; k < TPG; ++k) ph[k] = 0.f;
        c.template put<-1>(P, x);
        __syncthreads();
        if (c.active) c.op_ldr(P, x, ph);
        c.putg(phi, ph);
#pragma unroll
        for (int k = 0; k < TPG; ++k) v[k] = c.active ? gam[c.gl(k)] + a.rho * ph[k] : 0.f;
        __syncthreads();
    }

    // ---- RHS_x
    float rhs[TPG];
    {
        float l[TPG];
#pragma unroll
        for (int k = 0; k < TPG; ++k) l[k] = 0.f;
        if (a.has_phi) {
            c.template put<+1>(P, v);
            __syncthreads();
            stamp_rhs(9);
            if (c.active) c.op_ldrt(P, v, l);
            __syncthreads();
            stamp_rhs(10);
        }
#pragma unroll
        for (int k = 0; k < TPG; ++k) rhs[k] = c.active ? (a.has_phi ? l[k] / 2.f + o[k] : o[k]) : 0.f;
    }
    float* ah = a.record ? a.alpha_hist + b : nullptr;
    float* bh = a.record ? a.beta_hist + b : nullptr;
    const size_t hstride = (size_t)a.max_cg * a.Bp;

    stamp(2);
    // ---- x solve (ADMM.py:571)
    int itx;
    if (a.lhsx_kind == 1) itx = lds_cg<TPG, BAND, 1, SB, NU, ND>(c, br, x, rhs, mk, 1, a.t_in, a.cx1, a.cx2, a.max_cg, a.cg_tol, ah, bh, a.Bp, a.nonfinite);
    else itx = lds_cg<TPG, BAND, 0, SB, NU, ND>(c, br, x, rhs, mk, 1, a.t_in, a.cx1, 0.f, a.max_cg, a.cg_tol, ah, bh, a.Bp, a.nonfinite);
    stamp(3);
    c.putg(xn, x);

    // ---- batch 2: operands of the x metrics and of the zu solve
    float z[TPG], gq[TPG];
    {
        float xold[TPG], yv[TPG], mv[TPG];
#pragma unroll
        for (int k = 0; k < TPG; ++k) {
            xold[k] = yv[k] = mv[k] = z[k] = gq[k] = 0.f;
            if (c.active) {
                const int e = c.gl(k);
                const int t = c.t0 + k;
                xold[k] = xo[e];
                if (mk) { mv[k] = mk[e]; yv[k] = yb[t * a.N + c.i]; }
                else if (t < a.t_in) yv[k] = yb[t * a.N + c.i];
                z[k] = zu[e];
                gq[k] = gu[e];
            }
        }
        double m_xshift = 0, m_rec = 0;
        if (c.active) {
#pragma unroll
            for (int k = 0; k < TPG; ++k) {
                const int t = c.t0 + k;
                const double dx = (double)x[k] - (double)xold[k];
                m_xshift += dx * dx;
                if (mk) {
                    const double e = (double)(x[k] * mv[k] - yv[k]);
                    m_rec += e * e;
                } else if (t < a.t_in) {
                    const double e = (double)(x[k] - yv[k]);
                    m_rec += e * e;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < TPG; ++k) rhs[k] = c.active ? gq[k] / 2.f + a.rho_u / 2.f * x[k] : 0.f;     // x = the x_new just stored
        emit(MGADMM_M_XSHIFT, m_xshift, true);
        emit(MGADMM_M_RECOVER, m_rec, true);
    }

    stamp(4);
    // ---- zu solve + gamma_u update (ADMM.py:579-580, 595)
    int itzu = lds_cg<TPG, BAND, 2, SB, NU, ND>(c, br, z, rhs, nullptr, 0, 0, a.rho_u / 2.f, a.mu_u, a.max_cg, a.cg_tol, ah ? ah + hstride : nullptr,
                                        bh ? bh + hstride : nullptr, a.Bp, a.nonfinite);
    stamp(5);
    // ---- batch 3: x_new, old zu, gamma_u for the update + the operands of the next phase (zd solve, or the phi prox)
    float xr[TPG], zn[TPG], gn[TPG];
    {
        float zo[TPG], gv[TPG];
#pragma unroll
        for (int k = 0; k < TPG; ++k) {
            xr[k] = zn[k] = gn[k] = zo[k] = gv[k] = 0.f;
            if (c.active) {
                const int e = c.gl(k);
                xr[k] = xn[e];
                zo[k] = zu[e];
                gv[k] = gu[e];
                if (a.has_zd) { zn[k] = zd[e]; gn[k] = gd[e]; }
                else if (a.has_phi) { zn[k] = phi[e]; gn[k] = gam[e]; }
            }
        }
        double m_prizu = 0, m_dualzu = 0;
        if (c.active) {
#pragma unroll
            for (int k = 0; k < TPG; ++k) {
                const int e = c.gl(k);
                const float pz = xr[k] - z[k], dz = z[k] - zo[k];
                m_prizu += (double)pz * pz;
                m_dualzu += (double)dz * dz;
                zu[e] = z[k];
                gu[e] = gv[k] + a.rho_u * pz;
            }
        }
        emit(MGADMM_M_PRI_ZU, m_prizu, true);
        emit(MGADMM_M_DUAL_ZU, m_dualzu, true);
    }
    // ---- zd solve + gamma_d update (ADMM.py:586-588, 597)
    int itzd = 0;
    if (a.has_zd) {
#pragma unroll
        for (int k = 0; k < TPG; ++k) {
            z[k] = zn[k];
            rhs[k] = c.active ? gn[k] / 2.f + a.rho_d / 2.f * xr[k] : 0.f;
        }
        stamp(6);
        itzd = lds_cg<TPG, BAND, 1, SB, NU, ND>(c, br, z, rhs, nullptr, 0, 0, a.rho_d / 2.f, a.mu_d2, a.max_cg, a.cg_tol,
                              ah ? ah + 2 * hstride : nullptr, bh ? bh + 2 * hstride : nullptr, a.Bp, a.nonfinite);
        stamp(7);
        // ---- batch 4: x_new, old zd, gamma_d + the operands of the phi prox
        float zo[TPG], gv[TPG];
#pragma unroll
        for (int k = 0; k < TPG; ++k) {
            xr[k] = zo[k] = gv[k] = zn[k] = gn[k] = 0.f;
            if (c.active) {
                const int e = c.gl(k);
                xr[k] = xn[e];
                zo[k] = zd[e];
                gv[k] = gd[e];
                if (a.has_phi) { zn[k] = phi[e]; gn[k] = gam[e]; }
            }
        }
        double m_prizd = 0, m_dualzd = 0;
        if (c.active) {
#pragma unroll
            for (int k = 0; k < TPG; ++k) {
                const int e = c.gl(k);
                const float pz = xr[k] - z[k], dz = z[k] - zo[k];
                m_prizd += (double)pz * pz;
                m_dualzd += (double)dz * dz;
                zd[e] = z[k];
                gd[e] = gv[k] + a.rho_d * pz;
            }
        }
        emit(MGADMM_M_PRI_ZD, m_prizd, true);
        emit(MGADMM_M_DUAL_ZD, m_dualzd, true);
    } else {
        emit(MGADMM_M_PRI_ZD, 0.0, false);
        emit(MGADMM_M_DUAL_ZD, 0.0, false);
    }

    // ---- phi prox, gamma update, Ldr/Lu based diagnostics (ADMM.py:600-606, 619, 627-637); xr = x_new, zn = phi_old, gn = gamma
    double m_priphi = 0, m_dualphi = 0, m_dgtv = 0, m_dglr = 0, m_glr = 0;
    stamp_late(8);
    __syncthreads();          // every LDS read of the last CG is done
    c.template put<-1>(P, xr);        // read by Ldr
    if (!SB) c.template put<0>(Q, xr);   // read by Lu (GLR); SB: Q aliases P, see below
    __syncthreads();
    if (c.active) {
        float l[TPG];
        c.op_ldr(P, xr, l);
        const float thr = a.mu_d1 / a.rho;
#pragma unroll
        for (int k = 0; k < TPG; ++k) {
            m_dgtv += fabs((double)l[k]);
            m_dglr += (double)l[k] * l[k];
            if (a.has_phi) {
                const int e = c.gl(k);
                const float gv = gn[k];
                const float s = l[k] - gv / a.rho;
                const float u = fabsf(s) - thr;
                const float pn = (u > 0.f) ? (s > 0.f ? u : -u) : 0.f;
                const float dd = pn - l[k], dp = pn - zn[k];
                m_priphi += (double)dd * dd;
                m_dualphi += (double)dp * dp;
                phi[e] = pn;
                gam[e] = gv + a.rho * dd;
            }
        }
        if (!SB) {
            c.op_lu(Q, xr, l);
#pragma unroll
            for (int k = 0; k < TPG; ++k) m_glr += (double)xr[k] * (double)l[k];
        }
    }
    if (SB) {                 // one LDS vector: the unshifted image replaces the shifted one after every Ldr gather is done
        __syncthreads();
        c.template put<0>(P, xr);
        __syncthreads();
        if (c.active) {
            float l[TPG];
            c.op_lu(P, xr, l);
#pragma unroll
            for (int k = 0; k < TPG; ++k) m_glr += (double)xr[k] * (double)l[k];
        }
    }

    stamp_late(9);
    emit(MGADMM_M_PRI_PHI, m_priphi, a.has_phi);
    emit(MGADMM_M_DUAL_PHI, m_dualphi, a.has_phi);
    emit(MGADMM_M_DGTV, m_dgtv, a.has_phi);
    emit(MGADMM_M_DGLR, m_dglr, a.has_zd);
    emit(MGADMM_M_GLR, m_glr, true);
    stamp_late(10);
    if (tid == 0) {
        a.cg_iters[b] = itx;
        a.cg_iters[a.Bp + b] = itzu;
        a.cg_iters[2 * a.Bp + b] = itzd;
    }
}

// initial state in sample-major layout: x0 (regression), zu = zd = x0, gamma* = 0.1   (ADMM.py:528-544)
template <bool MASKED>
__global__ __launch_bounds__(256) void k_init_lds(int T, int t_in, int N, int B, float tm, float den, const float* __restrict__ y,
                                                  const float* __restrict__ mask, float* __restrict__ x, float* __restrict__ zu,
                                                  float* __restrict__ zd, float* __restrict__ gam, float* __restrict__ gu,
                                                  float* __restrict__ gd, int* __restrict__ nonfinite) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= N) return;
    const size_t sb = (size_t)b * T * N;
    float w, c0;
    if (!MASKED) {
        const float* yb = y + (size_t)b * t_in * N;
        float sy = 0.f, sty = 0.f;
        for (int t = 0; t < t_in; ++t) {
            const float v = yb[t * N + i];
            sy += v;
            sty += (float)t * v;
        }
        const float ym = sy / (float)t_in;
        w = (sty / (float)t_in - tm * ym) / den;
        c0 = ym - w * tm;
        for (int t = 0; t < T; ++t) {
            const float v = (t < t_in) ? yb[t * N + i] : w * (float)t + c0;
            const size_t e = sb + (size_t)t * N + i;
            x[e] = v; zu[e] = v; zd[e] = v;
            gam[e] = 0.1f; gu[e] = 0.1f; gd[e] = 0.1f;
        }
    } else {
        const float* yb = y + sb;
        const float* mb = mask + sb;
        float n = 0, ts = 0, t2s = 0, ys = 0, tys = 0;
        for (int t = 0; t < T; ++t) {
            const float m = mb[t * N + i], v = yb[t * N + i];
            n += m; ts += (float)t * m; t2s += (float)(t * t) * m;
            ys += v * m; tys += (float)t * v * m;
        }
        const float tmean = ts / n, t2mean = t2s / n, ymean = ys / n, tymean = tys / n;
        float tm2 = tmean * tmean;
        asm volatile("" : "+v"(tm2));
        const float dn = t2mean - tm2;
        w = (tymean - tmean * ymean) / dn;
        c0 = ymean - w * tmean;
        if (!(fabsf(w) <= 3.0e38f) || !(fabsf(c0) <= 3.0e38f)) *nonfinite = 1;
        for (int t = 0; t < T; ++t) {
            const float v = (w * (float)t + c0) * (1.f - mb[t * N + i]) + yb[t * N + i];
            const size_t e = sb + (size_t)t * N + i;
            x[e] = v; zu[e] = v; zd[e] = v;
            gam[e] = 0.1f; gu[e] = 0.1f; gd[e] = 0.1f;
        }
    }
}

// delta_x_per_step on the sample-major layout.  Pass 1: workgroup (e-block, b-slice) sums x - x_old over
// its 64 samples -> part[slice][e].  Pass 2: fixed-order sum over the slices, mean, square -> m2[e].
__global__ __launch_bounds__(256) void k_dxps_sm(int TN, int B, const float* __restrict__ x, const float* __restrict__ xo,
                                                 double* __restrict__ part, const int* __restrict__ stop) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= TN || (stop != nullptr && *stop != 0)) return;
    const int b0 = blockIdx.y * 64, b1 = min(B, b0 + 64);
    double s = 0.0;
    for (int b = b0; b < b1; ++b) s += (double)x[(size_t)b * TN + e] - (double)xo[(size_t)b * TN + e];
    part[(size_t)blockIdx.y * TN + e] = s;
}

// the same with four consecutive elements per thread (16-byte loads; T*N a multiple of 4 and 16-byte aligned vectors):
// 61 -> 4x fewer load instructions for the 2 * B * T*N * 4 bytes this pass streams
// One-wave workgroups: in the overlapped schedule of the outer loop (Engine::solve_lds) this kernel runs beside the k_admm_lds
// launch of the next iteration, whose workgroup leaves room for single waves only (its 15 waves fill three of the four SIMDs
// of a CU; the fourth has 128 free VGPRs)
__global__ __launch_bounds__(64) void k_dxps_sm4(int TN, int B, const float* __restrict__ x, const float* __restrict__ xo,
                                                 double* __restrict__ part, const int* __restrict__ stop) {
    const int e = (blockIdx.x * 64 + threadIdx.x) * 4;
    if (e >= TN || (stop != nullptr && *stop != 0)) return;
    const int b0 = blockIdx.y * 64, b1 = min(B, b0 + 64);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int b = b0;
    for (; b + 4 <= b1; b += 4) {
        lds_f4 a[4], o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = *reinterpret_cast<const lds_f4*>(x + (size_t)(b + u) * TN + e);
            o[u] = *reinterpret_cast<const lds_f4*>(xo + (size_t)(b + u) * TN + e);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s0 += (double)a[u].x - (double)o[u].x;
            s1 += (double)a[u].y - (double)o[u].y;
            s2 += (double)a[u].z - (double)o[u].z;
            s3 += (double)a[u].w - (double)o[u].w;
        }
    }
    for (; b < b1; ++b) {
        const lds_f4 a = *reinterpret_cast<const lds_f4*>(x + (size_t)b * TN + e);
        const lds_f4 o = *reinterpret_cast<const lds_f4*>(xo + (size_t)b * TN + e);
        s0 += (double)a.x - (double)o.x;
        s1 += (double)a.y - (double)o.y;
        s2 += (double)a.z - (double)o.z;
        s3 += (double)a.w - (double)o.w;
    }
    double* dst = part + (size_t)blockIdx.y * TN + e;
    dst[0] = s0; dst[1] = s1; dst[2] = s2; dst[3] = s3;
}

__global__ __launch_bounds__(256) void k_dxps_sm_mean(int TN, int B, int nslices, const double* __restrict__ part,
                                                      double* __restrict__ m2, const int* __restrict__ stop) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= TN || (stop != nullptr && *stop != 0)) return;
    double s = 0.0;
    int k = 0;
    for (; k + 8 <= nslices; k += 8) {        // eight slices per trip in flight (one per trip: 64 dependent L2 round trips, 17 us); same order of additions
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(k + u) * TN + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < nslices; ++k) s += part[(size_t)k * TN + e];
    s /= (double)B;
    m2[e] = s * s;
}

__global__ void k_dxps_sm_final(int T, int N, const double* __restrict__ m2, double* __restrict__ out, const int* __restrict__ stop) {
    const int t = blockIdx.x;
    if (stop != nullptr && *stop != 0) return;          // uniform
    __shared__ double sm[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) s += m2[(size_t)t * N + i];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[t] = sqrt(sm[0]);
}

// Stop test of one ADMM iteration, on the device (the host's test of the synchronous loop, Engine::solve_lds): row = the
// whole-batch metrics of iteration `it` (k_batch_metrics).  One lane.
__global__ void k_lds_stop_test(const double* __restrict__ row, const int* __restrict__ nonfinite, int has_phi, int has_zd, double tol,
                                int it, int* __restrict__ stop) {
    if (threadIdx.x != 0 || *stop != 0) return;
    bool finite = *nonfinite == 0;
    for (int k = 0; k < MGADMM_NMETRIC; ++k) finite = finite && (fabs(row[k]) <= 1.7976931348623157e308);
    if (!finite) { *stop = -(it + 1); return; }
    double pri = row[MGADMM_M_PRI_ZU], dual = row[MGADMM_M_DUAL_ZU];
    if (has_phi) { pri = fmax(pri, row[MGADMM_M_PRI_PHI]); dual = fmax(dual, row[MGADMM_M_DUAL_PHI]); }
    if (has_zd) { pri = fmax(pri, row[MGADMM_M_PRI_ZD]); dual = fmax(dual, row[MGADMM_M_DUAL_ZD]); }
    if (pri < tol && dual < tol) *stop = it + 1;
}

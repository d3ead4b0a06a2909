// LDS-resident fused ADMM iterations for small graphs (PEMS-size: T*N*8 B + tables fit the 160 KiB LDS).
//
// One workgroup = one sample.  One launch = J full ADMM iterations of reference ADMM.py:546-646 on every sample (the
// workgroup keeps its sample for the J trips): RHS_x, the three CG solves (x, zu, zd) with all their iterations, the dual
// updates, the phi prox and every residual / regulariser of the history.  Inside a CG solve nothing touches HBM:
//   * thread (g, i) owns node i at TPG consecutive time steps t0 = g*TPG ...; x, r, p, Ap of those
//     elements live in registers for the whole solve;
//   * a copy of the CG direction p (and of q = Ldr p) lives in LDS so that neighbours can be gathered; every gather is an
//     aligned ds_read_b128 run (lds_apply: shifted ownership of q) -- the only LDS traffic of an iteration is those
//     gathers and one store of p and q;
//   * a thread's rows of W_u, W_d and the leading entries of its W_d^T row ({LDS row offset, weight} pairs) sit in registers
//     during a solve, the rest of the W_d^T rows in an LDS table padded to a uniform width;
//   * idle threads of the last wave are GHOSTS that own rows of zeros: one instruction stream for everybody (LdsCtx);
//   * p.Ap and r.r are reduced with wave shuffles + a 16-entry LDS exchange in fixed order (repeatable);
//     alpha, beta and the convergence test are computed redundantly by every thread (workgroup-uniform).
// The iterates x live in HBM in the reference's own sample-major (B, T*N) layout (the ABI's x needs no conversion); zu, zd,
// phi and the dual variables in a thread-major layout (lds_state_index), converted only when a caller asks for them.
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
    // row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave total.  Written as the one
    // instruction each of them is (the rows outside the row mask keep their value): through update_dpp the compiler emits a
    // zeroing move, a v_mov_b32_dpp and an add per step.  The s_nop are the wait states a DPP operand needs after the vector
    // instruction that wrote it (the hazard recogniser does not look into an asm statement).
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1" : "+v"(v));
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
    // (explicitly global: a base that went through scalar_ptr has lost its address space, and a FLAT access has no scalar-base form)
    const __attribute__((address_space(1))) char* g = (const __attribute__((address_space(1))) char*)base;
    return *reinterpret_cast<const __attribute__((address_space(1))) float*>(g + boff);
}
__device__ __forceinline__ void stg(float* base, unsigned boff, float v) {
    __attribute__((address_space(1))) char* g = (__attribute__((address_space(1))) char*)base;
    *reinterpret_cast<__attribute__((address_space(1))) float*>(g + boff) = v;
}
// REQUEST fence: the loads before it are issued together and waited for together, nothing crosses it.
#define MG_REQ_FENCE() asm volatile("" ::: "memory")
// a workgroup-uniform value the optimiser must keep in a scalar register from here on (it would otherwise re-read a launch
// argument from the kernarg segment inside a CG loop: a scalar load shares its counter with the LDS gathers in flight)
#define MG_PIN_S(x) asm volatile("" : "+s"(x))
#define MG_PIN_V(x) asm volatile("" : "+v"(x))      // ... a value that was computed (no scalar float unit): stays in a vector register

typedef int lds_i4 __attribute__((ext_vector_type(4)));
// a workgroup-uniform pointer as a scalar register pair the optimiser cannot merge with a per-thread offset (readfirstlane of
// the two halves: a plain copy when the value already sits in scalar registers)
template <typename P>
__device__ __forceinline__ P* scalar_ptr(P* p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (P*)(((unsigned long long)hi << 32) | lo);
}
// PACKED ARITHMETIC BY HAND.  This translation unit is compiled with -fno-slp-vectorize and the element loops of the CG
// solves are written on pairs of floats (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on the register pairs (2j, 2j+1) of a
// thread's vector): the SLP vectoriser pairs whatever looks profitable locally -- in the first branch-free build of round 3
// it packed the shifted self terms of lds_apply (q[k] = v[k+1] - acc[k]) along v's pairs, which put every gathered 16-byte
// piece one element off its accumulator pair (six register moves per table entry, 75 v_mov per CG iteration), and, once
// that was blocked, packed ACROSS the table entries instead (moves, multiplies and adds in place of fused multiply-adds).
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 mk2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
// y[k] += a * x[k]
template <int TPG>
__device__ __forceinline__ void axpy(float a, const float (&x)[TPG], float (&y)[TPG]) {
    if constexpr (TPG % 2 == 0) {
#pragma unroll
        for (int j = 0; j < TPG / 2; ++j) {
            const f2 r = __builtin_elementwise_fma(mk2(a, a), mk2(x[2 * j], x[2 * j + 1]), mk2(y[2 * j], y[2 * j + 1]));
            y[2 * j] = r.x; y[2 * j + 1] = r.y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < TPG; ++k) y[k] = __builtin_fmaf(a, x[k], y[k]);
    }
}
// y[k] = x[k] + b * y[k]
template <int TPG>
__device__ __forceinline__ void xpby(const float (&x)[TPG], float b, float (&y)[TPG]) {
    if constexpr (TPG % 2 == 0) {
#pragma unroll
        for (int j = 0; j < TPG / 2; ++j) {
            const f2 r = __builtin_elementwise_fma(mk2(b, b), mk2(y[2 * j], y[2 * j + 1]), mk2(x[2 * j], x[2 * j + 1]));
            y[2 * j] = r.x; y[2 * j + 1] = r.y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < TPG; ++k) y[k] = __builtin_fmaf(b, y[k], x[k]);
    }
}
// sum_k x[k] * y[k]: two running sums over the even / odd elements, then their sum (fixed order)
template <int TPG>
__device__ __forceinline__ float dot(const float (&x)[TPG], const float (&y)[TPG]) {
    if constexpr (TPG % 2 == 0) {
        f2 acc = mk2(x[0], x[1]) * mk2(y[0], y[1]);
#pragma unroll
        for (int j = 1; j < TPG / 2; ++j) acc = __builtin_elementwise_fma(mk2(x[2 * j], x[2 * j + 1]), mk2(y[2 * j], y[2 * j + 1]), acc);
        return acc.x + acc.y;
    } else {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < TPG; ++k) acc = __builtin_fmaf(x[k], y[k], acc);
        return acc;
    }
}
// y[k] = a * x[k]
template <int TPG>
__device__ __forceinline__ void scal(float a, const float (&x)[TPG], float (&y)[TPG]) {
    if constexpr (TPG % 2 == 0) {
#pragma unroll
        for (int j = 0; j < TPG / 2; ++j) {
            const f2 r = mk2(a, a) * mk2(x[2 * j], x[2 * j + 1]);
            y[2 * j] = r.x; y[2 * j + 1] = r.y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < TPG; ++k) y[k] = a * x[k];
    }
}
// A TABLE ENTRY IN REGISTERS is the pair {LDS byte address of the 16-byte run(s) the thread gathers, weight}: the address feeds
// ds_read_b128 as it stands, and v_pk_fma_f32 takes the weight for both halves straight from the pair (op_sel), so an
// entry costs two registers.  (A weight splatted by hand -- mk2(w, w) -- is hoisted out of the CG loop as a second register
// pair per entry, and the LDS addresses then went to scratch: five dependent scratch reloads per CG iteration.)
typedef int i2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 wsplat(i2v e) {
    const f2 f = __builtin_bit_cast(f2, e);
    return __builtin_shufflevector(f, f, 1, 1);
}
// y[k] = a[k mod 2] * x[k]
template <int TPG>
__device__ __forceinline__ void scal2(f2 a, const float (&x)[TPG], float (&y)[TPG]) {
    if constexpr (TPG % 2 == 0) {
#pragma unroll
        for (int j = 0; j < TPG / 2; ++j) {
            const f2 r = a * mk2(x[2 * j], x[2 * j + 1]);
            y[2 * j] = r.x; y[2 * j + 1] = r.y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < TPG; ++k) y[k] = a.x * x[k];
    }
}
// acc[k] += w_e * va[k]
template <int TPG>
__device__ __forceinline__ void wacc(i2v e, const float (&va)[TPG], float (&acc)[TPG]) {
    if constexpr (TPG % 2 == 0) {
        const f2 w = wsplat(e);
#pragma unroll
        for (int j = 0; j < TPG / 2; ++j) {
            const f2 r = __builtin_elementwise_fma(w, mk2(va[2 * j], va[2 * j + 1]), mk2(acc[2 * j], acc[2 * j + 1]));
            acc[2 * j] = r.x; acc[2 * j + 1] = r.y;
        }
    } else {
        const float w = __int_as_float(e.y);
#pragma unroll
        for (int k = 0; k < TPG; ++k) acc[k] = __builtin_fmaf(w, va[k], acc[k]);
    }
}
// LDS byte address of a pointer into the dynamic LDS block, and vector access through such an address
__device__ __forceinline__ int lds_addr(const float* p) {
    return (int)(unsigned)(size_t)(const __attribute__((address_space(3))) float*)p;
}
template <int TPG>
__device__ __forceinline__ void lds_load_a(int addr, float (&v)[TPG]) {
    const __attribute__((address_space(3))) float* p = (const __attribute__((address_space(3))) float*)(size_t)(unsigned)addr;
    if constexpr (TPG % 4 == 0) {
#pragma unroll
        for (int j = 0; j < TPG / 4; ++j) {
            const lds_f4 q = reinterpret_cast<const __attribute__((address_space(3))) lds_f4*>(p)[j];
            v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < TPG; ++k) v[k] = p[k];
    }
}

// the two halves of RHS_x that do not need a neighbour, and the operand of its Ldr^T term (ADMM.py:556-564): ONE expression
// each, used wherever the value is formed (from the state in HBM at the first trip of a launch, from the registers of the
// previous trip afterwards), so that the result does not depend on how the iterations are cut into launches
__device__ __forceinline__ float rhs_half(float rho, float z, float g) { return __builtin_fmaf(rho, z, -g) * 0.5f; }
__device__ __forceinline__ float ldrt_operand(float rho, float phi, float gam) { return __builtin_fmaf(rho, phi, gam); }

// Per-thread view: LDS row `i`, time steps t0 .. t0+TPG-1.
//   Threads tid < N*G own node i = tid mod N at time group tid / N.  The other threads of the workgroup are GHOSTS: ghost k
//   owns LDS row N + k (all zeros, written only by itself, with zeros) at time group 0, its table rows hold zero weights and
//   point at its own row, its HBM accesses are clamped to node 0 and masked.  Ghosts run the same instruction stream as
//   everybody else -- no `if (active)` regions, no per-lane trip counts inside the solves -- and add exact zeros to every
//   sum.  (Until round 3 idle threads were masked off: an exec-mask region and a set of zeroing moves around every operator
//   application; worse, the divergent tail loop of the ragged W_d^T gather made this kernel fragile under register pressure:
//   hipcc 7.2 gave the exec-mask accumulator of that loop the SGPR pair that held the LDS address of the W_d^T table --
//   profiles/r03/miscompile_s63_evidence.txt.)
//   LDS vectors are row-major, time innermost: A[row*TS + t] (row stride TS >= T, an odd number of 16-B slots so that rows
//   start on all banks) -> the TPG time steps of any node are one contiguous, aligned run (vector ds_read), also for a
//   gathered neighbour; HBM state is the reference's (T, N) order per sample: element (t, i) at t*N + i (coalesced over i).
template <int TPG, bool BAND, int NU = 0, int ND = 0, int TP = -1>
struct LdsCtx {
    int T, TS, N, t0, i, ig;   // i: LDS row, ig: node in HBM (ghosts: 0)
    bool active;
    float* P;
    float* Q;
    const i2v* en_u; const i2v* en_d;        // entries {col * TS, weight} of W_u / W_d rows (LDS image, or the global image: uniform instances)
    const i2v* lead_t;                       // W_d^T: [NR][LDS_NLEAD] leading entries (LDS or global image)
    const i2v* tail_t;                       // W_d^T: [NR][2 * tail_pairs] further entries (LDS), rows padded with {own row, 0}
    int u0, u1, d0, d1;                      // CSR row bounds (ragged gathers)
    int tail_pairs;
    unsigned so0;                            // byte offset of the thread's TPG elements in a thread-major state vector (ghosts: 0, masked)
    unsigned kN4[TPG];                       // byte offset k * N * 4 of element k in a state vector (scalar registers, set before any divergent region)
    f2 wdiag;                                // diagonal weights {W_d[i][i] (uniform instances: the table rows hold the others), W_d^T[i][i]}
    int skip, q1;
    const float* band_w;

    __device__ __forceinline__ unsigned glb0() const { return 4u * (unsigned)(t0 * N + ig); }   // HBM byte offset of element k = 0 (element k: + 4 k N, added to the uniform base)
    __device__ __forceinline__ int own() const { return i * TS + t0; }            // LDS index of element k = 0

    // The thread's rows of the fixed-width tables as REGISTER ENTRIES (uniform instances: requested by the kernel ahead of the
    // phase that precedes their use; the other instances read their tables at every application and carry an unused dummy).
    // The addresses of d and u point into the image P, those of t into Q (what the CG solves gather from); an application
    // to another image adds the distance of the two images.
    static constexpr bool UNI = NU > 0 && ND > 0;
    static constexpr int NDO = ND > 0 ? ND - 1 : 0;          // entries of a uniform W_d row besides the diagonal one
    struct Rows {
        i2v u[NU > 0 ? NU : 1];
        i2v d[NDO > 0 ? NDO : 1];
        i2v t[LDS_NLEAD];
    };
    // table entry {col * TS, w} -> register entry for image IMG
    __device__ __forceinline__ i2v reg_entry(i2v e, const float* IMG) const {
        i2v r;
        r.x = lds_addr(IMG) + 4 * (e.x + t0);
        r.y = e.y;
        return r;
    }
    template <int NE>
    __device__ __forceinline__ void load_rows(const i2v* EN, int e0, const float* IMG, i2v (&en)[NE]) const {
        i2v raw[NE];
#pragma unroll
        for (int u = 0; u < NE; ++u) raw[u] = EN[e0 + u];
#pragma unroll
        for (int u = 0; u < NE; ++u) en[u] = reg_entry(raw[u], IMG);
    }

    // acc[k] += sum_e w_e * IMG[col_e + t0 + k] over NE register entries (`delta`: byte distance of the image read from the
    // image the entries point into).  Every gather of a neighbour's TPG-step window is the ALIGNED run [t0, t0+TPG-1]:
    // conflict-free ds_read_b128 only.  The sum runs in entry order.
    template <int NE>
    __device__ __forceinline__ void gather_regs(const i2v (&en)[NE], int delta, float (&acc)[TPG]) const {
#pragma unroll
        for (int u = 0; u + 1 < NE; u += 2) {
            float va[TPG], vb[TPG];
            lds_load_a<TPG>(en[u].x + delta, va);
            lds_load_a<TPG>(en[u + 1].x + delta, vb);
            wacc<TPG>(en[u], va, acc);
            wacc<TPG>(en[u + 1], vb, acc);
        }
        if (NE & 1) {
            float va[TPG];
            lds_load_a<TPG>(en[NE - 1].x + delta, va);
            wacc<TPG>(en[NE - 1], va, acc);
        }
    }
    // ragged rows of the generic instances (W_u / W_d with pads): two entries in flight per trip, per-lane trip count
    __device__ __forceinline__ void gather(const float* SRC, const i2v* EN, int e0, int e1, float (&acc)[TPG]) const {
        const float* base = SRC + t0;
        i2v na = EN[e0], nb = EN[e0 + 1];          // the arrays are padded by 3 entries: reads past e1 are safe
        int e = e0;
        for (; e + 1 < e1; e += 2) {
            const i2v ea = na, eb = nb;
            float va[TPG], vb[TPG];
            lds_load<TPG>(base + ea.x, va);
            lds_load<TPG>(base + eb.x, vb);
            na = EN[e + 2];
            nb = EN[e + 3];
            wacc<TPG>(ea, va, acc);
            wacc<TPG>(eb, vb, acc);
        }
        if (e < e1) {
            float va[TPG];
            lds_load<TPG>(base + na.x, va);
            wacc<TPG>(na, va, acc);
        }
    }
    // the tail of the W_d^T rows: the same number of pairs of entries for EVERY lane (two entries = one 16-byte read); a row
    // that ends earlier holds {own row, weight 0} there -- consecutive lanes read consecutive rows, which cannot collide.
    // Same instruction slots as the per-lane loop it replaces (every wave of cfg2 holds a 9- or 10-entry row), but no
    // exec-mask bookkeeping, and with the pair count a compile-time constant (uniform instances) every pair is requested first.
    __device__ __forceinline__ void gather_tail(const float* SRC, float (&acc)[TPG]) const {
        const float* base = SRC + t0;
        if constexpr (TP >= 0) {
            const lds_i4* row = reinterpret_cast<const lds_i4*>(tail_t + (size_t)i * 2 * TP);
            lds_i4 e2[TP > 0 ? TP : 1];
#pragma unroll
            for (int j = 0; j < TP; ++j) e2[j] = row[j];
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                float va[TPG], vb[TPG];
                lds_load<TPG>(base + e2[j].x, va);
                lds_load<TPG>(base + e2[j].z, vb);
                wacc<TPG>(e2[j].xy, va, acc);
                wacc<TPG>(e2[j].zw, vb, acc);
            }
        } else {
            const lds_i4* row = reinterpret_cast<const lds_i4*>(tail_t + (size_t)i * 2 * tail_pairs);
            lds_i4 nxt = row[0];                        // (the table is padded by one pair)
            for (int j = 0; j < tail_pairs; ++j) {
                const lds_i4 e2 = nxt;
                float va[TPG], vb[TPG];
                lds_load<TPG>(base + e2.x, va);
                lds_load<TPG>(base + e2.z, vb);
                nxt = row[j + 1];
                wacc<TPG>(e2.xy, va, acc);
                wacc<TPG>(e2.zw, vb, acc);
            }
        }
    }

    // acc = W_u src / W_d src / W_d^T src on the own elements (src: an LDS image read by that operator; own_run: the thread's
    // OWN elements of that image -- what an entry with the thread's own row as column reads -- or nullptr: read them here).
    // The diagonal entry of W_d^T (and of W_d in the uniform instances) is not in the tables: its weight sits in a register,
    // and inside a CG solve its operand is the thread's own vector (lds_apply).
    __device__ __forceinline__ void mul_wu(const float* SRC, float (&acc)[TPG], const Rows& R) const {
#pragma unroll
        for (int k = 0; k < TPG; ++k) acc[k] = 0.f;
        if constexpr (UNI) gather_regs<NU>(R.u, lds_addr(SRC) - lds_addr(P), acc);
        else gather(SRC, en_u, u0, u1, acc);
    }
    __device__ __forceinline__ void mul_wd(const float* SRC, float (&acc)[TPG], const Rows& R, const float* own_run = nullptr) const {
        if constexpr (UNI) {
            float self[TPG];
            if (own_run) {
#pragma unroll
                for (int k = 0; k < TPG; ++k) self[k] = own_run[k];
            } else lds_load<TPG>(SRC + own(), self);
            scal2<TPG>(__builtin_shufflevector(wdiag, wdiag, 0, 0), self, acc);
            gather_regs<NDO>(R.d, lds_addr(SRC) - lds_addr(P), acc);
        } else {
#pragma unroll
            for (int k = 0; k < TPG; ++k) acc[k] = 0.f;
            gather(SRC, en_d, d0, d1, acc);
        }
    }
    __device__ __forceinline__ void mul_wdt(const float* SRC, float (&acc)[TPG], const Rows& R, const float* own_run = nullptr) const {
        float self[TPG];
        if (own_run) {
#pragma unroll
            for (int k = 0; k < TPG; ++k) self[k] = own_run[k];
        } else lds_load<TPG>(SRC + own(), self);
        scal2<TPG>(__builtin_shufflevector(wdiag, wdiag, 1, 1), self, acc);
        if constexpr (UNI) gather_regs<LDS_NLEAD>(R.t, lds_addr(SRC) - lds_addr(Q), acc);
        else {
            i2v en[LDS_NLEAD];
            load_rows<LDS_NLEAD>(lead_t, i * LDS_NLEAD, SRC, en);
            gather_regs<LDS_NLEAD>(en, 0, acc);
        }
        gather_tail(SRC, acc);
    }
    // band (line-graph) stencils on the row's own time axis
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
    // l = Lu(src): neighbours from SRC (LDS, unshifted image), the thread's own elements of src from registers (self)  ADMM.py:138-148
    __device__ __forceinline__ void op_lu(const float* SRC, const float (&self)[TPG], float (&l)[TPG], const Rows& R) const {
        float acc[TPG];
        mul_wu(SRC, acc, R);
#pragma unroll
        for (int k = 0; k < TPG; ++k) l[k] = self[k] - acc[k];
    }
    // l = Ldr(src), SRC: image stored with put<-1>      ADMM.py:150-177
    __device__ __forceinline__ void op_ldr(const float* SRC, const float (&self)[TPG], float (&l)[TPG], const Rows& R) const {
        float acc[TPG];
        if constexpr (BAND) band_back(SRC, acc);
        else mul_wd(SRC, acc, R);
#pragma unroll
        for (int k = 0; k < TPG; ++k) l[k] = ((k >= 1 || t0 >= 1) ? self[k] : 0.f) - acc[k];
    }
    // l = Ldr_T(src), SRC: image stored with put<+1>    ADMM.py:179-223 (q1: identity kept on the t=0 block)
    __device__ __forceinline__ void op_ldrt(const float* SRC, const float (&self)[TPG], float (&l)[TPG], const Rows& R) const {
        float acc[TPG];
        if constexpr (!BAND) mul_wdt(SRC, acc, R);
        else band_fwd(SRC, acc);
#pragma unroll
        for (int k = 0; k < TPG; ++k) l[k] = ((k > 0 || t0 > 0 || q1) ? self[k] : 0.f) - acc[k];
    }
    // own elements -> LDS image read by an operator that gathers time t + SH (SH = 0: Lu and the band stencils, -1: Ldr,
    // +1: Ldr^T): index j of a row holds time j + SH, 0 outside [0,T).  (Only the operator applications outside the CG
    // solves use the shifted forms; inside a solve lds_apply shifts the OWNERSHIP of q instead.)
    template <int SH>
    __device__ __forceinline__ void put(float* DST, const float (&v)[TPG]) const {
        if constexpr (SH == 0 || BAND) {
            lds_store<TPG>(DST + own(), v);
        } else {
            float* row = DST + i * TS + t0 - SH;            // index of element k = 0
            // the 16-byte groups of the image that lie entirely inside this thread's elements go out as one aligned
            // ds_write_b128 each (SH = -1: v[3..6], v[7..10], ...; SH = +1: v[1..4], v[5..8], ...); the elements at the two
            // ends are scalar stores
            constexpr int K0 = SH < 0 ? 3 : 1;              // first element of the first whole group
            constexpr int NG = TPG % 4 == 0 ? (TPG - K0) / 4 : 0;
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                lds_f4 q;
                q.x = v[K0 + 4 * j]; q.y = v[K0 + 4 * j + 1]; q.z = v[K0 + 4 * j + 2]; q.w = v[K0 + 4 * j + 3];
                *static_cast<lds_f4*>(__builtin_assume_aligned(row + K0 + 4 * j, 16)) = q;
            }
#pragma unroll
            for (int k = 0; k < TPG; ++k) {
                if (k >= K0 && k < K0 + 4 * NG) continue;
                const bool ok = SH < 0 ? (k < TPG - 1 || t0 + TPG < T) : (k > 0 || t0 > 0);   // index T resp. -1 does not exist
                if (ok) row[k] = v[k];
            }
            if (SH < 0 && t0 == 0) DST[i * TS] = 0.f;                      // time -1
            if (SH > 0 && t0 + TPG == T) DST[i * TS + T - 1] = 0.f;        // time T
        }
    }
    // own elements -> thread-major state vector in HBM (sample base already applied)
    __device__ __forceinline__ void puts(float* DST, const float (&v)[TPG]) const {
        if (active) {
            unsigned o = so0;
            MG_PIN_V(o);
            __attribute__((address_space(1))) char* g = (__attribute__((address_space(1))) char*)DST;
            if constexpr (TPG % 4 == 0) {
#pragma unroll
                for (int j = 0; j < TPG / 4; ++j) {
                    lds_f4 q;
                    q.x = v[4 * j]; q.y = v[4 * j + 1]; q.z = v[4 * j + 2]; q.w = v[4 * j + 3];
                    *reinterpret_cast<__attribute__((address_space(1))) lds_f4*>(g + o + 16 * j) = q;
                }
            } else {
#pragma unroll
                for (int k = 0; k < TPG; ++k) *reinterpret_cast<__attribute__((address_space(1))) float*>(g + o + 4 * k) = v[k];
            }
        }
    }
    // own elements -> HBM vector in the reference's (T, N) layout (sample base already applied)
    __device__ __forceinline__ void putg(float* DST, const float (&v)[TPG]) const {
        if (active) {
            unsigned o0 = glb0();
            MG_PIN_V(o0);            // (offsets formed next to the accesses: see the kernel's request helpers)
#pragma unroll
            for (int k = 0; k < TPG; ++k) stg(DST, o0 + kN4[k], v[k]);
        }
    }
};

// a per-thread operand vector parked in LDS across a CG solve (registers cannot hold it): element k of thread `tid`, 16-byte
// pieces of consecutive threads adjacent (conflict-free ds_read_b128 / ds_write_b128)
template <int TPG>
__device__ __forceinline__ void slot_put(float* S, int tid, int nthr, const float (&v)[TPG]) {
    if constexpr (TPG % 4 == 0) {
#pragma unroll
        for (int j = 0; j < TPG / 4; ++j) {
            lds_f4 q;
            q.x = v[4 * j]; q.y = v[4 * j + 1]; q.z = v[4 * j + 2]; q.w = v[4 * j + 3];
            static_cast<lds_f4*>(__builtin_assume_aligned(S, 16))[j * nthr + tid] = q;
        }
    } else {
#pragma unroll
        for (int k = 0; k < TPG; ++k) S[k * nthr + tid] = v[k];
    }
}
template <int TPG>
__device__ __forceinline__ void slot_get(const float* S, int tid, int nthr, float (&v)[TPG]) {
    if constexpr (TPG % 4 == 0) {
#pragma unroll
        for (int j = 0; j < TPG / 4; ++j) {
            const lds_f4 q = static_cast<const lds_f4*>(__builtin_assume_aligned(S, 16))[j * nthr + tid];
            v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < TPG; ++k) v[k] = S[k * nthr + tid];
    }
}

// av = (A v) on the thread's own elements, for the vector v held in ctx.P (LDS, unshifted; own elements also in v):
//   KIND 1: dc*v + c2*Ldr_T(Ldr v) ; KIND 2: dc*v + c2*Lu v ; KIND 0: dc*v
// dc = diagonal coefficient of the own elements (H^T H or mask value, plus the rho/2 terms), see lds_diag.
// Returns sum_k v_k * (A v)_k of the own elements.  Uses ctx.Q as scratch; contains a barrier for KIND 1.
// Callers separate successive calls by barriers.  R: the thread's table rows in registers (uniform instances).
template <int TPG, bool BAND, int KIND, bool SB, int NU, int ND, int TP>
__device__ __forceinline__ float lds_apply(LdsCtx<TPG, BAND, NU, ND, TP>& c, const float (&v)[TPG], float (&av)[TPG], const float (&dc)[TPG],
                                           float c2, typename LdsCtx<TPG, BAND, NU, ND, TP>::Rows& R) {
    // the register entries, made opaque IN PLACE once per application: their weights must be splatted INSIDE the CG loop, where
    // the instruction selector folds the splat into the packed multiply-add (hoisted, it becomes a register pair per weight)
    typedef LdsCtx<TPG, BAND, NU, ND, TP> Ctx;
    if constexpr (Ctx::UNI) {
        if (KIND == 1) {
#pragma unroll
            for (int u = 0; u < Ctx::NDO; ++u) MG_PIN_V(R.d[u]);
#pragma unroll
            for (int u = 0; u < LDS_NLEAD; ++u) MG_PIN_V(R.t[u]);
        } else if (KIND == 2) {
#pragma unroll
            for (int u = 0; u < NU; ++u) MG_PIN_V(R.u[u]);
        }
    }
    if (KIND == 1 && !BAND) MG_PIN_V(c.wdiag);
    float l[TPG];
#pragma unroll
    for (int k = 0; k < TPG; ++k) l[k] = 0.f;
    if (KIND == 1) {
        if constexpr (BAND) {
            float q[TPG];
            c.op_ldr(c.P, v, q, R);
            if (SB) __syncthreads();           // single LDS vector (Q aliases P): every gather of p is done before q replaces it
            c.template put<+1>(c.Q, q);
            __syncthreads();
            c.op_ldrt(c.Q, q, l, R);
        } else {
            // SHIFTED OWNERSHIP of q = Ldr v.  Ldr reads time t-1 and Ldr^T time t+1, so with one partition of the time axis for
            // both vectors either the store or the gather of an image is off by one float against the 16-byte groups.  Here
            // the thread that owns v at times t0 .. t0+TPG-1 computes and owns q at times t0+1 .. t0+TPG: its gather for
            // q[t0+1+k] is the aligned run v[col][t0+k] of the UNSHIFTED image of v, it stores its q values as the aligned
            // run at positions t0 .. (position j of the q image holds q[j+1]; time T does not exist: 0), and the Ldr^T
            // gather for time t0+k reads q[col][t0+k+1] = that aligned run again.  Both images go out as whole
            // ds_write_b128, every gather is an aligned run; the price is one read per operator for the self term that sits
            // in the neighbouring time group (v[t0+TPG] resp. q[t0]) -- taken from its aligned 16-byte group (a scalar read
            // of the same column of 32 consecutive rows is a 4-way bank conflict, the ds_read_b128 of consecutive rows none),
            // through a clamped address and a select where the neighbouring group does not exist (no exec-mask region).
            float q[TPG];        // q[k] = (Ldr v)[t0 + k + 1]
            {
                float acc[TPG];
                c.mul_wd(c.P, acc, R, v);           // diagonal term of the uniform rows: the thread's own v (registers)
                const bool has_next = c.t0 + TPG < c.T;
                float vnext;
                if constexpr (TPG % 4 == 0) vnext = static_cast<const lds_f4*>(__builtin_assume_aligned(c.P + c.own() + (has_next ? TPG : 0), 16))[0].x;
                else vnext = c.P[c.own() + (has_next ? TPG : 0)];
#pragma unroll
                for (int k = 0; k + 1 < TPG; ++k) q[k] = v[k + 1] - acc[k];
                q[TPG - 1] = (has_next ? vnext : acc[TPG - 1]) - acc[TPG - 1];      // time T does not exist: 0
            }
            if (SB) __syncthreads();           // single LDS vector (Q aliases P): every gather of p is done before q replaces it
            lds_store<TPG>(c.Q + c.own(), q);
            __syncthreads();
            {
                float acc[TPG];
                c.mul_wdt(c.Q, acc, R, q);          // diagonal term: position t0+k of the thread's own q row = q[k] (registers)
                const bool has_prev = c.t0 > 0;
                float qprev;          // q[t0]: the last value of the previous group (q[0] = 0)
                if constexpr (TPG % 4 == 0) qprev = static_cast<const lds_f4*>(__builtin_assume_aligned(c.Q + c.own() - (has_prev ? 4 : 0), 16))[0].w;
                else qprev = c.Q[c.own() - (has_prev ? 1 : 0)];
                l[0] = (has_prev ? qprev : 0.f) - acc[0];
#pragma unroll
                for (int k = 1; k < TPG; ++k) l[k] = q[k - 1] - acc[k];
            }
        }
    } else if (KIND == 2) {
        c.op_lu(c.P, v, l, R);
    }
    // av = dc * v (+ c2 * l);  v . av
    if constexpr (TPG % 2 == 0) {
#pragma unroll
        for (int j = 0; j < TPG / 2; ++j) {
            f2 r = mk2(dc[2 * j], dc[2 * j + 1]) * mk2(v[2 * j], v[2 * j + 1]);
            if (KIND != 0) r = __builtin_elementwise_fma(mk2(c2, c2), mk2(l[2 * j], l[2 * j + 1]), r);
            av[2 * j] = r.x; av[2 * j + 1] = r.y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < TPG; ++k) av[k] = (KIND == 0) ? dc[k] * v[k] : __builtin_fmaf(c2, l[k], dc[k] * v[k]);
    }
    return dot<TPG>(v, av);
}
// diagonal coefficient d + c1 of the own elements: d = dg[el] when dg != nullptr (mask values, global
// memory), else [hth && t < t_in]   (ADMM.py:371-379: H^T H x resp. mask * x; ADMM.py:381-399: none)
template <int TPG, bool BAND, int NU, int ND, int TP>
__device__ __forceinline__ void lds_diag(const LdsCtx<TPG, BAND, NU, ND, TP>& c, const float* dg, int hth, int t_in, float c1, float (&dc)[TPG]) {
    if (dg) {          // mask values: one branch-free batch of loads (ghosts read node 0, their value is dropped)
        unsigned o = c.glb0();
        MG_PIN_V(o);
        float mv[TPG];
#pragma unroll
        for (int k = 0; k < TPG; ++k) mv[k] = ldg(dg, o + c.kN4[k]);
        MG_REQ_FENCE();
#pragma unroll
        for (int k = 0; k < TPG; ++k) dc[k] = (c.active ? mv[k] : 0.f) + c1;
    } else {
        const int lim = hth ? t_in - c.t0 : 0;
#pragma unroll
        for (int k = 0; k < TPG; ++k) dc[k] = (k < lim ? 1.f : 0.f) + c1;
    }
}

// CG_solver (ADMM.py:329-368) for one sample.  x, r of the own elements live in registers, the direction
// p in registers with a copy in LDS (ctx.P) for the neighbours' gathers, A p in registers.  x holds x0 on entry and the solution on exit.  dmask: diagonal
// of the initial residual when a mask is given (global memory); the iterations always use [t<t_in]
// (quirk Q2).  Returns the iteration count (k+1) or -1.  Entry requirement: no thread still reads P/Q.
// Ghost threads enter with x = rhs = 0 and stay at 0.  R: the thread's rows of the tables this solve gathers with (uniform
// instances: W_d without its diagonal and the leading W_d^T entries for the cLdr solves, W_u for the zu solve), requested by
// the caller ahead of the phase that precedes the solve.
template <int TPG, bool BAND, int KIND, bool SB, int NU, int ND, int TP>
__device__ __forceinline__ int lds_cg(LdsCtx<TPG, BAND, NU, ND, TP>& c, BlockRed& br, float (&x)[TPG], const float (&rhs)[TPG], const float* dmask,
                      int hth, int t_in, float c1, float c2, int max_cg, double tol2, float* ah, float* bh, int Bp,
                      int* nonfinite, typename LdsCtx<TPG, BAND, NU, ND, TP>::Rows& R) {
    float r[TPG], pv[TPG], av[TPG], dc[TPG];
    // ONE loop, one inlined operator application: its first trip forms the initial residual (p = x0: r = rhs - A x0, then
    // p = r), the later trips are the CG iterations.  (With the initial residual in code of its own every solve carried two
    // copies of the operator application, and the registers of the first one were spilled around it.)
#pragma unroll
    for (int k = 0; k < TPG; ++k) { pv[k] = x[k]; r[k] = 0.f; }
    c.template put<0>(c.P, pv);
    lds_diag<TPG, BAND, NU, ND, TP>(c, dmask, hth, t_in, c1, dc);
    float rr = 0.f;
    // (one exit: with `break`s the compiler keeps a second copy of x for the exit paths and moves all of it once or twice per trip)
    int iters = -1;
    bool done = false, init = true;
    for (int it = -1; it < max_cg && !done; ++it) {
        __syncthreads();                 // p complete in LDS
        const float part = lds_apply<TPG, BAND, KIND, SB, NU, ND, TP>(c, pv, av, dc, c2, R);
        if (init) {                      // workgroup-uniform
            init = false;
            if (dmask != nullptr) lds_diag<TPG, BAND, NU, ND, TP>(c, nullptr, hth, t_in, c1, dc);     // quirk Q2: iterations use [t < t_in]
#pragma unroll
            for (int k = 0; k < TPG; ++k) {
                r[k] = rhs[k] - av[k];
                pv[k] = r[k];            // p = r
            }
            rr = br.sumf(dot<TPG>(r, r));        // barrier: every read of P (= x0) and Q is done
            c.template put<0>(c.P, pv);
            continue;
        }
        const float pAp = br.sumf(part);         // barrier: every gather from P/Q of this iteration is done
        // alpha, beta through v_rcp_f32 (1 ulp) instead of the correctly rounded division (a chain of ~10 dependent
        // instructions every thread waits for, twice per iteration: -2.9 % per launch).  The coefficients differ from
        // the IEEE quotient by at most 1.5 ulp -- below the rounding noise of the dot products they are formed from;
        // the streaming path keeps the IEEE division (its coefficients come from a separate tiny kernel).
        const float alpha = rr * __builtin_amdgcn_rcpf(pAp);
        axpy<TPG>(alpha, pv, x);
        axpy<TPG>(-alpha, av, r);
        const float rrn = br.sumf(dot<TPG>(r, r));
        const float beta = rrn * __builtin_amdgcn_rcpf(rr);
        rr = rrn;
        if (ah != nullptr && threadIdx.x == 0) {
            ah[(size_t)it * Bp] = alpha;
            bh[(size_t)it * Bp] = beta;
        }
        const bool bad = !(fabsf(rrn) <= 3.0e38f);      // NaN / Inf: report and stop this sample
        const bool conv = (double)rrn < tol2;           // sqrt(r.r) < CG_tol (ADMM.py:360) without the square root
        if (bad && threadIdx.x == 0) *nonfinite = 1;
        if (conv && !bad) iters = it + 1;
        done = bad || conv;
        xpby<TPG>(r, beta, pv);          // (also after the last iteration: a conditional update keeps two copies of p alive)
        c.template put<0>(c.P, pv);
    }
    return iters;
}

// MAXT: workgroup-size class the kernel is compiled for (register budget).  SB: single LDS vector (q = Ldr p replaces p
// in place, one more barrier per cLdr application); with the 640-thread class it is compiled for TWO resident
// workgroups per CU (5 waves per SIMD, <= 96 VGPRs): two samples in flight per CU overlap each other's barriers.
// NU / ND > 0: every row of W_u / W_d holds exactly that many entries (a kNN table without pads): unrolled gathers, the
// rows are read from the global image (L2) into registers once per solve and only the W_d^T tail table lives in LDS.
// SLOTS: two more LDS vectors park per-thread operands across the solves (see the trip body).
template <int TPG, bool BAND, int MAXT, bool SB, int NU = 0, int ND = 0, bool SLOTS = false, int TP = -1>
__global__ __launch_bounds__(MAXT, (SB && MAXT == 640) ? 5 : 1) void k_admm_lds(LdsArgs a_in) {
    constexpr bool ENTG = NU > 0 && ND > 0;       // table rows from the global image
    constexpr int NMRED = 12;                     // metric slots per wave (MGADMM_NMETRIC = 11)
    extern __shared__ __align__(16) unsigned char lds_raw[];
    float* P = reinterpret_cast<float*>(lds_raw);
    const int LN = a_in.NR * a_in.TS;                                  // floats per LDS vector (TS % 4 == 0 or TS == T)
    float* Q = SB ? P : P + LN;                                      // SB: one LDS vector serves p and q = Ldr p in turn
    float* red = Q + LN + ((4 - (LN & 3)) & 3);                       // 16-byte aligned: 2 x 16 floats of the CG reductions ...
    float* mred = red + 32;                                           // ... and 16 x NMRED wave totals of the metrics
    float* slot0 = mred + 16 * NMRED;
    const int nthr = (int)blockDim.x;
    float* slot1 = slot0 + (SLOTS ? nthr * TPG : 0);
    int* img = reinterpret_cast<int*>(slot1 + (SLOTS ? nthr * TPG : 0));
    int tid = threadIdx.x;
    int b = blockIdx.x;
    // ADMM outer loop without host round trips: iterations are enqueued ahead of the host's look at the stop test, a launch
    // that follows the stopping iteration must leave the state alone (workgroup-uniform scalar load)
    if (a_in.stop != nullptr && *a_in.stop != 0) return;
    // STAGGERED START.  Samples of one batch need (nearly) the same CG iteration counts, so the workgroups of a launch
    // run in lock step: all CUs store their results and request the next sample's operands at the same moment, while HBM
    // idles during the CG solves.  The workgroups of the first round (one per CU) start spread over `stagger_ticks`,
    // their successors inherit the offset: the operand requests of different CUs no longer coincide.
    if (a_in.stagger_ticks > 0 && b < a_in.stagger_wgs) {
        const unsigned long long wait = (unsigned long long)a_in.stagger_ticks * (unsigned)b / (unsigned)a_in.stagger_wgs;
        const unsigned long long t_in0 = wall_clock64();
        while (wall_clock64() - t_in0 < wait) __builtin_amdgcn_s_sleep(16);
    }
    // graph image -> LDS once per launch, eight words per thread in flight
    {
        const int* src = a_in.csr + a_in.lds_img0;
        const int nimg = a_in.lds_img_ints;
        for (int k0 = tid; k0 < nimg; k0 += 8 * nthr) {
            int wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u * nthr;
                wv[u] = src[k < nimg ? k : nimg - 1];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u * nthr;
                if (k < nimg) img[k] = wv[u];
            }
        }
    }
    for (int k = tid; k < 32 + 16 * NMRED; k += nthr) red[k] = 0.f;          // slots of non-existent waves must read as 0
    {   // the ghosts' rows of both images: zeros
        const int g0 = a_in.N * a_in.TS, gn = (a_in.NR - a_in.N) * a_in.TS;
        for (int k = tid; k < gn; k += nthr) { P[g0 + k] = 0.f; Q[g0 + k] = 0.f; }
    }
    // SEVERAL ADMM ITERATIONS PER LAUNCH (a_in.J trips): a workgroup keeps its sample.  Every thread reads back only state it
    // stored itself (no fence needed), the state of the samples in flight stays in L2 / Infinity Cache, x and gamma + rho phi
    // pass from one trip to the next in registers, the graph image is copied once and the staggered start and the drain of
    // the launch are paid once per J iterations.  The trip body must compile like a loop-free kernel: inside a loop the
    // compiler hoists the launch arguments, every element address and the index arithmetic out of the loop and keeps them
    // alive across the three solves (92 spilled SGPRs, 118 spilled VGPRs in round 3's first attempt).  So every trip re-reads
    // its arguments from the kernarg segment through a copy of the segment pointer the optimiser cannot see through, and
    // re-derives its indices from opaque copies of the thread and workgroup number.
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const LdsArgs __attribute__((address_space(4))) * KernargPtr;     // the kernarg segment: constant address space, scalar loads
    KernargPtr kp = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();       // LdsArgs is the kernel's only argument: offset 0
#else
    const LdsArgs* kp = &a_in;                                                  // (host pass of the single-source compile)
#endif
    const int J = a_in.J;
    float xc[TPG], vc[TPG];          // carried from trip to trip: the iterate and gamma + rho phi
#pragma unroll
    for (int k = 0; k < TPG; ++k) xc[k] = vc[k] = 0.f;
    for (int trip = 0; trip < J; ++trip) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(kp));
    asm volatile("" : "+v"(tid));
    asm volatile("" : "+s"(b));
    const LdsArgs __attribute__((address_space(4)))& a = *kp;
#else
    const LdsArgs& a = *kp;
#endif
    const bool first = a.first && trip == 0;      // phi = Ldr x0 is formed by the first trip of a cold start
    // diagnostic build (make EXTRA=-DMGADMM_PHASE_CLOCK): the per-sample metric slots receive the 100 MHz clock at the phase
    // boundaries of the trip instead of the metrics (tools/lds_phase_clock.py)
#ifdef MGADMM_PHASE_CLOCK
    unsigned long long stamps[MGADMM_NMETRIC];
#define MG_STAMP(m) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); stamps[m] = wall_clock64(); } while (0)
#else
#define MG_STAMP(m) do { } while (0)
#endif
    MG_STAMP(0);
    LdsCtx<TPG, BAND, NU, ND, TP> c;
    c.T = a.T; c.TS = a.TS; c.N = a.N;
    c.active = tid < a.nthreads;
    {
        const int g = c.active ? tid / a.N : 0;
        c.ig = c.active ? tid - g * a.N : 0;
        c.i = c.active ? c.ig : a.N + (tid - a.nthreads);
        c.t0 = g * TPG;
    }
    c.P = P; c.Q = Q;
    c.skip = a.skip; c.q1 = a.q1; c.band_w = a.band_w;
    c.tail_pairs = a.tail_pairs;
    // (k N 4 in scalar registers BEFORE the first exec-masked region: a uniform value first computed inside one -- the masked
    // stores of putg -- and used after it is kept in vector registers, and every element base after it with it)
#pragma unroll
    for (int k = 0; k < TPG; ++k) { c.kN4[k] = 4u * (unsigned)k * (unsigned)a.N; MG_PIN_S(c.kN4[k]); }
    c.so0 = c.active ? 4u * TPG * (unsigned)tid : 0u;
    {
        const int* tab = ENTG ? a.csr : img - a.lds_img0;          // where offsets into the image point
        c.en_u = reinterpret_cast<const i2v*>(tab + a.off_en_u);
        c.en_d = reinterpret_cast<const i2v*>(tab + a.off_en_d);
        c.lead_t = reinterpret_cast<const i2v*>(tab + a.off_lead_t);
        c.tail_t = reinterpret_cast<const i2v*>(img - a.lds_img0 + a.off_tail_t);
    }
    c.wdiag = mk2(0.f, 0.f);
    if constexpr (!BAND) {
        const float* dg = reinterpret_cast<const float*>(a.csr + a.off_diag);      // [NR] W_d diagonal, [NR] W_d^T diagonal
        c.wdiag = mk2(dg[c.i], dg[a.NR + c.i]);
    }
    // the thread's rows of the fixed-width tables (uniform instances), requested ahead of the phase that precedes their use
    constexpr bool UNI = NU > 0 && ND > 0;
    constexpr int NDO = ND > 0 ? ND - 1 : 0;
    typename LdsCtx<TPG, BAND, NU, ND, TP>::Rows R;
    // (pinned: the optimiser would otherwise re-read a table row from the global image inside the CG loop instead of keeping it)
    auto fetch_u = [&]() {
        if constexpr (UNI) {
            c.template load_rows<NU>(c.en_u, c.i * NU, P, R.u);
#pragma unroll
            for (int u = 0; u < NU; ++u) MG_PIN_V(R.u[u]);
        }
    };
    auto fetch_d = [&]() {
        if constexpr (UNI) {
            c.template load_rows<NDO>(c.en_d, c.i * NDO, P, R.d);
#pragma unroll
            for (int u = 0; u < NDO; ++u) MG_PIN_V(R.d[u]);
        }
    };
    auto fetch_t = [&]() {
        if constexpr (UNI) {
            c.template load_rows<LDS_NLEAD>(c.lead_t, c.i * LDS_NLEAD, Q, R.t);
#pragma unroll
            for (int u = 0; u < LDS_NLEAD; ++u) MG_PIN_V(R.t[u]);
        }
    };
    if (first) fetch_d();
    fetch_t();
    const size_t sb = (size_t)b * a.TN;
    const float* xo = kp->xs[trip] + sb;          // iteration k reads xs[k] and writes xs[k+1]
    float* xn = kp->xs[trip + 1] + sb;
    float *zu = a.zu + sb, *zd = a.zd + sb, *phi = a.phi + sb, *gam = a.gam + sb, *gu = a.gu + sb, *gd = a.gd + sb;
    const float* mk = a.mask ? a.mask + sb : nullptr;
    const int ty = a.mask ? a.T : a.t_in;
    const float* yb = a.y + (size_t)b * ty * a.N;
    const bool has_phi = a.has_phi, has_zd = a.has_zd;
    const float rho = a.rho, rho_u = a.rho_u, rho_d = a.rho_d;
    const int Nn = a.N;

    // REQUEST of two operand vectors (TPG elements each per thread): 2 * TPG unconditional loads issued together and
    // waited for together; nothing crosses the fence, so the destination registers of one request are all a batch needs.
    // (an operand the ablation does not use is read through a pointer to a vector that exists, ghosts read node 0, and the
    // values are selected afterwards: branch-free).  Element k of a state vector: uniform base + k N floats, ONE per-thread
    // byte offset for all of them (a register per element offset stayed alive through the whole trip: 16 VGPRs).
    // Addressing: uniform vector base (scalar register pair) + a 32-bit per-thread byte offset formed NEXT TO the access
    // (off0 + k N 4: one add) -- the scalar-base form of global_load / global_store.  An offset that is formed once and kept
    // is widened to 64 bits in another basic block, where instruction selection no longer sees that it fits the 32-bit
    // offset operand: every access then gets a 64-bit address in a register pair (16 VGPRs per vector, which went to scratch
    // and were reloaded one by one in front of the loads and stores of the state).
    auto request1 = [&](const float* A, float (&va)[TPG]) {
        unsigned o = c.glb0();
        MG_PIN_V(o);
#pragma unroll
        for (int k = 0; k < TPG; ++k) va[k] = ldg(A, o + c.kN4[k]);
        MG_REQ_FENCE();
    };
    // the same for the thread-major state vectors (zu, zd, phi, gamma*): TPG/4 16-byte loads per vector
    auto load_state = [&](const float* A, float (&va)[TPG]) {
        unsigned o = c.so0;
        MG_PIN_V(o);
        const __attribute__((address_space(1))) char* g = (const __attribute__((address_space(1))) char*)A;
        if constexpr (TPG % 4 == 0) {
#pragma unroll
            for (int j = 0; j < TPG / 4; ++j) {
                const lds_f4 q = *reinterpret_cast<const __attribute__((address_space(1))) lds_f4*>(g + o + 16 * j);
                va[4 * j] = q.x; va[4 * j + 1] = q.y; va[4 * j + 2] = q.z; va[4 * j + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < TPG; ++k) va[k] = *reinterpret_cast<const __attribute__((address_space(1))) float*>(g + o + 4 * k);
        }
    };
    auto request2s = [&](const float* A, const float* B, float (&va)[TPG], float (&vb)[TPG]) {
        load_state(A, va);
        load_state(B, vb);
        MG_REQ_FENCE();
    };
    auto request1s = [&](const float* A, float (&va)[TPG]) {
        load_state(A, va);
        MG_REQ_FENCE();
    };
    // y, masked: 0 for the rows past its end (t >= ty: read through offset 0) and for ghosts.  (Predicates on the time index are
    // written k < limit - t0: a per-element t0 + k is formed once, kept in TPG registers for the whole trip and spilled.)
    auto ylim_of = [&]() { return c.active ? ty - c.t0 : 0; };
    auto request_y = [&](float (&va)[TPG]) {
        const int ylim = ylim_of();
        unsigned o = 4u * (unsigned)(c.t0 * Nn + c.ig);
        MG_PIN_V(o);
#pragma unroll
        for (int k = 0; k < TPG; ++k) va[k] = ldg(yb, k < ylim ? o + c.kN4[k] : 0u);
        MG_REQ_FENCE();
#pragma unroll
        for (int k = 0; k < TPG; ++k) va[k] = k < ylim ? va[k] : 0.f;
    };

    // ---- operands of RHS_x (ADMM.py:556-564): o = (rho_u zu - gamma_u)/2 + (rho_d zd - gamma_d)/2 + H^T y, v = gamma + rho phi
    // and x_old.  First trip of a launch: from the state in HBM.  Later trips: x and v arrive in registers, o from its LDS
    // slot (SLOTS) or from the state this thread stored in the trip before.
    float x[TPG], o[TPG], v[TPG];
    if (trip == 0 || !SLOTS) {
        const float* zdp = has_zd ? zd : zu;
        const float* gdp = has_zd ? gd : gu;
        float ta[TPG], tb[TPG];
        request2s(zu, gu, ta, tb);
#pragma unroll
        for (int k = 0; k < TPG; ++k) o[k] = rhs_half(rho_u, ta[k], tb[k]);
        request2s(zdp, gdp, ta, tb);
#pragma unroll
        for (int k = 0; k < TPG; ++k) o[k] = has_zd ? o[k] + rhs_half(rho_d, ta[k], tb[k]) : o[k];
        request_y(ta);
#pragma unroll
        for (int k = 0; k < TPG; ++k) o[k] = c.active ? o[k] + ta[k] : 0.f;
    } else {
        slot_get<TPG>(slot1, tid, nthr, o);
    }
    if (trip == 0) {
        const bool use_v = has_phi && !first;
        const float* gamp = use_v ? gam : gu;
        const float* phip = use_v ? phi : zu;
        float ta[TPG], tb[TPG];
        request2s(gamp, phip, ta, tb);
#pragma unroll
        for (int k = 0; k < TPG; ++k) v[k] = (c.active && use_v) ? ldrt_operand(rho, tb[k], ta[k]) : 0.f;
        request1(xo, ta);
#pragma unroll
        for (int k = 0; k < TPG; ++k) x[k] = c.active ? ta[k] : 0.f;
    } else {
#pragma unroll
        for (int k = 0; k < TPG; ++k) { x[k] = xc[k]; v[k] = vc[k]; }
    }
    MG_STAMP(1);
    __syncthreads();          // graph image and zeroed rows (first trip); every LDS read of the previous trip is done
    c.u0 = c.u1 = c.d0 = c.d1 = 0;
    if constexpr (NU > 0) c.u0 = c.i * NU;
    else { c.u0 = img[a.off_rp_u - a.lds_img0 + c.i]; c.u1 = img[a.off_rp_u - a.lds_img0 + c.i + 1]; }
    if constexpr (!BAND) {
        if constexpr (ND > 0) c.d0 = c.i * NDO;
        else { c.d0 = img[a.off_rp_d - a.lds_img0 + c.i]; c.d1 = img[a.off_rp_d - a.lds_img0 + c.i + 1]; }
    }
    BlockRed br;
    br.red = red; br.par = 0; br.lane = tid & 63; br.wave = tid >> 6; br.nwaves = (nthr + 63) >> 6;
    // per-sample metric sums: wave totals (DPP) parked in LDS as soon as they are known (keeps no accumulator alive across
    // the CG solves), combined in a fixed order at the end of the trip; the whole-batch values are formed by k_batch_metrics
    auto mput = [&](int m, float val) {
        const float w = wave_sum(val);
        if (br.lane == 0) mred[br.wave * NMRED + m] = w;
    };

    // PHASE FENCE after every solve: the thread's index values become opaque again and what is derived from them is formed
    // anew.  Without it the optimiser keeps every address it formed before a solve for the phases after it (slot and state
    // offsets, table-row addresses, lane / wave numbers): values that are live across a CG loop they are not used in.
    auto refresh = [&]() {
        MG_PIN_V(tid);
        MG_PIN_V(c.i); MG_PIN_V(c.ig); MG_PIN_V(c.t0);
        c.active = tid < a.nthreads;
        c.so0 = c.active ? 4u * TPG * (unsigned)tid : 0u;
        br.lane = tid & 63; br.wave = tid >> 6;
    };

    // ---- first iteration only: phi = Ldr x0 (ADMM.py:541); the dual variables were filled by k_init_lds
    if (has_phi && first) {
        float ph[TPG];
        c.template put<-1>(P, x);
        __syncthreads();
        c.op_ldr(P, x, ph, R);
        c.puts(phi, ph);
        float gq[TPG];
        request1s(gam, gq);
#pragma unroll
        for (int k = 0; k < TPG; ++k) v[k] = c.active ? ldrt_operand(rho, ph[k], gq[k]) : 0.f;
        __syncthreads();
    }
    fetch_d();                // rows of the x solve

    // ---- RHS_x
    float rhs[TPG];
    {
        float l[TPG];
#pragma unroll
        for (int k = 0; k < TPG; ++k) l[k] = 0.f;
        if (has_phi) {
            c.template put<+1>(P, v);
            __syncthreads();
            c.op_ldrt(P, v, l, R);
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < TPG; ++k) rhs[k] = has_phi ? l[k] * 0.5f + o[k] : o[k];
    }
    if (SLOTS) slot_put<TPG>(slot0, tid, nthr, x);         // x_old, for the x-shift metric after the solve
    MG_STAMP(2);
    float* ah = a.record ? a.alpha_hist + b : nullptr;
    float* bh = a.record ? a.beta_hist + b : nullptr;
    int max_cg = a.max_cg;
    double tol = a.cg_tol2;                    // the solves compare r.r with CG_tol^2
    int Bp = a.Bp;
    int* nonfinite = a.nonfinite;
    MG_PIN_S(max_cg); MG_PIN_S(tol); MG_PIN_S(Bp); MG_PIN_S(nonfinite); MG_PIN_S(ah); MG_PIN_S(bh);
    const size_t hstride = (size_t)max_cg * Bp;

    // ---- x solve (ADMM.py:571)
    int itx;
    {
        float cx1 = a.cx1, cx2 = a.cx2;
        int t_in = a.t_in;
        MG_PIN_S(cx1); MG_PIN_S(cx2); MG_PIN_S(t_in);
        if (a.lhsx_kind == 1) itx = lds_cg<TPG, BAND, 1, SB, NU, ND, TP>(c, br, x, rhs, mk, 1, t_in, cx1, cx2, max_cg, tol, ah, bh, Bp, nonfinite, R);
        else itx = lds_cg<TPG, BAND, 0, SB, NU, ND, TP>(c, br, x, rhs, mk, 1, t_in, cx1, 0.f, max_cg, tol, ah, bh, Bp, nonfinite, R);
    }
    MG_STAMP(3);
    refresh();
    fetch_u();                // rows of the zu solve
    c.putg(xn, x);

    // ---- x metrics, operands of the zu solve
    float z[TPG];
    {
        float xold[TPG], yv[TPG], gq[TPG];
        request_y(yv);
        if (SLOTS) {
            request2s(gu, zu, gq, z);
            slot_get<TPG>(slot0, tid, nthr, xold);
            slot_put<TPG>(slot0, tid, nthr, x);            // x_new, for the updates after the zu / zd solves
        } else {
            request2s(gu, zu, gq, z);
            request1(xo, xold);
        }
        float m_xshift = 0.f, m_rec = 0.f;
        if (mk) {
            float mv[TPG];
            request1(mk, mv);
#pragma unroll
            for (int k = 0; k < TPG; ++k) {
                const float e = c.active ? x[k] * mv[k] - yv[k] : 0.f;
                m_rec += e * e;
            }
        } else {
#pragma unroll
            for (int k = 0; k < TPG; ++k) {
                const float e = k < ylim_of() ? x[k] - yv[k] : 0.f;
                m_rec += e * e;
            }
        }
#pragma unroll
        for (int k = 0; k < TPG; ++k) {
            const float dx = c.active ? x[k] - xold[k] : 0.f;
            m_xshift += dx * dx;
            z[k] = c.active ? z[k] : 0.f;
            gq[k] = c.active ? gq[k] : 0.f;
            rhs[k] = gq[k] * 0.5f + rho_u * 0.5f * x[k];      // x = the x_new just stored
        }
        mput(MGADMM_M_XSHIFT, m_xshift);
        mput(MGADMM_M_RECOVER, m_rec);
        if (SLOTS) {                       // gamma_u and the old zu wait in LDS for the update after the solve
            slot_put<TPG>(slot1, tid, nthr, gq);
            if (!SB) lds_store<TPG>(Q + c.own(), z);        // (the zu solve does not use Q)
        }
    }

    MG_STAMP(4);
    // ---- zu solve + gamma_u update (ADMM.py:579-580, 595)
    int itzu;
    {
        float c1 = rho_u * 0.5f, c2 = a.mu_u;
        MG_PIN_V(c1); MG_PIN_S(c2);
        itzu = lds_cg<TPG, BAND, 2, SB, NU, ND, TP>(c, br, z, rhs, nullptr, 0, 0, c1, c2, max_cg, tol, ah ? ah + hstride : nullptr,
                                                 bh ? bh + hstride : nullptr, Bp, nonfinite, R);
    }
    MG_STAMP(5);
    refresh();
    fetch_d();                // rows of the zd solve / of the Ldr of the phi prox
    if (has_zd) fetch_t();
    float xr[TPG], zn[TPG], gn[TPG];
    {
        float zo[TPG], gv[TPG];
        // operands of the next phase (zd solve, or the phi prox)
        const float* znp = has_zd ? zd : (has_phi ? phi : zu);
        const float* gnp = has_zd ? gd : (has_phi ? gam : gu);
        request2s(znp, gnp, zn, gn);
        if (SLOTS && !SB) {
            slot_get<TPG>(slot0, tid, nthr, xr);
            slot_get<TPG>(slot1, tid, nthr, gv);
            lds_load<TPG>(Q + c.own(), zo);
        } else {
            request2s(zu, gu, zo, gv);
            if (SLOTS) slot_get<TPG>(slot0, tid, nthr, xr);
            else request1(xn, xr);
        }
        float m_pri = 0.f, m_dual = 0.f;
        float ou[TPG];
#pragma unroll
        for (int k = 0; k < TPG; ++k) {
            xr[k] = c.active ? xr[k] : 0.f;
            zo[k] = c.active ? zo[k] : 0.f;
            gv[k] = c.active ? gv[k] : 0.f;
            zn[k] = c.active ? zn[k] : 0.f;
            gn[k] = c.active ? gn[k] : 0.f;
            const float pz = xr[k] - z[k], dz = z[k] - zo[k];
            m_pri += pz * pz;
            m_dual += dz * dz;
            gv[k] = gv[k] + rho_u * pz;
            ou[k] = rhs_half(rho_u, z[k], gv[k]);
        }
        c.puts(zu, z);
        c.puts(gu, gv);
        if (SLOTS) slot_put<TPG>(slot1, tid, nthr, ou);
        mput(MGADMM_M_PRI_ZU, m_pri);
        mput(MGADMM_M_DUAL_ZU, m_dual);
    }
    MG_STAMP(6);
    // ---- zd solve + gamma_d update (ADMM.py:586-588, 597)
    int itzd = 0;
    if (has_zd) {
#pragma unroll
        for (int k = 0; k < TPG; ++k) {
            z[k] = zn[k];
            rhs[k] = gn[k] * 0.5f + rho_d * 0.5f * xr[k];
        }
        {
            float c1 = rho_d * 0.5f, c2 = a.mu_d2;
            MG_PIN_V(c1); MG_PIN_S(c2);
            itzd = lds_cg<TPG, BAND, 1, SB, NU, ND, TP>(c, br, z, rhs, nullptr, 0, 0, c1, c2, max_cg, tol, ah ? ah + 2 * hstride : nullptr,
                                                     bh ? bh + 2 * hstride : nullptr, Bp, nonfinite, R);
        }
        MG_STAMP(7);
        refresh();
        fetch_u();            // rows of the Lu of the GLR term
        fetch_d();            // the W_d rows of the prox's Ldr again: rows that stay across the solve cannot be pinned in place inside
                              // its loop -- the loop then works on copies and both sets are alive (8 registers)
        float zo[TPG], gv[TPG];
        request2s(zd, gd, zo, gv);
        // (unconditional, through vectors that exist: a conditional request keeps the zn / gn of before the solve alive across
        // it -- 16 registers; without the phi term the prox below computes values nobody reads)
        request2s(has_phi ? phi : zd, has_phi ? gam : gd, zn, gn);
        if (SLOTS) slot_get<TPG>(slot0, tid, nthr, xr);
        else request1(xn, xr);
        float m_pri = 0.f, m_dual = 0.f;
#pragma unroll
        for (int k = 0; k < TPG; ++k) {
            xr[k] = c.active ? xr[k] : 0.f;
            zo[k] = c.active ? zo[k] : 0.f;
            gv[k] = c.active ? gv[k] : 0.f;
            zn[k] = c.active ? zn[k] : 0.f;
            gn[k] = c.active ? gn[k] : 0.f;
            const float pz = xr[k] - z[k], dz = z[k] - zo[k];
            m_pri += pz * pz;
            m_dual += dz * dz;
            gv[k] = gv[k] + rho_d * pz;
        }
        c.puts(zd, z);
        c.puts(gd, gv);
        if (SLOTS) {                        // o of the next trip: (o_u + o_d) + H^T y
            float ou[TPG], yv[TPG];
            request_y(yv);
            slot_get<TPG>(slot1, tid, nthr, ou);
#pragma unroll
            for (int k = 0; k < TPG; ++k) ou[k] = c.active ? (ou[k] + rhs_half(rho_d, z[k], gv[k])) + yv[k] : 0.f;
            slot_put<TPG>(slot1, tid, nthr, ou);
        }
        mput(MGADMM_M_PRI_ZD, m_pri);
        mput(MGADMM_M_DUAL_ZD, m_dual);
    } else {
        fetch_u();
        if (SLOTS) {                        // o of the next trip: o_u + H^T y
            float ou[TPG], yv[TPG];
            request_y(yv);
            slot_get<TPG>(slot1, tid, nthr, ou);
#pragma unroll
            for (int k = 0; k < TPG; ++k) ou[k] = c.active ? ou[k] + yv[k] : 0.f;
            slot_put<TPG>(slot1, tid, nthr, ou);
        }
        mput(MGADMM_M_PRI_ZD, 0.f);
        mput(MGADMM_M_DUAL_ZD, 0.f);
    }

    MG_STAMP(8);
    // ---- phi prox, gamma update, Ldr/Lu based diagnostics (ADMM.py:600-606, 619, 627-637); xr = x_new, zn = phi_old, gn = gamma
    float m_priphi = 0.f, m_dualphi = 0.f, m_dgtv = 0.f, m_dglr = 0.f, m_glr = 0.f;
    __syncthreads();          // every LDS read of the last CG is done
    c.template put<-1>(P, xr);        // read by Ldr
    if (!SB) c.template put<0>(Q, xr);   // read by Lu (GLR); SB: Q aliases P, see below
    __syncthreads();
    {
        float l[TPG];
        c.op_ldr(P, xr, l, R);
        const float thr = a.mu_d1 / rho;
        float pn[TPG], gnew[TPG];
#pragma unroll
        for (int k = 0; k < TPG; ++k) {
            m_dgtv += fabsf(l[k]);
            m_dglr += l[k] * l[k];
            const float gv = gn[k];
            const float s = l[k] - gv / rho;
            const float u = fabsf(s) - thr;
            pn[k] = (u > 0.f) ? (s > 0.f ? u : -u) : 0.f;
            const float dd = pn[k] - l[k], dp = pn[k] - zn[k];
            m_priphi += dd * dd;
            m_dualphi += dp * dp;
            gnew[k] = gv + rho * dd;
            vc[k] = has_phi ? ldrt_operand(rho, pn[k], gnew[k]) : 0.f;       // v of the next trip (ghosts: gn = zn = l = 0 -> 0)
            xc[k] = xr[k];                                                    // x_old of the next trip
        }
        if (has_phi) {
            c.puts(phi, pn);
            c.puts(gam, gnew);
        }
        if (!SB) {
            c.op_lu(Q, xr, l, R);
#pragma unroll
            for (int k = 0; k < TPG; ++k) m_glr += xr[k] * l[k];
        }
    }
    if (SB) {                 // one LDS vector: the unshifted image replaces the shifted one after every Ldr gather is done
        __syncthreads();
        c.template put<0>(P, xr);
        __syncthreads();
        float l[TPG];
        c.op_lu(P, xr, l, R);
#pragma unroll
        for (int k = 0; k < TPG; ++k) m_glr += xr[k] * l[k];
    }
    mput(MGADMM_M_PRI_PHI, has_phi ? m_priphi : 0.f);
    mput(MGADMM_M_DUAL_PHI, has_phi ? m_dualphi : 0.f);
    mput(MGADMM_M_DGTV, has_phi ? m_dgtv : 0.f);
    mput(MGADMM_M_DGLR, has_zd ? m_dglr : 0.f);
    mput(MGADMM_M_GLR, m_glr);
    MG_STAMP(9);
    __syncthreads();          // wave totals of every metric are in LDS; the P / Q reads of this trip are done
    if (tid < MGADMM_NMETRIC) {
        // fixed association: four partial sums over the waves w = j, j+4, j+8, j+12, then (s0 + s1) + (s2 + s3)
        float sj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            sj[j] = (mred[j * NMRED + tid] + mred[(j + 4) * NMRED + tid]) + (mred[(j + 8) * NMRED + tid] + mred[(j + 12) * NMRED + tid]);
        double* ps = a.ps + (size_t)trip * MGADMM_NMETRIC * a.Bp;     // per-sample metric sums of this iteration
#ifndef MGADMM_PHASE_CLOCK
        ps[(size_t)tid * a.Bp + b] = (double)((sj[0] + sj[1]) + (sj[2] + sj[3]));
#else
        (void)sj; (void)ps;
#endif
    }
#ifdef MGADMM_PHASE_CLOCK
    MG_STAMP(10);
    if (tid == 0) {
        double* ps = a.ps + (size_t)trip * MGADMM_NMETRIC * a.Bp;
        for (int m = 0; m < MGADMM_NMETRIC; ++m) ps[(size_t)m * a.Bp + b] = (double)stamps[m];
    }
#endif
#undef MG_STAMP
    if (tid == 0) {
        int* const cgi = a.cg_iters + (size_t)trip * 3 * a.Bp;
        cgi[b] = itx;
        cgi[a.Bp + b] = itzu;
        cgi[2 * a.Bp + b] = itzd;
    }
    }       // trips
}

// initial state in sample-major layout: x0 (regression), zu = zd = x0, gamma* = 0.1   (ADMM.py:528-544)
// element (t, i) of a sample in the THREAD-MAJOR layout of the LDS path's state vectors (zu, zd, phi, gamma*): the TPG elements
// of thread (g, i) are contiguous, threads in thread order -- one thread reads / writes a vector as TPG/4 16-byte accesses
// (global_load_dwordx4, lanes 48 B apart at TPG = 12) instead of TPG 4-byte ones in the reference's (T, N) order.  The 4-byte
// form is bound by the rate at which a CU takes vector-memory instructions (64 lanes x 4 B per instruction: 38 GB/s per CU,
// 200 such instructions per thread and ADMM iteration for the state); x, y and the mask keep the reference's layout.
__host__ __device__ __forceinline__ int lds_state_index(int t, int i, int N, int TPG) { return ((t / TPG) * N + i) * TPG + t % TPG; }

template <bool MASKED>
__global__ __launch_bounds__(256) void k_init_lds(int T, int t_in, int N, int TPG, int B, float tm, float den, const float* __restrict__ y,
                                                  const float* __restrict__ mask, float* __restrict__ x, float* __restrict__ zu,
                                                  float* __restrict__ zd, float* __restrict__ gam, float* __restrict__ gu,
                                                  float* __restrict__ gd, int* __restrict__ nonfinite) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= N) return;
    const size_t sb = (size_t)b * T * N;
    float w, c0;
    const float* yb = MASKED ? y + sb : y + (size_t)b * t_in * N;
    const float* mb = MASKED ? mask + sb : nullptr;
    if (!MASKED) {
        float sy = 0.f, sty = 0.f;
        for (int t = 0; t < t_in; ++t) {
            const float v = yb[t * N + i];
            sy += v;
            sty += (float)t * v;
        }
        const float ym = sy / (float)t_in;
        w = (sty / (float)t_in - tm * ym) / den;
        c0 = ym - w * tm;
    } else {
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
    }
    auto value = [&](int t) {
        if (!MASKED) return (t < t_in) ? yb[t * N + i] : w * (float)t + c0;
        return (w * (float)t + c0) * (1.f - mb[t * N + i]) + yb[t * N + i];
    };
    // a thread writes the TPG elements of one (time group, node) of the thread-major vectors as one contiguous run (consecutive
    // nodes: consecutive runs), x in the reference's layout element by element (consecutive nodes: consecutive addresses)
    for (int t0 = 0; t0 < T; t0 += TPG) {
        const size_t es = sb + (size_t)((t0 / TPG) * N + i) * TPG;
        if (TPG % 4 == 0) {
            for (int k = 0; k < TPG; k += 4) {
                lds_f4 q;
                q.x = value(t0 + k); q.y = value(t0 + k + 1); q.z = value(t0 + k + 2); q.w = value(t0 + k + 3);
                x[sb + (size_t)(t0 + k) * N + i] = q.x; x[sb + (size_t)(t0 + k + 1) * N + i] = q.y;
                x[sb + (size_t)(t0 + k + 2) * N + i] = q.z; x[sb + (size_t)(t0 + k + 3) * N + i] = q.w;
                *reinterpret_cast<lds_f4*>(zu + es + k) = q;
                *reinterpret_cast<lds_f4*>(zd + es + k) = q;
                lds_f4 c; c.x = c.y = c.z = c.w = 0.1f;
                *reinterpret_cast<lds_f4*>(gam + es + k) = c;
                *reinterpret_cast<lds_f4*>(gu + es + k) = c;
                *reinterpret_cast<lds_f4*>(gd + es + k) = c;
            }
        } else {
            for (int k = 0; k < TPG; ++k) {
                const float v = value(t0 + k);
                x[sb + (size_t)(t0 + k) * N + i] = v;
                zu[es + k] = v; zd[es + k] = v;
                gam[es + k] = 0.1f; gu[es + k] = 0.1f; gd[es + k] = 0.1f;
            }
        }
    }
}

// a state vector between the reference's (B, T, N) layout and the thread-major layout (warm start in, exported state out)
template <bool TO_THREAD_MAJOR>
__global__ __launch_bounds__(256) void k_state_layout(int T, int N, int TPG, const float* __restrict__ src, float* __restrict__ dst) {
    const int e = blockIdx.x * 256 + threadIdx.x;        // index in the reference layout
    if (e >= T * N) return;
    const size_t sb = (size_t)blockIdx.y * T * N;
    const int t = e / N, i = e - t * N;
    const int es = lds_state_index(t, i, N, TPG);
    if (TO_THREAD_MAJOR) dst[sb + es] = src[sb + e];
    else dst[sb + e] = src[sb + es];
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

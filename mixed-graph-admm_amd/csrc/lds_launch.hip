// LDS-resident fused ADMM path: kernel instantiations and launches (own translation unit: the library builds in parallel).
#include <mutex>
#include <utility>
#include <vector>

#include <cstdint>
#include "lds_kernels.h"

namespace {

std::mutex g_attr_mu;
std::vector<std::pair<const void*, int>> g_attr_done;     // (kernel, device) pairs whose dynamic-LDS limit has been raised

int allow_lds(const void* fn, int bytes) {
    int dev = 0;
    MG_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_attr_mu);
    for (auto& e : g_attr_done)
        if (e.first == fn && e.second == dev) return MGADMM_OK;
    MG_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));   // a per-DEVICE property
    g_attr_done.push_back({fn, dev});
    return MGADMM_OK;
}

template <int TPG, bool BAND, int MAXT, bool SB, int NU = 0, int ND = 0, bool SLOTS = false, int TP = -1>
int launch(const LdsLaunch& L, const LdsArgs& a, int B, hipStream_t st) {
    auto fn = k_admm_lds<TPG, BAND, MAXT, SB, NU, ND, SLOTS, TP>;
    MG_TRY(allow_lds((const void*)fn, 160 * 1024));
    hipLaunchKernelGGL(fn, dim3(B), dim3(L.block), L.lds_bytes, st, a);
    MG_HIP(hipGetLastError());
    return MGADMM_OK;
}

template <int TPG, int MAXT, bool SB>
int launch_b(const LdsLaunch& L, const LdsArgs& a, int B, hipStream_t st) {
    return a.band ? launch<TPG, true, MAXT, SB>(L, a, B, st) : launch<TPG, false, MAXT, SB>(L, a, B, st);
}

}  // namespace

int mg_lds_iteration(const LdsLaunch& L, const LdsArgs& a, int B, hipStream_t st) {
    if (a.J < 1 || a.J > LDS_MAXJ || a.NR * 1 < a.N || L.block - a.nthreads != a.NR - a.N) {
        mg_set_error("lds: launch geometry (J %d, rows %d for %d nodes, %d of %d threads own elements)", a.J, a.NR, a.N, a.nthreads, L.block);
        return MGADMM_ERR_INVALID;
    }
    if (L.uniform45 && (a.band || L.sb || !((L.tpg == 8 && L.maxt == 1024) || L.tpg == 12))) {
        mg_set_error("lds: the uniform-row instances exist for TPG 8 (1024-thread class) and TPG 12");
        return MGADMM_ERR_UNSUPPORTED;
    }
    if (L.uniform45 && L.tpg == 12 && L.maxt == 1024) {        // 342 .. 512 nodes: two time groups in a workgroup of up to 1024 threads (no room for slots)
        if (L.slots) { mg_set_error("lds: no slot instance in the 1024-thread class of TPG 12"); return MGADMM_ERR_UNSUPPORTED; }
        switch (a.tail_pairs) {
            case 0: return launch<12, false, 1024, false, 4, 5, false, 0>(L, a, B, st);
            case 1: return launch<12, false, 1024, false, 4, 5, false, 1>(L, a, B, st);
            case 2: return launch<12, false, 1024, false, 4, 5, false, 2>(L, a, B, st);
            case 3: return launch<12, false, 1024, false, 4, 5, false, 3>(L, a, B, st);
        }
        return launch<12, false, 1024, false, 4, 5, false, -1>(L, a, B, st);        // longer tails: pair count at run time
    }
    if (L.uniform45 && L.tpg == 12) {
        switch (a.tail_pairs * 2 + (L.slots ? 1 : 0)) {
            case 0: return launch<12, false, 640, false, 4, 5, false, 0>(L, a, B, st);
            case 1: return launch<12, false, 640, false, 4, 5, true, 0>(L, a, B, st);
            case 2: return launch<12, false, 640, false, 4, 5, false, 1>(L, a, B, st);
            case 3: return launch<12, false, 640, false, 4, 5, true, 1>(L, a, B, st);
            case 4: return launch<12, false, 640, false, 4, 5, false, 2>(L, a, B, st);
            case 5: return launch<12, false, 640, false, 4, 5, true, 2>(L, a, B, st);
            case 6: return launch<12, false, 640, false, 4, 5, false, 3>(L, a, B, st);
            case 7: return launch<12, false, 640, false, 4, 5, true, 3>(L, a, B, st);
        }
        return L.slots ? launch<12, false, 640, false, 4, 5, true, -1>(L, a, B, st) : launch<12, false, 640, false, 4, 5, false, -1>(L, a, B, st);
    }
    if (L.sb) {
        if (L.tpg == 12 && L.maxt == 640) return launch_b<12, 640, true>(L, a, B, st);
        if (L.tpg == 8 && L.maxt == 1024) return launch_b<8, 1024, true>(L, a, B, st);
        mg_set_error("lds: single-buffer mode exists for TPG 12 (<= 640 threads) and TPG 8 only");
        return MGADMM_ERR_UNSUPPORTED;
    }
    if (L.maxt == 640 && L.tpg == 12) return launch_b<12, 640, false>(L, a, B, st);
    switch (L.tpg) {
        case 1: return launch_b<1, 1024, false>(L, a, B, st);
        case 2: return launch_b<2, 1024, false>(L, a, B, st);
        case 3: return launch_b<3, 1024, false>(L, a, B, st);
        case 4: return launch_b<4, 1024, false>(L, a, B, st);
        case 6: return launch_b<6, 1024, false>(L, a, B, st);
        case 8:
            if (L.uniform45 && !a.band) {       // uniform-row instances: the pair count of the W_d^T tail table is a compile-time constant
                switch (a.tail_pairs * 2 + (L.slots ? 1 : 0)) {
                    case 0: return launch<8, false, 1024, false, 4, 5, false, 0>(L, a, B, st);
                    case 1: return launch<8, false, 1024, false, 4, 5, true, 0>(L, a, B, st);
                    case 2: return launch<8, false, 1024, false, 4, 5, false, 1>(L, a, B, st);
                    case 3: return launch<8, false, 1024, false, 4, 5, true, 1>(L, a, B, st);
                    case 4: return launch<8, false, 1024, false, 4, 5, false, 2>(L, a, B, st);
                    case 5: return launch<8, false, 1024, false, 4, 5, true, 2>(L, a, B, st);
                    case 6: return launch<8, false, 1024, false, 4, 5, false, 3>(L, a, B, st);
                    case 7: return launch<8, false, 1024, false, 4, 5, true, 3>(L, a, B, st);
                }
                // longer tails: pair count at run time
                return L.slots ? launch<8, false, 1024, false, 4, 5, true, -1>(L, a, B, st) : launch<8, false, 1024, false, 4, 5, false, -1>(L, a, B, st);
            }
            return launch_b<8, 1024, false>(L, a, B, st);
        case 12: return launch_b<12, 1024, false>(L, a, B, st);
    }
    mg_set_error("lds: no kernel for TPG %d", L.tpg);
    return MGADMM_ERR_UNSUPPORTED;
}

int mg_lds_init(bool masked, int T, int t_in, int N, int TPG, int B, float tm, float den, const float* y, const float* mask, float* x,
                float* zu, float* zd, float* gam, float* gu, float* gd, int* nonfinite, hipStream_t st) {
    dim3 grid((N + 255) / 256, B);
    if (masked) hipLaunchKernelGGL((k_init_lds<true>), grid, dim3(256), 0, st, T, t_in, N, TPG, B, tm, den, y, mask, x, zu, zd, gam, gu, gd, nonfinite);
    else hipLaunchKernelGGL((k_init_lds<false>), grid, dim3(256), 0, st, T, t_in, N, TPG, B, tm, den, y, (const float*)nullptr, x, zu, zd, gam, gu, gd, nonfinite);
    MG_HIP(hipGetLastError());
    return MGADMM_OK;
}

int mg_lds_state_layout(bool to_thread_major, int T, int N, int TPG, int B, const float* src, float* dst, hipStream_t st) {
    dim3 grid((T * N + 255) / 256, B);
    if (to_thread_major) hipLaunchKernelGGL((k_state_layout<true>), grid, dim3(256), 0, st, T, N, TPG, src, dst);
    else hipLaunchKernelGGL((k_state_layout<false>), grid, dim3(256), 0, st, T, N, TPG, src, dst);
    MG_HIP(hipGetLastError());
    return MGADMM_OK;
}

int mg_lds_dxps(int T, int N, int B, const float* x, const float* xo, double* scratch, double* out, const int* stop, hipStream_t st) {
    const int TN = T * N, nsl = (B + 63) / 64;
    if (TN % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)xo % 16) == 0)
        hipLaunchKernelGGL(k_dxps_sm4, dim3((TN / 4 + 63) / 64, nsl), dim3(64), 0, st, TN, B, x, xo, scratch + TN, stop);
    else
        hipLaunchKernelGGL(k_dxps_sm, dim3((TN + 255) / 256, nsl), dim3(256), 0, st, TN, B, x, xo, scratch + TN, stop);
    hipLaunchKernelGGL(k_dxps_sm_mean, dim3((TN + 255) / 256), dim3(256), 0, st, TN, B, nsl, (const double*)(scratch + TN), scratch, stop);
    hipLaunchKernelGGL(k_dxps_sm_final, dim3(T), dim3(256), 0, st, T, N, (const double*)scratch, out, stop);
    MG_HIP(hipGetLastError());
    return MGADMM_OK;
}

int mg_lds_stop_test(const double* metrics_row, const int* nonfinite, int has_phi, int has_zd, double tol, int it, int* stop, hipStream_t st) {
    hipLaunchKernelGGL(k_lds_stop_test, dim3(1), dim3(64), 0, st, metrics_row, nonfinite, has_phi, has_zd, tol, it, stop);
    MG_HIP(hipGetLastError());
    return MGADMM_OK;
}

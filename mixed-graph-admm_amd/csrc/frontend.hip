// Data front end on the GPU: what TrafficDataset does to the (time, node) series before the
// solver sees it (utils.py:54-134) -- per-node statistics, standardize / normalize (and recover_data), and the
// sliding windows data[i : i + 24] / data[i : i + 12] of get_predict_data / get_interpolated_data, for a whole
// batch of start indices at once.  HBM-bound byte work: coalesced over the node axis, fixed-order reductions.
#include <algorithm>

#include "common.h"

namespace {

constexpr int ST_ROWS = 256;    // time steps per partial (one workgroup row chunk)

// Stage 1: per (row chunk, column) partial min / max / sum.  Block = 64 columns x 4 row lanes; a row lane walks
// its quarter of the chunk, lanes are combined through LDS in fixed order.
template <typename S>
__global__ __launch_bounds__(256) void k_col_partials(long n_steps, int n_cols, const S* __restrict__ x, const double* __restrict__ mean,
                                                      double* __restrict__ psum, S* __restrict__ pmin, S* __restrict__ pmax) {
    __shared__ double s_sum[4][64];
    __shared__ S s_min[4][64], s_max[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cx;
    const long r0 = (long)blockIdx.y * ST_ROWS, r1 = min(r0 + ST_ROWS, n_steps);
    double sum = 0.0;
    S mn = 0, mx = 0;
    bool any = false;
    if (col < n_cols) {
        const double m = mean ? mean[col] : 0.0;
        for (long r = r0 + ry; r < r1; r += 4) {
            const S v = x[r * n_cols + col];
            if (mean) {                       // second pass: sum of squared deviations
                const double d = (double)v - m;
                sum += d * d;
            } else {
                sum += (double)v;
                mn = any ? min(mn, v) : v;
                mx = any ? max(mx, v) : v;
                any = true;
            }
        }
    }
    s_sum[ry][cx] = sum; s_min[ry][cx] = mn; s_max[ry][cx] = mx;
    __shared__ int s_any[4][64];
    s_any[ry][cx] = any;
    __syncthreads();
    if (ry == 0 && col < n_cols) {
        double t = s_sum[0][cx];
        S a = s_min[0][cx], b = s_max[0][cx];
        bool have = s_any[0][cx];
        for (int j = 1; j < 4; ++j) {
            t += s_sum[j][cx];
            if (s_any[j][cx]) {
                a = have ? min(a, s_min[j][cx]) : s_min[j][cx];
                b = have ? max(b, s_max[j][cx]) : s_max[j][cx];
                have = true;
            }
        }
        const size_t o = (size_t)blockIdx.y * n_cols + col;
        psum[o] = t;
        if (!mean) { pmin[o] = a; pmax[o] = b; }
    }
}

// Stage 2: combine the row-chunk partials of a column in chunk order.
//   mode 0: mean = sum / n, min, max          mode 1: std = sqrt(sum_sq_dev / (n - 1))  (torch.std: unbiased)
template <typename S>
__global__ __launch_bounds__(256) void k_col_finish(int nchunk, long n_steps, int n_cols, int mode, const double* __restrict__ psum,
                                                    const S* __restrict__ pmin, const S* __restrict__ pmax, double* __restrict__ mean_d,
                                                    S* __restrict__ o_min, S* __restrict__ o_max, S* __restrict__ o_mean, S* __restrict__ o_std) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= n_cols) return;
    double t = 0.0;
    for (int c = 0; c < nchunk; ++c) t += psum[(size_t)c * n_cols + col];
    if (mode == 0) {
        S a = pmin[col], b = pmax[col];
        for (int c = 1; c < nchunk; ++c) {
            a = min(a, pmin[(size_t)c * n_cols + col]);
            b = max(b, pmax[(size_t)c * n_cols + col]);
        }
        const double m = t / (double)n_steps;
        mean_d[col] = m;
        if (o_min) o_min[col] = a;
        if (o_max) o_max[col] = b;
        if (o_mean) o_mean[col] = (S)m;
    } else if (o_std) {
        o_std[col] = (S)sqrt(t / (double)(n_steps - 1));
    }
}

// x = (x - shift[col]) / scale[col]   (forward)      x = x * scale[col] + shift[col]   (inverse = recover_data)
template <typename S>
__global__ __launch_bounds__(256) void k_affine(long total, int n_cols, int inverse, const S* __restrict__ shift, const S* __restrict__ scale,
                                                const S* __restrict__ scale_lo, S* __restrict__ x) {
#pragma clang fp contract(off)
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % n_cols);
        const S sc = scale_lo ? scale[c] - scale_lo[c] : scale[c];      // 'normalize': max - min
        x[i] = inverse ? x[i] * sc + shift[c] : (x[i] - shift[c]) / sc;
    }
}

// out[b, t, c] = series[starts[b] + t, c] (* mask[t, c])     windows are contiguous slabs of the series
template <typename S>
__global__ __launch_bounds__(256) void k_windows(long n_steps, int n_cols, int win, const S* __restrict__ series, const long long* __restrict__ starts,
                                                 const float* __restrict__ mask, S* __restrict__ out, int* __restrict__ bad) {
    const int b = blockIdx.y;
    const long long s0 = starts[b];
    if (s0 < 0 || s0 + win > n_steps) {
        if (threadIdx.x == 0 && blockIdx.x == 0) *bad = 1;
        return;
    }
    const long per = (long)win * n_cols;
    const S* src = series + s0 * n_cols;
    S* dst = out + (size_t)b * per;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < per; i += (long)gridDim.x * 256) {
        S v = src[i];
        if (mask) v = v * (S)mask[i];
        dst[i] = v;
    }
}

template <typename S>
int series_stats(const S* x, int64_t n_steps, int n_cols, S* o_min, S* o_max, S* o_mean, S* o_std, hipStream_t st) {
    const int nchunk = (int)((n_steps + ST_ROWS - 1) / ST_ROWS);
    double *psum = nullptr, *mean_d = nullptr;
    S *pmin = nullptr, *pmax = nullptr;
    MG_HIP(hipMallocAsync((void**)&psum, sizeof(double) * (size_t)nchunk * n_cols, st));
    MG_HIP(hipMallocAsync((void**)&mean_d, sizeof(double) * n_cols, st));
    MG_HIP(hipMallocAsync((void**)&pmin, sizeof(S) * (size_t)nchunk * n_cols, st));
    MG_HIP(hipMallocAsync((void**)&pmax, sizeof(S) * (size_t)nchunk * n_cols, st));
    const dim3 g1((n_cols + 63) / 64, nchunk), g2((n_cols + 255) / 256);
    hipLaunchKernelGGL(k_col_partials<S>, g1, dim3(256), 0, st, (long)n_steps, n_cols, x, (const double*)nullptr, psum, pmin, pmax);
    hipLaunchKernelGGL(k_col_finish<S>, g2, dim3(256), 0, st, nchunk, (long)n_steps, n_cols, 0, psum, pmin, pmax, mean_d, o_min, o_max,
                       o_mean, (S*)nullptr);
    if (o_std) {
        hipLaunchKernelGGL(k_col_partials<S>, g1, dim3(256), 0, st, (long)n_steps, n_cols, x, (const double*)mean_d, psum, pmin, pmax);
        hipLaunchKernelGGL(k_col_finish<S>, g2, dim3(256), 0, st, nchunk, (long)n_steps, n_cols, 1, psum, pmin, pmax, mean_d, (S*)nullptr,
                           (S*)nullptr, (S*)nullptr, o_std);
    }
    MG_HIP(hipGetLastError());
    MG_HIP(hipFreeAsync(psum, st));
    MG_HIP(hipFreeAsync(mean_d, st));
    MG_HIP(hipFreeAsync(pmin, st));
    MG_HIP(hipFreeAsync(pmax, st));
    return MGADMM_OK;
}

}  // namespace

// The entry points below take raw device pointers and a stream but no device ordinal: make the device that owns
// `series` current (a caller with several GPUs in one process may have another one selected).
static int use_device_of(const void* ptr, const char* who) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, ptr) != hipSuccess) {
        (void)hipGetLastError();
        mg_set_error("%s: `series` is not a HIP device pointer", who);
        return MGADMM_ERR_INVALID;
    }
    MG_HIP(hipSetDevice(at.device));
    return MGADMM_OK;
}

extern "C" int mgadmm_series_stats(const void* series, int64_t n_steps, int32_t n_cols, int32_t dtype, void* o_min, void* o_max,
                                   void* o_mean, void* o_std, void* stream) {
    MG_REQUIRE(series && n_steps >= 1 && n_cols >= 1, "series_stats: bad arguments");
    MG_REQUIRE(!(o_std && n_steps < 2), "series_stats: the unbiased std needs at least 2 time steps");
    MG_TRY(use_device_of(series, "series_stats"));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MGADMM_F64)
        return series_stats<double>((const double*)series, n_steps, n_cols, (double*)o_min, (double*)o_max, (double*)o_mean, (double*)o_std, st);
    if (dtype == MGADMM_F32)
        return series_stats<float>((const float*)series, n_steps, n_cols, (float*)o_min, (float*)o_max, (float*)o_mean, (float*)o_std, st);
    mg_set_error("series_stats: bad dtype %d", dtype);
    return MGADMM_ERR_INVALID;
}

extern "C" int mgadmm_series_affine(void* series, int64_t n_steps, int32_t n_cols, int32_t dtype, const void* shift, const void* scale,
                                    const void* scale_lo, int32_t inverse, void* stream) {
    MG_REQUIRE(series && shift && scale && n_steps >= 1 && n_cols >= 1, "series_affine: bad arguments");
    MG_TRY(use_device_of(series, "series_affine"));
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)n_steps * n_cols;
    const dim3 grid((unsigned)std::min<long>((total + 255) / 256, 8192));
    if (dtype == MGADMM_F64)
        hipLaunchKernelGGL(k_affine<double>, grid, dim3(256), 0, st, total, n_cols, inverse, (const double*)shift, (const double*)scale,
                           (const double*)scale_lo, (double*)series);
    else if (dtype == MGADMM_F32)
        hipLaunchKernelGGL(k_affine<float>, grid, dim3(256), 0, st, total, n_cols, inverse, (const float*)shift, (const float*)scale,
                           (const float*)scale_lo, (float*)series);
    else {
        mg_set_error("series_affine: bad dtype %d", dtype);
        return MGADMM_ERR_INVALID;
    }
    MG_HIP(hipGetLastError());
    return MGADMM_OK;
}

extern "C" int mgadmm_gather_windows(const void* series, int64_t n_steps, int32_t n_cols, int32_t dtype, const int64_t* starts, int32_t B,
                                     int32_t win, const float* mask, void* out, void* stream) {
    MG_REQUIRE(series && starts && out && n_steps >= 1 && n_cols >= 1 && B >= 1 && win >= 1, "gather_windows: bad arguments");
    MG_REQUIRE(win <= n_steps, "gather_windows: window of %d steps, series of %lld", win, (long long)n_steps);
    MG_REQUIRE(B <= 65535, "gather_windows: at most 65535 windows per call, got %d", B);
    MG_TRY(use_device_of(series, "gather_windows"));
    hipStream_t st = (hipStream_t)stream;
    int* bad = nullptr;
    MG_HIP(hipMallocAsync((void**)&bad, sizeof(int), st));
    MG_HIP(hipMemsetAsync(bad, 0, sizeof(int), st));
    const long per = (long)win * n_cols;
    const dim3 grid((unsigned)std::min<long>((per + 255) / 256, 64), B);
    if (dtype == MGADMM_F64)
        hipLaunchKernelGGL(k_windows<double>, grid, dim3(256), 0, st, (long)n_steps, n_cols, win, (const double*)series,
                           (const long long*)starts, mask, (double*)out, bad);
    else if (dtype == MGADMM_F32)
        hipLaunchKernelGGL(k_windows<float>, grid, dim3(256), 0, st, (long)n_steps, n_cols, win, (const float*)series,
                           (const long long*)starts, mask, (float*)out, bad);
    else {
        mg_set_error("gather_windows: bad dtype %d", dtype);
        return MGADMM_ERR_INVALID;
    }
    MG_HIP(hipGetLastError());
    int h_bad = 0;
    MG_HIP(hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, st));
    MG_HIP(hipStreamSynchronize(st));
    MG_HIP(hipFreeAsync(bad, st));
    MG_REQUIRE(!h_bad, "gather_windows: a start index is outside [0, n_steps - win]");
    return MGADMM_OK;
}

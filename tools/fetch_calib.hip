// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on this box for the access widths the kernels of this repository use
// (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of 16 B/lane streaming reads; other widths are
// "uncalibrated: calibrate on a known byte count in your own access pattern").  Three kernels stream the same 1 GiB buffer
// once, coalesced, with 4, 8 and 16 bytes per lane and write 64 MiB; tools/fetch_calib.sh runs them under
// `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` and records counter / known bytes per width.
//   hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o tools/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename V>
__global__ __launch_bounds__(256) void k_stream_read(const V* __restrict__ src, float* __restrict__ dst, size_t n, int per_thread) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    for (int j = 0; j < per_thread; ++j) {
        const size_t i = (size_t)j * gridDim.x * 256 + t;        // consecutive lanes: consecutive vectors
        if (i < n) {
            const V v = src[i];
            const float* f = reinterpret_cast<const float*>(&v);
            for (unsigned c = 0; c < sizeof(V) / 4; ++c) acc += f[c];
        }
    }
    dst[t] = acc;
}

int main() {
    const size_t bytes = 1ull << 30;
    void* src = nullptr;
    float* dst = nullptr;
    const int blocks = 65536, per = 16;        // 16 M threads
    if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst, sizeof(float) * (size_t)blocks * 256) != hipSuccess) return 1;
    hipMemset(src, 0, bytes);
    hipDeviceSynchronize();
    // every kernel reads the whole buffer: vectors per thread = bytes / sizeof(V) / threads
    hipLaunchKernelGGL((k_stream_read<float>), dim3(blocks), dim3(256), 0, 0, (const float*)src, dst, bytes / 4, (int)(bytes / 4 / ((size_t)blocks * 256)));
    hipLaunchKernelGGL((k_stream_read<float2>), dim3(blocks), dim3(256), 0, 0, (const float2*)src, dst, bytes / 8, (int)(bytes / 8 / ((size_t)blocks * 256)));
    hipLaunchKernelGGL((k_stream_read<float4>), dim3(blocks), dim3(256), 0, 0, (const float4*)src, dst, bytes / 16, (int)(bytes / 16 / ((size_t)blocks * 256)));
    (void)per;
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("read_bytes %zu write_bytes %zu\n", bytes, sizeof(float) * (size_t)blocks * 256);
    return 0;
}

// How long does ONE workgroup take to pull a cold 221 KB operand batch (8 vectors x 7368 floats) from HBM while the
// other CUs are mostly busy with something else?  Compares the access shapes open to k_admm_lds:
//   mode 0  (T,N) layout: 8 rows x 8 vectors of coalesced 4-byte loads per thread (64 loads in flight)
//   mode 1  [N][T] layout: 2 x 16-byte loads per vector and thread (16 loads in flight)
//   mode 2  as 0, but one time step (8 loads) per dependent round trip
// One 960-thread workgroup per CU, 16 batches per workgroup, each from a fresh region (footprint 0.9 GB >> MALL),
// pseudo-random busy-waits of 0..150 us between batches so that the CUs are not in lock step (as in the solver).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_batch_load tools/probe_batch_load.hip && tools/probe_batch_load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int N = 307, T = 24, TN = N * T, NV = 8, NB = 4096, TPG = 8;

__device__ __forceinline__ void spin_us(unsigned us) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(8);
}

template <int MODE>
__global__ __launch_bounds__(1024) void k_probe(const float* __restrict__ pool, float* __restrict__ sink, unsigned* __restrict__ ticks, int reps,
                                                int lockstep, int stagger_us) {
    const int tid = threadIdx.x;
    const bool active = tid < 3 * N;
    const int g = active ? tid / N : 0, i = active ? tid - g * N : 0, t0 = g * TPG;
    float acc = 0.f;
    if (stagger_us) spin_us((unsigned)(blockIdx.x % 256) * (unsigned)stagger_us / 256u);
    for (int r = 0; r < reps; ++r) {
        const int b = r * gridDim.x + blockIdx.x;
        const float* base = pool + (size_t)b * NV * TN;
        __syncthreads();
        const unsigned long long c0 = wall_clock64();
        if (MODE == 0) {
            float v[NV][TPG];
#pragma unroll
            for (int k = 0; k < TPG; ++k)
#pragma unroll
                for (int q = 0; q < NV; ++q) v[q][k] = base[(size_t)q * TN + (t0 + k) * N + i];
#pragma unroll
            for (int k = 0; k < TPG; ++k)
#pragma unroll
                for (int q = 0; q < NV; ++q) acc += v[q][k];
        } else if (MODE == 1) {
            float4 v[NV][2];
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const float4* p = reinterpret_cast<const float4*>(base + (size_t)q * TN + i * T + t0);
                v[q][0] = p[0];
                v[q][1] = p[1];
            }
#pragma unroll
            for (int q = 0; q < NV; ++q) acc += v[q][0].x + v[q][0].y + v[q][0].z + v[q][0].w + v[q][1].x + v[q][1].y + v[q][1].z + v[q][1].w;
        } else {
#pragma unroll
            for (int k = 0; k < TPG; ++k) {
                float v[NV];
#pragma unroll
                for (int q = 0; q < NV; ++q) v[q] = base[(size_t)q * TN + (t0 + k) * N + i];
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < NV; ++q) s += v[q];
                acc += s;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned long long c1 = wall_clock64();
        if (tid == 0) ticks[r * gridDim.x + blockIdx.x] = (unsigned)(c1 - c0);
        spin_us(lockstep ? 150u : (unsigned)((blockIdx.x * 37u + r * 101u) % 150u));   // lockstep: every workgroup computes equally long
    }
    if (acc == 123.456f) sink[blockIdx.x] = acc;
}

__global__ void k_write(float* __restrict__ pool, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) pool[i] = v;
}

template <int MODE>
static void run(const char* name, const float* pool, float* sink, unsigned* d_ticks, int grid, int reps, bool quiet, int lockstep = 0,
                int stagger_us = 0) {
    std::vector<unsigned> h((size_t)grid * reps);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(960), 0, 0, pool, sink, d_ticks, reps, lockstep, stagger_us);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d_ticks, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    double mean = 0;
    for (unsigned v : h) mean += v;
    mean /= h.size();
    if (!quiet)
        printf("%-58s mean %6.2f us  median %6.2f  p10 %6.2f  p90 %6.2f   (%.1f GB/s per CU)\n", name, mean * 0.01, h[h.size() / 2] * 0.01,
               h[h.size() / 10] * 0.01, h[h.size() * 9 / 10] * 0.01, NV * TN * 4.0 / (mean * 0.01e-6) / 1e9);
}

int main() {
    float *pool, *sink;
    unsigned* ticks;
    const size_t elems = (size_t)NB * NV * TN;
    hipMalloc(&pool, elems * sizeof(float));
    hipMalloc(&sink, 4096 * sizeof(float));
    hipMalloc(&ticks, 65536 * sizeof(unsigned));
    hipMemset(pool, 0, elems * sizeof(float));
    const int grid = 256, reps = 16;
    run<0>("warm-up", pool, sink, ticks, grid, reps, true);
    for (int pass = 0; pass < 2; ++pass) {
        run<0>("(T,N): 64 x 4-byte loads per thread in flight", pool, sink, ticks, grid, reps, false);
        run<1>("[N][T]: 16 x 16-byte loads per thread in flight", pool, sink, ticks, grid, reps, false);
        run<2>("(T,N): 8 round trips of 8 x 4-byte loads", pool, sink, ticks, grid, reps, false);
    }
    // the same reads of data a previous kernel has just WRITTEN (as in the solver: the state of iteration i-1)
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, 0, pool, elems, 1.0f + pass);
        run<0>("after a full rewrite of the pool: 64 x 4-byte loads", pool, sink, ticks, grid, reps, false);
        hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, 0, pool, elems, 2.0f + pass);
        run<2>("after a full rewrite of the pool: 8 round trips", pool, sink, ticks, grid, reps, false);
    }
    // every workgroup computes for the same time between its batches (the solver: equal CG iteration counts), so all 256
    // CUs request their batch at the same moment; then the same with the first workgroups' start staggered over 40 us
    for (int pass = 0; pass < 2; ++pass) {
        run<0>("lock step: 64 x 4-byte loads", pool, sink, ticks, grid, reps, false, 1, 0);
        run<2>("lock step: 8 round trips", pool, sink, ticks, grid, reps, false, 1, 0);
        run<0>("lock step, start staggered over 40 us: 64 loads", pool, sink, ticks, grid, reps, false, 1, 40);
        run<2>("lock step, start staggered over 40 us: 8 round trips", pool, sink, ticks, grid, reps, false, 1, 40);
    }
    return 0;
}

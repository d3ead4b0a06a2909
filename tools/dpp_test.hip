#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_step(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_step<0x111, 0xf>(v);
    v = dpp_step<0x112, 0xf>(v);
    v = dpp_step<0x114, 0xf>(v);
    v = dpp_step<0x118, 0xf>(v);
    v = dpp_step<0x142, 0xa>(v);
    v = dpp_step<0x143, 0xc>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__global__ void k(float* out, float* steps) {
    float v = (float)(threadIdx.x + 1);
    float a = dpp_step<0x111, 0xf>(v);
    steps[threadIdx.x] = a;
    float b = dpp_step<0x112, 0xf>(a);
    steps[64 + threadIdx.x] = b;
    float c = dpp_step<0x114, 0xf>(b);
    float d = dpp_step<0x118, 0xf>(c);
    steps[128 + threadIdx.x] = d;
    float e = dpp_step<0x142, 0xa>(d);
    steps[192 + threadIdx.x] = e;
    float f = dpp_step<0x143, 0xc>(e);
    steps[256 + threadIdx.x] = f;
    out[threadIdx.x] = wave_sum(v);
}
int main() {
    float *d, *s; hipMalloc(&d, 64 * 4); hipMalloc(&s, 320 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, s);
    float h[64], hs[320]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost); hipMemcpy(hs, s, 1280, hipMemcpyDeviceToHost);
    printf("sum lane0 %g lane63 %g (expect 2080)\n", h[0], h[63]);
    for (int st = 0; st < 5; ++st) { printf("step %d:", st); for (int i = 0; i < 64; ++i) printf(" %g", hs[st * 64 + i]); printf("\n"); }
    return 0;
}

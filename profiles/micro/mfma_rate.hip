// mfma_rate.hip — sustained fp32 matrix rate of the two fp32 MFMA shapes on register-resident operands (no memory
// traffic): does the clock the chip holds under a full fp32 matrix load depend on the instruction shape?
//   hipcc -O3 --offload-arch=gfx950 -o mfma_rate mfma_rate.hip && ./mfma_rate
// 256 blocks x 512 threads (two waves per SIMD, as the batch kernels), independent accumulators per wave, N MFMAs each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512) void k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(512) void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
// the same with eight pseudo-random operand pairs per lane cycled through (data-dependent switching power)
__global__ __launch_bounds__(512) void k32r(float* out, int iters, unsigned seed) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[8], b[8];
    unsigned s = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < 8; ++i) {
        s = s * 1664525u + 1013904223u; a[i] = (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
        s = s * 1664525u + 1013904223u; b[i] = (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
    }
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc[j & 3], 0, 0, 0);
    }
    float t = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) t += acc[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = t;
}
__global__ __launch_bounds__(512) void k16r(float* out, int iters, unsigned seed) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a[8], b[8];
    unsigned s = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < 8; ++i) {
        s = s * 1664525u + 1013904223u; a[i] = (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
        s = s * 1664525u + 1013904223u; b[i] = (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc[j], 0, 0, 0);
    }
    float t = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) t += acc[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = t;
}
template <typename F>
double run(F launch, double flops_per_launch, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch();                                   // warm-up
    hipDeviceSynchronize();
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-34s %8.3f ms  %7.1f TFLOP/s  (%.1f %% of 157.3)\n", name, best, flops_per_launch / best / 1e9, flops_per_launch / best / 1e9 / 157.3 * 100);
    return best;
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * sizeof(float));
    const int blocks = 256;
    for (int ms_target : {1, 20}) {             // a short and a long launch (the clock may sag with time)
        const int iters = ms_target == 1 ? 2000 : 40000;
        printf("-- %d iterations per wave\n", iters);
        run([&] { hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(512), 0, 0, out, iters, 1.0f, 1.0f); },
            (double)blocks * 8 * iters * 4 * 4096.0, "32x32x2 f32, 4 accumulators");
        run([&] { hipLaunchKernelGGL(k32<2>, dim3(blocks), dim3(512), 0, 0, out, iters * 2, 1.0f, 1.0f); },
            (double)blocks * 8 * iters * 2 * 2 * 4096.0, "32x32x2 f32, 2 accumulators");
        run([&] { hipLaunchKernelGGL(k16<8>, dim3(blocks), dim3(512), 0, 0, out, iters * 2, 1.0f, 1.0f); },
            (double)blocks * 8 * iters * 2 * 8 * 2048.0, "16x16x4 f32, 8 accumulators");
        run([&] { hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(512), 0, 0, out, iters * 4, 1.0f, 1.0f); },
            (double)blocks * 8 * iters * 4 * 4 * 2048.0, "16x16x4 f32, 4 accumulators");
        run([&] { hipLaunchKernelGGL(k32r, dim3(blocks), dim3(512), 0, 0, out, iters, 12345u); },
            (double)blocks * 8 * iters * 4 * 4096.0, "32x32x2 f32, random operands");
        run([&] { hipLaunchKernelGGL(k16r, dim3(blocks), dim3(512), 0, 0, out, iters, 12345u); },
            (double)blocks * 8 * iters * 8 * 2048.0, "16x16x4 f32, random operands");
    }
    return 0;
}

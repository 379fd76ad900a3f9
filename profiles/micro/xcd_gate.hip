// xcd_gate.hip — gate for a single-XCD persistent PointMaze step:
//   (1) hand-rolled barrier among the 32 workgroups that share an XCD (equal blockIdx % 8), in us;
//   (2) what those 32 workgroups stream from a 64 MB buffer (Infinity-Cache resident on the second
//       pass), in GB/s;  and the same for a chip-wide 256-workgroup barrier / stream, for reference.
// Build:  hipcc -O3 --offload-arch=gfx950 -o xcd_gate profiles/micro/xcd_gate.hip ; run on the GPU box.
// Every spin is bounded (a timeout sets a flag and every block leaves).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// group = blocks with blockIdx % nxcd == 0 (nxcd = 8: one XCD under round-robin placement; 1: all)
__global__ void barrier_kernel(unsigned* counter, unsigned* timeout, unsigned long long* out, int* xcc, int iters, int stride) {
    if (blockIdx.x % stride != 0) return;
    const int members = gridDim.x / stride;
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[blockIdx.x] = (int)(id & 0xf);
    }
    unsigned long long t0 = 0;
    for (int it = 0; it <= iters; ++it) {
        if (it == 1 && threadIdx.x == 0) t0 = __builtin_amdgcn_s_memrealtime();
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)members * (unsigned)(it + 1);
            unsigned spins = 0;
            while (ld_sc1(counter) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 2000000u || ld_sc1(timeout) != 0u) { atomicExch(timeout, 1u); break; }
            }
        }
        __syncthreads();
        if (ld_sc1(timeout) != 0u) return;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
}

__global__ void stream_kernel(const float4* src, float* sink, long n4, int stride, unsigned long long* out) {
    if (blockIdx.x % stride != 0) return;
    const int member = blockIdx.x / stride, members = gridDim.x / stride;
    const long per = n4 / members;
    const float4* p = src + member * per;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long i = threadIdx.x; i + 3 * blockDim.x < per; i += 4 * blockDim.x) {
        const float4 a = p[i], b = p[i + blockDim.x], c = p[i + 2 * blockDim.x], d = p[i + 3 * blockDim.x];
        acc.x += a.x + b.x + c.x + d.x; acc.y += a.y + b.y + c.y + d.y;
        acc.z += a.z + b.z + c.z + d.z; acc.w += a.w + b.w + c.w + d.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
}

int main() {
    const int G = 256, iters = 2000;
    unsigned *counter, *timeout; unsigned long long* out; int* xcc; float4* buf; float* sink;
    const long n4 = (64L << 20) / 16;
    CHECK(hipMalloc(&counter, 64)); CHECK(hipMalloc(&timeout, 64));
    CHECK(hipMalloc(&out, G * sizeof(unsigned long long))); CHECK(hipMalloc(&xcc, G * sizeof(int)));
    CHECK(hipMalloc(&buf, n4 * 16)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 0, n4 * 16));
    std::vector<unsigned long long> h(G); std::vector<int> hx(G);
    for (int stride : {8, 1}) {
        CHECK(hipMemset(counter, 0, 64)); CHECK(hipMemset(timeout, 0, 64)); CHECK(hipMemset(xcc, 0xff, G * sizeof(int)));
        hipLaunchKernelGGL(barrier_kernel, dim3(G), dim3(512), 0, 0, counter, timeout, out, xcc, iters, stride);
        CHECK(hipDeviceSynchronize());
        unsigned to = 0; CHECK(hipMemcpy(&to, timeout, 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(h.data(), out, G * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hx.data(), xcc, G * 4, hipMemcpyDeviceToHost));
        unsigned long long worst = 0; int distinct = 0; bool seen[16] = {false};
        for (int b = 0; b < G; b += stride) { worst = std::max(worst, h[b]); if (!seen[hx[b] & 15]) { seen[hx[b] & 15] = true; ++distinct; } }
        printf("barrier over %3d workgroups (blockIdx %% %d == 0): %.2f us per barrier, %d distinct XCC ids%s\n", G / stride, stride,
               worst * 0.01 / iters, distinct, to ? "  [TIMEOUT]" : "");
    }
    for (int stride : {8, 1}) {
        for (int pass = 0; pass < 3; ++pass) {
            hipLaunchKernelGGL(stream_kernel, dim3(G), dim3(512), 0, 0, buf, sink, n4, stride, out);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(h.data(), out, G * 8, hipMemcpyDeviceToHost));
            unsigned long long worst = 0;
            for (int b = 0; b < G; b += stride) worst = std::max(worst, h[b]);
            printf("stream 64 MB by %3d workgroups, pass %d: %.1f us = %.0f GB/s\n", G / stride, pass, worst * 0.01,
                   (double)(n4 * 16) / (worst * 0.01e-6) / 1e9);
        }
    }
    return 0;
}

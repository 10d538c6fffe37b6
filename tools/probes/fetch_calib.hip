// fetch_calib.hip -- what does the FETCH_SIZE counter (KB) report for the access patterns of this library?  Every kernel reads
// each byte of a 1 GiB buffer region exactly once (cold: the buffer is far larger than the 256 MiB Infinity Cache and the L2s),
// so FETCH_SIZE * 1024 / bytes read is the counter's calibration factor for that pattern.
//   k_lane16     16 bytes per lane, a wave reads 1024 contiguous bytes          (A kernels: 128-byte segments)
//   k_lane8      8 bytes per lane, 512 contiguous bytes per wave               (B kernels)
//   k_lane4      4 bytes per lane, 256 contiguous bytes per wave
//   k_chunk64    64-byte chunks, four lanes x 16 bytes, global -> LDS DMA, one chunk per 256 bytes of the buffer (the polar
//                gather's staging: isolated 64-byte pieces)
//   k_chunk64d   the same with the four chunks of every 256 bytes all read (dense: every byte once)
// Build: hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib
// Run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- ./fetch_calib      (tools/fetch_calib.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
__global__ __launch_bounds__(256) void k_lane16(const float4* __restrict__ p, size_t n, float* out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_lane8(const float2* __restrict__ p, size_t n, float* out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float2 v = p[i]; acc += v.x + v.y; }
    if (acc == 12345.f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_lane4(const float* __restrict__ p, size_t n, float* out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc == 12345.f) out[0] = acc;
}
// chunk c of the buffer = bytes [step*c, step*c + 64); four lanes per chunk, 64 chunks per workgroup pass
template <int STEP> __global__ __launch_bounds__(256) void k_chunk64(const char* __restrict__ p, size_t n_chunks, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[1024];
    float acc = 0.f;
    for (size_t c0 = (size_t)blockIdx.x * 64; c0 < n_chunks; c0 += (size_t)gridDim.x * 64) {
        const size_t c = c0 + (threadIdx.x >> 2);
        if (c < n_chunks)
            __builtin_amdgcn_global_load_lds((glb_void*)(p + c * STEP + 16 * (threadIdx.x & 3)), (lds_void*)(lds + (threadIdx.x & ~63) * 4), 16, 0, 0);
        __syncthreads();
        acc += lds[threadIdx.x * 4];
        __syncthreads();
    }
    if (acc == 12345.f) out[0] = acc;
}
int main() {
    const size_t bytes = (size_t)1 << 30;
    char* buf; float* out;
    hipMalloc(&buf, 5 * bytes); hipMalloc(&out, 4);
    hipMemset(buf, 1, 5 * bytes); hipDeviceSynchronize();
    const int G = 256 * 16;
    hipLaunchKernelGGL(k_lane16, dim3(G), dim3(256), 0, 0, (const float4*)(buf + 0 * bytes), bytes / 16, out);
    hipLaunchKernelGGL(k_lane8, dim3(G), dim3(256), 0, 0, (const float2*)(buf + 1 * bytes), bytes / 8, out);
    hipLaunchKernelGGL(k_lane4, dim3(G), dim3(256), 0, 0, (const float*)(buf + 2 * bytes), bytes / 4, out);
    hipLaunchKernelGGL(k_chunk64<256>, dim3(G), dim3(256), 0, 0, (const char*)(buf + 3 * bytes), bytes / 256, out);
    hipLaunchKernelGGL(k_chunk64<64>, dim3(G), dim3(256), 0, 0, (const char*)(buf + 4 * bytes), bytes / 64, out);
    hipDeviceSynchronize();
    printf("bytes read: k_lane16 %zu k_lane8 %zu k_lane4 %zu k_chunk64<256> %zu k_chunk64<64> %zu\n", bytes, bytes, bytes, bytes / 4, bytes);
    return 0;
}

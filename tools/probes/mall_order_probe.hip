// mall_order_probe.hip -- does the ORDER in which a consumer kernel walks a buffer that the producer kernel has just written
// decide how much of it comes back from the 256 MiB Infinity Cache?  A producer copies X -> Y front to back (item order), the
// consumer then copies Y -> Z either front to back (the items written FIRST are read first: they are the ones an LRU-like
// cache has already dropped) or back to front (the items written LAST are read first).  Items are 1.4 MB like a polar
// spectrum; both kernels walk items in blockIdx order (the dispatcher issues blocks in linear order).
// Build: hipcc --offload-arch=gfx950 -O3 mall_order_probe.hip -o mall_order_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k_copy_items(const float4* __restrict__ a, float4* __restrict__ b, int n_items, size_t item_f4, int chunks, int reverse) {
    // grid = n_items * chunks blocks; block L handles chunk L % chunks of item L / chunks (or the mirrored item)
    int item = blockIdx.x / chunks; const int ch = blockIdx.x % chunks;
    if (reverse) item = n_items - 1 - item;
    const size_t per = item_f4 / chunks;
    const float4* s = a + (size_t)item * item_f4 + (size_t)ch * per;
    float4* d = b + (size_t)item * item_f4 + (size_t)ch * per;
    for (size_t i = threadIdx.x; i < per; i += 256) d[i] = s[i];
}
__global__ void k_read(const float4* p, size_t n, float* out) {
    float s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.f) out[0] = s;
}
int main() {
    const size_t ITEM = 1392640;                 // bytes per item (multiple of 16 * 85 chunks)
    const int CH = 85;                            // blocks per item (16 KB per block)
    const size_t item_f4 = ITEM / 16;
    const int MAXI = 1024;
    float4 *x, *y, *z, *cold; float* out;
    (void)hipMalloc(&x, ITEM * MAXI); (void)hipMalloc(&y, ITEM * MAXI); (void)hipMalloc(&z, ITEM * MAXI); (void)hipMalloc(&cold, (size_t)2048 << 20); (void)hipMalloc(&out, 4);
    hipMemset(x, 0, ITEM * MAXI); hipMemset(y, 0, ITEM * MAXI); hipMemset(z, 0, ITEM * MAXI); hipMemset(cold, 0, (size_t)2048 << 20);
    hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    printf("items  MB/plane   producer GB/s(r+w)   consumer fwd GB/s   consumer rev GB/s   3-stage chain fwd / alternating (GB/s r+w)\n");
    for (int n : { 32, 64, 96, 128, 192, 256, 384, 512, 1024 }) {
        double tp = 0, tf = 0, tr = 0, tcf = 0, tca = 0; const int REP = 6;
        for (int r = 0; r < REP + 1; ++r) {
            float ms;
            for (int rev = 0; rev < 2; ++rev) {
                k_read<<<4096, 256>>>(cold, ((size_t)2048 << 20) / 16, out);
                hipEventRecord(e0); k_copy_items<<<n * CH, 256>>>(x, y, n, item_f4, CH, 0); hipEventRecord(e1);
                k_copy_items<<<n * CH, 256>>>(y, z, n, item_f4, CH, rev); hipEventRecord(e2); hipEventSynchronize(e2);
                if (r) { hipEventElapsedTime(&ms, e0, e1); tp += ms / 2; hipEventElapsedTime(&ms, e1, e2); (rev ? tr : tf) += ms; }
            }
            // a chain of 6 dependent copies x->y->z->x..., all forward vs alternating direction
            for (int alt = 0; alt < 2; ++alt) {
                k_read<<<4096, 256>>>(cold, ((size_t)2048 << 20) / 16, out);
                float4* bufs[3] = { x, y, z };
                hipEventRecord(e0);
                for (int s = 0; s < 6; ++s) k_copy_items<<<n * CH, 256>>>(bufs[s % 3], bufs[(s + 1) % 3], n, item_f4, CH, alt ? (s & 1) : 0);
                hipEventRecord(e1); hipEventSynchronize(e1);
                if (r) { hipEventElapsedTime(&ms, e0, e1); (alt ? tca : tcf) += ms; }
            }
        }
        const double bytes = (double)ITEM * n;
        printf("%5d  %7.1f   %8.0f   %8.0f   %8.0f   %8.0f / %8.0f\n", n, bytes / 1e6, 2 * bytes / (tp / REP * 1e-3) / 1e9, 2 * bytes / (tf / REP * 1e-3) / 1e9,
               2 * bytes / (tr / REP * 1e-3) / 1e9, 12 * bytes / (tcf / REP * 1e-3) / 1e9, 12 * bytes / (tca / REP * 1e-3) / 1e9);
    }
    return 0;
}

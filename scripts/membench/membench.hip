// Per-CU vector-memory delivery rate on MI355X: every block streams its own region `iters` times with 16-byte (or 8-byte) loads,
// 8 loads in flight per thread.  Region small (all blocks' regions fit the 4 MiB L2 of their XCD) -> L2-hit rate; region large ->
// HBM rate.  Usage: membench   (prints a table).  Build: hipcc --offload-arch=gfx950 -O3 membench.hip -o membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <typename V>
__global__ void __launch_bounds__(256) stream_kernel(const V* __restrict__ p, long region_elems, int iters, unsigned* out) {
    const V* base = p + (long)blockIdx.x * region_elems;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        for (long i = threadIdx.x; i + 7 * 256 < region_elems; i += 8 * 256) {
            V v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(base + i + u * 256) , (void)0;
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= ((const unsigned*)&v[u])[0];
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <typename V>
__global__ void __launch_bounds__(256) stream_kernel_cached(const V* __restrict__ p, long region_elems, int iters, unsigned* out) {
    const V* base = p + (long)blockIdx.x * region_elems;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        for (long i = threadIdx.x; i + 7 * 256 < region_elems; i += 8 * 256) {
            V v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = base[i + u * 256];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= ((const unsigned*)&v[u])[0];
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <typename V>
static void run(const char* tag, int blocks, long region_bytes, int iters, void* buf, unsigned* out) {
    long elems = region_bytes / (long)sizeof(V);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((stream_kernel_cached<V>), dim3(blocks), dim3(256), 0, 0, (const V*)buf, elems, iters, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_kernel_cached<V>), dim3(blocks), dim3(256), 0, 0, (const V*)buf, elems, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double bytes = (double)blocks * region_bytes * iters;
    double tbs = bytes / (ms * 1e-3) / 1e12;
    printf("%-28s blocks %5d  region %8ld B  iters %4d  %8.3f ms  %6.2f TB/s  %5.1f B/clk/CU (2.4 GHz, 256 CUs)\n", tag, blocks, region_bytes,
           iters, ms, tbs, tbs * 1e12 / 256 / 2.4e9);
}

int main() {
    const size_t total = (size_t)6 << 30;
    void* buf; unsigned* out;
    if (hipMalloc(&buf, total) != hipSuccess) return 1;
    hipMalloc(&out, 64);
    hipMemset(buf, 1, total);
    for (int bpc : {1, 2, 3, 4, 8}) {
        int blocks = 256 * bpc;
        // L2-resident: all regions of an XCD (blocks/8 of them) fit 3 MiB
        long region = (3l << 20) / (blocks / 8);
        region = region / (8 * 256 * 16) * (8 * 256 * 16);
        if (region < 8 * 256 * 16) region = 8 * 256 * 16;
        char tag[64];
        snprintf(tag, sizeof tag, "L2-resident 16B/lane x%d/CU", bpc);
        run<uint4>(tag, blocks, region, (int)((64l << 20) / region), buf, out);
        snprintf(tag, sizeof tag, "L2-resident  8B/lane x%d/CU", bpc);
        run<uint2>(tag, blocks, region, (int)((64l << 20) / region), buf, out);
    }
    for (int bpc : {1, 3, 8}) {
        int blocks = 256 * bpc;
        long region = (long)(total / blocks) / (8 * 256 * 16) * (8 * 256 * 16);
        char tag[64];
        snprintf(tag, sizeof tag, "HBM stream 16B/lane x%d/CU", bpc);
        run<uint4>(tag, blocks, region, 1, buf, out);
    }
    return 0;
}

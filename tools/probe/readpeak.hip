// readpeak.hip — what a plain coalesced read stream achieves on this GPU once the clocks are up (measurement support
// only): the yardstick next to the 8 TB/s data-sheet peak for the "fraction of achievable" figure in bench.py.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
__global__ __launch_bounds__(256) void read_stream(const uint4* __restrict__ in, size_t units_per_block, uint32_t* out) {
    const uint4* p = in + (size_t)blockIdx.x * units_per_block;
    uint32_t acc = 0;
    for (size_t u = threadIdx.x; u < units_per_block; u += 256) { uint4 v = p[u]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const size_t N = (size_t)8 << 30;
    uint4* d; uint32_t* o; CK(hipMalloc(&d, N)); CK(hipMalloc(&o, 64)); CK(hipMemset(d, 1, N));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const size_t tile = 32768;
    for (int round = 0; round < 6; round++) {
        CK(hipEventRecord(a));
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL(read_stream, dim3((unsigned)(N / tile)), dim3(256), 0, 0, d, tile / 16, o);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("launches %2d-%2d: coalesced read of 8 GiB %7.3f ms  %7.1f GB/s\n", round * 10 + 1, round * 10 + 10, ms / 10, N / (ms / 10) / 1e6);
    }
    return 0;
}

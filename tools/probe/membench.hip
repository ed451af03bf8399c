// membench.hip — access-pattern probe for the text feed of the scan kernel (measurement support only).
//  A: workgroup reads a contiguous tile with coalesced 16 B/lane loads (what LDS tile staging does)
//  B: lane-stripes: each lane owns STRIPE contiguous bytes; per round 4 lanes fetch one 64 B piece of a
//     stripe (16 pieces per wave-instruction), staged through LDS
//  C: lane-stripes, direct: each lane loads 16 B of its own stripe per round (64 lines per wave-instruction)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

__global__ __launch_bounds__(256) void patA(const uint4* __restrict__ in, size_t units_per_block, uint32_t* out) {
    const uint4* p = in + (size_t)blockIdx.x * units_per_block;
    uint32_t acc = 0;
    for (size_t u = threadIdx.x; u < units_per_block; u += 256) { uint4 v = p[u]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
// C: stripe per lane, direct 16 B loads; UNROLL independent loads in flight
template <int UNROLL>
__global__ __launch_bounds__(256) void patC(const uint4* __restrict__ in, size_t stripe_units, uint32_t* out) {
    const size_t lane_global = (size_t)blockIdx.x * 256 + threadIdx.x;
    const uint4* p = in + lane_global * stripe_units;
    uint32_t acc = 0;
    for (size_t u = 0; u < stripe_units; u += UNROLL) {
        uint4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) v[k] = p[u + k];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// B: stripes via LDS: block owns 256 stripes; per round every stripe advances by PIECE bytes; PIECE/16 lanes per piece
template <int PIECE>
__global__ __launch_bounds__(256) void patB(const uint4* __restrict__ in, size_t stripe_units, uint32_t* out) {
    __shared__ uint4 lds[256 * PIECE / 16];
    constexpr int LPP = PIECE / 16;                 // lanes per piece
    constexpr int PPI = 256 / LPP;                  // pieces per block-wide load instruction
    const uint4* base = in + (size_t)blockIdx.x * 256 * stripe_units;
    uint32_t acc = 0;
    for (size_t r = 0; r < stripe_units / LPP; r++) {
#pragma unroll
        for (int i = 0; i < LPP; i++) {             // LPP block-wide instructions cover 256 pieces
            int piece = i * PPI + threadIdx.x / LPP;       // stripe index
            int sub = threadIdx.x % LPP;
            uint4 v = base[(size_t)piece * stripe_units + r * LPP + sub];
            lds[piece * LPP + sub] = v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < LPP; k++) { uint4 v = lds[threadIdx.x * LPP + k]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
        __syncthreads();
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <class F> float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main() {
    const size_t N = (size_t)8 << 30;
    uint4* d; uint32_t* o; CK(hipMalloc(&d, N)); CK(hipMalloc(&o, 64)); CK(hipMemset(d, 1, N));
    auto report = [&](const char* name, float ms) { printf("%-40s %8.3f ms  %8.1f GB/s\n", name, ms, N / ms / 1e6); fflush(stdout); };
    for (size_t tile : {32768, 65536}) {
        size_t upb = tile / 16; unsigned blocks = (unsigned)(N / tile);
        char nm[64]; snprintf(nm, 64, "A coalesced tile %zuK", tile >> 10);
        report(nm, timeit([&] { hipLaunchKernelGGL(patA, dim3(blocks), dim3(256), 0, 0, d, upb, o); }, 5));
    }
    for (size_t stripe : {512, 1024, 4096, 16384}) {
        size_t su = stripe / 16; unsigned blocks = (unsigned)(N / (256 * stripe));
        char nm[64];
        snprintf(nm, 64, "C direct stripe %zu B unroll 4", stripe);
        report(nm, timeit([&] { hipLaunchKernelGGL(patC<4>, dim3(blocks), dim3(256), 0, 0, d, su, o); }, 5));
        snprintf(nm, 64, "C direct stripe %zu B unroll 8", stripe);
        report(nm, timeit([&] { hipLaunchKernelGGL(patC<8>, dim3(blocks), dim3(256), 0, 0, d, su, o); }, 5));
    }
    for (size_t stripe : {1024, 4096}) {
        size_t su = stripe / 16; unsigned blocks = (unsigned)(N / (256 * stripe));
        char nm[64];
        snprintf(nm, 64, "B LDS-staged stripe %zu B piece 64", stripe);
        report(nm, timeit([&] { hipLaunchKernelGGL(patB<64>, dim3(blocks), dim3(256), 0, 0, d, su, o); }, 5));
        snprintf(nm, 64, "B LDS-staged stripe %zu B piece 128", stripe);
        report(nm, timeit([&] { hipLaunchKernelGGL(patB<128>, dim3(blocks), dim3(256), 0, 0, d, su, o); }, 5));
    }
    return 0;
}

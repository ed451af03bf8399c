// feedbench.hip — how fast can lanes stream their own stripes when the loads are spaced out by per-byte work?
// (measurement support only).  Every lane owns a contiguous stripe; per byte it does `WORK` dependent VALU ops.
// Variants of how a lane requests its next 128-byte line:
//   0  one burst of 8 x 16 B after the previous line is consumed            (the product kernel's feed)
//   1  register double buffer: burst for line r+1 issued before line r is processed
//   2  two 64-byte half-line bursts, the second issued HALF-way through the first half's processing
//   3  64-byte rounds: burst of 4 after the previous 64 bytes are consumed (second half a full round later)
//   4  as 2 but the second half is requested right after the first half has ARRIVED
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

template <int WORK>
__device__ __forceinline__ uint32_t chew(uint32_t acc, uint4 v) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t c = (w[q] >> (8 * k)) & 0xffu;
#pragma unroll
            for (int j = 0; j < WORK; j++) acc = (acc ^ c) * 0x9E3779B1u + j;      // dependent chain, ~2 VALU each
        }
    return acc;
}

template <int VARIANT, int WORK, int THREADS>
__global__ __launch_bounds__(THREADS) void feed(const uint4* __restrict__ in, uint32_t stripe_units, uint32_t* out) {
    const size_t g = (size_t)blockIdx.x * THREADS + threadIdx.x;
    const uint4* src = in + g * stripe_units;
    uint32_t acc = (uint32_t)g;
    if (VARIANT == 0) {
        uint4 b[8];
        for (int i = 0; i < 8; i++) b[i] = src[i];
        for (uint32_t u = 0; u < stripe_units; u += 8) {
#pragma unroll
            for (int i = 0; i < 8; i++) acc = chew<WORK>(acc, b[i]);
            if (u + 8 < stripe_units) {
#pragma unroll
                for (int i = 0; i < 8; i++) b[i] = src[u + 8 + i];
            }
        }
    } else if (VARIANT == 1) {
        uint4 b[8], n[8];
        for (int i = 0; i < 8; i++) b[i] = src[i];
        for (uint32_t u = 0; u < stripe_units; u += 8) {
            const bool more = u + 8 < stripe_units;
            if (more) {
#pragma unroll
                for (int i = 0; i < 8; i++) n[i] = src[u + 8 + i];
            }
#pragma unroll
            for (int i = 0; i < 8; i++) acc = chew<WORK>(acc, b[i]);
#pragma unroll
            for (int i = 0; i < 8; i++) b[i] = n[i];
        }
    } else if (VARIANT == 2 || VARIANT == 4) {
        uint4 a[4], b[4];
        for (int i = 0; i < 4; i++) a[i] = src[i];
        for (int i = 0; i < 4; i++) b[i] = src[4 + i];
        for (uint32_t u = 0; u < stripe_units; u += 8) {
            const bool more = u + 8 < stripe_units;
            // first half
            acc = chew<WORK>(acc, a[0]); acc = chew<WORK>(acc, a[1]);
            acc = chew<WORK>(acc, a[2]); acc = chew<WORK>(acc, a[3]);
            if (more) {
#pragma unroll
                for (int i = 0; i < 4; i++) a[i] = src[u + 8 + i];              // first half of the next line
                if (VARIANT == 4) asm volatile("" ::"v"(a[0].x), "v"(a[1].x), "v"(a[2].x), "v"(a[3].x));
            }
            if (VARIANT == 4 && more) {
                uint4 t[4];
#pragma unroll
                for (int i = 0; i < 4; i++) t[i] = src[u + 12 + i];
                acc = chew<WORK>(acc, b[0]); acc = chew<WORK>(acc, b[1]);
                acc = chew<WORK>(acc, b[2]); acc = chew<WORK>(acc, b[3]);
#pragma unroll
                for (int i = 0; i < 4; i++) b[i] = t[i];
            } else {
                acc = chew<WORK>(acc, b[0]); acc = chew<WORK>(acc, b[1]);
                if (more && VARIANT == 2) {                                      // second half of the next line, half a half later
                    uint4 t0 = src[u + 12], t1 = src[u + 13], t2 = src[u + 14], t3 = src[u + 15];
                    acc = chew<WORK>(acc, b[2]); acc = chew<WORK>(acc, b[3]);
                    b[0] = t0; b[1] = t1; b[2] = t2; b[3] = t3;
                } else { acc = chew<WORK>(acc, b[2]); acc = chew<WORK>(acc, b[3]); }
            }
        }
    } else if (VARIANT == 3) {
        uint4 b[4];
        for (int i = 0; i < 4; i++) b[i] = src[i];
        for (uint32_t u = 0; u < stripe_units; u += 4) {
#pragma unroll
            for (int i = 0; i < 4; i++) acc = chew<WORK>(acc, b[i]);
            if (u + 4 < stripe_units) {
#pragma unroll
                for (int i = 0; i < 4; i++) b[i] = src[u + 4 + i];
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <class F> float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

template <int VARIANT, int WORK, int THREADS>
void run(const uint4* d, uint32_t* o, size_t N, uint32_t stripe) {
    const uint32_t su = stripe / 16;
    const unsigned blocks = (unsigned)(N / ((size_t)THREADS * stripe));
    float ms = timeit([&] { hipLaunchKernelGGL((feed<VARIANT, WORK, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, d, su, o); }, 3);
    printf("variant %d work %d threads %4d stripe %5u : %8.3f ms %8.1f GB/s\n", VARIANT, WORK, THREADS, stripe, ms, N / ms / 1e6);
    fflush(stdout);
}

int main() {
    const size_t N = (size_t)8 << 30;
    uint4* d; uint32_t* o; CK(hipMalloc(&d, N)); CK(hipMalloc(&o, 64)); CK(hipMemset(d, 1, N));
    // WORK 0: pure feed; WORK 1: ~2 VALU/byte; WORK 2: ~4 VALU/byte (the product kernel's level)
    run<0, 0, 1024>(d, o, N, 16384); run<1, 0, 1024>(d, o, N, 16384); run<2, 0, 1024>(d, o, N, 16384); run<3, 0, 1024>(d, o, N, 16384); run<4, 0, 1024>(d, o, N, 16384);
    run<0, 2, 1024>(d, o, N, 16384); run<1, 2, 1024>(d, o, N, 16384); run<2, 2, 1024>(d, o, N, 16384); run<3, 2, 1024>(d, o, N, 16384); run<4, 2, 1024>(d, o, N, 16384);
    run<0, 2, 256>(d, o, N, 16384); run<1, 2, 256>(d, o, N, 16384); run<2, 2, 256>(d, o, N, 16384); run<3, 2, 256>(d, o, N, 16384); run<4, 2, 256>(d, o, N, 16384);
    run<0, 1, 1024>(d, o, N, 16384); run<3, 1, 1024>(d, o, N, 16384); run<2, 1, 1024>(d, o, N, 16384);
    return 0;
}

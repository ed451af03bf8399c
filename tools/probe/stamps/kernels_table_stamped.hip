// PROBE BUILD (not shipped): kernels_table.hip plus a stride-2 match kernel whose phase hook writes a timestamp per workgroup
// and phase - where a launch's fixed cost goes (VERDICT r2 "next" #3).  The clock is s_memrealtime (100 MHz, one counter for
// the chip).  stamps[workgroup][wave][kPhases + 2]: the phases of kernels_table.hip's enum, then XCC_ID and the CU id.
#include "../../../roaringregex_amd/csrc/kernels_table.hip"

namespace rrx {
namespace dev {
namespace {
constexpr int kMaxRounds = 32;      // round stamps: the time a wave STARTS round r (stripes up to 4 KiB)
struct StampHook {
    uint64_t *row;
    uint64_t *rounds;               // [wave][kMaxRounds] of this workgroup
    __device__ __forceinline__ void round(int r) const {
        if ((threadIdx.x & 63) == 0 && r < kMaxRounds) rounds[(threadIdx.x >> 6) * kMaxRounds + r] = wall_clock64();
    }
    __device__ __forceinline__ void operator()(int k) const {
        if ((threadIdx.x & 63) == 0) row[(threadIdx.x >> 6) * (kPhases + 2) + k] = wall_clock64();
    }
};
__global__ __launch_bounds__(kThreads) void match_stripes2_stamped_kernel(Dfa2Device prog, const uint8_t *__restrict__ bytes, size_t nbytes,
                                                                           uint32_t stripe, const uint64_t *__restrict__ stripe_base,
                                                                           uint32_t *__restrict__ accept_bits, uint64_t *__restrict__ stamps, uint64_t *__restrict__ rounds) {
    uint64_t *row = stamps + (size_t)blockIdx.x * (kThreads / 64) * (kPhases + 2);      // a row per wave
    if ((threadIdx.x & 63) == 0) {
        uint64_t *mine = row + (threadIdx.x >> 6) * (kPhases + 2);
        uint32_t xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        mine[kPhases] = xcc & 0xf;
        mine[kPhases + 1] = hw;
    }
    dfa2_body<false, StampHook>(prog, bytes, nbytes, stripe, stripe_base, accept_bits, nullptr, nullptr, StampHook{row, rounds + (size_t)blockIdx.x * (kThreads / 64) * kMaxRounds});
}
__global__ __launch_bounds__(kThreads) void match_units2_stamped_kernel(Dfa2Device prog, const uint8_t *__restrict__ bytes, size_t nbytes,
                                                                         uint32_t stripe, const uint64_t *__restrict__ stripe_base,
                                                                         uint32_t *__restrict__ accept_bits, uint32_t units_per_wg, uint64_t *__restrict__ stamps,
                                                                         uint64_t *__restrict__ rounds) {
    uint64_t *row = stamps + (size_t)blockIdx.x * (kThreads / 64) * (kPhases + 2);      // a row per wave (the last unit of a wave overwrites)
    if ((threadIdx.x & 63) == 0) {
        uint64_t *mine = row + (threadIdx.x >> 6) * (kPhases + 2);
        uint32_t xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        mine[kPhases] = xcc & 0xf;
        mine[kPhases + 1] = hw;
    }
    dfa2_units_body<StampHook>(prog, bytes, nbytes, stripe, stripe_base, accept_bits, units_per_wg, StampHook{row, rounds + (size_t)blockIdx.x * (kThreads / 64) * kMaxRounds});
}
}  // namespace
int match_units_dfa2_stamped(const Dfa2Device &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base,
                             size_t nstripes, uint32_t *accept, uint32_t units_per_wg, uint64_t *stamps, uint64_t *rounds, void *stream) {
    if (!nstripes) return 0;
    const size_t units = (nstripes + 63) / 64, blocks = (units + units_per_wg - 1) / units_per_wg;
    hipLaunchKernelGGL(match_units2_stamped_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, p, bytes, nbytes, stripe, stripe_base, accept, units_per_wg, stamps, rounds);
    return (int)hipGetLastError();
}
int match_stripes_dfa2_stamped(const Dfa2Device &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base,
                               size_t nstripes, uint32_t *accept, uint64_t *stamps, uint64_t *rounds, void *stream) {
    if (!nstripes) return 0;
    if (Dfa2::lds_bytes(p) > kDfa2MaxTable) return (int)hipErrorInvalidValue;
    size_t blocks = (nstripes + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(match_stripes2_stamped_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, p, bytes, nbytes, stripe, stripe_base,
                       accept, stamps, rounds);
    return (int)hipGetLastError();
}
int stamp_columns() { return kPhases + 2; }
}  // namespace dev
}  // namespace rrx

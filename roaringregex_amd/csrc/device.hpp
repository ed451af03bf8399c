// device.hpp — host-visible interface of the HIP kernels (kernels.hip).  gfx950 only.
#pragma once
#include <cstddef>
#include <cstdint>

namespace rrx {
namespace dev {

// Geometry of the batch kernel: a workgroup scans one TILE of the corpus; lane l owns the lines that
// START in bytes [l*SEG, (l+1)*SEG) of the tile and follows its last line past the segment end.
constexpr int kThreads = 256;
constexpr int kSeg = 128;
constexpr int kTile = kThreads * kSeg;          // 32 KiB of text per workgroup
constexpr int kHalo = 2048;                     // look-ahead staged in LDS; longer lines read HBM directly
constexpr int kMaxNfaWords = 8;                 // 256 positions per lane-resident state set
constexpr uint32_t kDfaLdsBudget = 96 * 1024;   // table bytes that may live in LDS next to the text tile

struct NfaMasks {                               // passed by value -> SGPRs
    uint32_t init[kMaxNfaWords], fin[kMaxNfaWords], chain[kMaxNfaWords], self[kMaxNfaWords], excm[kMaxNfaWords];
};

struct NfaDevice {                              // tables in HBM (copied to LDS by every workgroup)
    uint32_t W = 0, nbits = 0, any_exc = 0;
    NfaMasks masks;
    const uint32_t *B = nullptr;                // [256][W]
    const uint32_t *X = nullptr;                // [nbits][W]
};

struct DfaDevice {
    uint32_t nstates = 0, ncls = 0, start = 0;
    const uint8_t *cls = nullptr;               // [256]
    const uint16_t *next = nullptr;             // [nstates][ncls]
    const uint8_t *acc = nullptr;               // [nstates]
};

// All launchers are asynchronous on `stream` and return a hipError_t value (0 = success).
int count_newlines_per_tile(const uint8_t *bytes, size_t nbytes, uint32_t *tile_counts, size_t ntiles, void *stream);
int scan_tile_counts(const uint32_t *tile_counts, uint64_t *tile_base, size_t ntiles, void *stream);

int match_tiles_nfa(const NfaDevice &p, const uint8_t *bytes, size_t nbytes, const uint64_t *tile_base,
                    size_t ntiles, uint8_t *accept, void *stream);
int match_tiles_dfa(const DfaDevice &p, const uint8_t *bytes, size_t nbytes, const uint64_t *tile_base,
                    size_t ntiles, uint8_t *accept, void *stream);

// items i = bytes[off[i] .. off[i+1] - trim) ; trim = 1 drops a trailing delimiter byte per item
int match_extents_nfa(const NfaDevice &p, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim,
                      uint8_t *accept, void *stream);
int match_extents_dfa(const DfaDevice &p, const uint8_t *bytes, const uint64_t *off, size_t nitems, uint32_t trim,
                      uint8_t *accept, void *stream);

}  // namespace dev
}  // namespace rrx

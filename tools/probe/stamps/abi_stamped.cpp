// PROBE BUILD (not shipped): the C ABI plus one entry that runs the stamped stride-2 kernel on a corpus.
#include "../../../roaringregex_amd/csrc/abi.cpp"
namespace rrx { namespace dev {
int match_stripes_dfa2_stamped(const Dfa2Device &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base,
                               size_t nstripes, uint32_t *accept, uint64_t *stamps, uint64_t *rounds, void *stream);
int match_units_dfa2_stamped(const Dfa2Device &p, const uint8_t *bytes, size_t nbytes, uint32_t stripe, const uint64_t *stripe_base,
                             size_t nstripes, uint32_t *accept, uint32_t units_per_wg, uint64_t *stamps, uint64_t *rounds, void *stream);
int stamp_columns();
} }
extern "C" int rrx_probe_stamp_columns(void) { return rrx::dev::stamp_columns(); }
// d_stamps: workgroups x 16 waves x rrx_probe_stamp_columns() u64; d_rounds: workgroups x 16 waves x 32 u64 (start of round r); workgroups = ceil(stripes / 1024)
extern "C" int rrx_probe_match_stamped(const rrx_regex *re, const rrx_corpus *c, uint32_t *d_accept_bits, uint64_t *d_stamps, uint64_t *d_rounds, void *stream) {
    const DeviceTables *t;
    int rc = re->tables(c->device, &t);
    if (rc) return rc;
    if (!re->has_dfa2 || c->has_high) return fail(RRX_ERR_UNSUPPORTED, "stamps: the stride-2 engine only");
    HIP_TRY(hipMemsetAsync(d_accept_bits, 0, rrx_corpus_bitmap_words(c) * sizeof(uint32_t), (hipStream_t)stream));
    const uint32_t upw = (uint32_t)re->opt_units_per_wg.load();
    int e = upw ? dev::match_units_dfa2_stamped(t->dfa2, c->d_bytes, c->nbytes, c->stripe, c->d_base, c->nstripes, d_accept_bits, upw, d_stamps, d_rounds, stream)
                : dev::match_stripes_dfa2_stamped(t->dfa2, c->d_bytes, c->nbytes, c->stripe, c->d_base, c->nstripes, d_accept_bits, d_stamps, d_rounds, stream);
    if (e) return hip_fail((hipError_t)e, "stamped launch");
    return RRX_OK;
}

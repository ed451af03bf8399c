// PROBE BUILD (not shipped): the C ABI plus an entry that sets the order of the stride-2 table's rows and columns (from
// tools/probe/lds_layout_opt.py) before the tables are uploaded - does a frequency-aware layout pay on the GPU?
#include "../../../roaringregex_amd/csrc/abi.cpp"
extern "C" int rrx_probe_set_t2_order(rrx_regex *re, const uint32_t *row_slot, uint32_t nrows, const uint32_t *col_slot, uint32_t ncols) {
    if (!re->has_dfa2 || nrows != re->dfa2.nstates || ncols != re->dfa2.ncols || row_slot[0] != 0) return fail(RRX_ERR_ARG, "order does not fit the table");
    std::lock_guard<std::mutex> lock(re->mu);
    if (!re->on_device.empty()) return fail(RRX_ERR_ARG, "tables already uploaded");
    re->t2_row_slot.assign(row_slot, row_slot + nrows);
    re->t2_col_slot.assign(col_slot, col_slot + ncols);
    return RRX_OK;
}

// wc_int_rows.hip -- per-channel (whole-row group) INT8 / INT4 lowering over many tensors: instantiations of the
// row kernel (wc_row_kernels.h), kept in a translation unit of their own for build time.
#include <vector>

#include "wc_int_policy.h"
#include "wc_row_kernels.h"

namespace wc {

bool int_rows_fit(int w_dtype, const BatchItem* items, int n, int64_t group_size) {
    return rows_fit(w_dtype, items, n, group_size);
}

// items[i].packed: 1-byte codes (8-bit modes) or nibble-packed codes (4-bit modes)
int int_rows_run(int w_dtype, const BatchItem* items, int n, int mode, const char* what, hipStream_t s) {
    switch (mode) {
        case NNCF_AMD_INT8_ASYM: return run_rows<IntPolicyT<NNCF_AMD_INT8_ASYM>, false>(w_dtype, items, n, mode, what, s);
        case NNCF_AMD_INT8_SYM: return run_rows<IntPolicyT<NNCF_AMD_INT8_SYM>, false>(w_dtype, items, n, mode, what, s);
        case NNCF_AMD_INT4_ASYM: return run_rows<IntPolicyT<NNCF_AMD_INT4_ASYM>, true>(w_dtype, items, n, mode, what, s);
        case NNCF_AMD_INT4_SYM: return run_rows<IntPolicyT<NNCF_AMD_INT4_SYM>, true>(w_dtype, items, n, mode, what, s);
        default:
            set_error("%s: mode %d is not an integer mode", what, mode);
            return NNCF_AMD_ERR_MODE;
    }
}

}  // namespace wc

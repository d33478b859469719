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

// Per-channel error metric of the tensors of one launch (nncf_amd_int_quant_error[_batched] with group_size -1 or
// the row length): the row kernel without its stores.  The mode is read at run time here (one set of kernels).
bool err_rows_launch(int w_dtype, const ErrBatchArgs& a, float* ws, hipStream_t s) {
    if (a.n < 1 || a.row_start[a.n] == 0) return false;
    for (int i = 0; i < a.n; ++i)
        if (a.gs != -1 && a.gs != (int64_t)a.cols[i]) return false;
    switch (w_dtype) {
        case NNCF_AMD_F32: return run_err_rows<IntPolicy, NNCF_AMD_F32>(a, ws, s);
        case NNCF_AMD_F16: return run_err_rows<IntPolicy, NNCF_AMD_F16>(a, ws, s);
        case NNCF_AMD_BF16: return run_err_rows<IntPolicy, NNCF_AMD_BF16>(a, ws, s);
        default: return false;
    }
}

}  // namespace wc

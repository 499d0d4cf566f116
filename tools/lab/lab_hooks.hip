// lab_hooks.hip -- tools/bin/libts2d_lab.so only (csrc/ts2d_lab.h): the sort / scan test hooks and their rocPRIM comparators.
// The product library links no rocPRIM; the hand-written passes under test are the product's own objects (binning.hip), reached
// through the same internal entry point that knn.hip uses.
#pragma GCC visibility push(default)
#include "ts2d_lab.h"
#pragma GCC visibility pop
#include "ts2d_common.h"
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace
{
int sort_pairs_rocprim(const uint32_t *keys_in, const uint32_t *vals_in, uint32_t *keys_out, uint32_t *vals_out, size_t n, int end_bit, hipStream_t s)
{
    size_t bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, (unsigned)end_bit, s) != hipSuccess) return 2;
    void *tmp = nullptr;
    if (hipMalloc(&tmp, bytes ? bytes : 1) != hipSuccess) return 2;
    hipError_t e = rocprim::radix_sort_pairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, (unsigned)end_bit, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(tmp);
    return e == hipSuccess ? 0 : 2;
}

// the hand-written passes on copies of the caller's arrays, scratch from hipMalloc
int sort_pairs_handwritten(const uint32_t *keys_in, const uint32_t *vals_in, uint32_t *keys_out, uint32_t *vals_out, size_t n, int end_bit,
                           bool force_tickets, hipStream_t s)
{
    if (n == 0) return 0;
    const size_t arr = ts_align_up(n * 4), scratch = ts_radix_scratch_bytes(n);
    char *base = nullptr;
    if (hipMalloc((void **)&base, 4 * arr + scratch + TS_ALIGN) != hipSuccess) return 2;
    uint32_t *k[2] = {(uint32_t *)base, (uint32_t *)(base + arr)}, *v[2] = {(uint32_t *)(base + 2 * arr), (uint32_t *)(base + 3 * arr)};
    hipError_t e = hipMemcpyAsync(k[0], keys_in, n * 4, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(v[0], vals_in, n * 4, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess)
    {
        const int src = ts_radix_sort_pairs(k, v, n, end_bit, base + 4 * arr, s, force_tickets);
        e = hipMemcpyAsync(keys_out, k[src], n * 4, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(vals_out, v[src], n * 4, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e == hipSuccess) e = hipGetLastError();
    }
    (void)hipFree(base);
    return e == hipSuccess ? 0 : 2;
}
} // namespace

extern "C" {
int ts2d_test_sort_pairs(const uint32_t *keys_in, const uint32_t *vals_in, uint32_t *keys_out, uint32_t *vals_out, size_t n, int32_t end_bit,
                         int32_t which, void *stream)
{
    if (end_bit < 1 || end_bit > 32) return TS2D_ERR_INVALID;
    if (n && (!keys_in || !vals_in || !keys_out || !vals_out)) return TS2D_ERR_INVALID;
    const int rc = which == 1 ? sort_pairs_rocprim(keys_in, vals_in, keys_out, vals_out, n, end_bit, (hipStream_t)stream)
                              : sort_pairs_handwritten(keys_in, vals_in, keys_out, vals_out, n, end_bit, which == 2, (hipStream_t)stream);
    return rc ? TS2D_ERR_HIP : TS2D_OK;
}

// torch.quantile(keys as non-negative floats, q) by the library's radix select (select.hip).  `scratch`: ts2d_test_quantile_scratch_bytes() bytes;
// `out`: one float, both in device memory.
size_t ts2d_test_quantile_scratch_bytes(void) { return ts_quantile_scratch_bytes(); }
int ts2d_test_quantile(const uint32_t *keys, size_t n, float q, void *scratch, float *out, void *stream)
{
    if (n == 0 || !keys || !scratch || !out) return TS2D_ERR_INVALID;
    ts_quantile_threshold(keys, n, q, scratch, out, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? TS2D_OK : TS2D_ERR_HIP;
}

int ts2d_test_inclusive_scan_rocprim(const uint32_t *in, uint32_t *out, size_t n, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    size_t bytes = 0;
    if (rocprim::inclusive_scan(nullptr, bytes, in, out, n, rocprim::plus<uint32_t>(), s) != hipSuccess) return TS2D_ERR_HIP;
    void *tmp = nullptr;
    if (hipMalloc(&tmp, bytes ? bytes : 1) != hipSuccess) return TS2D_ERR_HIP;
    hipError_t e = rocprim::inclusive_scan(tmp, bytes, in, out, n, rocprim::plus<uint32_t>(), s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(tmp);
    return e == hipSuccess ? TS2D_OK : TS2D_ERR_HIP;
}
}

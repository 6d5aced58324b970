// Host side of the input stage: gathers the crop boxes of a batch (strided views into decoded images) into the contiguous pinned
// staging buffer the H2D copy reads.  Plain C++ threads, no GPU call; the Python loader calls it through ctypes, which drops the
// GIL for the duration, so the copy runs at memory speed beside the interpreter instead of inside it.
#include "common.h"
#include <atomic>
#include <string.h>
#include <thread>
#include <vector>

extern "C" int sc_host_gather_rows(int64_t n, const void* const* src, const int64_t* src_row_stride, const int64_t* rows, const int64_t* row_bytes,
                                   void* dst, const int64_t* dst_offset, int64_t dst_bytes, int threads) {
    SC_REQUIRE(n >= 0 && (n == 0 || (src && src_row_stride && rows && row_bytes && dst && dst_offset)), SC_ERR_ARG, "sc_host_gather_rows: bad argument");
    for (int64_t k = 0; k < n; ++k) {
        SC_REQUIRE(src[k] && rows[k] >= 0 && row_bytes[k] >= 0 && dst_offset[k] >= 0 && dst_offset[k] + rows[k] * row_bytes[k] <= dst_bytes, SC_ERR_SHAPE,
                   "sc_host_gather_rows: item %lld does not fit the staging buffer", (long long)k);
    }
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    std::atomic<int64_t> next{0};
    auto work = [&] {
        for (;;) {
            const int64_t k = next.fetch_add(1, std::memory_order_relaxed);
            if (k >= n) return;
            const char* s = (const char*)src[k];
            char* d = (char*)dst + dst_offset[k];
            const int64_t rb = row_bytes[k], st = src_row_stride[k];
            if (st == rb) memcpy(d, s, (size_t)(rows[k] * rb));
            else
                for (int64_t r = 0; r < rows[k]; ++r) memcpy(d + r * rb, s + r * st, (size_t)rb);
        }
    };
    if (threads == 1 || n < 2) {
        work();
        return SC_OK;
    }
    std::vector<std::thread> pool;
    pool.reserve(threads - 1);
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    return SC_OK;
}

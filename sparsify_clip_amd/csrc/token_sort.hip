// Index bookkeeping of the token-embedding scatter-add (sc_text_embed_bwd): the flat positions b * S + s sorted by token id.
// Key = token id for the positions at or before a caption's EOT, `vocab` for the positions behind it (their gradient is exactly zero
// under the causal mask; the scatter kernel ignores that key), then a stable LSD radix sort over the bits a key can have - the
// rocPRIM device primitive through its hipCUB front end (header-only, compiled into this library).  Rounds 1-2 built the keys with
// torch.where and sorted with torch.sort on the training hot path (sparsify_clip_amd/model.py).
#include "common.h"
#include <hipcub/hipcub.hpp>

namespace {

__global__ __launch_bounds__(256) void token_keys_kernel(const int64_t* tokens, const int32_t* eot, int64_t n, int seq, int64_t vocab, int64_t* keys,
                                                         int64_t* idx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t b = i / seq;
    const int s = (int)(i - b * seq);
    keys[i] = s <= eot[b] ? tokens[i] : vocab;
    idx[i] = i;
}

int key_bits(int64_t vocab) {   // keys are 0 .. vocab
    int bits = 1;
    while (bits < 63 && (int64_t(1) << bits) <= vocab) ++bits;
    return bits;
}

size_t sort_temp_bytes(int64_t n, int bits) {
    size_t bytes = 0;
    const hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const int64_t*)nullptr, (int64_t*)nullptr, (const int64_t*)nullptr, (int64_t*)nullptr,
                                                            (int)n, 0, bits, (hipStream_t) nullptr);
    return e == hipSuccess ? bytes : 0;
}

size_t align256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

extern "C" size_t sc_token_sort_workspace_bytes(int64_t n, int64_t vocab) {
    if (n <= 0 || vocab <= 0 || n >= (int64_t(1) << 31)) return 0;
    return 2 * align256((size_t)n * sizeof(int64_t)) + align256(sort_temp_bytes(n, key_bits(vocab))) + 256;
}

extern "C" int sc_token_sort(const int64_t* tokens, const int32_t* eot, int64_t batch, int64_t seq, int64_t vocab, int64_t* sorted_keys, int64_t* order,
                             void* ws, size_t ws_bytes, void* stream) {
    SC_REQUIRE(tokens && eot && sorted_keys && order && ws, SC_ERR_ARG, "sc_token_sort: null argument");
    const int64_t n = batch * seq;
    SC_REQUIRE(batch > 0 && seq > 0 && vocab > 0 && n < (int64_t(1) << 31), SC_ERR_SHAPE, "sc_token_sort: bad shape");
    SC_REQUIRE(ws_bytes >= sc_token_sort_workspace_bytes(n, vocab) && sc_aligned(ws, 16), SC_ERR_WORKSPACE, "sc_token_sort: workspace too small or misaligned");
    const int bits = key_bits(vocab);
    char* base = (char*)ws;
    int64_t* keys = (int64_t*)base;
    int64_t* idx = (int64_t*)(base + align256((size_t)n * sizeof(int64_t)));
    void* temp = base + 2 * align256((size_t)n * sizeof(int64_t));
    size_t temp_bytes = sort_temp_bytes(n, bits);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(token_keys_kernel, dim3((unsigned)sc_cdiv(n, 256)), dim3(256), 0, st, tokens, eot, n, (int)seq, vocab, keys, idx);
    SC_CHECK_LAUNCH();
    const hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, (const int64_t*)keys, sorted_keys, (const int64_t*)idx, order, (int)n, 0, bits, st);
    if (e != hipSuccess) return sc_set_error((int)e, "sc_token_sort: radix sort: %s", hipGetErrorString(e));
    return SC_OK;
}

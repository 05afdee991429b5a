// Host emulation of rocprim::segmented_radix_sort_keys -- TEST INFRASTRUCTURE ONLY (see ../hip/hip_runtime.h).
#pragma once
#include <algorithm>
#include <vector>
#include <hip/hip_runtime.h>

namespace rocprim {
template <class Key, class Off>
inline hipError_t segmented_radix_sort_keys(void* temp, size_t& bytes, const Key* in, Key* out, unsigned size,
                                            unsigned segments, Off begin, Off end, unsigned = 0, unsigned = 64,
                                            hipStream_t = nullptr, bool = false) {
    if (temp == nullptr) { bytes = 16; return hipSuccess; }
    std::copy(in, in + size, out);
    // radix order: NaN (positive quiet NaN, as the engine writes it) sorts after every number
    auto less = [](const Key& a, const Key& b) { return (a == a) && (!(b == b) || a < b); };
    for (unsigned s = 0; s < segments; s++) std::sort(out + begin[s], out + end[s], less);
    return hipSuccess;
}
template <class Key>
inline hipError_t radix_sort_keys(void* temp, size_t& bytes, const Key* in, Key* out, size_t size, unsigned = 0, unsigned = 64,
                                  hipStream_t = nullptr, bool = false) {
    if (temp == nullptr) { bytes = 16; return hipSuccess; }
    std::copy(in, in + size, out);
    std::sort(out, out + size);
    return hipSuccess;
}
template <class Key, class Val>
inline hipError_t radix_sort_pairs(void* temp, size_t& bytes, const Key* kin, Key* kout, const Val* vin, Val* vout, size_t size,
                                   unsigned begin_bit = 0, unsigned end_bit = 64, hipStream_t = nullptr, bool = false) {
    if (temp == nullptr) { bytes = 16; return hipSuccess; }
    std::vector<size_t> order(size);
    for (size_t q = 0; q < size; q++) order[q] = q;
    const Key mask = end_bit - begin_bit >= 8 * sizeof(Key) ? ~Key(0) : (((Key(1) << (end_bit - begin_bit)) - 1) << begin_bit);
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return (kin[a] & mask) < (kin[b] & mask); });   // stable, only the named bits
    for (size_t q = 0; q < size; q++) { kout[q] = kin[order[q]]; vout[q] = vin[order[q]]; }
    return hipSuccess;
}
}  // namespace rocprim

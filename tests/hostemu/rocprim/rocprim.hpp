// Host emulation of rocprim::segmented_radix_sort_keys -- TEST INFRASTRUCTURE ONLY (see ../hip/hip_runtime.h).
#pragma once
#include <algorithm>
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
}  // namespace rocprim

// aten_shim.h -- STAND-IN for at::TensorAccessor (esac_types.h:45-46), TEST INFRASTRUCTURE (oracle/_ref only).
#pragma once
#include <cstddef>
#include <cstdint>
namespace at {
template <typename T, size_t N> class TensorAccessor {
public:
    TensorAccessor(T* d, const int64_t* sizes, const int64_t* strides) : d_(d), sizes_(sizes), strides_(strides) {}
    TensorAccessor<T, N - 1> operator[](int64_t i) const { return TensorAccessor<T, N - 1>(d_ + i * strides_[0], sizes_ + 1, strides_ + 1); }
    int64_t size(int i) const { return sizes_[i]; }
private:
    T* d_; const int64_t* sizes_; const int64_t* strides_;
};
template <typename T> class TensorAccessor<T, 1> {
public:
    TensorAccessor(T* d, const int64_t* sizes, const int64_t* strides) : d_(d), sizes_(sizes), strides_(strides) {}
    T& operator[](int64_t i) const { return d_[i * strides_[0]]; }
    int64_t size(int i) const { return sizes_[i]; }
private:
    T* d_; const int64_t* sizes_; const int64_t* strides_;
};
}  // namespace at

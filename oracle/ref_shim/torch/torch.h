// torch/torch.h -- STAND-IN, TEST INFRASTRUCTURE (oracle/_ref only): at::Tensor with accessor<T,N>() over caller
// memory and a no-op PYBIND11_MODULE, so that the reference's esac.cpp compiles unmodified without libtorch.
#pragma once
#include <cstdint>
#include <vector>
#include "../aten_shim.h"
namespace at {
class Tensor {
public:
    Tensor() : data_(nullptr) {}
    Tensor(void* data, std::vector<int64_t> sizes) : data_(data), sizes_(std::move(sizes)), strides_(sizes_.size()) {
        int64_t s = 1;
        for (int i = (int)sizes_.size() - 1; i >= 0; i--) { strides_[i] = s; s *= sizes_[i]; }
    }
    template <typename T, size_t N> TensorAccessor<T, N> accessor() const {
        return TensorAccessor<T, N>(static_cast<T*>(data_), sizes_.data(), strides_.data());
    }
private:
    void* data_;
    std::vector<int64_t> sizes_, strides_;
};
}  // namespace at
struct PyModuleStub_ { template <typename F> void def(const char*, F, const char*) {} };
#define PYBIND11_MODULE(name, var) static void esac_pybind_stub_(PyModuleStub_& var)

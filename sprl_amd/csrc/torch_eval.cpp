// torch_eval.cpp — libsprl_amd_torch.so: the reference's GridNetwork (networks/GridNetwork.hpp:37-102) on the GPU.
//
// Loads the traced TorchScript policy/value CNN with LibTorch-ROCm and runs it on the dense leaf batch the
// tree kernel has already symmetrised and plane-encoded in HBM (no host round trip, no per-element tensor
// writes as in GridNetwork.hpp:72-97,104-107).  The MFMA work of the whole engine lives inside this call
// (MIOpen / hipBLASLt convolutions and GEMMs).  Kept in its own shared object so the core library has no
// torch dependency; C ABI, plain pointers only.
#include <torch/script.h>
#include <torch/torch.h>

#include <cstring>
#include <string>

namespace {
struct Model {
    torch::jit::Module module;
    int device = 0;
};

void put_err(char* err, int errlen, const std::string& msg) {
    if (err && errlen > 0) {
        strncpy(err, msg.c_str(), (size_t)errlen - 1);
        err[errlen - 1] = 0;
    }
}
}  // namespace

extern "C" {

void* sprl_torch_load(const char* path, int device, char* err, int errlen) {
    try {
        if (!torch::cuda::is_available()) {
            put_err(err, errlen, "LibTorch reports no ROCm device");
            return nullptr;
        }
        auto* m = new Model();
        m->device = device;
        m->module = torch::jit::load(path, torch::Device(torch::kCUDA, (c10::DeviceIndex)device));
        m->module.eval();                       // GridNetwork.hpp:67
        return m;
    } catch (const std::exception& e) {
        put_err(err, errlen, e.what());
        return nullptr;
    }
}

int sprl_torch_forward(void* handle, const float* planes, int batch, int nplanes, int rows, int cols,
                       float* logits, int actions, float* value, char* err, int errlen) {
    try {
        auto* m = static_cast<Model*>(handle);
        c10::InferenceMode guard;
        auto opts = torch::TensorOptions().dtype(torch::kFloat32).device(torch::kCUDA, (c10::DeviceIndex)m->device);
        auto in = torch::from_blob(const_cast<float*>(planes), { batch, nplanes, rows, cols }, opts);
        auto out = m->module.forward({ in }).toTuple();             // GridNetwork.hpp:99-102
        auto lo = out->elements()[0].toTensor();
        auto va = out->elements()[1].toTensor();
        if (lo.numel() != (int64_t)batch * actions || va.numel() != batch) {
            put_err(err, errlen, "model output shape does not match (logits[B,A], value[B,1])");
            return -1;
        }
        torch::from_blob(logits, { batch, actions }, opts).copy_(lo.reshape({ batch, actions }));
        torch::from_blob(value, { batch }, opts).copy_(va.reshape({ batch }));
        return 0;
    } catch (const std::exception& e) {
        put_err(err, errlen, e.what());
        return -1;
    }
}

void sprl_torch_free(void* handle) { delete static_cast<Model*>(handle); }

}  // extern "C"

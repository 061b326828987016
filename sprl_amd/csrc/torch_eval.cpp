// torch_eval.cpp — libsprl_amd_torch.so: the reference's GridNetwork (networks/GridNetwork.hpp:37-102) on the GPU.
//
// Loads the traced TorchScript policy/value CNN with LibTorch-ROCm and runs it on the dense leaf batch the
// tree kernel has already symmetrised and plane-encoded in HBM (no host round trip, no per-element tensor
// writes as in GridNetwork.hpp:72-97,104-107).  The MFMA work of the whole engine lives inside this call
// (MIOpen / hipBLASLt convolutions and GEMMs).  Kept in its own shared object so the core library has no
// torch dependency; C ABI, plain pointers only.
#include <torch/csrc/jit/api/module.h>
#include <torch/csrc/jit/ir/ir.h>
#include <torch/csrc/jit/passes/freeze_module.h>
#include <torch/script.h>
#include <torch/torch.h>

#include <cstring>
#include <string>

namespace {
struct Model {
    torch::jit::Module module;
    int device = 0;
};

// After freezing, every convolution carries its bias as a constant and ATen applies it with a separate full-tensor
// elementwise kernel (the convolution library does not add it).  Move the bias into an explicit broadcast add right
// behind the convolution: the JIT fuser then folds it into the BatchNorm/ReLU kernel that follows, so each
// convolution costs one elementwise pass over the activations instead of two.  Values are unchanged up to fp32
// rounding order ((conv + b) then BN, exactly as before).
int hoist_conv_bias(torch::jit::Module& module) {
    using namespace torch::jit;
    auto graph = module.get_method("forward").graph();
    std::vector<Node*> convs;
    for (Node* n : graph->nodes())
        if (n->kind() == aten::_convolution || n->kind() == aten::conv2d) convs.push_back(n);
    int moved = 0;
    for (Node* n : convs) {
        Value* bias = n->input(2);
        auto iv = toIValue(bias);
        if (!iv || !iv->isTensor()) continue;
        at::Tensor b = iv->toTensor();
        if (!b.defined() || b.dim() != 1) continue;
        WithInsertPoint guard(n);               // constants go in front of the convolution
        Value* none = graph->insertConstant(IValue());
        Value* b4 = graph->insertConstant(b.reshape({ 1, -1, 1, 1 }).contiguous());
        Value* one = graph->insertConstant(1);
        Node* add = graph->create(aten::add, { n->output(), b4, one });
        add->output()->setType(n->output()->type());
        add->insertAfter(n);
        n->output()->replaceAllUsesAfterNodeWith(add, add->output());
        n->replaceInput(2, none);
        ++moved;
    }
    return moved;
}

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
        // device < 0: host tensors — used only by the CPU unit test of the graph rewrite, never by the engine
        if (device >= 0 && !torch::cuda::is_available()) {
            put_err(err, errlen, "LibTorch reports no ROCm device");
            return nullptr;
        }
        auto* m = new Model();
        m->device = device;
        m->module = torch::jit::load(path, device >= 0 ? torch::Device(torch::kCUDA, (c10::DeviceIndex)device)
                                                       : torch::Device(torch::kCPU));
        m->module.eval();                       // GridNetwork.hpp:67
        if (!getenv("SPRL_TORCH_NO_REWRITE")) {
            try {
                torch::jit::Module frozen = torch::jit::freeze_module(m->module);
                hoist_conv_bias(frozen);
                m->module = frozen;
            } catch (const std::exception&) {
                // keep the module as loaded: the rewrite is an optimisation only
            }
        }
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
        auto opts = m->device >= 0
                        ? torch::TensorOptions().dtype(torch::kFloat32).device(torch::kCUDA, (c10::DeviceIndex)m->device)
                        : torch::TensorOptions().dtype(torch::kFloat32).device(torch::kCPU);
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

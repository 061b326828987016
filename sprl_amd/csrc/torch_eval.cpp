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
#include <hip/hip_runtime_api.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/script.h>
#include <torch/torch.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "busy_log.h"

extern "C" int sprl_bn_relu_inplace(float* x, const float* residual, const float* scale, const float* shift,
                                    int64_t numel, int channels, int hw, void* stream);
extern "C" int sprl_wino_conv64_dev(const float* x, const float* u, const float* scale, const float* shift, const float* res,
                                    float* y, int batch, int H, int W, int relu, const unsigned* batch_dev, void* stream);
extern "C" int sprl_wino_weight_layout(void);
extern "C" int sprl_wino_conv64_nchw(const float* x, const float* u, const float* scale, const float* shift, const float* res,
                                     float* y, int batch, int H, int W, int relu, void* stream);
extern "C" int sprl_wino_conv64_nchw_tiled(const float* x, const float* u, const float* scale, const float* shift, const float* res,
                                           float* y, int batch, int H, int W, int relu, int tile, const unsigned* batch_dev, void* stream);
extern "C" int sprl_wino_nchw_tile(int H, int W);
extern "C" int sprl_wino_t_board_floats(int H, int W, int tile);
extern "C" int sprl_wino_conv64_t(const float* x, const float* u, const float* scale, const float* shift, const float* res, float* y,
                                  int batch, int H, int W, int relu, int tile, const unsigned* batch_dev, void* stream);
extern "C" int sprl_stem_conv3x3_t(const float* planes, const float* w, const float* scale, const float* shift, float* y,
                                   long long batch, int P, int H, int W, int tile, const unsigned* batch_dev, void* stream);
extern "C" int sprl_tail_t(const float* x, const float* hw, const float* hb, const float* pfc_w, const float* pfc_b,
                           const float* vfc1_w, const float* vfc1_b, const float* vfc2_w, const float* vfc2_b, float* pmaps,
                           float* logits, float* value, int batch, int H, int W, int PC, int VC, int A, int HID, int tile,
                           const unsigned* batch_dev, void* stream);
extern "C" int sprl_stem_conv3x3_nchw_dev(const float* planes, const float* w, const float* scale, const float* shift, float* y,
                                          long long batch, int P, int H, int W, const unsigned* batch_dev, void* stream);
extern "C" int sprl_tail_nchw(const float* x, const float* hw, const float* hb, const float* pfc_w, const float* pfc_b,
                              const float* vfc1_w, const float* vfc1_b, const float* vfc2_w, const float* vfc2_b, float* pmaps,
                              float* logits, float* value, int batch, int H, int W, int PC, int VC, int A, int HID,
                              const unsigned* batch_dev, void* stream);
extern "C" int sprl_stem_conv3x3_w(const float* planes, const float* w, const float* scale, const float* shift, float* y,
                                   int batch, int P, int H, int W, const unsigned* batch_dev, void* stream);
extern "C" int sprl_stem_conv3x3_nchw(const float* planes, const float* w, const float* scale, const float* shift, float* y,
                                      long long batch, int P, int H, int W, void* stream);
extern "C" int sprl_wino_conv64_heads(const float* x, const float* u, const float* scale, const float* shift, const float* res,
                                      int batch, int H, int W, const unsigned* batch_dev, const float* hw, const float* hb,
                                      float* maps_out, void* stream);
extern "C" int sprl_tail_fc_form(const float* x, const float* maps_in, const float* hw, const float* hb, const float* pfc_w,
                                 const float* pfc_b, const float* vfc1_w, const float* vfc1_b, const float* vfc2_w, const float* vfc2_b,
                                 float* logits, float* value, int batch, int H, int W, int PC, int VC, int A, int HID, int no_mfma,
                                 const unsigned* batch_dev, void* stream);
extern "C" int sprl_wino_conv64_heads_fc(const float* x, const float* u, const float* scale, const float* shift, const float* res,
                                         int batch, int H, int W, const unsigned* batch_dev, const float* hw, const float* hb,
                                         const float* pfc_w, const float* pfc_b, const float* vfc1_w, const float* vfc1_b,
                                         const float* vfc2_w, const float* vfc2_b, float* logits, float* value, int A, int HID,
                                         void* stream);
extern "C" int sprl_wino_conv64_stem(const float* planes, const float* stem_w, const float* stem_scale, const float* stem_shift, float* x0,
                                     const float* u, const float* scale, const float* shift, float* y, int batch, int H, int W,
                                     const unsigned* batch_dev, void* stream);
extern "C" int sprl_wino_conv64_t_occ(const float* x, const float* u, const float* scale, const float* shift, const float* res, float* y,
                                      int batch, int H, int W, int relu, int tile, int occ, const unsigned* batch_dev, void* stream);
extern "C" int sprl_stem_conv3x3_w_form(const float* planes, const float* w, const float* scale, const float* shift, float* y,
                                        int batch, int P, int H, int W, int valu, const unsigned* batch_dev, void* stream);
extern "C" int sprl_stem_conv3x3_nchw_valu(const float* planes, const float* w, const float* scale, const float* shift, float* y,
                                           long long batch, int P, int H, int W, void* stream);
extern "C" int sprl_tail_heads_fc(const float* x, const float* hw, const float* hb, const float* pfc_w, const float* pfc_b,
                                  const float* vfc1_w, const float* vfc1_b, const float* vfc2_w, const float* vfc2_b,
                                  float* logits, float* value, int batch, int H, int W, int PC, int VC, int A, int HID,
                                  const unsigned* batch_dev, void* stream);
extern "C" int sprl_heads_conv1x1_relu(const float* x, const float* w, const float* bias, float* out_p, float* out_v,
                                       int64_t batch, int C, int HW, int PC, int VC, int board_w, void* stream);

namespace {
// Lab switches: environment toggles that move a forward off the hand-written path or pick a variant of it (A/B measurements,
// tests of the fallback paths).  They are read ONCE, when a model is loaded - never per forward or per launch - kept in the model
// and reported by sprl_torch_path_info, so a timed run can refuse to start on anything but the default path (bench.py does).
struct PathSwitches {
    bool no_native = false, no_rewrite = false, no_winograd = false, no_conv_tail = false, no_conv_fc = false, no_conv_stem = false, no_nchw_native = false,
         no_nchw_stem = false, no_winograd_nchw = false, no_fused_tail = false, stem_valu = false, tail_no_mfma = false;
    int nchw_tile = 0, f3_occ = 0;
    std::string active;                  // the names that were set, space separated ("" = the default path)
    static PathSwitches from_env() {
        PathSwitches sw;
        auto flag = [&](const char* name, bool& dst) {
            if (getenv(name)) {
                dst = true;
                sw.active += std::string(sw.active.empty() ? "" : " ") + name;
            }
        };
        auto num = [&](const char* name, int& dst) {
            if (const char* v = getenv(name)) {
                dst = atoi(v);
                sw.active += std::string(sw.active.empty() ? "" : " ") + name + "=" + v;
            }
        };
        flag("SPRL_TORCH_NO_NATIVE", sw.no_native);
        flag("SPRL_TORCH_NO_REWRITE", sw.no_rewrite);
        flag("SPRL_TORCH_NO_WINOGRAD", sw.no_winograd);
        flag("SPRL_TORCH_NO_CONV_TAIL", sw.no_conv_tail);
        flag("SPRL_TORCH_NO_CONV_FC", sw.no_conv_fc);
        flag("SPRL_TORCH_NO_CONV_STEM", sw.no_conv_stem);
        flag("SPRL_TORCH_NO_NCHW_NATIVE", sw.no_nchw_native);
        flag("SPRL_TORCH_NO_NCHW_STEM", sw.no_nchw_stem);
        flag("SPRL_TORCH_NO_WINOGRAD_NCHW", sw.no_winograd_nchw);
        flag("SPRL_TORCH_NO_FUSED_TAIL", sw.no_fused_tail);
        flag("SPRL_STEM_VALU", sw.stem_valu);
        flag("SPRL_TAIL_NO_MFMA", sw.tail_no_mfma);
        num("SPRL_WINO_NCHW_TILE", sw.nchw_tile);
        num("SPRL_WINO_F3_OCC", sw.f3_occ);
        if (sw.nchw_tile != 0 && sw.nchw_tile != 3) sw.nchw_tile = 4;
        return sw;
    }
};

// The network of src/networks/grid_networks.py:30-79 with its parameters taken from the traced module, evaluated as
// MIOpen convolutions (no bias) + one hand-written fused epilogue pass per convolution (cnn_epilogue.hip).
struct NativeNet {
    bool ok = false;
    at::Tensor stem_w, stem_scale, stem_shift;
    struct Block { at::Tensor w1, s1, t1, w2, s2, t2, u1, u2, u1t, u2t; };     // u*: Winograd-domain weights (cnn_wino.hip);
                                                                               // u*t: for the any-board kernel on layout T in the
                                                                               // tiling `ut_tile`, built on first use
    int ut_tile = 0;
    bool wino = false;              // every trunk convolution is 64 -> 64: the hand-written Winograd/MFMA kernel applies
    std::vector<Block> blocks;
    at::Tensor pconv_w, pconv_b, pfc_w, pfc_b, vconv_w, vconv_b, vfc1_w, vfc1_b, vfc2_w, vfc2_b;
    at::Tensor heads_w, heads_b;     // [PC + VC][C] and [PC + VC]: both 1x1 head convolutions as one fused pass
    int pc = 0, vc = 0;
};

// live timing of the trunk convolution kernel (profile mode): HIP event pairs on the stream the kernel is launched on
busy::Log g_conv_busy;

struct ConvProfile {
    bool on = false;
    bool per_launch = false;             // mode 2: one event pair per convolution launch (kernel durations comparable with a profiler's
                                         // per-kernel average; costs two queue packets per launch - never used in a timed region)
    std::vector<hipEvent_t> ev;          // pairs (start, end), resolved lazily
    std::vector<int> ev_kind;            // per pair: the kind of its (last) launch - meaningful in per-launch mode
    double ms = 0.0;
    int64_t launches = 0, boards = 0;
    // per-launch mode: time and launches by kind of trunk-convolution launch: 0 = plain (no residual), 1 = with residual,
    // 2 = with the stem in its prologue, 3 = with the heads (and FC layers) behind it
    double kind_ms[4] = { 0.0, 0.0, 0.0, 0.0 };
    int64_t kind_launches[4] = { 0, 0, 0, 0 };
    int open_kind = 0;
    busy::Chain chain;                   // the same intervals on the process-wide clock (several models on several streams)
    // The trunk convolutions of a forward follow each other on the stream with nothing between them, so ONE pair of events brackets
    // all of them (open before the first, note per launch, close behind the last): an event record is a packet of its own in the
    // hardware queue, and two per convolution cost the bench 4 % (round 3: 442.9 games/s without the events, 426 with them).
    hipEvent_t open_ev = nullptr;
    int64_t open_launches = 0, open_boards = 0;
    void open(hipStream_t st) {
        if (!on) return;
        if (open_ev) busy::put_event(open_ev);       // (a forward that failed half way)
        open_ev = busy::get_event();
        if (open_ev) (void)hipEventRecord(open_ev, st);
        open_launches = open_boards = 0;
    }
    void note(int64_t nboards, int kind = 0) {
        if (!open_ev) return;
        open_launches++;
        open_boards += nboards;
        open_kind = kind;
    }
    void close(hipStream_t st) {
        if (!open_ev) return;
        hipEvent_t e1 = open_launches ? busy::get_event() : nullptr;
        if (!e1) {
            busy::put_event(open_ev);
            open_ev = nullptr;
            return;
        }
        (void)hipEventRecord(e1, st);
        ev.push_back(open_ev);
        ev.push_back(e1);
        ev_kind.push_back(open_kind);
        open_ev = nullptr;
        launches += open_launches;
        boards += open_boards;
        if (ev.size() >= 8192) resolve();
    }
    void resolve() {
        std::vector<std::pair<double, double>> iv;
        iv.reserve(ev.size() / 2);
        for (size_t i = 0; i + 1 < ev.size(); i += 2) {
            double t = 0.0;
            (void)hipEventSynchronize(ev[i + 1]);
            iv.push_back(chain.resolve(ev[i], ev[i + 1], &t));
            ms += t;
            if (per_launch) {
                kind_ms[ev_kind[i / 2] & 3] += t;
                kind_launches[ev_kind[i / 2] & 3]++;
            }
            busy::put_event(ev[i + 1]);
        }
        ev.clear();
        ev_kind.clear();
        g_conv_busy.add(iv);
    }
    ~ConvProfile() {
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        if (open_ev) (void)hipEventDestroy(open_ev);
        chain.release();
    }
};

struct Model {
    torch::jit::Module module;
    int device = 0;
    PathSwitches sw;             // resolved once at load
    NativeNet native;
    ConvProfile prof;
    at::Tensor act[3], maps;     // activation buffers of the hand-written path, kept across calls (no allocator traffic, any stream)
    at::Tensor nact[3], npmaps;  // the same for boards wider than 8 (NCHW with the slack sprl_wino_conv64_nchw needs; policy head maps)
    int64_t nact_floats = 0, nact_pmaps = 0;     // allocated sizes: activations (floats per buffer), policy maps (floats)
    int last_stem = 0;           // 1: the last 8x8 forward ran the stem inside the first trunk convolution (no stem launch)
    int last_tail = 0;           // what the last 8x8 forward ended in: 2 = conv + heads + FC in one launch, 1 = heads fused + FC kernel, 0 = separate tail
};

// scale = gamma / sqrt(var + eps), shift = (conv_bias - mean) * scale + beta  (BatchNorm2d eval, eps = 1e-5)
bool fold_bn(const std::map<std::string, at::Tensor>& t, const std::string& conv, const std::string& bn, at::Tensor& w,
             at::Tensor& scale, at::Tensor& shift) {
    auto W = t.find(conv + ".weight"), B = t.find(conv + ".bias"), g = t.find(bn + ".weight"), b = t.find(bn + ".bias"),
         m = t.find(bn + ".running_mean"), v = t.find(bn + ".running_var");
    if (W == t.end() || B == t.end() || g == t.end() || b == t.end() || m == t.end() || v == t.end()) return false;
    if (W->second.dim() != 4 || W->second.size(2) != 3 || W->second.size(3) != 3) return false;
    w = W->second.contiguous();
    scale = (g->second / at::sqrt(v->second + 1e-5)).contiguous();
    shift = ((B->second - m->second) * scale + b->second).contiguous();
    return true;
}

// U = G g G^T for F(4x4, 3x3), computed in double and stored in the lane order of the kernel's A operand:
// U2[p = xi * 6 + nu][s][kb][lane] with output channel k = 16 kb + lane % 16, input channel = slot lane / 16 of group s.
// F(3x3, 3x3) (interpolation points 0, 1, -1, 2, inf; the kernel's B^T and A^T are in cnn_wino.hip): U = G g G^T is 5x5, stored
// as U4[p / 4][s][kb][lane][p % 4] with 7 quads (positions 25..27 are zero and never multiplied).
void wino_transform_f3(const float* g, float* up, bool quad_order = false) {
    static const double G[5][3] = { { 1.0 / 2, 0, 0 }, { 1.0 / 2, 1.0 / 2, 1.0 / 2 }, { 1.0 / 6, -1.0 / 6, 1.0 / 6 }, { 1.0 / 6, 1.0 / 3, 2.0 / 3 }, { 0, 0, 1 } };
    for (size_t i = 0; i < (size_t)7 * 16 * 4 * 64 * 4; ++i) up[i] = 0.0f;
    for (int k = 0; k < 64; ++k)
        for (int c = 0; c < 64; ++c) {
            const float* gk = g + ((size_t)k * 64 + c) * 9;
            double t[5][3];
            for (int a = 0; a < 5; ++a)
                for (int j = 0; j < 3; ++j) t[a][j] = G[a][0] * gk[j] + G[a][1] * gk[3 + j] + G[a][2] * gk[6 + j];
            for (int a = 0; a < 5; ++a)
                for (int b = 0; b < 5; ++b) {
                    const double v = t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2];
                    // K step s reads input channels 16 (s >> 2) + 4 slot + (s & 3) (layout W / NCHW kernels) or, with quad_order,
                    // 4 s + slot (layout T: a K step is one channel quad)
                    const int s = quad_order ? c >> 2 : 4 * (c >> 4) + (c & 3), slot = quad_order ? c & 3 : (c >> 2) & 3;
                    const int p = a * 5 + b, kb = k >> 4, lane = slot * 16 + (k & 15);
                    up[((((size_t)(p >> 2) * 16 + s) * 4 + kb) * 64 + lane) * 4 + (p & 3)] = (float)v;
                }
        }
}

void wino_transform(const float* g, float* up, bool quad_order = false) {
    const bool packed4 = sprl_wino_weight_layout() == 2;       // U4[p / 4][s][kb][lane][p % 4] instead of U2[p][s][kb][lane]
    static const double G[6][3] = { { 1.0 / 4, 0, 0 },         { -1.0 / 6, -1.0 / 6, -1.0 / 6 }, { -1.0 / 6, 1.0 / 6, -1.0 / 6 },
                                    { 1.0 / 24, 1.0 / 12, 1.0 / 6 }, { 1.0 / 24, -1.0 / 12, 1.0 / 6 }, { 0, 0, 1 } };
    for (int k = 0; k < 64; ++k)
        for (int c = 0; c < 64; ++c) {
            const float* gk = g + ((size_t)k * 64 + c) * 9;
            double t[6][3];
            for (int a = 0; a < 6; ++a)
                for (int j = 0; j < 3; ++j) t[a][j] = G[a][0] * gk[j] + G[a][1] * gk[3 + j] + G[a][2] * gk[6 + j];
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 6; ++b) {
                    const double v = t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2];
                    // K-loop step s reads group s of layout W: input channel c = 16 (s >> 2) + 4 slot + (s & 3)
                    const int s = quad_order ? c >> 2 : 4 * (c >> 4) + (c & 3), slot = quad_order ? c & 3 : (c >> 2) & 3;
                    const int p = a * 6 + b, kb = k >> 4, lane = slot * 16 + (k & 15);
                    if (packed4) up[((((size_t)(p >> 2) * 16 + s) * 4 + kb) * 64 + lane) * 4 + (p & 3)] = (float)v;
                    else up[(((size_t)p * 16 + s) * 4 + kb) * 64 + lane] = (float)v;
                }
        }
}

at::Tensor wino_weights(const at::Tensor& w_dev, int tile = 4, bool quad_order = false) {
    at::Tensor w = w_dev.to(at::kCPU, at::kFloat).contiguous();
    at::Tensor u = at::empty({ (tile == 3 ? 28 : 36) * 64 * 64 }, at::TensorOptions().dtype(at::kFloat));
    if (tile == 3) wino_transform_f3(w.data_ptr<float>(), u.data_ptr<float>(), quad_order);
    else wino_transform(w.data_ptr<float>(), u.data_ptr<float>(), quad_order);
    return u.to(w_dev.device());
}

void build_native(Model* m) {
    std::map<std::string, at::Tensor> t;
    for (const auto& p : m->module.named_parameters()) t[p.name] = p.value.detach();
    for (const auto& b : m->module.named_buffers()) t[b.name] = b.value.detach();
    NativeNet n;
    if (!fold_bn(t, "conv", "bn", n.stem_w, n.stem_scale, n.stem_shift)) return;
    for (int i = 0;; ++i) {
        const std::string pre = "residual_blocks." + std::to_string(i);
        if (t.find(pre + ".conv1.weight") == t.end()) break;
        NativeNet::Block blk;
        if (!fold_bn(t, pre + ".conv1", pre + ".bn1", blk.w1, blk.s1, blk.t1)) return;
        if (!fold_bn(t, pre + ".conv2", pre + ".bn2", blk.w2, blk.s2, blk.t2)) return;
        n.blocks.push_back(blk);
    }
    const char* need[] = { "policy_conv.weight", "policy_conv.bias", "policy_fc.weight", "policy_fc.bias", "value_conv.weight",
                           "value_conv.bias", "value_fc1.weight", "value_fc1.bias", "value_fc2.weight", "value_fc2.bias" };
    for (const char* k : need)
        if (t.find(k) == t.end()) return;
    n.pconv_w = t["policy_conv.weight"]; n.pconv_b = t["policy_conv.bias"];
    n.pfc_w = t["policy_fc.weight"].t().contiguous(); n.pfc_b = t["policy_fc.bias"];
    n.vconv_w = t["value_conv.weight"]; n.vconv_b = t["value_conv.bias"];
    n.vfc1_w = t["value_fc1.weight"].t().contiguous(); n.vfc1_b = t["value_fc1.bias"];
    n.vfc2_w = t["value_fc2.weight"].t().contiguous(); n.vfc2_b = t["value_fc2.bias"];
    n.pc = (int)n.pconv_w.size(0);
    n.vc = (int)n.vconv_w.size(0);
    if (n.pconv_w.size(2) != 1 || n.vconv_w.size(2) != 1 || n.pc + n.vc < 2 || n.pc + n.vc > 4) return;
    n.heads_w = at::cat({ n.pconv_w.reshape({ n.pc, -1 }), n.vconv_w.reshape({ n.vc, -1 }) }, 0).contiguous();
    n.heads_b = at::cat({ n.pconv_b, n.vconv_b }, 0).contiguous();
    // every tensor of the traced module must be accounted for, otherwise this is not the architecture we know
    size_t expected = 6 + n.blocks.size() * 12 + 10 + (1 + 2 * n.blocks.size());   // + num_batches_tracked buffers
    if (t.size() != expected) return;
    n.wino = !n.blocks.empty() && n.stem_w.size(0) == 64 && !m->sw.no_winograd;
    for (const auto& b : n.blocks)
        n.wino = n.wino && b.w1.size(0) == 64 && b.w1.size(1) == 64 && b.w2.size(0) == 64 && b.w2.size(1) == 64;
    if (n.wino)
        for (auto& b : n.blocks) {
            b.u1 = wino_weights(b.w1);
            b.u2 = wino_weights(b.w2);
        }
    n.ok = true;
    m->native = n;
}

bool epilogue(at::Tensor& x, const at::Tensor& scale, const at::Tensor& shift, const at::Tensor* residual) {
    const int C = (int)x.size(1), hw = (int)(x.size(2) * x.size(3));
    return sprl_bn_relu_inplace(x.data_ptr<float>(), residual ? residual->data_ptr<float>() : nullptr,
                                scale.data_ptr<float>(), shift.data_ptr<float>(), x.numel(), C, hw, nullptr) == 0;
}

// Trunk entirely in hand-written kernels: stem (VALU) -> residual blocks (Winograd on fp32 MFMA, cnn_wino.hip) -> both
// 1x1 heads, activations in layout W; only the three small fully connected layers go through the BLAS library.
bool forward_wino(Model* mdl, const at::Tensor& in, at::Tensor& p, at::Tensor& v, ConvProfile* prof, float* logits_out,
                  float* value_out, bool* wrote_outputs, const unsigned* batch_dev = nullptr, void* stream = nullptr) {
    const NativeNet& n = mdl->native;
    const int B = (int)in.size(0), P = (int)in.size(1), H = (int)in.size(2), W = (int)in.size(3);
    auto opts = in.options();
    if (!mdl->act[0].defined() || mdl->act[0].size(0) < B) {
        for (auto& t : mdl->act) t = at::empty({ B, 4096 }, opts);
        mdl->maps = at::empty({ B, (int64_t)3 * 64 }, opts);
        (void)hipDeviceSynchronize();            // the buffers may be used on another stream than the one that allocated them
    }
    at::Tensor x = mdl->act[0], y = mdl->act[1], z = mdl->act[2];
    // 3 input planes: the stem runs in the prologue of the first trunk convolution (round 4) - no stem launch
    const bool stem_in_conv = P == 3 && !mdl->sw.stem_valu && !mdl->sw.no_conv_stem && in.is_contiguous();
    mdl->last_stem = stem_in_conv ? 1 : 0;
    if (!stem_in_conv &&
        sprl_stem_conv3x3_w_form(in.data_ptr<float>(), n.stem_w.data_ptr<float>(), n.stem_scale.data_ptr<float>(),
                                 n.stem_shift.data_ptr<float>(), x.data_ptr<float>(), B, P, H, W, mdl->sw.stem_valu ? 1 : 0, batch_dev, stream) != 0)
        return false;
    if (prof) prof->open((hipStream_t)stream);       // one event pair around all the trunk convolutions of this forward
    auto conv = [&](const at::Tensor& src, const at::Tensor& u, const at::Tensor& sc, const at::Tensor& sh, const float* res,
                    at::Tensor& dst) {
        const int rc = sprl_wino_conv64_dev(src.data_ptr<float>(), u.data_ptr<float>(), sc.data_ptr<float>(), sh.data_ptr<float>(),
                                            res, dst.data_ptr<float>(), B, H, W, 1, batch_dev, stream);
        if (prof) {
            prof->note(B, res ? 1 : 0);
            if (prof->per_launch) {                  // close this launch's pair, open the next one's
                prof->close((hipStream_t)stream);
                prof->open((hipStream_t)stream);
            }
        }
        return rc == 0;
    };
    const int A0 = (int)n.pfc_w.size(1), HID0 = (int)n.vfc1_w.size(1);
    // the last convolution can carry the whole tail (heads + FC layers) behind its inverse transform
    const bool fuse_last = logits_out && value_out && n.pc == 2 && n.vc == 1 && HID0 <= 64 && n.vfc2_w.numel() == HID0 &&
                           !mdl->sw.no_conv_tail;
    for (size_t bi = 0; bi < n.blocks.size(); ++bi) {
        const auto& b = n.blocks[bi];
        if (bi == 0 && stem_in_conv) {
            const int rc = sprl_wino_conv64_stem(in.data_ptr<float>(), n.stem_w.data_ptr<float>(), n.stem_scale.data_ptr<float>(),
                                                 n.stem_shift.data_ptr<float>(), x.data_ptr<float>(), b.u1.data_ptr<float>(), b.s1.data_ptr<float>(),
                                                 b.t1.data_ptr<float>(), y.data_ptr<float>(), B, H, W, batch_dev, stream);
            if (prof) {
                prof->note(B, 2);
                if (prof->per_launch) {
                    prof->close((hipStream_t)stream);
                    prof->open((hipStream_t)stream);
                }
            }
            if (rc != 0) return false;
        } else if (!conv(x, b.u1, b.s1, b.t1, nullptr, y)) return false;
        if (fuse_last && bi + 1 == n.blocks.size()) {
            // last convolution + both head convolutions + the FC layers in ONE kernel: the forward ends in this launch (round 4)
            if (!mdl->sw.no_conv_fc) {
                const int rc = sprl_wino_conv64_heads_fc(y.data_ptr<float>(), b.u2.data_ptr<float>(), b.s2.data_ptr<float>(),
                                                         b.t2.data_ptr<float>(), x.data_ptr<float>(), B, H, W, batch_dev,
                                                         n.heads_w.data_ptr<float>(), n.heads_b.data_ptr<float>(), n.pfc_w.data_ptr<float>(),
                                                         n.pfc_b.data_ptr<float>(), n.vfc1_w.data_ptr<float>(), n.vfc1_b.data_ptr<float>(),
                                                         n.vfc2_w.data_ptr<float>(), n.vfc2_b.data_ptr<float>(), logits_out, value_out, A0, HID0,
                                                         stream);
                if (rc == 0) {
                    if (prof) {
                        prof->note(B, 3);
                        prof->close((hipStream_t)stream);
                    }
                    mdl->last_tail = 2;
                    *wrote_outputs = true;
                    return true;
                }
                if (rc != -1) return false;      // (-1: shape not covered - the two-kernel form below)
            }
            // last convolution + both head convolutions in one kernel (the trunk output is never written), then the FC layers
            at::Tensor maps = mdl->maps;
            const int rc = sprl_wino_conv64_heads(y.data_ptr<float>(), b.u2.data_ptr<float>(), b.s2.data_ptr<float>(),
                                                  b.t2.data_ptr<float>(), x.data_ptr<float>(), B, H, W, batch_dev,
                                                  n.heads_w.data_ptr<float>(), n.heads_b.data_ptr<float>(), maps.data_ptr<float>(),
                                                  stream);
            if (prof) {
                prof->note(B, 3);
                prof->close((hipStream_t)stream);
            }
            if (rc != 0) return false;
            if (sprl_tail_fc_form(nullptr, maps.data_ptr<float>(), n.heads_w.data_ptr<float>(), n.heads_b.data_ptr<float>(),
                                  n.pfc_w.data_ptr<float>(), n.pfc_b.data_ptr<float>(), n.vfc1_w.data_ptr<float>(), n.vfc1_b.data_ptr<float>(),
                                  n.vfc2_w.data_ptr<float>(), n.vfc2_b.data_ptr<float>(), logits_out, value_out, B, H, W, n.pc, n.vc, A0, HID0,
                                  mdl->sw.tail_no_mfma ? 1 : 0, batch_dev, stream) != 0)
                return false;
            mdl->last_tail = 1;
            *wrote_outputs = true;
            return true;
        }
        if (!conv(y, b.u2, b.s2, b.t2, x.data_ptr<float>(), z)) return false;
        std::swap(x, z);
    }
    if (prof) prof->close((hipStream_t)stream);
    const int HW = H * W;
    const int A = (int)n.pfc_w.size(1), HID = (int)n.vfc1_w.size(1);
    if (logits_out && value_out && n.vfc2_w.numel() == HID &&
        sprl_tail_heads_fc(x.data_ptr<float>(), n.heads_w.data_ptr<float>(), n.heads_b.data_ptr<float>(), n.pfc_w.data_ptr<float>(),
                           n.pfc_b.data_ptr<float>(), n.vfc1_w.data_ptr<float>(), n.vfc1_b.data_ptr<float>(),
                           n.vfc2_w.data_ptr<float>(), n.vfc2_b.data_ptr<float>(), logits_out, value_out, B, H, W, n.pc, n.vc, A,
                           HID, batch_dev, stream) == 0) {
        mdl->last_tail = 0;
        *wrote_outputs = true;                   // heads + FC layers fused, results already in the caller's buffers
        return true;
    }
    if (batch_dev || stream) return false;       // the device-side count / a private stream need the fused tail
    p = at::empty({ B, (int64_t)n.pc * HW }, opts);
    v = at::empty({ B, (int64_t)n.vc * HW }, opts);
    return sprl_heads_conv1x1_relu(x.data_ptr<float>(), n.heads_w.data_ptr<float>(), n.heads_b.data_ptr<float>(),
                                   p.data_ptr<float>(), v.data_ptr<float>(), B, 64, HW, n.pc, n.vc, W, nullptr) == 0;
}

// NCHW activation [B][64][H][W] with 16 readable bytes in front of it and 32 behind it (sprl_wino_conv64_nchw fetches patch
// rows 16 + 8 bytes at a time, starting one column left of the tile)
at::Tensor nchw_act(int64_t B, int H, int W, const at::TensorOptions& opts) {
    const int64_t n = B * 64 * H * W;
    return at::empty({ n + 12 }, opts).narrow(0, 4, n).view({ B, 64, H, W });
}

// Boards wider than 8 (Go 9x9, 19x19; any H x W up to 64) with a 64-channel trunk, END TO END in hand-written kernels and
// without the host: NCHW stem on the matrix cores -> any-board Winograd/MFMA trunk -> heads + FC tail (cnn_epilogue.hip:
// tail_nchw_kernel), every kernel taking the board count from device memory when `batch_dev` is given (`cap` is then the
// capacity the grids and the buffers are sized for).  The activation buffers are kept across calls; logits and values are
// written in place.  No library call, no allocation, no copy, no synchronisation per forward (the earlier
// wide-board path needed the batch size on the host for three rocBLAS GEMMs and allocated / copied per round).
// Returns false when this network / board is not covered (the caller falls back to the library path).
bool nchw_covered(const Model* mdl, int P, int H, int W, int actions) {
    const NativeNet& n = mdl->native;
    const int HID = (int)n.vfc1_w.size(1);
    return n.wino && (P == 3 || P == 17) && H >= 1 && W >= 1 && H <= 64 && W <= 64 && n.pc == 2 && n.vc == 1 &&
           n.pfc_w.size(0) == (int64_t)2 * H * W && n.pfc_w.size(1) == actions && n.vfc1_w.size(0) == (int64_t)H * W && HID <= 64 &&
           n.vfc2_w.numel() == HID && (size_t)(3 * 64 + 3 + (3 * H * W > 512 ? 8 : 16) * (3 * H * W + 5 * 64)) * 4 <= 64 * 1024 &&
           (size_t)((3 * H * W > 512 ? 8 : 16) * (2 * H * W + 128)) * 4 <= 64 * 1024 && !mdl->sw.no_nchw_native;
}

bool forward_nchw(Model* mdl, const float* planes, int cap, int P, int H, int W, ConvProfile* prof, float* logits_out,
                  float* value_out, const unsigned* batch_dev, void* stream, const at::TensorOptions& opts) {
    const int tile = mdl->sw.nchw_tile ? mdl->sw.nchw_tile : sprl_wino_nchw_tile(H, W);
    if (mdl->native.ut_tile != tile) {               // first board of this tiling on this model: filters for the layout-T kernel
        for (auto& b : mdl->native.blocks) {
            b.u1t = wino_weights(b.w1, tile, true);
            b.u2t = wino_weights(b.w2, tile, true);
        }
        mdl->native.ut_tile = tile;
        (void)hipDeviceSynchronize();
    }
    const NativeNet& n = mdl->native;
    const int64_t board_floats = sprl_wino_t_board_floats(H, W, tile);
    const int A = (int)n.pfc_w.size(1), HID = (int)n.vfc1_w.size(1);
    if ((long long)cap * board_floats * 4 >= 0x7fffff00LL || (long long)cap * H * W * P * 4 >= 0x40000000LL) return false;
    // the cached buffers are keyed on what they HOLD (floats), not on (boards, H * W): 4x16 and 8x8, or another tiling of the same
    // board, have other padded sizes
    if (!mdl->nact[0].defined() || mdl->nact_floats < (int64_t)cap * board_floats || mdl->nact_pmaps < (int64_t)cap * 2 * H * W) {
        for (auto& t : mdl->nact) t = at::empty({ (int64_t)cap * board_floats }, opts);      // layout T (cnn_wino.hip)
        mdl->npmaps = at::empty({ (int64_t)cap * 2 * H * W }, opts);
        mdl->nact_floats = (int64_t)cap * board_floats;
        mdl->nact_pmaps = (int64_t)cap * 2 * H * W;
        (void)hipDeviceSynchronize();            // the buffers may be used on another stream than the one that allocated them
    }
    float *x = mdl->nact[0].data_ptr<float>(), *ya = mdl->nact[1].data_ptr<float>(), *za = mdl->nact[2].data_ptr<float>();
    if (sprl_stem_conv3x3_t(planes, n.stem_w.data_ptr<float>(), n.stem_scale.data_ptr<float>(), n.stem_shift.data_ptr<float>(), x, cap, P, H, W,
                            tile, batch_dev, stream) != 0)
        return false;
    if (prof) prof->open((hipStream_t)stream);       // one event pair around all the trunk convolutions of this forward
    auto conv = [&](const float* src, const at::Tensor& u, const at::Tensor& sc, const at::Tensor& sh, const float* res, float* dst) {
        const int rc = sprl_wino_conv64_t_occ(src, u.data_ptr<float>(), sc.data_ptr<float>(), sh.data_ptr<float>(), res, dst, cap, H, W, 1, tile,
                                              mdl->sw.f3_occ, batch_dev, stream);
        if (prof) {
            prof->note(cap);
            if (prof->per_launch) {
                prof->close((hipStream_t)stream);
                prof->open((hipStream_t)stream);
            }
        }
        return rc == 0;
    };
    for (const auto& b : n.blocks) {
        if (!conv(x, b.u1t, b.s1, b.t1, nullptr, ya) || !conv(ya, b.u2t, b.s2, b.t2, x, za)) return false;
        std::swap(x, za);
    }
    if (prof) prof->close((hipStream_t)stream);
    return sprl_tail_t(x, n.heads_w.data_ptr<float>(), n.heads_b.data_ptr<float>(), n.pfc_w.data_ptr<float>(), n.pfc_b.data_ptr<float>(),
                       n.vfc1_w.data_ptr<float>(), n.vfc1_b.data_ptr<float>(), n.vfc2_w.data_ptr<float>(), n.vfc2_b.data_ptr<float>(),
                       mdl->npmaps.data_ptr<float>(), logits_out, value_out, cap, H, W, n.pc, n.vc, A, HID, tile, batch_dev, stream) == 0;
}

bool forward_native(Model* mdl, const at::Tensor& in, at::Tensor& logits, at::Tensor& value, ConvProfile* prof,
                    float* logits_out, float* value_out, bool* wrote_outputs, void* stream = nullptr) {
    const NativeNet& n = mdl->native;
    at::Tensor p, v;
    const int H0 = (int)in.size(2), W0 = (int)in.size(3), P0 = (int)in.size(1);
    const bool wino = n.wino && (P0 == 3 || P0 == 17) && ((H0 == 8 && W0 == 8) || (H0 == 6 && W0 == 7) || (H0 == 7 && W0 == 7));
    if (wino) {
        if (!forward_wino(mdl, in, p, v, prof, logits_out, value_out, wrote_outputs, nullptr, stream)) return false;
        if (*wrote_outputs) return true;
    } else if (logits_out && value_out && in.is_contiguous() && nchw_covered(mdl, P0, H0, W0, (int)n.pfc_w.size(1)) &&
               forward_nchw(mdl, in.data_ptr<float>(), (int)in.size(0), P0, H0, W0, prof, logits_out, value_out, nullptr, stream, in.options())) {
        *wrote_outputs = true;                   // the whole forward in our kernels, results already in the caller's buffers
        return true;
    } else {
        at::Tensor x;
        if ((P0 == 3 || P0 == 17) && n.stem_w.size(0) == 64 && in.is_contiguous() && !mdl->sw.no_nchw_stem) {
            // hand-written stem for any board size (Go 9x9 / 19x19): conv + folded BN + ReLU in one kernel
            x = nchw_act(in.size(0), H0, W0, in.options());
            if ((mdl->sw.stem_valu ? sprl_stem_conv3x3_nchw_valu : sprl_stem_conv3x3_nchw)(
                    in.data_ptr<float>(), n.stem_w.data_ptr<float>(), n.stem_scale.data_ptr<float>(), n.stem_shift.data_ptr<float>(),
                    x.data_ptr<float>(), in.size(0), P0, H0, W0, stream) != 0)
                return false;
        } else {
            x = at::conv2d(in, n.stem_w, {}, 1, 1);
            if (!x.is_contiguous() || !epilogue(x, n.stem_scale, n.stem_shift, nullptr)) return false;
        }
        // boards wider than 8: the trunk still runs on the hand-written Winograd/MFMA kernel (any-board variant, NCHW);
        // the stem (P planes) and the heads keep the library convolution / the NCHW heads kernel
        const bool trunk_wino = n.wino && H0 <= 64 && W0 <= 64 && !mdl->sw.no_winograd_nchw;
        at::Tensor ya, za;
        if (trunk_wino) {
            if (x.storage_offset() < 4 || x.storage().nbytes() < (size_t)(x.storage_offset() + x.numel()) * 4 + 32) {   // (library stem: no slack yet)
                at::Tensor xs = nchw_act(x.size(0), H0, W0, x.options());
                xs.copy_(x);
                x = xs;
            }
            ya = nchw_act(x.size(0), H0, W0, x.options());
            za = nchw_act(x.size(0), H0, W0, x.options());
        }
        // profile mode: HIP event pairs around every trunk-convolution launch, as in forward_wino
        auto timed_conv = [&](const at::Tensor& src, const at::Tensor& u, const at::Tensor& sc, const at::Tensor& sh, const float* res,
                              at::Tensor& dst) {
            const int B0 = (int)src.size(0);
            if (prof) prof->open((hipStream_t)stream);      // (other kernels sit between these launches: a pair per launch)
            const int rc = sprl_wino_conv64_nchw(src.data_ptr<float>(), u.data_ptr<float>(), sc.data_ptr<float>(), sh.data_ptr<float>(), res,
                                                 dst.data_ptr<float>(), B0, H0, W0, 1, stream);
            if (prof) {
                prof->note(B0);
                prof->close((hipStream_t)stream);
            }
            return rc == 0;
        };
        for (const auto& b : n.blocks) {
            if (trunk_wino) {
                if (!timed_conv(x, b.u1, b.s1, b.t1, nullptr, ya) || !timed_conv(ya, b.u2, b.s2, b.t2, x.data_ptr<float>(), za)) return false;
                std::swap(x, za);
                continue;
            }
            at::Tensor y = at::conv2d(x, b.w1, {}, 1, 1);
            if (!y.is_contiguous() || !epilogue(y, b.s1, b.t1, nullptr)) return false;
            at::Tensor z = at::conv2d(y, b.w2, {}, 1, 1);
            if (!z.is_contiguous() || !epilogue(z, b.s2, b.t2, &x)) return false;
            x = z;
        }
        const int64_t B = x.size(0);
        const int C = (int)x.size(1), HW = (int)(x.size(2) * x.size(3));
        p = at::empty({ B, (int64_t)n.pc * HW }, x.options());
        v = at::empty({ B, (int64_t)n.vc * HW }, x.options());
        if (sprl_heads_conv1x1_relu(x.data_ptr<float>(), n.heads_w.data_ptr<float>(), n.heads_b.data_ptr<float>(),
                                    p.data_ptr<float>(), v.data_ptr<float>(), B, C, HW, n.pc, n.vc, 0, stream) != 0)
            return false;
    }
    logits = at::addmm(n.pfc_b, p, n.pfc_w);
    v = at::relu(at::addmm(n.vfc1_b, v, n.vfc1_w));
    value = at::tanh(at::addmm(n.vfc2_b, v, n.vfc2_w));
    return true;
}

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

static void* load_common(const char* path, const void* bytes, long long nbytes, int device, char* err, int errlen);

void* sprl_torch_load(const char* path, int device, char* err, int errlen) { return load_common(path, nullptr, 0, device, err, errlen); }

// the same from a TorchScript archive held in memory (torch.jit.save into a buffer): the trainer hands a new model to the
// engine without a file
void* sprl_torch_load_buffer(const void* bytes, long long nbytes, int device, char* err, int errlen) {
    return load_common(nullptr, bytes, nbytes, device, err, errlen);
}

static void* load_common(const char* path, const void* bytes, long long nbytes, int device, char* err, int errlen) {
    // MIOpen's solver search benchmarks its im2col+GEMM family one image at a time (2.2 M tiny launches for our 16
    // batch shapes, ~20 s) and never picks it for these 3x3 convolutions; leave it out of the search unless the user
    // has set the variable.
    setenv("MIOPEN_DEBUG_CONV_GEMM", "0", 0);
    try {
        // device < 0: host tensors — used only by the CPU unit test of the graph rewrite, never by the engine
        if (device >= 0 && !torch::cuda::is_available()) {
            put_err(err, errlen, "LibTorch reports no ROCm device");
            return nullptr;
        }
        auto* m = new Model();
        m->device = device;
        const torch::Device where = device >= 0 ? torch::Device(torch::kCUDA, (c10::DeviceIndex)device) : torch::Device(torch::kCPU);
        if (path) {
            m->module = torch::jit::load(path, where);
        } else {
            std::istringstream in(std::string((const char*)bytes, (size_t)nbytes), std::ios::binary);
            m->module = torch::jit::load(in, where);
        }
        m->module.eval();                       // GridNetwork.hpp:67
        m->sw = PathSwitches::from_env();       // the only place the environment is read
        if (device >= 0 && !m->sw.no_native) build_native(m);
        if (!m->sw.no_rewrite) {
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

static int forward_common(void* handle, const float* planes, int batch, int nplanes, int rows, int cols, float* logits,
                          int actions, float* value, void* stream, char* err, int errlen);

int sprl_torch_forward(void* handle, const float* planes, int batch, int nplanes, int rows, int cols,
                       float* logits, int actions, float* value, char* err, int errlen) {
    return forward_common(handle, planes, batch, nplanes, rows, cols, logits, actions, value, nullptr, err, errlen);
}

// The same forward with every kernel (ours and the library's) on the caller's HIP stream: an engine on a private stream
// (sprl_config.own_stream) whose evaluator needs the batch size on the host - boards wider than 8, other architectures.
int sprl_torch_forward_on(void* handle, const float* planes, int batch, int nplanes, int rows, int cols,
                          float* logits, int actions, float* value, void* stream, char* err, int errlen) {
    return forward_common(handle, planes, batch, nplanes, rows, cols, logits, actions, value, stream, err, errlen);
}

static int forward_common(void* handle, const float* planes, int batch, int nplanes, int rows, int cols, float* logits,
                          int actions, float* value, void* stream, char* err, int errlen) {
    try {
        auto* m = static_cast<Model*>(handle);
        c10::InferenceMode guard;
        std::unique_ptr<c10::hip::HIPStreamGuardMasqueradingAsCUDA> on_stream;       // LibTorch's own kernels follow
        if (stream && m->device >= 0)
            on_stream.reset(new c10::hip::HIPStreamGuardMasqueradingAsCUDA(
                c10::hip::getStreamFromExternalMasqueradingAsCUDA((hipStream_t)stream, (c10::DeviceIndex)m->device)));
        auto opts = m->device >= 0
                        ? torch::TensorOptions().dtype(torch::kFloat32).device(torch::kCUDA, (c10::DeviceIndex)m->device)
                        : torch::TensorOptions().dtype(torch::kFloat32).device(torch::kCPU);
        auto in = torch::from_blob(const_cast<float*>(planes), { batch, nplanes, rows, cols }, opts);
        at::Tensor lo, va;
        bool wrote = false;
        const bool fuse_tail = m->device >= 0 && m->native.ok && m->native.pfc_w.size(1) == actions && !m->sw.no_fused_tail;
        if (m->native.ok && forward_native(m, in, lo, va, &m->prof, fuse_tail ? logits : nullptr,
                                           fuse_tail ? value : nullptr, &wrote, stream) && wrote)
            return 0;
        if (!lo.defined()) {
            auto out = m->module.forward({ in }).toTuple();         // GridNetwork.hpp:99-102 (generic TorchScript path)
            lo = out->elements()[0].toTensor();
            va = out->elements()[1].toTensor();
        }
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

// host -> host: conv weight [64][64][3][3] to the Winograd-domain layout sprl_wino_conv64 takes (36*64*64 floats)
void sprl_wino_transform_weights(const float* w, float* u) { wino_transform(w, u); }
// the same for the F(3x3,3x3) tiling of the any-board kernel (28*64*64 floats)
void sprl_wino_transform_weights_f3(const float* w, float* u) { wino_transform_f3(w, u); }
// for the any-board kernel on layout T (a K step = one channel quad): tile 4 -> 36*64*64 floats, tile 3 -> 28*64*64
void sprl_wino_transform_weights_t(const float* w, float* u, int tile) {
    if (tile == 3) wino_transform_f3(w, u, true);
    else wino_transform(w, u, true);
}

// profile mode: time every trunk-convolution launch with HIP events; totals since load (ms, launches, boards)
// on = 1: one event pair around all trunk convolutions of a forward (what a timed run uses: the interval includes the few
// microseconds between the launches); on = 2: one pair per convolution launch (kernel durations, for a sample outside a timed region)
void sprl_torch_profile_enable(void* handle, int on) {
    ConvProfile& p = static_cast<Model*>(handle)->prof;
    p.on = on != 0;
    p.per_launch = on == 2;
}
// time with >= 1 trunk-convolution launch executing, over all models of the process (busy_log.h); read after profile_read
double sprl_torch_profile_busy(double* sum_ms) { return g_conv_busy.union_ms(sum_ms); }
void sprl_torch_profile_busy_reset() { g_conv_busy.reset(); }
void sprl_torch_profile_read(void* handle, double* conv_ms, int64_t* launches, int64_t* boards) {
    ConvProfile& p = static_cast<Model*>(handle)->prof;
    p.resolve();
    if (conv_ms) *conv_ms = p.ms;
    if (launches) *launches = p.launches;
    if (boards) *boards = p.boards;
}

// per-launch mode (profile 2): time and launches by kind of trunk-convolution launch - 0 plain, 1 with residual, 2 with the stem in
// its prologue, 3 with the heads / FC layers behind it (boards up to 8x8; the any-board path counts everything as 0 / 1)
void sprl_torch_profile_read_kinds(void* handle, double* ms4, int64_t* launches4) {
    ConvProfile& p = static_cast<Model*>(handle)->prof;
    p.resolve();
    for (int k = 0; k < 4; ++k) {
        if (ms4) ms4[k] = p.kind_ms[k];
        if (launches4) launches4[k] = p.kind_launches[k];
    }
}

// The whole forward with the batch size read ON THE DEVICE (`batch_dev`, <= max_batch): nothing here depends on the host
// knowing how many leaves the round queued, so the engine can enqueue rounds without a synchronisation in between.
// `stream`: the HIP stream every kernel of the forward goes to (null = the null stream).
// Only the hand-written paths (kind 2: layout-W kernels up to 8x8, NCHW kernels for wider boards) can do this; returns -2 when
// the model or the board is not covered.
int sprl_torch_forward_dev(void* handle, const float* planes, const unsigned* batch_dev, int max_batch, int nplanes, int rows,
                           int cols, float* logits, int actions, float* value, void* stream, char* err, int errlen) {
    try {
        auto* m = static_cast<Model*>(handle);
        const NativeNet& n = m->native;
        if (!(n.ok && n.wino) || m->device < 0 || n.pfc_w.size(1) != actions || m->sw.no_fused_tail) return -2;
        c10::InferenceMode guard;
        auto opts = torch::TensorOptions().dtype(torch::kFloat32).device(torch::kCUDA, (c10::DeviceIndex)m->device);
        if (!((nplanes == 3 || nplanes == 17) && ((rows == 8 && cols == 8) || (rows == 6 && cols == 7) || (rows == 7 && cols == 7)))) {
            // boards wider than 8: the NCHW kernels, same contract
            if (!nchw_covered(m, nplanes, rows, cols, actions)) return -2;
            return forward_nchw(m, planes, max_batch, nplanes, rows, cols, &m->prof, logits, value, batch_dev, stream, opts) ? 0 : -2;
        }
        auto in = torch::from_blob(const_cast<float*>(planes), { max_batch, nplanes, rows, cols }, opts);
        at::Tensor p, v;
        bool wrote = false;
        if (!forward_wino(m, in, p, v, &m->prof, logits, value, &wrote, batch_dev, stream) || !wrote) return -2;
        return 0;
    } catch (const std::exception& e) {
        put_err(err, errlen, e.what());
        return -1;
    }
}

void sprl_torch_free(void* handle) { delete static_cast<Model*>(handle); }

// What the model's forward resolves to: "kind=<0|1|2> stem=<1|0> tail=<2|1|0> lab=[<switches that were set when it was loaded>]".
// kind as sprl_torch_is_native; boards up to 8x8, after the first forward: stem 1 = the stem runs inside the first trunk
// convolution (3 input planes), 0 = its own launch; tail 2 = last convolution + heads + FC layers in one launch, 1 = heads fused
// into the last convolution + FC kernel, 0 = separate tail kernel.  An empty lab list = the default path.  The default 8x8 forward
// of the 2-block network is FOUR launches: conv(stem) - conv - conv - conv(heads, FC).
int sprl_torch_path_info(void* handle, char* buf, int len) {
    if (!handle || !buf || len < 1) return -1;
    const Model* m = static_cast<Model*>(handle);
    snprintf(buf, (size_t)len, "kind=%d stem=%d tail=%d lab=[%s]", m->native.ok ? (m->native.wino ? 2 : 1) : 0, m->last_stem, m->last_tail,
             m->sw.active.c_str());
    return 0;
}

// 2: recognised architecture with a 64-channel trunk — stem, Winograd/MFMA trunk and heads all in the hand-written kernels;
// 1: recognised, convolutions in MIOpen + the hand-written fused epilogue; 0: generic (rewritten) TorchScript graph
int sprl_torch_is_native(void* handle) {
    const NativeNet& n = static_cast<Model*>(handle)->native;
    return n.ok ? (n.wino ? 2 : 1) : 0;
}

}  // extern "C"

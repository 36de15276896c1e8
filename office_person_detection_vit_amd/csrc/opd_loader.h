// opd_loader.h — native safetensors reader + DETR state-dict schema (host side, no HIP).
#pragma once
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

namespace opd {

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;  // always widened to fp32 on load
    int64_t numel() const {
        int64_t n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};

// name (HF 5.x spelling) -> tensor
using StateDict = std::map<std::string, HostTensor>;

// Reads a .safetensors file (F32 / F16 / BF16 tensors).  Keys are normalised to the HF 5.x DETR names
// (4.x `conv_encoder` / `out_proj` / `fc1` spellings and the timm ResNet layout are renamed).
// Returns 0 or a negative OPD_E* code with `err` filled.
int load_safetensors(const std::string& path, StateDict* out, std::string* err);

// HF 4.x / timm key -> 5.x key (identity for 5.x keys).  Exposed for tests.
std::string normalise_key(const std::string& key);

// IEEE fp32 -> fp16 bits, round-to-nearest-even, overflow -> inf, subnormals handled.
uint16_t f32_to_f16(float f);
float f16_to_f32(uint16_t h);
// IEEE fp32 -> bfloat16 bits, round-to-nearest-even (NaN stays NaN), and back.
uint16_t f32_to_bf16(float f);
float bf16_to_f32(uint16_t h);

struct Arch {
    int depths[4] = {0, 0, 0, 0};
    int hidden[4] = {256, 512, 1024, 2048};
    int embed = 64;
    int d_model = 256, heads = 8, ffn = 2048, enc_layers = 0, dec_layers = 0, queries = 0, ncls = 0;
};

// Infers the architecture from the tensors present and validates every shape of the detect path.
int infer_arch(const StateDict& sd, Arch* arch, std::string* err);

}  // namespace opd

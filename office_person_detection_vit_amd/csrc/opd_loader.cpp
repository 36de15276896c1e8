// opd_loader.cpp — safetensors parsing, key normalisation, architecture inference (host only).
//
// Replaces the checkpoint half of `ViTDetector.load_model` (deleted vit_detector.py 81-99 -> HF
// `DetrForObjectDetection.from_pretrained`).  File format: u64 little-endian header length, a JSON object
// {name: {"dtype","shape","data_offsets":[begin,end]}, "__metadata__": {...}}, then the raw tensor bytes.
#include "opd_loader.h"

#include <string.h>

#include <cstdio>
#include <cstdlib>

#include "../../include/opd_detr.h"

namespace opd {

uint16_t f32_to_f16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t expo = (x >> 23) & 0xffu;
    uint32_t mant = x & 0x7fffffu;
    if (expo == 0xff) return (uint16_t)(sign | 0x7c00u | (mant ? 0x200u : 0u));  // inf / nan
    int e = (int)expo - 127 + 15;
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);  // overflow -> inf
    if (e <= 0) {                                     // subnormal or zero
        if (e < -10) return (uint16_t)sign;
        mant |= 0x800000u;
        const int shift = 14 - e;  // 14..24
        uint32_t h = mant >> shift;
        const uint32_t rem = mant & ((1u << shift) - 1u);
        const uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1u))) ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)e << 10) | (mant >> 13);
    const uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;  // may carry into the exponent: still correct
    return (uint16_t)(sign | h);
}

uint16_t f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);   // NaN: keep it quiet
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
float bf16_to_f32(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t expo = (h >> 10) & 0x1fu;
    uint32_t mant = h & 0x3ffu;
    uint32_t x;
    if (expo == 0) {
        if (mant == 0) {
            x = sign;
        } else {
            int e = -1;
            do { mant <<= 1; ++e; } while (!(mant & 0x400u));
            x = sign | ((uint32_t)(127 - 15 - e) << 23) | ((mant & 0x3ffu) << 13);
        }
    } else if (expo == 31) {
        x = sign | 0x7f800000u | (mant << 13);
    } else {
        x = sign | ((expo + 127 - 15) << 23) | (mant << 13);
    }
    float f;
    memcpy(&f, &x, 4);
    return f;
}

// ---- minimal JSON reader for the safetensors header ---------------------------------------------------------------
namespace {

struct Cursor {
    const char* p;
    const char* end;
    bool fail = false;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    bool eat(char c) {
        ws();
        if (p < end && *p == c) { ++p; return true; }
        return false;
    }
    std::string str() {
        ws();
        std::string s;
        if (p >= end || *p != '"') { fail = true; return s; }
        ++p;
        while (p < end && *p != '"') {
            if (*p == '\\' && p + 1 < end) {
                ++p;
                switch (*p) {
                    case 'n': s.push_back('\n'); break;
                    case 't': s.push_back('\t'); break;
                    case 'u': s.push_back('?'); p += 4; break;  // names are ASCII; keep position sane
                    default: s.push_back(*p);
                }
                ++p;
            } else {
                s.push_back(*p++);
            }
        }
        if (p >= end) { fail = true; return s; }
        ++p;
        return s;
    }
    int64_t integer() {
        ws();
        char* e = nullptr;
        const long long v = strtoll(p, &e, 10);
        if (e == p) fail = true;
        p = e;
        return v;
    }
    void skip_value() {  // skips any JSON value (used for __metadata__)
        ws();
        if (p >= end) { fail = true; return; }
        if (*p == '"') { str(); return; }
        if (*p == '{' || *p == '[') {
            const char open = *p, close = (*p == '{') ? '}' : ']';
            int depth = 0;
            bool in_str = false;
            for (; p < end; ++p) {
                if (in_str) {
                    if (*p == '\\') ++p;
                    else if (*p == '"') in_str = false;
                } else if (*p == '"') in_str = true;
                else if (*p == open) ++depth;
                else if (*p == close && --depth == 0) { ++p; return; }
            }
            fail = true;
            return;
        }
        while (p < end && *p != ',' && *p != '}' && *p != ']') ++p;
    }
};

struct Entry {
    std::string dtype;
    std::vector<int64_t> shape;
    int64_t begin = 0, end = 0;
};

bool parse_entry(Cursor& c, Entry* e) {
    if (!c.eat('{')) return false;
    while (true) {
        const std::string k = c.str();
        if (c.fail || !c.eat(':')) return false;
        if (k == "dtype") {
            e->dtype = c.str();
        } else if (k == "shape") {
            if (!c.eat('[')) return false;
            if (!c.eat(']')) {
                do { e->shape.push_back(c.integer()); } while (c.eat(','));
                if (!c.eat(']')) return false;
            }
        } else if (k == "data_offsets") {
            if (!c.eat('[')) return false;
            e->begin = c.integer();
            if (!c.eat(',')) return false;
            e->end = c.integer();
            if (!c.eat(']')) return false;
        } else {
            c.skip_value();
        }
        if (c.fail) return false;
        if (c.eat(',')) continue;
        return c.eat('}');
    }
}

void replace_all(std::string& s, const std::string& a, const std::string& b) {
    size_t pos = 0;
    while ((pos = s.find(a, pos)) != std::string::npos) {
        s.replace(pos, a.size(), b);
        pos += b.size();
    }
}

}  // namespace

std::string normalise_key(const std::string& key_in) {
    std::string k = key_in;
    // HF 4.x -> 5.x (HF:conversion_mapping.py:1036-1041)
    replace_all(k, "model.backbone.conv_encoder.", "model.backbone.");
    replace_all(k, ".out_proj.", ".o_proj.");
    if ((k.rfind("model.encoder.layers.", 0) == 0 || k.rfind("model.decoder.layers.", 0) == 0) &&
        k.find(".mlp.") == std::string::npos) {
        replace_all(k, ".fc1.", ".mlp.fc1.");
        replace_all(k, ".fc2.", ".mlp.fc2.");
    }
    // timm ResNet layout (4.x `use_timm_backbone=True` checkpoints) -> HF ResNetBackbone layout
    const std::string bb = "model.backbone.model.";
    if (k.rfind(bb, 0) == 0 && k.find("embedder") == std::string::npos && k.find("encoder.stages") == std::string::npos) {
        std::string r = k.substr(bb.size());
        std::string out;
        if (r.rfind("conv1.", 0) == 0) out = "embedder.embedder.convolution." + r.substr(6);
        else if (r.rfind("bn1.", 0) == 0) out = "embedder.embedder.normalization." + r.substr(4);
        else if (r.rfind("layer", 0) == 0 && r.size() > 8) {
            // layer{S}.{L}.conv{J}.weight | bn{J}.x | downsample.0.weight | downsample.1.x
            const int stage = r[5] - '1';
            const size_t d1 = r.find('.', 0), d2 = r.find('.', d1 + 1);
            const std::string layer = r.substr(d1 + 1, d2 - d1 - 1);
            std::string rest = r.substr(d2 + 1);
            std::string pre = "encoder.stages." + std::to_string(stage) + ".layers." + layer + ".";
            if (rest.rfind("conv", 0) == 0) out = pre + "layer." + std::to_string(rest[4] - '1') + ".convolution." + rest.substr(6);
            else if (rest.rfind("bn", 0) == 0) out = pre + "layer." + std::to_string(rest[2] - '1') + ".normalization." + rest.substr(4);
            else if (rest.rfind("downsample.0.", 0) == 0) out = pre + "shortcut.convolution." + rest.substr(13);
            else if (rest.rfind("downsample.1.", 0) == 0) out = pre + "shortcut.normalization." + rest.substr(13);
        }
        if (!out.empty()) k = bb + out;
    }
    return k;
}

int load_safetensors(const std::string& path, StateDict* out, std::string* err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { *err = "cannot open weight file '" + path + "'"; return OPD_EIO; }
    uint64_t hl = 0;
    if (fread(&hl, 8, 1, f) != 1 || hl == 0 || hl > (1ull << 30)) {
        fclose(f);
        *err = "'" + path + "' is not a safetensors file (bad header length)";
        return OPD_EIO;
    }
    std::string header(hl, '\0');
    if (fread(&header[0], 1, hl, f) != hl) { fclose(f); *err = "truncated safetensors header in '" + path + "'"; return OPD_EIO; }
    fseek(f, 0, SEEK_END);
    const int64_t fsize = ftell(f);
    const int64_t base = 8 + (int64_t)hl;
    Cursor c{header.data(), header.data() + header.size()};
    if (!c.eat('{')) { fclose(f); *err = "malformed safetensors header (no object)"; return OPD_EIO; }
    std::vector<unsigned char> raw;
    while (true) {
        const std::string name = c.str();
        if (c.fail || !c.eat(':')) { fclose(f); *err = "malformed safetensors header near '" + name + "'"; return OPD_EIO; }
        if (name == "__metadata__") {
            c.skip_value();
        } else {
            Entry e;
            if (!parse_entry(c, &e)) { fclose(f); *err = "malformed safetensors entry '" + name + "'"; return OPD_EIO; }
            int64_t n = 1;
            bool bad_shape = e.shape.size() > 8;
            for (auto s : e.shape) {   // negative sizes and products beyond the file size are refused before anything is allocated
                if (s < 0 || (s > 0 && n > fsize / s)) { bad_shape = true; break; }
                n *= s;
            }
            if (bad_shape) { fclose(f); *err = "tensor '" + name + "' has an invalid shape"; return OPD_EIO; }
            const int esz = (e.dtype == "F32") ? 4 : (e.dtype == "F16" || e.dtype == "BF16") ? 2 : 0;
            if (esz == 0) {  // integer tensors (e.g. num_batches_tracked) are not part of the path: skip
                if (c.eat(',')) continue;
                break;
            }
            if (e.end - e.begin != n * esz || e.begin < 0 || base + e.end > fsize) {
                fclose(f);
                *err = "tensor '" + name + "' has inconsistent dtype/shape/offsets";
                return OPD_EIO;
            }
            raw.resize((size_t)(n * esz));
            fseek(f, base + e.begin, SEEK_SET);
            if (n && fread(raw.data(), 1, raw.size(), f) != raw.size()) { fclose(f); *err = "short read for '" + name + "'"; return OPD_EIO; }
            HostTensor t;
            t.shape = e.shape;
            t.data.resize((size_t)n);
            if (e.dtype == "F32") {
                memcpy(t.data.data(), raw.data(), raw.size());
            } else if (e.dtype == "F16") {
                const uint16_t* h = reinterpret_cast<const uint16_t*>(raw.data());
                for (int64_t i = 0; i < n; ++i) t.data[i] = f16_to_f32(h[i]);
            } else {  // BF16
                const uint16_t* h = reinterpret_cast<const uint16_t*>(raw.data());
                for (int64_t i = 0; i < n; ++i) {
                    const uint32_t x = (uint32_t)h[i] << 16;
                    memcpy(&t.data[i], &x, 4);
                }
            }
            (*out)[normalise_key(name)] = std::move(t);
        }
        if (c.fail) { fclose(f); *err = "malformed safetensors header"; return OPD_EIO; }
        if (c.eat(',')) continue;
        break;
    }
    fclose(f);
    return OPD_OK;
}

namespace {
bool has(const StateDict& sd, const std::string& k) { return sd.find(k) != sd.end(); }
int want(const StateDict& sd, const std::string& k, std::initializer_list<int64_t> shape, std::string* err) {
    auto it = sd.find(k);
    if (it == sd.end()) { *err = "weight file lacks tensor '" + k + "'"; return OPD_ESCHEMA; }
    std::vector<int64_t> s(shape);
    if (it->second.shape != s) {
        std::string got = "[";
        for (auto v : it->second.shape) got += std::to_string(v) + ",";
        *err = "tensor '" + k + "' has shape " + got + "] which does not match the DETR detect path";
        return OPD_ESCHEMA;
    }
    return OPD_OK;
}
}  // namespace

int infer_arch(const StateDict& sd, Arch* a, std::string* err) {
    const std::string bb = "model.backbone.model.";
    for (int s = 0; s < 4; ++s) {
        int d = 0;
        while (has(sd, bb + "encoder.stages." + std::to_string(s) + ".layers." + std::to_string(d) + ".layer.0.convolution.weight")) ++d;
        if (d == 0) { *err = "weight file has no ResNet stage " + std::to_string(s) + " (not a DETR-ResNet checkpoint?)"; return OPD_ESCHEMA; }
        a->depths[s] = d;
    }
    while (has(sd, "model.encoder.layers." + std::to_string(a->enc_layers) + ".self_attn.q_proj.weight")) ++a->enc_layers;
    while (has(sd, "model.decoder.layers." + std::to_string(a->dec_layers) + ".self_attn.q_proj.weight")) ++a->dec_layers;
    if (a->enc_layers == 0 || a->dec_layers == 0) { *err = "weight file has no transformer encoder/decoder layers"; return OPD_ESCHEMA; }
    auto q = sd.find("model.query_position_embeddings.weight");
    auto c = sd.find("class_labels_classifier.weight");
    if (q == sd.end() || c == sd.end() || q->second.shape.size() != 2 || c->second.shape.size() != 2) {
        *err = "weight file lacks query_position_embeddings / class_labels_classifier";
        return OPD_ESCHEMA;
    }
    a->queries = (int)q->second.shape[0];
    a->ncls = (int)c->second.shape[0];
    if (q->second.shape[1] != 256 || c->second.shape[1] != 256 || a->queries > 128 || a->ncls > 256) {
        *err = "unsupported d_model / num_queries / num_labels (kernels are built for d_model 256, <=128 queries)";
        return OPD_ESCHEMA;
    }
    int rc;
    const int64_t d = a->d_model, f = a->ffn;
    // backbone
    if ((rc = want(sd, bb + "embedder.embedder.convolution.weight", {64, 3, 7, 7}, err))) return rc;
    for (const char* nm : {"weight", "bias", "running_mean", "running_var"})
        if ((rc = want(sd, bb + "embedder.embedder.normalization." + nm, {64}, err))) return rc;
    int64_t cin = 64;
    for (int s = 0; s < 4; ++s) {
        const int64_t cout = a->hidden[s], mid = cout / 4;
        for (int l = 0; l < a->depths[s]; ++l) {
            const std::string p = bb + "encoder.stages." + std::to_string(s) + ".layers." + std::to_string(l) + ".";
            if (l == 0) {
                if ((rc = want(sd, p + "shortcut.convolution.weight", {cout, cin, 1, 1}, err))) return rc;
                for (const char* nm : {"weight", "bias", "running_mean", "running_var"})
                    if ((rc = want(sd, p + "shortcut.normalization." + nm, {cout}, err))) return rc;
            }
            if ((rc = want(sd, p + "layer.0.convolution.weight", {mid, cin, 1, 1}, err))) return rc;
            if ((rc = want(sd, p + "layer.1.convolution.weight", {mid, mid, 3, 3}, err))) return rc;
            if ((rc = want(sd, p + "layer.2.convolution.weight", {cout, mid, 1, 1}, err))) return rc;
            for (int j = 0; j < 3; ++j)
                for (const char* nm : {"weight", "bias", "running_mean", "running_var"})
                    if ((rc = want(sd, p + "layer." + std::to_string(j) + ".normalization." + nm, {j == 2 ? cout : mid}, err))) return rc;
            cin = cout;
        }
    }
    if ((rc = want(sd, "model.input_projection.weight", {d, 2048, 1, 1}, err))) return rc;
    if ((rc = want(sd, "model.input_projection.bias", {d}, err))) return rc;
    auto attn = [&](const std::string& p) -> int {
        for (const char* nm : {"q_proj", "k_proj", "v_proj", "o_proj"}) {
            if ((rc = want(sd, p + "." + nm + ".weight", {d, d}, err))) return rc;
            if ((rc = want(sd, p + "." + nm + ".bias", {d}, err))) return rc;
        }
        return 0;
    };
    auto ln = [&](const std::string& p) -> int {
        if ((rc = want(sd, p + ".weight", {d}, err))) return rc;
        return want(sd, p + ".bias", {d}, err);
    };
    auto mlp = [&](const std::string& p) -> int {
        if ((rc = want(sd, p + ".mlp.fc1.weight", {f, d}, err))) return rc;
        if ((rc = want(sd, p + ".mlp.fc1.bias", {f}, err))) return rc;
        if ((rc = want(sd, p + ".mlp.fc2.weight", {d, f}, err))) return rc;
        return want(sd, p + ".mlp.fc2.bias", {d}, err);
    };
    for (int i = 0; i < a->enc_layers; ++i) {
        const std::string p = "model.encoder.layers." + std::to_string(i);
        if ((rc = attn(p + ".self_attn")) || (rc = ln(p + ".self_attn_layer_norm")) || (rc = mlp(p)) || (rc = ln(p + ".final_layer_norm"))) return rc;
    }
    for (int i = 0; i < a->dec_layers; ++i) {
        const std::string p = "model.decoder.layers." + std::to_string(i);
        if ((rc = attn(p + ".self_attn")) || (rc = ln(p + ".self_attn_layer_norm")) || (rc = attn(p + ".encoder_attn")) ||
            (rc = ln(p + ".encoder_attn_layer_norm")) || (rc = mlp(p)) || (rc = ln(p + ".final_layer_norm")))
            return rc;
    }
    if ((rc = ln("model.decoder.layernorm"))) return rc;
    if ((rc = want(sd, "model.query_position_embeddings.weight", {(int64_t)a->queries, d}, err))) return rc;
    if ((rc = want(sd, "class_labels_classifier.weight", {(int64_t)a->ncls, d}, err))) return rc;
    if ((rc = want(sd, "class_labels_classifier.bias", {(int64_t)a->ncls}, err))) return rc;
    if ((rc = want(sd, "bbox_predictor.layers.0.weight", {d, d}, err))) return rc;
    if ((rc = want(sd, "bbox_predictor.layers.1.weight", {d, d}, err))) return rc;
    if ((rc = want(sd, "bbox_predictor.layers.2.weight", {4, d}, err))) return rc;
    for (int j = 0; j < 3; ++j)
        if ((rc = want(sd, "bbox_predictor.layers." + std::to_string(j) + ".bias", {j == 2 ? 4 : d}, err))) return rc;
    return OPD_OK;
}

}  // namespace opd

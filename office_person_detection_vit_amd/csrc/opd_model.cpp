// opd_model.cpp — device model, per-resolution plan, forward graph and the C-ABI of include/opd_detr.h.
//
// Host-side orchestration of the DETR detect path:
//   preprocess -> stem 7x7 -> maxpool -> 16/33 bottlenecks -> input_projection -> 6 x encoder layer ->
//   (memory K/V of all decoder layers in one GEMM) -> 6 x decoder layer -> final LN -> heads -> post-process.
// Data layout in HBM: activations NHWC fp16 ([B*H*W][C] matrices), FrozenBN folded into fp16 [Cout][KH][KW][Cin]
// kernels + fp32 bias, transformer residual stream fp32 [tokens][256] with an fp16 shadow feeding the MFMA GEMMs.
// The sine position embedding is constant per (h, w), so pos.Wq / pos.Wk (+ biases) are folded into row-periodic
// fp32 bias matrices at plan-build time and q/k/v become ONE GEMM over x per layer.
#include <new>
#include <stdexcept>

#include "opd_model.h"

using namespace opd;


namespace opd {

std::atomic<int> g_alloc_poison{-1};   // opd_model.h: diagnostic allocation mode

static int upload_f32(opd_detr* m, float** dst, const std::vector<float>& v) {
    RCCHK(dalloc(m, dst, v.size(), true));
    HIPCHK(hipMemcpy(*dst, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    return OPD_OK;
}
static int upload_f16(opd_detr* m, f16_t** dst, const std::vector<float>& v) {
    std::vector<f16_t> h(v.size());   // the 16-bit operand type of this handle: fp16, or bf16 under OPD_FLAG_BF16
    if (m->dtype == OPD_DT_BF16) for (size_t i = 0; i < v.size(); ++i) h[i] = f32_to_bf16(v[i]);
    else for (size_t i = 0; i < v.size(); ++i) h[i] = f32_to_f16(v[i]);
    RCCHK(dalloc(m, dst, h.size(), true));
    HIPCHK(hipMemcpy(*dst, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    return OPD_OK;
}

// a linear layer's weights [N][K] as the fused decoder's split pair in MFMA-fragment order (opd_split_f16_frag)
static int upload_frag(opd_detr* m, f16_t** dst, const std::vector<float>& v, int N, int K) {
    if ((size_t)N * K != v.size() || N % 16 || K % 32) return fail(OPD_ESCHEMA, "decoder weight matrix does not tile into 16 x 32 fragments");
    std::vector<f16_t> f(v.size() * 2);
    opd_split_f16_frag(v.data(), N, K, f.data());
    RCCHK(dalloc(m, dst, f.size(), true));
    HIPCHK(hipMemcpy(*dst, f.data(), f.size() * 2, hipMemcpyHostToDevice));
    return OPD_OK;
}

static const HostTensor& T(const StateDict& sd, const std::string& k) { return sd.at(k); }

// the encoder FFN's weights (+ the rows of its tail projection, pass order) as enc_ffn_kernel's per-wave fragment streams, in the handle's 16-bit
// operand type
static int upload_encffn(opd_detr* m, unsigned char** dst, const std::vector<float>& w1, const std::vector<float>& b1, const std::vector<float>& w2, int F,
                         const std::vector<float>& wt, const std::vector<float>& bt, int tail, const std::vector<float>& wo) {
    std::vector<uint16_t> h1(w1.size()), h2(w2.size()), ht(wt.size()), ho(wo.size());
    const bool bf = m->dtype == OPD_DT_BF16;
    auto cv = [&](const std::vector<float>& v, std::vector<uint16_t>& h) { for (size_t i = 0; i < v.size(); ++i) h[i] = bf ? f32_to_bf16(v[i]) : f32_to_f16(v[i]); };
    cv(w1, h1); cv(w2, h2); cv(wt, ht); cv(wo, ho);
    if (!ho.empty() && ho.size() != (size_t)256 * 256) return fail(OPD_ESCHEMA, "encoder front projection: unexpected weight shape");
    if (ht.size() != (size_t)tail * 256 * 256 || bt.size() != (size_t)tail * 256) return fail(OPD_ESCHEMA, "encoder tail projection: unexpected weight shape");
    std::vector<unsigned char> pk(opd_encffn_pack_bytes(F, tail, ho.empty() ? 0 : 1));
    opd_encffn_pack(h1.data(), b1.data(), h2.data(), F, ht.data(), bt.data(), tail, ho.empty() ? nullptr : ho.data(), pk.data());
    RCCHK(dalloc(m, dst, pk.size(), true));
    HIPCHK(hipMemcpy(*dst, pk.data(), pk.size(), hipMemcpyHostToDevice));
    return OPD_OK;
}

// conv + FrozenBN -> folded fp16 [Cout][KH][KW][Cin] + fp32 bias (HF:models/detr/modeling_detr.py:207-215)
static int make_conv(opd_detr* m, const StateDict& sd, const std::string& prefix, int stride, Conv* c) {
    const HostTensor& w = T(sd, prefix + ".convolution.weight");
    const int Cout = (int)w.shape[0], Cin = (int)w.shape[1], KH = (int)w.shape[2], KW = (int)w.shape[3];
    const std::string n = prefix + ".normalization";
    const auto& g = T(sd, n + ".weight").data;
    const auto& bt = T(sd, n + ".bias").data;
    const auto& mu = T(sd, n + ".running_mean").data;
    const auto& var = T(sd, n + ".running_var").data;
    std::vector<float> scale(Cout), bias(Cout);
    for (int o = 0; o < Cout; ++o) {
        scale[o] = g[o] * (1.0f / sqrtf(var[o] + 1e-5f));
        bias[o] = bt[o] - mu[o] * scale[o];
    }
    c->Cin = Cin; c->Cout = Cout; c->KH = KH; c->KW = KW; c->stride = stride; c->pad = KH / 2;
    std::vector<float> wt;
    if (Cin == 3) {  // stem: [64][8][8][4], zero padded (kh = 7, kw = 7, c = 3)
        c->stem = true;
        c->K = 256;
        std::vector<float> cw((size_t)Cout * 147);   // folded, [o][kh][kw][ci] without the padding: what gets rounded
        for (int o = 0; o < Cout; ++o)
            for (int ci = 0; ci < 3; ++ci)
                for (int kh = 0; kh < 7; ++kh)
                    for (int kw = 0; kw < 7; ++kw)
                        cw[(size_t)o * 147 + (kh * 7 + kw) * 3 + ci] = w.data[(((size_t)o * 3 + ci) * 7 + kh) * 7 + kw] * scale[o];
        if (m->wround) round_f16_diffused(cw.data(), (size_t)Cout, 49, 3, m->dtype == OPD_DT_BF16);
        wt.assign((size_t)Cout * 256, 0.f);
        for (int o = 0; o < Cout; ++o)
            for (int kh = 0; kh < 7; ++kh)
                for (int kw = 0; kw < 7; ++kw)
                    for (int ci = 0; ci < 3; ++ci) wt[(size_t)o * 256 + kh * 32 + kw * 4 + ci] = cw[(size_t)o * 147 + (kh * 7 + kw) * 3 + ci];
    } else {
        c->K = KH * KW * Cin;
        wt.resize((size_t)Cout * c->K);
        for (int o = 0; o < Cout; ++o)
            for (int ci = 0; ci < Cin; ++ci)
                for (int kh = 0; kh < KH; ++kh)
                    for (int kw = 0; kw < KW; ++kw)
                        wt[(size_t)o * c->K + (size_t)(kh * KW + kw) * Cin + ci] =
                            w.data[(((size_t)o * Cin + ci) * KH + kh) * KW + kw] * scale[o];
        // the fp16 image of the folded kernel: error diffusion along the reduction (opd_host.h) instead of round-to-nearest
        if (m->wround) round_f16_diffused(wt.data(), (size_t)Cout, KH * KW, Cin, m->dtype == OPD_DT_BF16);
    }
    RCCHK(upload_f16(m, &c->w, wt));
    if (KH == 1 && KW == 1 && Cin % 32 == 0 && Cin <= 1024) {  // operands of kernels_btail.hip / kernels_btail3.hip (stages 1-3)
        std::vector<float> wp(wt.size());
        for (int o = 0; o < Cout; ++o)
            for (int b = 0; b < Cin; b += 32)
                for (int g = 0; g < 4; ++g)
                    for (int e = 0; e < 4; ++e) {
                        wp[(size_t)o * Cin + b + 8 * g + e] = wt[(size_t)o * Cin + b + 4 * g + e];
                        wp[(size_t)o * Cin + b + 8 * g + 4 + e] = wt[(size_t)o * Cin + b + 16 + 4 * g + e];
                    }
        RCCHK(upload_f16(m, &c->wp, wp));
    }
    RCCHK(upload_f32(m, &c->bias, bias));
    return OPD_OK;
}

static int make_lin(opd_detr* m, const StateDict& sd, const std::string& prefix, Lin* l) {
    const HostTensor& w = T(sd, prefix + ".weight");
    l->N = (int)w.shape[0];
    l->K = (int)w.shape[1];
    RCCHK(upload_f16(m, &l->w, w.data));
    RCCHK(upload_f32(m, &l->b, T(sd, prefix + ".bias").data));
    return OPD_OK;
}
static int make_ln(opd_detr* m, const StateDict& sd, const std::string& prefix, LNp* l) {
    RCCHK(upload_f32(m, &l->g, T(sd, prefix + ".weight").data));
    RCCHK(upload_f32(m, &l->b, T(sd, prefix + ".bias").data));
    return OPD_OK;
}
static void append(std::vector<float>& dst, const std::vector<float>& src) { dst.insert(dst.end(), src.begin(), src.end()); }

static int build_weights(opd_detr* m, const StateDict& sd) {
    const Arch& a = m->arch;
    const std::string bb = "model.backbone.model.";
    RCCHK(make_conv(m, sd, bb + "embedder.embedder", 2, &m->stem));
    for (int s = 0; s < 4; ++s) {
        m->stage_first.push_back((int)m->blocks.size());
        for (int l = 0; l < a.depths[s]; ++l) {
            const std::string p = bb + "encoder.stages." + std::to_string(s) + ".layers." + std::to_string(l);
            const int stride = (l == 0 && s > 0) ? 2 : 1;
            Block b;
            b.has_sc = sd.count(p + ".shortcut.convolution.weight") > 0;
            if (b.has_sc) RCCHK(make_conv(m, sd, p + ".shortcut", stride, &b.sc));
            RCCHK(make_conv(m, sd, p + ".layer.0", 1, &b.c0));
            RCCHK(make_conv(m, sd, p + ".layer.1", stride, &b.c1));
            RCCHK(make_conv(m, sd, p + ".layer.2", 1, &b.c2));
            if (b.has_sc && b.sc.Cout == b.c2.Cout) {
                std::vector<float> b2(b.c2.Cout), bs(b.c2.Cout);
                HIPCHK(hipMemcpy(b2.data(), b.c2.bias, b2.size() * 4, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(bs.data(), b.sc.bias, bs.size() * 4, hipMemcpyDeviceToHost));
                for (size_t j = 0; j < b2.size(); ++j) b2[j] += bs[j];
                RCCHK(upload_f32(m, &b.bias2sc, b2));
                if (b.sc.KH == 1 && b.c2.KH == 1 && b.c2.Cout % 128 == 0 && b.c2.Cin >= 128) {   // stages 2-4: [W2 | Wsc]
                    const size_t K1 = (size_t)b.c2.K, K2 = (size_t)b.sc.K, N = (size_t)b.c2.Cout;
                    std::vector<f16_t> h2(N * K1), hs(N * K2), cat(N * (K1 + K2));
                    HIPCHK(hipMemcpy(h2.data(), b.c2.w, h2.size() * 2, hipMemcpyDeviceToHost));
                    HIPCHK(hipMemcpy(hs.data(), b.sc.w, hs.size() * 2, hipMemcpyDeviceToHost));
                    for (size_t n = 0; n < N; ++n) {
                        memcpy(&cat[n * (K1 + K2)], &h2[n * K1], K1 * 2);
                        memcpy(&cat[n * (K1 + K2) + K1], &hs[n * K2], K2 * 2);
                    }
                    RCCHK(dalloc(m, &b.w2sc, cat.size(), true));
                    HIPCHK(hipMemcpy(b.w2sc, cat.data(), cat.size() * 2, hipMemcpyHostToDevice));
                }
            }
            m->blocks.push_back(b);
        }
    }
    {  // input_projection: plain 1x1 conv with bias, no BN
        const HostTensor& w = T(sd, "model.input_projection.weight");
        m->proj.Cin = (int)w.shape[1]; m->proj.Cout = (int)w.shape[0]; m->proj.K = m->proj.Cin;
        RCCHK(upload_f16(m, &m->proj.w, w.data));
        RCCHK(upload_f32(m, &m->proj.bias, T(sd, "model.input_projection.bias").data));
    }
    const int D = a.d_model;
    const std::vector<float> zerosW((size_t)D * D, 0.f);
    auto cat3 = [&](const std::string& p, std::vector<float>* w_full, std::vector<float>* w_pos, std::vector<float>* b_cat) {
        // w_full = [Wq;Wk;Wv] (GEMM weights), w_pos = [Wq;Wk;0] and b_cat = [bq;bk;bv] (row-bias fold)
        w_full->clear(); w_pos->clear(); b_cat->clear();
        append(*w_full, T(sd, p + ".q_proj.weight").data); append(*w_full, T(sd, p + ".k_proj.weight").data);
        append(*w_full, T(sd, p + ".v_proj.weight").data);
        append(*w_pos, T(sd, p + ".q_proj.weight").data); append(*w_pos, T(sd, p + ".k_proj.weight").data);
        append(*w_pos, zerosW);
        append(*b_cat, T(sd, p + ".q_proj.bias").data); append(*b_cat, T(sd, p + ".k_proj.bias").data);
        append(*b_cat, T(sd, p + ".v_proj.bias").data);
    };
    m->enc.resize(a.enc_layers);
    m->h_enc_cat_w.resize(a.enc_layers);
    m->h_enc_cat_b.resize(a.enc_layers);
    for (int i = 0; i < a.enc_layers; ++i) {
        const std::string p = "model.encoder.layers." + std::to_string(i);
        EncLayer& L = m->enc[i];
        std::vector<float> wfull;
        cat3(p + ".self_attn", &wfull, &m->h_enc_cat_w[i], &m->h_enc_cat_b[i]);
        RCCHK(upload_f16(m, &L.wqkv, wfull));
        RCCHK(upload_f32(m, &L.bqkv, m->h_enc_cat_b[i]));
        RCCHK(make_lin(m, sd, p + ".self_attn.o_proj", &L.o));
        RCCHK(make_ln(m, sd, p + ".self_attn_layer_norm", &L.ln1));
        RCCHK(make_lin(m, sd, p + ".mlp.fc1", &L.fc1));
        RCCHK(make_lin(m, sd, p + ".mlp.fc2", &L.fc2));
        RCCHK(make_ln(m, sd, p + ".final_layer_norm", &L.ln2));
    }
    // decoder: query-position folds are resolution independent -> build them now with the fp32 plan GEMM
    float* d_qpos = nullptr;
    RCCHK(upload_f32(m, &d_qpos, T(sd, "model.query_position_embeddings.weight").data));
    const int Q = a.queries;
    m->dec.resize(a.dec_layers);
    std::vector<float> kv_full;
    for (int i = 0; i < a.dec_layers; ++i) {
        const std::string p = "model.decoder.layers." + std::to_string(i);
        DecLayer& L = m->dec[i];
        std::vector<float> wfull, wpos, bcat;
        cat3(p + ".self_attn", &wfull, &wpos, &bcat);
        RCCHK(upload_f16(m, &L.wqkv, wfull));
        float *d_w = nullptr, *d_b = nullptr;
        RCCHK(upload_f32(m, &d_w, wpos));
        RCCHK(upload_f32(m, &d_b, bcat));
        RCCHK(dalloc(m, &L.rb_self, (size_t)Q * 768, true));
        HIPCHK(opd_launch_gemm_f32(d_qpos, d_w, d_b, L.rb_self, Q, 768, D, 768, m->stream));
        RCCHK(make_lin(m, sd, p + ".self_attn.o_proj", &L.so));
        RCCHK(make_ln(m, sd, p + ".self_attn_layer_norm", &L.ln1));
        // cross attention: q from the decoder state, k/v from the encoder memory
        RCCHK(upload_f16(m, &L.wq_c, T(sd, p + ".encoder_attn.q_proj.weight").data));
        float *d_wq = nullptr, *d_bq = nullptr;
        RCCHK(upload_f32(m, &d_wq, T(sd, p + ".encoder_attn.q_proj.weight").data));
        RCCHK(upload_f32(m, &d_bq, T(sd, p + ".encoder_attn.q_proj.bias").data));
        RCCHK(dalloc(m, &L.rb_q, (size_t)Q * D, true));
        HIPCHK(opd_launch_gemm_f32(d_qpos, d_wq, d_bq, L.rb_q, Q, D, D, D, m->stream));
        append(kv_full, T(sd, p + ".encoder_attn.k_proj.weight").data);
        append(kv_full, T(sd, p + ".encoder_attn.v_proj.weight").data);
        append(m->h_kv_cat_w, T(sd, p + ".encoder_attn.k_proj.weight").data);
        append(m->h_kv_cat_w, zerosW);
        append(m->h_kv_cat_b, T(sd, p + ".encoder_attn.k_proj.bias").data);
        append(m->h_kv_cat_b, T(sd, p + ".encoder_attn.v_proj.bias").data);
        RCCHK(make_lin(m, sd, p + ".encoder_attn.o_proj", &L.co));
        RCCHK(make_ln(m, sd, p + ".encoder_attn_layer_norm", &L.ln2));
        RCCHK(make_lin(m, sd, p + ".mlp.fc1", &L.fc1));
        RCCHK(make_lin(m, sd, p + ".mlp.fc2", &L.fc2));
        RCCHK(make_ln(m, sd, p + ".final_layer_norm", &L.ln3));
        // split pairs for the fused decoder
        if (D % 32 == 0 && a.ffn % 32 == 0) {
            RCCHK(upload_frag(m, &L.wqkv_f, wfull, 3 * D, D));
            RCCHK(upload_frag(m, &L.so_f, T(sd, p + ".self_attn.o_proj.weight").data, D, D));
            RCCHK(upload_frag(m, &L.wqc_f, T(sd, p + ".encoder_attn.q_proj.weight").data, D, D));
            RCCHK(upload_frag(m, &L.co_f, T(sd, p + ".encoder_attn.o_proj.weight").data, D, D));
            RCCHK(upload_frag(m, &L.fc1_f, T(sd, p + ".mlp.fc1.weight").data, a.ffn, D));
            RCCHK(upload_frag(m, &L.fc2_f, T(sd, p + ".mlp.fc2.weight").data, D, a.ffn));
        }
    }
    RCCHK(upload_f16(m, &m->wkv_all, kv_full));
    RCCHK(upload_f32(m, &m->bkv_all, m->h_kv_cat_b));
    if (a.d_model == 256 && a.ffn % 128 == 0) {   // the encoder FFN blocks as single launches, each with the projection that consumes its output
        const int D = 256, L = a.dec_layers;
        for (int i = 0; i < a.enc_layers; ++i) {
            EncLayer& E = m->enc[i];
            const std::string p = "model.encoder.layers." + std::to_string(i);
            std::vector<float> wt, bt;
            if (i + 1 < a.enc_layers) {   // the next layer's q, k (on x + pos), v
                const std::string n = "model.encoder.layers." + std::to_string(i + 1) + ".self_attn.";
                for (const char* pr : {"q_proj", "k_proj", "v_proj"}) { append(wt, T(sd, n + pr + ".weight").data); append(bt, T(sd, n + pr + ".bias").data); }
                E.tail = 3; E.tail_pos = 2; E.tail_ld = 3 * D;
                for (int t = 0; t < 3; ++t) E.tail_col[t] = t * D;
            } else if (2 * L <= 16 && kv_full.size() == (size_t)L * 2 * D * D && m->h_kv_cat_b.size() == (size_t)L * 2 * D) {
                // the decoder's memory projections, [k_l | v_l] per layer in wkv_all: passes k_0 .. k_{L-1} (on x + pos), then v_0 .. v_{L-1}
                for (int kv = 0; kv < 2; ++kv)
                    for (int l = 0; l < L; ++l) {
                        wt.insert(wt.end(), kv_full.begin() + (size_t)(2 * l + kv) * D * D, kv_full.begin() + (size_t)(2 * l + kv + 1) * D * D);
                        bt.insert(bt.end(), m->h_kv_cat_b.begin() + (size_t)(2 * l + kv) * D, m->h_kv_cat_b.begin() + (size_t)(2 * l + kv + 1) * D);
                        E.tail_col[kv * L + l] = (2 * l + kv) * D;
                    }
                E.tail = 2 * L; E.tail_pos = L; E.tail_ld = 2 * L * D;
            }
            RCCHK(upload_encffn(m, &E.ffn_pack, T(sd, p + ".mlp.fc1.weight").data, T(sd, p + ".mlp.fc1.bias").data, T(sd, p + ".mlp.fc2.weight").data, a.ffn, wt, bt, E.tail,
                                T(sd, p + ".self_attn.o_proj.weight").data));
            E.front = 1;
        }
    }
    {   // The decoder starts from h = 0 (HF:models/detr/modeling_detr.py:1243-1251), so in layer 0 the self-attention values are the
        // same row for every query, v = 0 . Wv^T + bv, the softmax weights of a row sum to one, and the block's output
        // LN(0 + Wo . bv + bo) is ONE vector, whatever the frame shows: computed here once in fp32, broadcast at run time instead of
        // two memsets, the QKV projection, the attention and the output projection + LayerNorm of that layer.
        const std::string p0 = "model.decoder.layers.0";
        const auto& bv = T(sd, p0 + ".self_attn.v_proj.bias").data;
        const auto& wo = T(sd, p0 + ".self_attn.o_proj.weight").data;
        const auto& bo = T(sd, p0 + ".self_attn.o_proj.bias").data;
        const auto& g = T(sd, p0 + ".self_attn_layer_norm.weight").data;
        const auto& be = T(sd, p0 + ".self_attn_layer_norm.bias").data;
        std::vector<float> x(D), c(D);
        for (int n = 0; n < D; ++n) {
            float acc = 0.f;
            for (int k = 0; k < D; ++k) acc += wo[(size_t)n * D + k] * bv[k];
            x[n] = acc + bo[n];
        }
        float mean = 0.f, var = 0.f;
        for (int n = 0; n < D; ++n) mean += x[n];
        mean /= (float)D;
        for (int n = 0; n < D; ++n) var += (x[n] - mean) * (x[n] - mean);
        var /= (float)D;
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        for (int n = 0; n < D; ++n) c[n] = (x[n] - mean) * rstd * g[n] + be[n];
        RCCHK(upload_f32(m, &m->dec0_h, c));
        // ... and so are layer 0's cross-attention queries, (h1 + qpos) . Wq_c^T + bq_c: one [Q][D] table (fp32 sums, stored as the fp16 operand
        // the attention kernel reads)
        const auto& qpos = T(sd, "model.query_position_embeddings.weight").data;
        const auto& wq = T(sd, p0 + ".encoder_attn.q_proj.weight").data;
        const auto& bq = T(sd, p0 + ".encoder_attn.q_proj.bias").data;
        std::vector<float> q0((size_t)Q * D);
        for (int q = 0; q < Q; ++q)
            for (int n = 0; n < D; ++n) {
                double acc = bq[n];
                for (int k = 0; k < D; ++k) acc += ((double)c[k] + qpos[(size_t)q * D + k]) * wq[(size_t)n * D + k];
                q0[(size_t)q * D + n] = (float)acc;
            }
        RCCHK(upload_f16(m, &m->qc0, q0));
    }
    RCCHK(make_ln(m, sd, "model.decoder.layernorm", &m->dec_ln));
    auto transposed = [&](const std::string& key) {  // [out][in] -> [in][out] (coalesced reads in heads_kernel)
        const HostTensor& w = T(sd, key);
        const int O = (int)w.shape[0], I = (int)w.shape[1];
        std::vector<float> t((size_t)O * I);
        for (int o = 0; o < O; ++o)
            for (int i = 0; i < I; ++i) t[(size_t)i * O + o] = w.data[(size_t)o * I + i];
        return t;
    };
    RCCHK(upload_f32(m, &m->wc, transposed("class_labels_classifier.weight")));
    RCCHK(upload_f32(m, &m->bc, T(sd, "class_labels_classifier.bias").data));
    if (a.d_model == 256 && a.ncls <= 128) {   // the heads on split fp16 operands (kernels_dec.hip::heads2_kernel): class matrix padded to 128 rows
        std::vector<float> wcp((size_t)128 * 256, 0.f);
        const auto& wcs = T(sd, "class_labels_classifier.weight").data;
        std::copy(wcs.begin(), wcs.end(), wcp.begin());
        RCCHK(upload_frag(m, &m->wc_f, wcp, 128, 256));
        RCCHK(upload_frag(m, &m->w1_f, T(sd, "bbox_predictor.layers.0.weight").data, 256, 256));
        RCCHK(upload_frag(m, &m->w2_f, T(sd, "bbox_predictor.layers.1.weight").data, 256, 256));
    }
    RCCHK(upload_f32(m, &m->w1, transposed("bbox_predictor.layers.0.weight")));
    RCCHK(upload_f32(m, &m->b1, T(sd, "bbox_predictor.layers.0.bias").data));
    RCCHK(upload_f32(m, &m->w2, transposed("bbox_predictor.layers.1.weight")));
    RCCHK(upload_f32(m, &m->b2, T(sd, "bbox_predictor.layers.1.bias").data));
    RCCHK(upload_f32(m, &m->w3, transposed("bbox_predictor.layers.2.weight")));
    RCCHK(upload_f32(m, &m->b3, T(sd, "bbox_predictor.layers.2.bias").data));
    RCCHK(upload_f32(m, &m->zero_bias, std::vector<float>(4096, 0.f)));
    HIPCHK(hipStreamSynchronize(m->stream));
    return OPD_OK;
}

static void compute_dims(int B, int H, int W, Dims* d) {
    d->B = B; d->H = H; d->W = W;
    d->H1 = down2(H); d->W1 = down2(W);
    d->H2 = down2(d->H1); d->W2 = down2(d->W1);
    d->sh[0] = d->H2; d->sw[0] = d->W2;
    for (int s = 1; s < 4; ++s) { d->sh[s] = down2(d->sh[s - 1]); d->sw[s] = down2(d->sw[s - 1]); }
}

// Layer 0's cross-attention queries do not depend on the frames (build_weights: qc0): the fused decoder reads them from layer 0's region of
// d_qd16, where opd_detr_attention_map finds them too; written once per handle, every frame slot (and again by the test hooks that
// switch between the fused and the unfused decoder, whose layer 0 writes its own rounding of the same values there).
int fill_qc0(opd_detr* m) {
    if (!m->qc0 || !m->d_qd16) return OPD_OK;
    const size_t QD = (size_t)m->arch.queries * m->arch.d_model;
    for (int b = 0; b < m->cfg.max_batch; ++b) HIPCHK(hipMemcpy(m->d_qd16 + b * QD, m->qc0, QD * 2, hipMemcpyDeviceToDevice));
    return OPD_OK;
}

static int conv_splits(const opd_detr* m, const Conv& c, int stage);   // (below, with run_conv)
static int build_workspace(opd_detr* m) {
    const Arch& a = m->arch;
    // Frames may come in either orientation (the HF size rule maps a portrait camera frame to about 1333 x 750): the handle
    // accepts every H x W with H, W <= max(max_height, max_width) and H * W <= max_height * max_width, so the buffers are sized by
    // bounds on the pixel COUNT of each pyramid level, not by one shape: down2(n) <= (n + 1) / 2, hence
    // H1 * W1 <= (H * W + H + W + 1) / 4 <= (n + 2 * L + 1) / 4 for a level with n pixels and sides <= L.
    const size_t B = m->cfg.max_batch;
    size_t edge = (size_t)std::max(m->cfg.max_height, m->cfg.max_width);
    const size_t npix = B * (size_t)m->cfg.max_height * m->cfg.max_width;
    size_t lvl[6], side[6];   // per-frame pixel bound / side bound of: image, stem, pool (= stage 1), stage 2, 3, 4
    lvl[0] = (size_t)m->cfg.max_height * m->cfg.max_width; side[0] = edge;
    for (int k = 1; k < 6; ++k) { lvl[k] = (lvl[k - 1] + 2 * side[k - 1] + 1) / 4 + 1; side[k] = (size_t)down2((int)side[k - 1]); }
    RCCHK(dalloc(m, &m->d_u8, npix * 3, false));
    RCCHK(dalloc(m, &m->d_pv, npix * 3, false));
    RCCHK(dalloc(m, &m->d_x4, B * (4 * lvl[1] + 24 * side[1] + 36) * 4, false));  // zero-bordered NHWC4: (2 H1 + 6) x (2 W1 + 6)
    RCCHK(dalloc(m, &m->d_stem, B * lvl[1] * 64, false));
    RCCHK(dalloc(m, &m->d_pool, B * lvl[2] * 64, false));
    size_t trunk = 0, mid = 0;
    for (int s = 0; s < 4; ++s) {
        const size_t hw = lvl[2 + s];
        trunk = std::max(trunk, B * hw * a.hidden[s]);
        // first block of a stage runs its 1x1 reduce at the INPUT resolution of the stage
        const size_t hw_in = s == 0 ? hw : lvl[1 + s];
        mid = std::max(mid, B * hw_in * (a.hidden[s] / 4));
    }
    RCCHK(dalloc(m, &m->d_t0, trunk, false));
    RCCHK(dalloc(m, &m->d_t1, trunk, false));
    RCCHK(dalloc(m, &m->d_sc, trunk, false));
    RCCHK(dalloc(m, &m->d_m0, mid, false));
    RCCHK(dalloc(m, &m->d_m1, mid, false));
    const size_t M = B * lvl[5];
    const size_t D = a.d_model, Md = B * a.queries;
    RCCHK(dalloc(m, &m->d_x32, M * D, false));
    RCCHK(dalloc(m, &m->d_y32, M * D, false));
    for (int s = 0; s < 4; ++s) m->stage_px[s] = lvl[2 + s];
    m->slab_floats = std::max(M * D * 8, Md * D * 8);   // (split-K slabs of the encoder side: 4 slices, 8 for small handles)
    for (size_t bi = 0; bi < m->blocks.size(); ++bi) {   // split-K plans of small handles (conv_splits): room for their fp32 slabs
        int s = 0;
        while (s + 1 < 4 && (int)bi >= m->stage_first[s + 1]) ++s;
        const Block& b = m->blocks[bi];
        const bool first = (int)bi == m->stage_first[s];
        // c0 runs at the stage's INPUT resolution in its first block, c1 / c2 at the output resolution
        const int s_c0 = first && s > 0 ? s - 1 : s;
        m->slab_floats = std::max(m->slab_floats, (size_t)conv_splits(m, b.c0, s_c0) * B * m->stage_px[s_c0] * b.c0.Cout);
        m->slab_floats = std::max(m->slab_floats, (size_t)conv_splits(m, b.c1, s) * B * m->stage_px[s] * b.c1.Cout);
    }
    RCCHK(dalloc(m, &m->d_slab, m->slab_floats, false));
    RCCHK(dalloc(m, &m->d_x16, M * D, false));
    RCCHK(dalloc(m, &m->d_xp16, M * D, false));
    RCCHK(dalloc(m, &m->d_qkv16, M * 3 * D, false));
    RCCHK(dalloc(m, &m->d_attn16, M * D, false));
    RCCHK(dalloc(m, &m->d_ffn16, M * a.ffn, false));
    RCCHK(dalloc(m, &m->d_memkv16, M * 2 * D * a.dec_layers, false));
    RCCHK(dalloc(m, &m->d_h32, Md * D, false));
    RCCHK(dalloc(m, &m->d_yd32, Md * D, false));
    RCCHK(dalloc(m, &m->d_hs32, Md * D, false));
    RCCHK(dalloc(m, &m->d_h16, Md * D, false));
    RCCHK(dalloc(m, &m->d_qkvd16, Md * 3 * D, false));
    RCCHK(dalloc(m, &m->d_qd16, Md * D * a.dec_layers, false));   // (every layer's cross-attention queries stay: opd_detr_attention_map)
    RCCHK(dalloc(m, &m->d_amap, (size_t)lvl[5] + 64, false));
    RCCHK(dalloc(m, &m->d_amap_stat, (size_t)a.queries * a.heads * 2, false));
    RCCHK(dalloc(m, &m->d_amap_sel, (size_t)a.queries, false));
    RCCHK(dalloc(m, &m->d_attnd16, Md * D, false));
    RCCHK(dalloc(m, &m->d_ffnd16, Md * a.ffn, false));
    RCCHK(dalloc(m, &m->d_dq16, Md * D, false));
    RCCHK(dalloc(m, &m->d_dk16, B * 8 * 8 * 512, false));   // (fragment order: 8 heads x 8 key tiles x 1 KiB per frame)
    RCCHK(dalloc(m, &m->d_dvT, B * 8 * 8 * 512, false));
    RCCHK(dalloc(m, &m->d_part_o, (size_t)m->dec_splits * Md * D, false));
    RCCHK(dalloc(m, &m->d_part_ml, (size_t)m->dec_splits * Md * a.heads * 2, false));
    RCCHK(dalloc(m, &m->d_ffn_part, (size_t)(a.ffn / OPD_DEC_FFN_CHUNK + 1) * Md * D, false));
    RCCHK(fill_qc0(m));
    RCCHK(dalloc(m, &m->d_logits, Md * a.ncls, false));
    RCCHK(dalloc(m, &m->d_boxes, Md * 4, false));
    // records of max_batch frames, the per-frame counts right behind them: ONE device-to-host copy fetches both
    // (and behind the counts, 32-byte aligned, room for one feature row per record: opd_detr_detect_frames_features fetches all three at once)
    const size_t cnt_units = ((size_t)B * 4 + sizeof(opd_det) - 1) / sizeof(opd_det);
    RCCHK(dalloc(m, &m->d_records, Md + cnt_units + Md * D * 4 / sizeof(opd_det), false));
    m->d_counts = reinterpret_cast<int32_t*>(m->d_records + Md);
    m->d_feat_all = reinterpret_cast<float*>(m->d_records + Md + cnt_units);
    RCCHK(dalloc(m, &m->d_orig_hw, B * 2, false));
    RCCHK(dalloc(m, &m->d_valid_hw, B * 2, false));
    RCCHK(dalloc(m, &m->d_key_valid, B * 2, false));
    RCCHK(dalloc(m, &m->d_bias_ptrs, (size_t)(a.enc_layers + 2) * B, false));
    RCCHK(dalloc(m, &m->d_rois, (size_t)128 * 4, false));
    RCCHK(dalloc(m, &m->d_roi_out, (size_t)128 * D, false));
    return OPD_OK;
}

static int get_plan(opd_detr* m, int fh, int fw, int vh, int vw, Plan** out) {
    // the cache lives with the weights; a fold is built once, on the calling handle's stream, and is complete (stream
    // synchronised) before the lock is released.  Its buffers belong to the WeightSet: they may outlive this handle.
    std::lock_guard<std::mutex> plan_lock(m->weights->plan_mu);
    auto& plans = m->weights->plans;
    for (auto& p : plans)
        if (p->fh == fh && p->fw == fw && p->vh == vh && p->vw == vw) { *out = p.get(); return OPD_OK; }
    struct Unseal {
        opd_detr* m;
        explicit Unseal(opd_detr* mm) : m(mm) { m->weights_sealed = false; }
        ~Unseal() { m->weights_sealed = true; }
    } unseal(m);
    const Arch& a = m->arch;
    const int D = a.d_model, hw = fh * fw;
    std::unique_ptr<Plan> p(new Plan());
    p->fh = fh; p->fw = fw; p->vh = vh; p->vw = vw;
    std::vector<float> pos;
    sine_pos_embed(fh, fw, vh, vw, D, &pos);
    float* d_pos = nullptr;
    RCCHK(upload_f32(m, &d_pos, pos));
    p->d_pos = d_pos;
    p->rb_enc.resize(a.enc_layers);
    for (int i = 0; i < a.enc_layers; ++i) {
        float *d_w = nullptr, *d_b = nullptr;
        RCCHK(upload_f32(m, &d_w, m->h_enc_cat_w[i]));
        RCCHK(upload_f32(m, &d_b, m->h_enc_cat_b[i]));
        RCCHK(dalloc(m, &p->rb_enc[i], (size_t)hw * 768, true));
        HIPCHK(opd_launch_gemm_f32(d_pos, d_w, d_b, p->rb_enc[i], hw, 768, D, 768, m->stream));
    }
    {
        const int NKV = a.dec_layers * 512;
        float *d_w = nullptr, *d_b = nullptr;
        RCCHK(upload_f32(m, &d_w, m->h_kv_cat_w));
        RCCHK(upload_f32(m, &d_b, m->h_kv_cat_b));
        RCCHK(dalloc(m, &p->rb_kv, (size_t)hw * NKV, true));
        HIPCHK(opd_launch_gemm_f32(d_pos, d_w, d_b, p->rb_kv, hw, NKV, D, NKV, m->stream));
    }
    HIPCHK(hipStreamSynchronize(m->stream));
    *out = p.get();
    plans.push_back(std::move(p));
    return OPD_OK;
}

// ---- per-launch timing --------------------------------------------------------------------------------------------
enum { CLS_CONV = 0, CLS_GEMM = 1, CLS_ATTN = 2, CLS_OTHER = 3 };

static int timed_begin(opd_detr* m, int cls, double flops) {
    if (m->profiling != 1) return OPD_OK;
    hipEvent_t e[2];
    for (int i = 0; i < 2; ++i) {
        if (m->pool_next == m->event_pool.size()) {
            hipEvent_t ne;
            HIPCHK(hipEventCreate(&ne));
            m->event_pool.push_back(ne);
        }
        e[i] = m->event_pool[m->pool_next++];
    }
    HIPCHK(hipEventRecord(e[0], m->stream));
    m->timed.push_back({cls, e[0], e[1], flops, nullptr});
    opd_last_kernel_name = nullptr;
    return OPD_OK;
}
static int timed_end(opd_detr* m) {
    if (m->profiling != 1) return OPD_OK;
    HIPCHK(hipEventRecord(m->timed.back().b, m->stream));
    m->timed.back().name = opd_last_kernel_name;   // what the launcher just launched (OPD_LAUNCH): the name rocprofv3 prints, minus namespace and signature
    return OPD_OK;
}
static void timed_reset(opd_detr* m) {
    m->timed.clear();
    m->pool_next = 0;
}
static void timed_collect(opd_detr* m) {
    for (int c = 0; c < 4; ++c) { m->class_ms[c] = 0.f; m->class_launches[c] = 0; m->class_flops[c] = 0.0; }
    m->ktable.clear();
    static const char* const cls_name[4] = {"(convolution)", "(linear layer)", "(attention)", "(other)"};
    for (const auto& t : m->timed) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
            m->class_ms[t.cls] += ms;
            m->class_launches[t.cls] += 1;
            m->class_flops[t.cls] += t.flops;
            std::string nm = t.name ? t.name : cls_name[t.cls];
            if (!nm.empty() && nm.front() == '(' && nm.back() == ')' && t.name) nm = nm.substr(1, nm.size() - 2);   // OPD_LAUNCH((kernel<a, b>), ...)
            opd_detr::KernelRow* row = nullptr;
            for (auto& r : m->ktable)
                if (r.name == nm) { row = &r; break; }
            if (!row) { m->ktable.push_back({nm, 0, 0.f, 0.0}); row = &m->ktable.back(); }
            row->launches += 1; row->ms += ms; row->flops += t.flops;
        }
    }
    std::sort(m->ktable.begin(), m->ktable.end(), [](const opd_detr::KernelRow& a, const opd_detr::KernelRow& b) { return a.ms > b.ms; });
}

static int tap(opd_detr* m, const char* name, const void* p, size_t bytes) {
    if (!m->taps || !m->d_taps || m->tap_next >= OPD_MAX_TAPS) return OPD_OK;
    HIPCHK(opd_launch_checksum(p, bytes, m->d_taps + (size_t)m->tap_next * OPD_TAP_BLOCKS, m->stream));
    if ((int)m->tap_names.size() <= m->tap_next) m->tap_names.resize(m->tap_next + 1);
    m->tap_names[m->tap_next++] = name;
    return OPD_OK;
}

// Split-K plan of a deep convolution for SMALL handles (round 5).  A max_batch = 1 handle at 800 x 1333 runs stage 4's 3x3 as 72 workgroups
// that each walk 72 k-steps (50 us: the launch lasts one workgroup's life, 184 CUs idle); cut eight ways it is 576 workgroups of 9 k-steps
// plus a 3-us reduction.  The split follows the handle's CONFIGURATION (max_batch, the frame-size bounds, the layer), never the batch or frame
// at hand -- a frame's low-order bits must not depend on the call it travels in -- and applies only where the unsplit launch would leave most
// CUs without a workgroup.  Returns 1 (no split) or a divisor of the k-step count.
static int conv_splits(const opd_detr* m, const Conv& c, int stage) {
    if (!m->small_splitk || stage < 0 || stage > 3 || c.stem || c.K % 64 != 0 || c.Cout % 64 != 0) return 1;
    const long long px = (long long)m->cfg.max_batch * (long long)m->stage_px[stage];
    const long long tiles = ((px + 127) / 128) * (c.Cout / 64);
    const int nk = c.K / 64;
    if (tiles > 160 || nk < 24) return 1;
    int best = 1;
    for (int s : {2, 3, 4, 6, 8})
        if (nk % s == 0 && nk / s >= 6 && tiles * s <= 640) best = s;
    return best;
}

static int run_conv(opd_detr* m, const Conv& c, const f16_t* x, int B, int H, int W, int OH, int OW, void* out, bool relu,
                    const f16_t* res16, int stage = -1) {
    ConvGemmParams p{}; p.dtype = m->dtype;
    p.x = x; p.w = c.w; p.bias = c.bias; p.res16 = res16; p.res32 = nullptr; p.out = out; p.out16_aux = nullptr; p.zero16 = m->zero_bias;
    p.B = B; p.H = H; p.W = W; p.Cin = c.Cin; p.OH = OH; p.OW = OW; p.N = c.Cout; p.KH = c.KH; p.KW = c.KW;
    p.stride = c.stride; p.pad = c.pad; p.M = B * OH * OW; p.K = c.K; p.relu = relu ? 1 : 0; p.bias_period = 0;
    p.out_f32 = 0; p.stem = c.stem ? 1 : 0; p.dbg = m->dbg_gemm; p.wprefetch = m->wprefetch & 1;
    // algorithmic FLOPs (2 x MAC over the real taps/channels; the stem's zero padding is not counted)
    if (const int splits = res16 ? 1 : conv_splits(m, c, stage); splits > 1 && (size_t)splits * p.M * c.Cout <= m->slab_floats) {
        p.out = m->d_slab; p.out_f32 = 1; p.relu = 0; p.split_k = splits;
        RCCHK(timed_begin(m, CLS_CONV, 2.0 * p.M * (double)c.Cout * c.KH * c.KW * c.Cin));
        HIPCHK(opd_launch_conv_gemm(p, m->stream));
        RCCHK(timed_end(m));
        RCCHK(timed_begin(m, CLS_OTHER, 0.0));
        HIPCHK(opd_launch_reduce_act16(m->d_slab, splits, (size_t)p.M * c.Cout, reinterpret_cast<f16_t*>(out), (size_t)p.M * c.Cout, relu ? 1 : 0, m->stream, m->dtype));
        RCCHK(timed_end(m));
        RCCHK(tap(m, c.KH == 3 ? "conv3x3" : "conv1x1", out, (size_t)p.M * c.Cout * 2));
        return OPD_OK;
    }
    RCCHK(timed_begin(m, CLS_CONV, 2.0 * p.M * (double)c.Cout * c.KH * c.KW * c.Cin));
    // Wide layers with few row tiles (stage 4) through the eight-wave kernel (kernels_w8.hip; identical bits).  The choice follows the handle's
    // CONFIGURATION (max_batch and the layer), never the batch at hand.  OPD_W8: bit 0 = 3x3, bit 1 = 1x1 with K >= 1024, bit 2 = 1x1 with K = 512.
    bool w8 = false;
    if (m->w8 && c.Cout % 256 == 0 && c.Cout >= 512) {
        const long long tiles = (((long long)OH * OW * m->cfg.max_batch + 127) / 128) * (c.Cout / 256);
        const bool few = tiles <= 3LL * m->num_cus;
        const int kind = c.KH == 3 ? (m->w8 & 1) : (c.K >= 1024 ? (m->w8 & 2) : (c.K == 512 ? (m->w8 & 4) : 0));
        w8 = few && kind && opd_conv_w8_supported(p);
    }
    HIPCHK(w8 ? opd_launch_conv_w8(p, m->stream) : opd_launch_conv_gemm(p, m->stream));
    RCCHK(timed_end(m));
    RCCHK(tap(m, c.KH == 3 ? "conv3x3" : "conv1x1", out, (size_t)p.M * c.Cout * 2));
    return OPD_OK;
}

// out[M][N] = x16[M][K] . w[N][K]^T + bias (+ res32), as a 1x1 "convolution" over M pixels
static int run_gemm(opd_detr* m, const f16_t* x, const f16_t* w, const float* bias, int bias_period, int M, int N, int K,
                    void* out, bool out_f32, bool relu, const float* res32, const float* const* bias_ptrs = nullptr, int bias_pmod = 0,
                    int bias_pcols = 0, const f16_t* x_alt = nullptr, int alt_mod = 0, int alt_cols = 0) {
    ConvGemmParams p{}; p.dtype = m->dtype;
    p.bias_ptrs = bias_ptrs; p.bias_pmod = bias_pmod; p.bias_pcols = bias_pcols;
    p.x_alt = x_alt; p.alt_mod = alt_mod; p.alt_cols = alt_cols;
    p.x = x; p.w = w; p.bias = bias; p.res16 = nullptr; p.res32 = res32; p.out = out; p.out16_aux = nullptr; p.zero16 = m->zero_bias;
    p.B = M; p.H = 1; p.W = 1; p.Cin = K; p.OH = 1; p.OW = 1; p.N = N; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0;
    p.M = M; p.K = K; p.relu = relu ? 1 : 0; p.bias_period = bias_period; p.out_f32 = out_f32 ? 1 : 0; p.stem = 0; p.dbg = m->dbg_gemm; p.wprefetch = m->wprefetch & 1;
    RCCHK(timed_begin(m, CLS_GEMM, 2.0 * M * (double)N * K));
    HIPCHK(opd_launch_conv_gemm(p, m->stream));
    RCCHK(timed_end(m));
    RCCHK(tap(m, "gemm", out, (size_t)M * N * (out_f32 ? 4 : 2)));
    return OPD_OK;
}

// Split-K flavour for skinny / deep-K linears: slices write fp32 slabs, then ONE fused kernel reduces them in slice
// order, adds the residual stream and applies the post-LayerNorm (gamma == nullptr: plain sum, e.g. input_projection).
struct PosShadow { const float* pos; const float* const* pos_ptrs; int period; f16_t* yp16; };   // second fp16 output of a reduce + LN
static int run_gemm_splitk_ln(opd_detr* m, const f16_t* x, const f16_t* w, const float* bias, int M, int N, int K, int splits,
                              const float* res32, const LNp* ln, float* y32, f16_t* y16, int cls, const PosShadow* ps = nullptr) {
    ConvGemmParams p{}; p.dtype = m->dtype;
    p.x = x; p.w = w; p.bias = bias; p.out = m->d_slab; p.zero16 = m->zero_bias;
    p.B = M; p.H = 1; p.W = 1; p.Cin = K; p.OH = 1; p.OW = 1; p.N = N; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0;
    p.M = M; p.K = K; p.out_f32 = 1; p.split_k = splits;
    RCCHK(timed_begin(m, cls, 2.0 * M * (double)N * K));
    HIPCHK(opd_launch_conv_gemm(p, m->stream));
    RCCHK(timed_end(m));
    RCCHK(tap(m, "splitk_slabs", m->d_slab, (size_t)splits * M * N * 4));
    RCCHK(timed_begin(m, CLS_OTHER, 0.0));
    HIPCHK(opd_launch_reduce_ln_pos(m->d_slab, splits, (size_t)M * N, res32, ln ? ln->g : nullptr, ln ? ln->b : nullptr, y32, y16, M,
                                    ps ? ps->pos : nullptr, ps ? ps->pos_ptrs : nullptr, ps ? ps->period : 0, ps ? ps->yp16 : nullptr, m->stream, m->dtype));
    RCCHK(timed_end(m));
    RCCHK(tap(m, "reduce_ln", y32, (size_t)M * N * 4));
    return OPD_OK;
}

// y = LayerNorm(x16 . w^T + bias + res32): one launch (kernels_rowln.hip); y32 may alias res32
static int run_gemm_ln(opd_detr* m, const f16_t* x, const f16_t* w, const float* bias, int M, int K, const float* res32,
                       const LNp& ln, float* y32, f16_t* y16) {
    GemmLnParams p{}; p.dtype = m->dtype;
    p.x = x; p.w = w; p.bias = bias; p.res32 = res32; p.gamma = ln.g; p.beta = ln.b; p.y32 = y32; p.y16 = y16; p.M = M; p.K = K;
    RCCHK(timed_begin(m, CLS_GEMM, 2.0 * M * 256.0 * K));
    HIPCHK(opd_launch_gemm_ln(p, m->stream));
    RCCHK(timed_end(m));
    RCCHK(tap(m, "gemm_ln", y32, (size_t)M * 256 * 4));
    return OPD_OK;
}

// Decoder-sized linear layer (M = B x queries, K a multiple of 256): the one-shot kernel of kernels_rowln.hip; K > 256 is
// cut into 256-wide slices whose fp32 slabs are summed by the fused reduce + residual + LayerNorm kernel.
static int run_small_gemm(opd_detr* m, const f16_t* x, const f16_t* w, const float* bias, int bias_period, int M, int N, int K,
                          f16_t* out16, bool relu) {
    GemmK256Params p{}; p.dtype = m->dtype;
    p.x = x; p.w = w; p.bias = bias; p.out16 = out16; p.M = M; p.N = N; p.ldx = K; p.ldw = K; p.slices = 1;
    p.bias_period = bias_period; p.relu = relu ? 1 : 0;
    RCCHK(timed_begin(m, CLS_GEMM, 2.0 * M * (double)N * K));
    HIPCHK(opd_launch_gemm_k256(p, m->stream));
    RCCHK(timed_end(m));
    RCCHK(tap(m, "gemm_k256", out16, (size_t)M * N * 2));
    return OPD_OK;
}
static int run_small_gemm_ln(opd_detr* m, const f16_t* x, const f16_t* w, const float* bias, int M, int K, const float* res32,
                             const LNp& ln, float* y32, f16_t* y16) {
    GemmK256Params p{}; p.dtype = m->dtype;
    p.x = x; p.w = w; p.bias = bias; p.out32 = m->d_slab; p.M = M; p.N = 256; p.ldx = K; p.ldw = K; p.slices = K / 256;
    RCCHK(timed_begin(m, CLS_GEMM, 2.0 * M * 256.0 * K));
    HIPCHK(opd_launch_gemm_k256(p, m->stream));
    RCCHK(timed_end(m));
    RCCHK(timed_begin(m, CLS_OTHER, 0.0));
    HIPCHK(opd_launch_reduce_ln(m->d_slab, p.slices, (size_t)M * 256, res32, ln.g, ln.b, y32, y16, M, m->stream, m->dtype));
    RCCHK(timed_end(m));
    RCCHK(tap(m, "small_gemm_ln", y32, (size_t)M * 256 * 4));
    return OPD_OK;
}

static int run_attn(opd_detr* m, const f16_t* q, int ldq, const f16_t* k, int ldk, const f16_t* v, int ldv, f16_t* o, int ldo,
                    int B, int Lq, int Lk, const int32_t* key_valid = nullptr, int key_row = 0) {
    AttnParams p{}; p.dtype = m->dtype;
    p.key_valid = key_valid; p.key_row = key_row;
    p.q = q; p.k = k; p.v = v; p.o = o; p.B = B; p.heads = m->arch.heads; p.Lq = Lq; p.Lk = Lk;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.scale = 1.0f / sqrtf((float)(m->arch.d_model / m->arch.heads));
    RCCHK(timed_begin(m, CLS_ATTN, 4.0 * B * (double)m->arch.heads * Lq * Lk * 32));
    HIPCHK(opd_launch_attention(p, m->stream));
    RCCHK(timed_end(m));
    if (ldo == m->arch.d_model) RCCHK(tap(m, "attention", o, (size_t)B * Lq * ldo * 2));
    return OPD_OK;
}

#define MARK(i)                                                   \
    do {                                                          \
        if (m->profiling) HIPCHK(hipEventRecord(m->ev[i], m->stream)); \
        opd_dbg_skip_launch = (m->dbg_skip >> (i)) & 1;           \
    } while (0)

// Enqueues the whole forward on m->stream.  `pixels` must already be on the device.
// True when some frame of the batch does not fill the H x W canvas (a ragged batch: padding mask path).
static bool is_ragged(const int32_t* valid_hw, int B, int H, int W) {
    if (!valid_hw) return false;
    for (int b = 0; b < B; ++b)
        if (valid_hw[2 * b] != H || valid_hw[2 * b + 1] != W) return true;
    return false;
}

// `valid_hw` (host, nullable): [B][2] = (h, w) of each frame inside the H x W canvas.
static int enqueue_forward(opd_detr* m, const void* d_pixels, int pixel_format, int B, int H, int W, const int32_t* valid_hw = nullptr) {
    const Arch& a = m->arch;
    Dims d;
    compute_dims(B, H, W, &d);
    const int fh = d.sh[3], fw = d.sw[3];
    const bool ragged = is_ragged(valid_hw, B, H, W);
    Plan* plan = nullptr;
    const float* const* enc_bias_ptrs[16] = {};   // per encoder layer: device array of B per-frame fold pointers (ragged only)
    const float* const* kv_bias_ptrs = nullptr;
    const float* const* pos_ptrs = nullptr;       // per-frame position embeddings (ragged only)
    const int32_t *d_valid = nullptr, *d_keyv = nullptr;
    if (!ragged) {
        RCCHK(get_plan(m, fh, fw, fh, fw, &plan));
    } else {
        if (a.enc_layers > 16) return fail(OPD_EINVAL, "ragged batches: at most 16 encoder layers");
        m->h_valid_hw.assign(valid_hw, valid_hw + 2 * B);
        m->h_key_valid.resize((size_t)2 * B);
        m->h_bias_ptrs.assign((size_t)(a.enc_layers + 2) * B, nullptr);
        for (int b = 0; b < B; ++b) {
            const int vh = valid_hw[2 * b], vw = valid_hw[2 * b + 1];
            if (vh < 1 || vw < 1 || vh > H || vw > W) return fail(OPD_EINVAL, "valid_hw outside the frame canvas");
            const int vfh = valid_prefix(vh, H, fh), vfw = valid_prefix(vw, W, fw);
            if (vfh < 1 || vfw < 1) return fail(OPD_EINVAL, "frame too small: no valid feature-map position");
            m->h_key_valid[2 * b] = vfh; m->h_key_valid[2 * b + 1] = vfw;
            Plan* pb = nullptr;
            RCCHK(get_plan(m, fh, fw, vfh, vfw, &pb));
            if (b == 0) plan = pb;
            for (int i = 0; i < a.enc_layers; ++i) m->h_bias_ptrs[(size_t)i * B + b] = pb->rb_enc[i];
            m->h_bias_ptrs[(size_t)a.enc_layers * B + b] = pb->rb_kv;
            m->h_bias_ptrs[(size_t)(a.enc_layers + 1) * B + b] = pb->d_pos;
        }
        // (member vectors: they outlive the asynchronous copies; every entry point synchronises before it returns)
        HIPCHK(hipMemcpyAsync(m->d_valid_hw, m->h_valid_hw.data(), (size_t)2 * B * 4, hipMemcpyHostToDevice, m->stream));
        HIPCHK(hipMemcpyAsync(m->d_key_valid, m->h_key_valid.data(), (size_t)2 * B * 4, hipMemcpyHostToDevice, m->stream));
        HIPCHK(hipMemcpyAsync(m->d_bias_ptrs, m->h_bias_ptrs.data(), m->h_bias_ptrs.size() * sizeof(float*), hipMemcpyHostToDevice, m->stream));
        for (int i = 0; i < a.enc_layers; ++i) enc_bias_ptrs[i] = m->d_bias_ptrs + (size_t)i * B;
        kv_bias_ptrs = m->d_bias_ptrs + (size_t)a.enc_layers * B;
        pos_ptrs = m->d_bias_ptrs + (size_t)(a.enc_layers + 1) * B;
        d_valid = m->d_valid_hw;
        d_keyv = m->d_key_valid;
    }
    timed_reset(m);
    m->tap_next = 0;
    if (pixel_format == OPD_PIXELS_U8_BGR_HWC) RCCHK(tap(m, "pixels_u8", d_pixels, (size_t)B * H * W * 3));
    MARK(0);
    const int Hp = 2 * d.H1 + 6, Wp = 2 * d.W1 + 6;  // padded image seen by the stem: rows/cols 2*o + k, k = 0..7
    const bool prep_in_stem = m->fuse_prep && m->fuse_stem_pool && pixel_format == OPD_PIXELS_U8_BGR_HWC;
    if (!prep_in_stem) {
        RCCHK(timed_begin(m, CLS_OTHER, 0.0));
        if (pixel_format == OPD_PIXELS_U8_BGR_HWC)
            HIPCHK(opd_launch_preprocess_u8(reinterpret_cast<const uint8_t*>(d_pixels), m->d_x4, B, H, W, Hp, Wp, d_valid, m->stream, m->dtype));
        else
            HIPCHK(opd_launch_preprocess_f32(reinterpret_cast<const float*>(d_pixels), m->d_x4, B, H, W, Hp, Wp, d_valid, m->stream, m->dtype));
        RCCHK(timed_end(m));
    }
    if (prep_in_stem) {
        RCCHK(timed_begin(m, CLS_CONV, 2.0 * B * d.H1 * d.W1 * 64.0 * 147.0));
        HIPCHK(opd_launch_stem_pool_u8(reinterpret_cast<const uint8_t*>(d_pixels), d_valid, m->stem.w, m->stem.bias, m->d_pool, B, H, W, d.H1, d.W1,
                                       d.H2, d.W2, m->stream, m->dtype));
        RCCHK(timed_end(m));
        RCCHK(tap(m, "stem_pool_u8", m->d_pool, (size_t)B * d.H2 * d.W2 * 64 * 2));
    } else if (m->fuse_stem_pool) {
        RCCHK(timed_begin(m, CLS_CONV, 2.0 * B * d.H1 * d.W1 * 64.0 * 147.0));
        HIPCHK(opd_launch_stem_pool(m->d_x4, m->stem.w, m->stem.bias, m->d_pool, B, Hp, Wp, d.H1, d.W1, d.H2, d.W2, m->stream, m->dtype));
        RCCHK(timed_end(m));
    } else {
        ConvGemmParams p{}; p.dtype = m->dtype;
        p.x = m->d_x4; p.w = m->stem.w; p.bias = m->stem.bias; p.out = m->d_stem; p.zero16 = m->zero_bias;
        p.B = B; p.H = Hp; p.W = Wp; p.Cin = 256; p.OH = d.H1; p.OW = d.W1; p.N = 64; p.KH = 1; p.KW = 1; p.stride = 2; p.pad = 0;
        p.M = B * d.H1 * d.W1; p.K = 256; p.relu = 1; p.stem = 2;
        RCCHK(timed_begin(m, CLS_CONV, 2.0 * p.M * 64.0 * 147.0));
        HIPCHK(opd_launch_conv_gemm(p, m->stream));
        RCCHK(timed_end(m));
        RCCHK(timed_begin(m, CLS_OTHER, 0.0));
        HIPCHK(opd_launch_maxpool(m->d_stem, m->d_pool, B, d.H1, d.W1, 64, d.H2, d.W2, m->stream, m->dtype));
        RCCHK(timed_end(m));
    }
    MARK(1);
    // ---- trunk.  Stages 1-2 may run in SUB-BATCHES (cfg: trunk_subbatch frames at a time through both stages, then the next
    // frames): their block outputs are 274 / 137 MB at batch 8 — each one written by a fused tail and read back by the next as its
    // residual — and the Infinity Cache holds 256 MiB, so at full batch that read comes from HBM; with 4 frames a tensor is
    // 137 / 68 MB and the consumer finds it on the die.  Same kernels, same per-row arithmetic (tiles are cut from the flattened
    // row index either way), so results do not change; buffers: every tensor of sub-batch [b0, b0 + nb) lives at frame offset b0
    // of the buffer the full batch would use, the finished stage-2 outputs of earlier sub-batches sit below the regions later
    // ones touch (per-frame sizes shrink from stage to stage).
    struct TrunkState { int cur_id; int ch, cw; int z_id; };   // cur_id 0 = pool, 1 = t0, 2 = t1; z_id -1 / 0 = m0 / 1 = m1
    int tail_no = 0;   // consecutive fused tails walk the tiles in alternating directions (tail_rev)
    int rc_prev = -1, rc_prev2 = -1;   // blocks whose tails stored a1 instead of y (their successors rebuild the residual: BtailParams::rc)
    const f16_t* rc_xs = nullptr;
    auto run_blocks = [&](int s_begin, int s_end, int b0, int nb, TrunkState& st, int l_begin = 0, int l_end = 1 << 30) -> int {
        auto trunk = [&](int id, size_t per_frame) { return (id == 0 ? m->d_pool : id == 1 ? m->d_t0 : m->d_t1) + (size_t)b0 * per_frame; };
        auto mid = [&](int id, size_t per_frame) { return (id ? m->d_m1 : m->d_m0) + (size_t)b0 * per_frame; };
        for (int s = s_begin; s < s_end; ++s) {
            for (int l = l_begin; l < a.depths[s] && l < l_end; ++l) {   // (a block range only makes sense with s_end == s_begin + 1)
                const int bi = m->stage_first[s] + l;
                const Block& b = m->blocks[bi];
                const Block* nbk = bi + 1 < (int)m->blocks.size() ? &m->blocks[bi + 1] : nullptr;
                const int ch = st.ch, cw = st.cw;
                const int oh = (b.c1.stride == 2) ? down2(ch) : ch, ow = (b.c1.stride == 2) ? down2(cw) : cw;
                const int C1 = b.c1.Cin, C2 = b.c2.Cout;
                const f16_t* cur = trunk(st.cur_id, (size_t)ch * cw * b.c0.Cin);
                const int out_id = st.cur_id == 1 ? 2 : 1;
                f16_t* out = trunk(out_id, (size_t)oh * ow * C2);
                const f16_t* res = cur;
                // first block of stage 1 (64 -> 256 channels, stride 1): the shortcut runs inside the fused tail (kernels_btail.hip, SC)
                const bool sc_in_tail = b.has_sc && m->fuse_shortcut && m->fuse_btail && b.bias2sc && b.sc.KH == 1 && b.sc.stride == 1 &&
                                        b.sc.Cin == 64 && b.c1.Cin == 64 && b.c1.stride == 1 && b.c2.Cout == 256 && nbk && nbk->c0.wp &&
                                        nbk->c0.Cin == 256 && nbk->c0.Cout == 64;
                // first block of stages 3 / 4: the shortcut is extra K of the 1x1 expand (conv_gemm_dma_kernel, DUAL)
                bool tail_kernel = m->fuse_btail && b.c1.KH == 3 && b.c1.Cout == C1 && b.c2.Cin == C1 && b.c2.Cout == 4 * C1 && b.c2.wp &&
                                   opd_btail_supported(C1, 0);
                // first block of stage 2: 3x3 + dual-source expand (+ the next reduce on its own) beats shortcut launch + fused tail
                // (stage 2: 0.674 -> 0.657 ms; OPD_DUAL_OVER_TAIL=0 restores the tail)
                if (m->dual_over_tail && tail_kernel && b.has_sc && !sc_in_tail && b.w2sc) tail_kernel = false;
                // Stage 3: the eight-wave tail holds one 160-KiB workgroup per CU, so a launch costs whole ROUNDS of ~70 us whatever they hold.
                // Rounds that are only partly filled because the batch does not divide into them are dealt with by the frame split below; what
                // remains is the case of too few tiles for even one round (small frames / batches: the three launches win there).  The choice is
                // made from the handle's configuration (max_batch and the frame size), never from the batch at hand: the two paths differ in
                // the last bit (kernels_btail3.hip), and a frame's low-order bits must not depend on the batch it travels in.  Handles that
                // keep several batches in flight (OPD_FLAG_MULTI_STREAM) always take the fused tail: other streams fill its idle CUs.
                if (C1 == 256 && tail_kernel) {
                    const long long tiles = ((long long)m->cfg.max_batch * oh * ow + 127) / 128;
                    const bool pays = tiles * 10 >= (long long)m->num_cus * 6;
                    if (m->tail3 == 0 || (m->tail3 == 1 && !pays && !(m->cfg.flags & OPD_FLAG_MULTI_STREAM))) tail_kernel = false;
                }
                const bool sc_in_expand = b.has_sc && m->fuse_shortcut && b.w2sc && !sc_in_tail && !tail_kernel;
                if (b.has_sc && !sc_in_tail && !sc_in_expand) {
                    f16_t* scb = m->d_sc + (size_t)b0 * oh * ow * C2;
                    RCCHK(run_conv(m, b.sc, cur, nb, ch, cw, oh, ow, scb, false, nullptr));
                    res = scb;
                }
                int x1_id = st.z_id;
                const f16_t* x1 = nullptr;
                if (x1_id >= 0) {
                    x1 = mid(x1_id, (size_t)ch * cw * C1);
                } else {
                    x1_id = 0;
                    f16_t* c0out = mid(0, (size_t)ch * cw * b.c0.Cout);
                    RCCHK(run_conv(m, b.c0, cur, nb, ch, cw, ch, cw, c0out, true, nullptr, (l == 0 && s > 0) ? s - 1 : s));
                    x1 = c0out;
                }
                st.z_id = -1;
                const bool tail_ok = tail_kernel && (size_t)nb * ch * cw * C1 * 2 < 0x7ff00000ull;
                if (tail_ok) {
                    int C3 = 0;
                    if (nbk && nbk->c0.wp && nbk->c0.Cin == 4 * C1 && opd_btail_supported(C1, nbk->c0.Cout)) C3 = nbk->c0.Cout;
                    // (a sub-batch pipeline ends with stage 2: its last tail cannot hand z to stage 3 anyway — 256 channels)
                    BtailParams p{}; p.dtype = m->dtype;
                    p.x1 = x1; p.w1 = b.c1.w; p.b1 = b.c1.bias; p.w2p = C1 == 256 ? b.c2.wp : b.c2.w; p.b2 = b.c2.bias; p.res = res; p.y = out;   // (K-permuted copies: stage-3 kernel only)
                    if (sc_in_tail) { p.res = nullptr; p.xs = cur; p.wsc = b.sc.w; p.b2 = b.bias2sc; }
                    f16_t* z = mid(1 - x1_id, (size_t)oh * ow * C3);
                    if (C3) { p.w3p = C1 == 256 ? nbk->c0.wp : nbk->c0.w; p.b3 = nbk->c0.bias; p.z = z; }
                    // Stage 1, residual rebuild (BtailParams::rc): block 0 stores its a1 (64 channels) instead of its output (256), block 1 rebuilds
                    // that output chunk by chunk from a1 and the pooled map (the shortcut's input) as its residual.  Bit-identical results
                    // (tests/test_kernels_gpu.py, OPD_TAIL_RC=0/1 end to end); 344 MB less HBM traffic per forward at batch 8.  The a1 tensor lives in
                    // the shortcut buffer, which a fused shortcut leaves unused.
                    // tail_rc == 2 (NOT the default: measured slower, opd_model.h) goes one block further: block 1 stores ITS a1 as well and block 2 (the last of stage 1, C3 = 128)
                    // rebuilds both outputs (btail_rc2_kernel): stage 1 then moves 64-channel tensors only, plus the quarter of its last output
                    // that the next stage's shortcut reads -- another 276 MB less per forward.
                    const bool rc_ok = m->tail_rc && s == 0 && C1 == 64 && a.depths[0] >= 2 && !m->taps;
                    const bool rc2_ok = rc_ok && m->tail_rc >= 2 && a.depths[0] == 3;
                    f16_t* const a1_keep0 = m->d_sc + (size_t)b0 * oh * ow * 64;                                          // block 0's a1
                    f16_t* const a1_keep1 = m->d_sc + (size_t)m->cfg.max_batch * oh * ow * 64 + (size_t)b0 * oh * ow * 64;   // block 1's a1 (behind the whole batch's block-0 tensor)
                    if (rc_ok && C3 == 64 && sc_in_tail && l == 0) { p.y = nullptr; p.a1_out = a1_keep0; rc_prev = bi; rc_xs = cur; }
                    else if (rc_ok && C3 == 64 && l == 1 && rc_prev == bi - 1 && !b.has_sc) {
                        const Block& pb = m->blocks[bi - 1];
                        p.res = nullptr; p.rc = 1; p.rc_a1[0] = a1_keep0; p.rc_xs = rc_xs; p.rc_w2[0] = pb.c2.w; p.rc_wsc = pb.sc.w; p.rc_b[0] = pb.bias2sc;
                        if (rc2_ok && nbk && !nbk->has_sc && nbk->c1.Cin == 64) { p.y = nullptr; p.a1_out = a1_keep1; rc_prev2 = bi; }
                    } else if (rc2_ok && C3 == 128 && l == 2 && rc_prev2 == bi - 1 && rc_prev == bi - 2 && !b.has_sc) {
                        const Block &p1 = m->blocks[bi - 1], &p0 = m->blocks[bi - 2];
                        p.res = nullptr; p.rc = 2; p.rc_xs = rc_xs; p.rc_wsc = p0.sc.w;
                        p.rc_a1[0] = a1_keep1; p.rc_w2[0] = p1.c2.w; p.rc_b[0] = p1.c2.bias;
                        p.rc_a1[1] = a1_keep0; p.rc_w2[1] = p0.c2.w; p.rc_b[1] = p0.bias2sc;
                    }
                    // The last block of stage 1 hands the next stage its reduce output z (fused above); the block output itself is then read
                    // by that stage's stride-2 shortcut only, i.e. at even (oh, ow): the other three quarters of its 274 MB are not stored.
                    if (m->y_stride2 && C3 && b.c1.stride == 1 && l + 1 == a.depths[s] && nbk && nbk->has_sc && nbk->sc.stride == 2 && nbk->sc.KH == 1 &&
                        nbk->c1.stride == 2 && !m->taps)
                        p.y_stride2 = 1;
                    p.B = nb; p.H = ch; p.W = cw; p.OH = oh; p.OW = ow; p.stride = b.c1.stride; p.M = nb * oh * ow; p.C1 = C1; p.C3 = C3;
                    p.rev = m->tail_rev ? (tail_no++ & 1) : 0;
                    p.dbg = m->dbg_btail;
                    p.nw = (C1 == 256 || p.rc == 2) ? 0 : m->tail_nw;
                    RCCHK(timed_begin(m, CLS_CONV, 2.0 * p.M * ((double)C1 * 9 * C1 + 4.0 * C1 * C1 + 4.0 * C1 * C3 + (sc_in_tail ? 64.0 * 256 : 0.0))));
                    HIPCHK(opd_launch_btail(p, m->stream));
                    RCCHK(timed_end(m));
                    if (p.y) RCCHK(tap(m, "btail_y", out, (size_t)p.M * 4 * C1 * 2));
                    if (C3) RCCHK(tap(m, "btail_z", z, (size_t)p.M * C3 * 2));
                    if (C3) st.z_id = 1 - x1_id;
                } else {
                    f16_t* a1 = mid(1 - x1_id, (size_t)oh * ow * C1);
                    RCCHK(run_conv(m, b.c1, x1, nb, ch, cw, oh, ow, a1, true, nullptr, s));
                    if (sc_in_expand) {
                        ConvGemmParams p{}; p.dtype = m->dtype;
                        p.x = a1; p.w = b.w2sc; p.bias = b.bias2sc; p.out = out; p.zero16 = m->zero_bias;
                        p.B = nb; p.H = oh; p.W = ow; p.Cin = C1; p.OH = oh; p.OW = ow; p.N = C2; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0;
                        p.M = nb * oh * ow; p.K1 = C1; p.K = C1 + b.sc.Cin; p.relu = 1;
                        p.x2 = cur; p.H2 = ch; p.W2 = cw; p.Cin2 = b.sc.Cin; p.stride2 = b.sc.stride; p.dbg = m->dbg_gemm; p.wprefetch = m->wprefetch & 1;
                        RCCHK(timed_begin(m, CLS_CONV, 2.0 * p.M * (double)C2 * p.K));
                        HIPCHK(opd_launch_conv_gemm(p, m->stream));
                        RCCHK(timed_end(m));
                        RCCHK(tap(m, "dual_expand", out, (size_t)p.M * C2 * 2));
                    } else {
                        RCCHK(run_conv(m, b.c2, a1, nb, oh, ow, oh, ow, out, true, res));
                    }
                }
                st.cur_id = out_id; st.ch = oh; st.cw = ow;
            }
            if (b0 + nb == B && l_end >= a.depths[s]) MARK(2 + s);
        }
        return OPD_OK;
    };
    TrunkState st{0, d.H2, d.W2, -1};
    {
        const int sub = (m->trunk_subbatch > 0 && m->trunk_subbatch < B && B % m->trunk_subbatch == 0) ? m->trunk_subbatch : B;
        TrunkState done = st;
        for (int b0 = 0; b0 < B; b0 += sub) {
            TrunkState t = st;
            RCCHK(run_blocks(0, 2, b0, sub, t));
            done = t;
        }
        st = done;
        st.z_id = -1;   // (stage 2's last tail has no fused reduce)
        // Stage 3.  Frame split: with the fused tails a launch of T tiles costs ceil(T / CUs) rounds, and batch 8 at 800x1333 is 263 tiles on
        // 256 CUs.  Frames are independent, so blocks 1 .. n-1 of the stage (all tensors there have one per-frame size, the buffers of the
        // two chains never overlap) run as TWO chains on two streams -- frames [0, nbA) = whole rounds, the rest on `stream2` -- whose
        // workgroups the hardware packs onto whatever CU is free: 5 tails of 263 workgroup-lives take ~5.3 rounds instead of 10.  Per-row
        // arithmetic does not depend on the tiling, so this is invisible in the results (any batch, any split).  Captured into the graph
        // as a fork / join; not under profiling (event pairs on one stream) or diagnostic taps, and not for handles that keep several batches
        // in flight (there other handles' kernels fill the idle CUs; measured on one box, 1000 steps x 2: three streams 2841 frames/s with
        // one launch per tail, 2783 with the split, 2793 unfused; one stream 2045 / 2112 / 2090).
        int nbA = B;
        {
            const Block& b1 = m->blocks[m->stage_first[2] + (a.depths[2] > 1 ? 1 : 0)];
            const int oh3 = down2(st.ch), ow3 = down2(st.cw);
            const bool fused3 = m->fuse_btail && a.depths[2] > 2 && b1.c1.Cin == 256 && opd_btail_supported(256, 0) && b1.c2.wp && m->profiling != 1 && !m->taps &&
                                m->stream2 && m->tail3_split && !(m->cfg.flags & OPD_FLAG_MULTI_STREAM) && B >= 2 && [&] {   // the policy of run_blocks, evaluated for this stage
                                    const long long tiles = ((long long)m->cfg.max_batch * oh3 * ow3 + 127) / 128;
                                    return m->tail3 == 2 || (m->tail3 == 1 && (tiles * 10 >= (long long)m->num_cus * 6 || (m->cfg.flags & OPD_FLAG_MULTI_STREAM)));
                                }();
            if (fused3) {
                auto tiles_of = [&](int frames) { return ((long long)frames * oh3 * ow3 + 127) / 128; };
                const long long total = tiles_of(B), cus = m->num_cus, last = total % cus;
                if (total > cus && last > 0 && last < cus / 2) {
                    const long long whole = (total / cus) * cus;
                    int bA = B - 1;
                    while (bA > 1 && tiles_of(bA) > whole) --bA;
                    if (tiles_of(bA) <= whole) nbA = bA;
                }
            }
        }
        RCCHK(run_blocks(2, 3, 0, B, st, 0, 1));   // first block (stride 2, shortcut): whole batch
        if (nbA < B) {
            TrunkState ta = st, tb = st;
            HIPCHK(hipEventRecord(m->ev_fork, m->stream));
            HIPCHK(hipStreamWaitEvent(m->stream2, m->ev_fork, 0));
            // whatever happens in either chain, `stream2` is rejoined before this function returns (an unjoined fork would leak into
            // the next forward's events, or leave a capture with a dangling branch) and m->stream is the main stream again
            hipStream_t main_stream = m->stream;
            const int rc_a = run_blocks(2, 3, 0, nbA, ta, 1);
            m->stream = m->stream2;
            const int rc_b = rc_a == OPD_OK ? run_blocks(2, 3, nbA, B - nbA, tb, 1) : OPD_OK;
            m->stream = main_stream;
            const hipError_t ej = hipEventRecord(m->ev_join, m->stream2);
            const hipError_t ew = ej == hipSuccess ? hipStreamWaitEvent(m->stream, m->ev_join, 0) : ej;
            RCCHK(rc_a);
            RCCHK(rc_b);
            HIPCHK(ew);
            MARK(4);   // (stage 3 ends at the join, not where the second chain's own mark fell)
            st = ta;
        } else {
            RCCHK(run_blocks(2, 3, 0, B, st, 1));
        }
        RCCHK(run_blocks(3, 4, 0, B, st));
    }
    const f16_t* cur = st.cur_id == 1 ? m->d_t0 : m->d_t1;
    const int ch = st.ch, cw = st.cw;
    // ---- input projection -> encoder ------------------------------------------------------------------------------
    const int hw = ch * cw, M = B * hw, D = a.d_model, F = a.ffn;
    // pos_shadow: whoever writes x (input projection, each layer's last LayerNorm) also writes fp16(x + pos); the fused QKV projection
    // reads that for its q / k column tiles and x for its v tiles, with a plain bias vector -- instead of x everywhere plus a [hw][768]
    // fp32 table W.pos + b added per output tile (two divisions and 16 dependent table loads per lane in front of the first MFMA:
    // 14.2 us per launch against 9.4 for the same GEMM with a bias vector; decoder K/V 44.7 against 24-27)
    const bool shadow = m->pos_shadow && D == 256 && m->enc[0].bqkv && m->bkv_all;
    const PosShadow psh{plan->d_pos, pos_ptrs, hw, m->d_xp16};
    const PosShadow* ps = shadow ? &psh : nullptr;
    // Deep-K row-owner launches (kernels_rowln.hip::gemm_ln256_ring_kernel) for the two K = 2048 -> 256 linears of the encoder side:
    // the input projection (no LayerNorm) and every layer's FFN-2 (+ residual + LayerNorm); both also write the position shadow
    auto run_deep = [&](const f16_t* x, const Lin* lin, const f16_t* w, const float* bias, int K, const float* res32, const LNp* ln, int cls) -> int {
        GemmLnParams gp{}; gp.dtype = m->dtype;
        gp.x = x; gp.w = w; gp.bias = bias; gp.res32 = res32; gp.gamma = ln ? ln->g : nullptr; gp.beta = ln ? ln->b : nullptr;
        gp.y32 = m->d_x32; gp.y16 = m->d_x16; gp.M = M; gp.K = K; gp.deep_k = 1;
        if (ps) { gp.pos = ps->pos; gp.pos_ptrs = ps->pos_ptrs; gp.pos_period = ps->period; gp.yp16 = ps->yp16; }
        RCCHK(timed_begin(m, cls, 2.0 * M * (double)D * K));
        HIPCHK(opd_launch_gemm_ln(gp, m->stream));
        RCCHK(timed_end(m));
        RCCHK(tap(m, ln ? "fc2_ln_ring" : "input_proj_ring", m->d_x32, (size_t)M * D * 4));
        (void)lin;
        return OPD_OK;
    };
    // Small handles (round 5): the row-owner launches of the encoder side own 48 / 64 rows per workgroup and stream a whole weight matrix through
    // each -- at max_batch = 1 that is 22 workgroups walking 2.2 MB apiece (42 us per layer, 27 us for the input projection).  Where the handle's
    // CONFIGURATION bounds the token count below ~1400 the same linears run as tiled GEMMs with the reduction split eight ways over workgroups
    // and the fixed-order reduce + LayerNorm kernel (the round-1 path): encoder stage 0.41 -> 0.32 ms at batch 1.  Never per batch.
    const bool small_enc = m->small_enc && (size_t)m->cfg.max_batch * m->stage_px[3] <= 1400;
    const int enc_splits = small_enc ? 8 : 4;
    const bool deep_ok = m->deep_fc2 && D == 256 && !small_enc;
    if (deep_ok && m->proj.K % 64 == 0 && (size_t)M * m->proj.K * 2 < 0x7fffff00ull)
        RCCHK(run_deep(cur, nullptr, m->proj.w, m->proj.bias, m->proj.K, nullptr, nullptr, CLS_CONV));
    else
        RCCHK(run_gemm_splitk_ln(m, cur, m->proj.w, m->proj.bias, M, D, m->proj.K, (m->proj.K / 64) % enc_splits == 0 ? enc_splits : 4, nullptr, nullptr, m->d_x32, m->d_x16, CLS_CONV, ps));
    bool qkv_done = false, kv_done = false;
    const int Q = a.queries, Md = B * Q, NKV = a.dec_layers * 2 * D;
    for (int i = 0; i < a.enc_layers; ++i) {
        const EncLayer& L = m->enc[i];
        if (qkv_done) {   // written by the previous layer's FFN launch (its tail projection)
        } else if (shadow)
            RCCHK(run_gemm(m, m->d_xp16, L.wqkv, L.bqkv, 0, M, 3 * D, D, m->d_qkv16, false, false, nullptr, nullptr, 0, 0, m->d_x16, 3 * D, 2 * D));
        else
            RCCHK(run_gemm(m, m->d_x16, L.wqkv, plan->rb_enc[i], hw, M, 3 * D, D, m->d_qkv16, false, false, nullptr, enc_bias_ptrs[i], 3 * D, 2 * D));   // (pos enters q and k only)
        RCCHK(run_attn(m, m->d_qkv16, 3 * D, m->d_qkv16 + D, 3 * D, m->d_qkv16 + 2 * D, 3 * D, m->d_attn16, D, B, hw, hw, d_keyv, cw));
        const bool ffn_fused = m->fused_enc_ffn && m->fuse_gemm_ln && L.ffn_pack && D == 256 && !small_enc;
        const bool front = ffn_fused && L.front && m->enc_front;   // the output projection + LayerNorm run inside the FFN launch
        if (front) {
        } else if (m->fuse_gemm_ln && D == 256) {
            RCCHK(run_gemm_ln(m, m->d_attn16, L.o.w, L.o.b, M, D, m->d_x32, L.ln1, m->d_x32, m->d_x16));
        } else {
            RCCHK(run_gemm(m, m->d_attn16, L.o.w, L.o.b, 0, M, D, D, m->d_y32, true, false, m->d_x32));
            RCCHK(timed_begin(m, CLS_OTHER, 0.0));
            HIPCHK(opd_launch_layernorm(m->d_y32, L.ln1.g, L.ln1.b, m->d_x32, m->d_x16, M, m->stream, m->dtype));
            RCCHK(timed_end(m));
        }
        if (ffn_fused) {
            // the whole FFN block as ONE row-owner launch: the [M][F] hidden tensor never leaves LDS (kernels_rowln.hip::enc_ffn_kernel)
            EncFfnParams fp{}; fp.dtype = m->dtype; fp.wprefetch = (m->wprefetch >> 1) & 1;
            fp.x = m->d_x16; fp.wpack = L.ffn_pack; fp.b2 = L.fc2.b; fp.res32 = m->d_x32; fp.gamma = L.ln2.g; fp.beta = L.ln2.b;
            fp.y32 = m->d_x32; fp.y16 = m->d_x16; fp.M = M; fp.F = F; fp.pack_tail = L.tail; fp.pack_front = L.front;
            if (front) { fp.attn = m->d_attn16; fp.bo = L.o.b; fp.gamma1 = L.ln1.g; fp.beta1 = L.ln1.b; }
            if (ps) { fp.pos = ps->pos; fp.pos_ptrs = ps->pos_ptrs; fp.pos_period = ps->period; fp.yp16 = ps->yp16; }
            const bool last = i + 1 == a.enc_layers;
            qkv_done = false;
            // the tail projection: what consumes this block's output (only on the position-shadow path: x + pos with plain bias vectors)
            if (ps && m->enc_tail && L.tail && L.tail_ld == (last ? NKV : 3 * D) && (last ? (size_t)M * NKV * 2 < (1ull << 32) : true)) {
                fp.tail = L.tail; fp.tail_pos = L.tail_pos; fp.tail_ld = L.tail_ld;
                fp.tail_out = last ? m->d_memkv16 : m->d_qkv16;
                for (int t = 0; t < L.tail; ++t) fp.tail_col[t] = L.tail_col[t];
                (last ? kv_done : qkv_done) = true;
            }
            RCCHK(timed_begin(m, CLS_GEMM, 4.0 * M * (double)D * F + 2.0 * M * (double)D * 256 * (fp.tail + (front ? 1 : 0))));
            HIPCHK(opd_launch_enc_ffn(fp, m->stream));
            RCCHK(timed_end(m));
            RCCHK(tap(m, "enc_ffn", m->d_x32, (size_t)M * D * 4));
        } else {
            qkv_done = false;
            RCCHK(run_gemm(m, m->d_x16, L.fc1.w, L.fc1.b, 0, M, F, D, m->d_ffn16, false, true, nullptr));
            // fc2 + residual + LayerNorm (+ the position shadow) as ONE row-owner launch of the three-stage ring kernel: 36.9 us in the
            // forward against 19.4 + 12.2 us for split-K slabs + reduce, but 514 MB less HBM traffic per forward and half the CUs left to
            // other batches (+1.3 % with three streams, -5 us per layer for a lone stream; profiles/NOTES.md "Deep-K row owners")
            if (deep_ok && F % 64 == 0 && (size_t)M * F * 2 < 0x7fffff00ull) {
                RCCHK(run_deep(m->d_ffn16, nullptr, L.fc2.w, L.fc2.b, F, m->d_x32, &L.ln2, CLS_GEMM));
            } else {
                RCCHK(run_gemm_splitk_ln(m, m->d_ffn16, L.fc2.w, L.fc2.b, M, D, F, (F / 64) % enc_splits == 0 ? enc_splits : 4, m->d_x32, &L.ln2, m->d_x32, m->d_x16, CLS_GEMM, ps));
            }
        }
    }
    MARK(6);
    // ---- decoder -----------------------------------------------------------------------------------------------
    if (kv_done) {   // written by the last encoder layer's FFN launch
    } else if (shadow)
        RCCHK(run_gemm(m, m->d_xp16, m->wkv_all, m->bkv_all, 0, M, NKV, D, m->d_memkv16, false, false, nullptr, nullptr, 0, 0, m->d_x16, 2 * D, D));
    else
        RCCHK(run_gemm(m, m->d_x16, m->wkv_all, plan->rb_kv, hw, M, NKV, D, m->d_memkv16, false, false, nullptr, kv_bias_ptrs, 2 * D, D));   // (per layer [k | v]: pos enters k only)
    const bool dec0 = m->fuse_dec0 && m->dec0_h && D == 256;
    // The fused decoder (kernels_dec.hip): five launches per layer on split fp16 operands; layer 0 starts at its cross-attention (its
    // self-attention block and its queries are constants of the weights).  Taken when the architecture fits the kernels' fixed shapes.
    const bool fused_dec = m->fused_dec && dec0 && m->qc0 && a.heads == 8 && Q <= 128 && (Q & 3) == 0 && F % OPD_DEC_FFN_CHUNK == 0 && F / OPD_DEC_FFN_CHUNK <= 16 && m->dec_splits <= 6 && D == 256 && m->dec[0].wqkv_f &&
                           (size_t)M * NKV * 2 < (1ull << 32);
    const float* dec_final_h = m->d_h32;   // the state the heads read (fused: before the last FFN, whose partial sums travel with it)
    if (fused_dec) {
        const int S = m->dec_splits, nchunk = F / OPD_DEC_FFN_CHUNK;
        float* hbuf[2] = {m->d_h32, m->d_yd32};
        int cur = 0;   // hbuf[cur] holds the layer's state from its self-attention block on
        for (int i = 0; i < a.dec_layers && i < m->dbg_dec_layers; ++i) {
            const DecLayer& L = m->dec[i];
            f16_t* qd = m->d_qd16 + (size_t)i * m->cfg.max_batch * Q * D;   // (per-layer regions of max_batch frames: layer 0's constants stay put)
            if (i > 0) {
                const DecLayer& P = m->dec[i - 1];
                DecQkvParams qp{};
                qp.h_in = hbuf[cur]; qp.partials = m->d_ffn_part; qp.nsplit = nchunk; qp.b2 = P.fc2.b; qp.ln_g = P.ln3.g; qp.ln_b = P.ln3.b;
                qp.h_out = hbuf[cur ^ 1]; qp.w = L.wqkv_f; qp.bias = L.rb_self;
                qp.q16 = m->d_dq16; qp.k16 = m->d_dk16; qp.vT = m->d_dvT; qp.M = Md; qp.Q = Q;
                RCCHK(timed_begin(m, CLS_GEMM, 2.0 * Md * 768.0 * D));
                HIPCHK(opd_launch_dec_qkv(qp, m->stream));
                RCCHK(timed_end(m));
                cur ^= 1;
                RCCHK(tap(m, "dec_qkv_h", hbuf[cur], (size_t)Md * D * 4));
                DecSelfParams sp{};
                sp.q16 = m->d_dq16; sp.k16 = m->d_dk16; sp.vT = m->d_dvT; sp.h = hbuf[cur]; sp.wo = L.so_f; sp.bo = L.so.b;
                sp.ln_g = L.ln1.g; sp.ln_b = L.ln1.b; sp.wq = L.wqc_f; sp.rbq = L.rb_q; sp.qc16 = qd; sp.qc_bf16 = m->dtype == OPD_DT_BF16; sp.B = B; sp.Q = Q;
                sp.scale = 1.0f / sqrtf((float)(D / a.heads));
                RCCHK(timed_begin(m, CLS_GEMM, 4.0 * Md * (double)D * D + 4.0 * B * (double)a.heads * Q * Q * 32));
                HIPCHK(opd_launch_dec_self(sp, m->stream));
                RCCHK(timed_end(m));
                RCCHK(tap(m, "dec_self_h", hbuf[cur], (size_t)Md * D * 4));
            }
            {   // cross-attention over S key ranges: unnormalised partials
                AttnParams ap{}; ap.dtype = m->dtype;
                ap.q = qd; ap.k = m->d_memkv16 + (size_t)i * 2 * D; ap.v = m->d_memkv16 + (size_t)i * 2 * D + D; ap.o = nullptr;
                ap.B = B; ap.heads = a.heads; ap.Lq = Q; ap.Lk = hw; ap.ldq = D; ap.ldk = NKV; ap.ldv = NKV; ap.ldo = D;
                ap.scale = 1.0f / sqrtf((float)(D / a.heads)); ap.key_valid = d_keyv; ap.key_row = cw;
                ap.splits = S; ap.part_o = m->d_part_o; ap.part_ml = m->d_part_ml;
                RCCHK(timed_begin(m, CLS_ATTN, 4.0 * B * (double)a.heads * Q * hw * 32));
                HIPCHK(opd_launch_attention(ap, m->stream));
                RCCHK(timed_end(m));
                RCCHK(tap(m, "dec_cross_part", m->d_part_o, (size_t)S * Md * D * 4));
            }
            {
                DecCrossOutParams cp{};
                cp.part_o = m->d_part_o; cp.part_ml = m->d_part_ml; cp.splits = S;
                cp.res = i == 0 ? m->dec0_h : hbuf[cur]; cp.res_period = i == 0 ? 1 : 0; cp.h = hbuf[cur];
                cp.wo = L.co_f; cp.bo = L.co.b; cp.ln_g = L.ln2.g; cp.ln_b = L.ln2.b; cp.M = Md;
                RCCHK(timed_begin(m, CLS_GEMM, 2.0 * Md * (double)D * D));
                HIPCHK(opd_launch_dec_cross_out(cp, m->stream));
                RCCHK(timed_end(m));
                RCCHK(tap(m, "dec_cross_h", hbuf[cur], (size_t)Md * D * 4));
            }
            {
                DecFfnParams fp{};
                fp.h = hbuf[cur]; fp.w1 = L.fc1_f; fp.b1 = L.fc1.b; fp.w2 = L.fc2_f;
                fp.partials = m->d_ffn_part; fp.M = Md; fp.F = F;
                RCCHK(timed_begin(m, CLS_GEMM, 4.0 * Md * (double)D * F));
                HIPCHK(opd_launch_dec_ffn(fp, m->stream));
                RCCHK(timed_end(m));
                RCCHK(tap(m, "dec_ffn_part", m->d_ffn_part, (size_t)nchunk * Md * D * 4));
            }
        }
        dec_final_h = hbuf[cur];
    } else {
    if (dec0) {
        RCCHK(timed_begin(m, CLS_OTHER, 0.0));
        HIPCHK(opd_launch_broadcast_rows(m->dec0_h, m->d_h32, m->d_h16, Md, m->stream, m->dtype));
        RCCHK(timed_end(m));
        RCCHK(tap(m, "dec0_broadcast", m->d_h32, (size_t)Md * D * 4));
    } else {
        HIPCHK(hipMemsetAsync(m->d_h32, 0, (size_t)Md * D * 4, m->stream));
        HIPCHK(hipMemsetAsync(m->d_h16, 0, (size_t)Md * D * 2, m->stream));
    }
    for (int i = 0; i < a.dec_layers && i < m->dbg_dec_layers; ++i) {
        const DecLayer& L = m->dec[i];
        const bool small = m->small_m_gemm && D == 256 && F % 256 == 0 && F / 256 <= 8;
        if (!(dec0 && i == 0)) {   // (layer 0's self-attention block is the broadcast above)
        if (small) RCCHK(run_small_gemm(m, m->d_h16, L.wqkv, L.rb_self, Q, Md, 3 * D, D, m->d_qkvd16, false));
        else RCCHK(run_gemm(m, m->d_h16, L.wqkv, L.rb_self, Q, Md, 3 * D, D, m->d_qkvd16, false, false, nullptr));
        RCCHK(run_attn(m, m->d_qkvd16, 3 * D, m->d_qkvd16 + D, 3 * D, m->d_qkvd16 + 2 * D, 3 * D, m->d_attnd16, D, B, Q, Q));
        if (m->fuse_gemm_ln && D == 256)
            RCCHK(run_gemm_ln(m, m->d_attnd16, L.so.w, L.so.b, Md, D, m->d_h32, L.ln1, m->d_h32, m->d_h16));
        else
            RCCHK(run_gemm_splitk_ln(m, m->d_attnd16, L.so.w, L.so.b, Md, D, D, 4, m->d_h32, &L.ln1, m->d_h32, m->d_h16, CLS_GEMM));
        }
        f16_t* qd = m->d_qd16 + (size_t)i * m->cfg.max_batch * Q * D;
        if (small) RCCHK(run_small_gemm(m, m->d_h16, L.wq_c, L.rb_q, Q, Md, D, D, qd, false));
        else RCCHK(run_gemm(m, m->d_h16, L.wq_c, L.rb_q, Q, Md, D, D, qd, false, false, nullptr));
        RCCHK(run_attn(m, qd, D, m->d_memkv16 + (size_t)i * 2 * D, NKV, m->d_memkv16 + (size_t)i * 2 * D + D, NKV,
                       m->d_attnd16, D, B, Q, hw, d_keyv, cw));
        if (m->fuse_gemm_ln && D == 256)
            RCCHK(run_gemm_ln(m, m->d_attnd16, L.co.w, L.co.b, Md, D, m->d_h32, L.ln2, m->d_h32, m->d_h16));
        else
            RCCHK(run_gemm_splitk_ln(m, m->d_attnd16, L.co.w, L.co.b, Md, D, D, 4, m->d_h32, &L.ln2, m->d_h32, m->d_h16, CLS_GEMM));
        if (small) {
            RCCHK(run_small_gemm(m, m->d_h16, L.fc1.w, L.fc1.b, 0, Md, F, D, m->d_ffnd16, true));
            RCCHK(run_small_gemm_ln(m, m->d_ffnd16, L.fc2.w, L.fc2.b, Md, F, m->d_h32, L.ln3, m->d_h32, m->d_h16));
        } else {
            RCCHK(run_gemm(m, m->d_h16, L.fc1.w, L.fc1.b, 0, Md, F, D, m->d_ffnd16, false, true, nullptr));
            RCCHK(run_gemm_splitk_ln(m, m->d_ffnd16, L.fc2.w, L.fc2.b, Md, D, F, 8, m->d_h32, &L.ln3, m->d_h32, m->d_h16, CLS_GEMM));
        }
    }
    }
    HeadParams hp{};
    if (fused_dec) {   // the last layer's FFN sum + LN3 and the final LayerNorm run inside the heads kernel
        const DecLayer& P = m->dec[a.dec_layers - 1];
        hp.hs = dec_final_h; hp.partials = m->d_ffn_part; hp.nsplit = F / OPD_DEC_FFN_CHUNK; hp.ffn_b2 = P.fc2.b; hp.ln3_gamma = P.ln3.g; hp.ln3_beta = P.ln3.b;
        hp.ln_gamma = m->dec_ln.g; hp.ln_beta = m->dec_ln.b;
    } else if (m->fuse_gemm_ln) {   // the final LayerNorm runs inside the heads kernel
        hp.hs = m->d_h32; hp.ln_gamma = m->dec_ln.g; hp.ln_beta = m->dec_ln.b;
    } else {
        RCCHK(timed_begin(m, CLS_OTHER, 0.0));
        HIPCHK(opd_launch_layernorm(m->d_h32, m->dec_ln.g, m->dec_ln.b, m->d_hs32, nullptr, Md, m->stream, m->dtype));
        RCCHK(timed_end(m));
        hp.hs = m->d_hs32;
    } hp.wc = m->wc; hp.bc = m->bc; hp.w1 = m->w1; hp.b1 = m->b1; hp.w2 = m->w2; hp.b2 = m->b2;
    hp.w3 = m->w3; hp.b3 = m->b3; hp.logits = m->d_logits; hp.boxes = m->d_boxes; hp.rows = Md; hp.ncls = a.ncls;
    if (m->heads2 && m->wc_f && m->w1_f && m->w2_f) { hp.wc_f = m->wc_f; hp.w1_f = m->w1_f; hp.w2_f = m->w2_f; }
    RCCHK(timed_begin(m, CLS_OTHER, 2.0 * Md * 256.0 * (a.ncls + 256 + 256 + 4)));
    HIPCHK(opd_launch_heads(hp, m->stream));
    RCCHK(timed_end(m));
    RCCHK(tap(m, "heads_logits", m->d_logits, (size_t)Md * a.ncls * 4));
    MARK(7);
    m->last_B = B; m->last_H = H; m->last_W = W; m->last_fh = ch; m->last_fw = cw;
    m->last_ragged = ragged;
    return OPD_OK;
}

std::shared_mutex g_api_mu;
std::atomic<unsigned> g_handle_epoch{0};
std::atomic<int> g_graph_guard{1};   // opd_test_set_graph_guard(0): leave stale-epoch graphs alone (diagnosis only)
thread_local std::shared_lock<std::shared_mutex>* tl_api_lock = nullptr;

// Forward through the graph cache.  First call of a (shape, pixel pointer) key runs eagerly (one-time function-attribute
// setup and plan building are not capturable); the second call captures the stream into a hipGraph; later calls replay it.
int run_forward(opd_detr* m, const void* d_pixels, int pixel_format, int B, int H, int W, const int32_t* valid_hw) {
    m->graph_marks = false;
    // ragged batches run eagerly: their launch sequence depends on per-call host data (fold pointers, valid sizes)
    if (is_ragged(valid_hw, B, H, W)) return enqueue_forward(m, d_pixels, pixel_format, B, H, W, valid_hw);
    if (m->profiling == 1 || (m->cfg.flags & OPD_FLAG_NO_GRAPH)) return enqueue_forward(m, d_pixels, pixel_format, B, H, W);
    opd_detr::GraphEntry* e = nullptr;
    for (auto& g : m->graphs)
        if (g.B == B && g.H == H && g.W == W && g.fmt == pixel_format && g.pixels == d_pixels) e = &g;
    if (!e) {
        if (m->graphs.size() >= 8) {  // bounded cache: drop the oldest entry
            if (m->graphs.front().exec) (void)hipGraphExecDestroy(m->graphs.front().exec);
            m->graphs.erase(m->graphs.begin());
        }
        m->graphs.push_back({B, H, W, pixel_format, 0, 0, d_pixels, 0, nullptr, 0u});
        e = &m->graphs.back();
    }
    if (e->exec && g_graph_guard.load() && e->epoch != g_handle_epoch.load()) {   // handles came or went since the capture: capture again (see g_handle_epoch)
        (void)hipGraphExecDestroy(e->exec);
        e->exec = nullptr;
        e->uses = 1;
    }
    if (e->exec) {
        HIPCHK(hipGraphLaunch(e->exec, m->stream));
        if (m->profiling == 2) { HIPCHK(hipEventRecord(m->ev[9], m->stream)); m->graph_marks = true; }   // eager mark behind the graph: the post-process stage is timed between eager events
        m->last_B = B; m->last_H = H; m->last_W = W; m->last_fh = e->fh; m->last_fw = e->fw; m->last_ragged = false;
        return OPD_OK;
    }
    if (e->uses++ == 0) return enqueue_forward(m, d_pixels, pixel_format, B, H, W);
    hipGraph_t graph = nullptr;
    int rc;
    hipError_t ec;
    {
        CaptureExclusive alone;
        HIPCHK(hipStreamBeginCapture(m->stream, hipStreamCaptureModeThreadLocal));
        rc = enqueue_forward(m, d_pixels, pixel_format, B, H, W);
        ec = hipStreamEndCapture(m->stream, &graph);
    }
    if (rc != OPD_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    // A refused capture or instantiation is an ERROR of the call, not a reason to go on eagerly without saying so (round 3 did): the
    // caller asked for the graph path, and a runtime that rejects the recorded launch sequence has a reason a caller should see.
    if (ec != hipSuccess || !graph) {
        e->uses = 1;   // (the next call tries again)
        (void)hipGetLastError();
        return fail(OPD_EHIP, std::string("hipStreamEndCapture refused the forward: ") + hipGetErrorString(ec) + " (OPD_FLAG_NO_GRAPH runs eagerly)");
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess) {
        e->uses = 1;
        (void)hipGetLastError();
        return fail(OPD_EHIP, std::string("hipGraphInstantiate failed: ") + hipGetErrorString(ei) + " (OPD_FLAG_NO_GRAPH runs eagerly)");
    }
    e->exec = exec; e->fh = m->last_fh; e->fw = m->last_fw; e->epoch = g_handle_epoch.load();
    HIPCHK(hipGraphLaunch(exec, m->stream));
    if (m->profiling == 2) { HIPCHK(hipEventRecord(m->ev[9], m->stream)); m->graph_marks = true; }
    return OPD_OK;
}

// mem_kind: where the pixels come from / where the outputs go
static inline bool pixels_on_device(int mem_kind) { return mem_kind == OPD_MEM_DEVICE; }
static inline bool outputs_on_device(int mem_kind) { return mem_kind != OPD_MEM_HOST; }

// A pointer that kernels will dereference must be memory the HIP runtime knows as device-accessible (device, page-locked host or managed):
// an ordinary host pointer handed over with a DEVICE mem_kind would make a kernel fault the GPU -- for every process on it -- instead of
// returning an error (hipPointerGetAttributes: 0.06 us per call).
static bool device_accessible(const void* p) {
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeHost || a.type == hipMemoryTypeManaged;
}
static int check_device_outputs(int mem_kind, const void* out, const void* counts, const char* who) {
    if (!outputs_on_device(mem_kind)) return OPD_OK;
    if (!device_accessible(out) || !device_accessible(counts))
        return fail(OPD_EINVAL, std::string(who) + ": this mem_kind takes DEVICE output pointers; the ones given are not device-accessible memory");
    return OPD_OK;
}

int check_shape(opd_detr* m, const void* pixels, int pixel_format, int mem_kind, int B, int H, int W) {
    if (!m) return fail(OPD_EINVAL, "null model handle");
    if (!pixels) return fail(OPD_EINVAL, "null pixel buffer");
    if (pixel_format != OPD_PIXELS_U8_BGR_HWC && pixel_format != OPD_PIXELS_F32_NCHW) return fail(OPD_EINVAL, "unknown pixel_format");
    if (mem_kind != OPD_MEM_HOST && mem_kind != OPD_MEM_DEVICE && mem_kind != OPD_MEM_HOST_PIXELS_DEVICE_OUT)
        return fail(OPD_EINVAL, "unknown mem_kind");
    if (pixels_on_device(mem_kind) && !device_accessible(pixels))
        return fail(OPD_EINVAL, "OPD_MEM_DEVICE: the pixel pointer is not device-accessible memory");
    const int edge = std::max(m->cfg.max_height, m->cfg.max_width);   // either orientation: see build_workspace
    if (B < 1 || B > m->cfg.max_batch || H < 32 || W < 32 || H > edge || W > edge ||
        (size_t)H * W > (size_t)m->cfg.max_height * m->cfg.max_width)
        return fail(OPD_EINVAL, "frame batch [" + std::to_string(B) + "," + std::to_string(H) + "," + std::to_string(W) +
                                    "] outside the configured maximum [" + std::to_string(m->cfg.max_batch) + "," +
                                    std::to_string(m->cfg.max_height) + "," + std::to_string(m->cfg.max_width) + "] (either orientation)");
    return OPD_OK;
}

int stage_pixels(opd_detr* m, const void* pixels, int pixel_format, int mem_kind, int B, int H, int W, const void** d_pixels) {
    if (pixels_on_device(mem_kind)) { *d_pixels = pixels; return OPD_OK; }
    const size_t n = (size_t)B * H * W * 3;
    if (pixel_format == OPD_PIXELS_U8_BGR_HWC) {
        HIPCHK(hipMemcpyAsync(m->d_u8, pixels, n, hipMemcpyHostToDevice, m->stream));
        *d_pixels = m->d_u8;
    } else {
        HIPCHK(hipMemcpyAsync(m->d_pv, pixels, n * 4, hipMemcpyHostToDevice, m->stream));
        *d_pixels = m->d_pv;
    }
    return OPD_OK;
}

// Frames at camera resolution -> m->d_u8 at model resolution (Pillow-exact bilinear, kernels_misc.hip).
// (`list` != null: one host pointer per frame instead of the contiguous block `frames`)
static int enqueue_resize(opd_detr* m, const uint8_t* frames, int mem_kind, int B, int h, int w, int oh, int ow, const uint8_t* const* list = nullptr) {
    if (h < 1 || w < 1 || (size_t)h * w > (size_t)1 << 26) return fail(OPD_EINVAL, "source frame size out of range");
    const size_t need = (size_t)B * h * w * 3;
    const uint8_t* d_in = frames;
    if (!pixels_on_device(mem_kind)) {
        if (need > m->src_bytes) {
            HIPCHK(hipStreamSynchronize(m->stream));
            if (m->d_src) (void)hipFree(m->d_src);
            m->d_src = nullptr; m->src_bytes = 0;
            void* q = nullptr;
            if (hipMalloc(&q, need) != hipSuccess) return fail(OPD_ENOMEM, "source frame staging allocation failed");
            m->d_src = reinterpret_cast<uint8_t*>(q);
            m->src_bytes = need;
        }
        if (list) {
            for (int b = 0; b < B; ++b) HIPCHK(hipMemcpyAsync(m->d_src + (size_t)b * h * w * 3, list[b], (size_t)h * w * 3, hipMemcpyHostToDevice, m->stream));
        } else {
            HIPCHK(hipMemcpyAsync(m->d_src, frames, need, hipMemcpyHostToDevice, m->stream));
        }
        d_in = m->d_src;
    }
    const opd_detr::ResizeTab* tab = nullptr;
    for (const auto& t : m->resize_tabs)
        if (t.h == h && t.w == w && t.oh == oh && t.ow == ow) tab = &t;
    if (!tab) {
        std::vector<int32_t> bh, kh, bv, kv;
        opd_detr::ResizeTab t{h, w, oh, ow, 0, 0, nullptr, nullptr, nullptr, nullptr};
        opd_resize_coeffs(w, ow, &bh, &kh, &t.ksh);
        opd_resize_coeffs(h, oh, &bv, &kv, &t.ksv);
        auto up = [&](const std::vector<int32_t>& v, int32_t** d) -> int {
            RCCHK(dalloc(m, d, v.size(), false));
            HIPCHK(hipMemcpy(*d, v.data(), v.size() * 4, hipMemcpyHostToDevice));
            return OPD_OK;
        };
        RCCHK(up(bh, &t.bh)); RCCHK(up(kh, &t.kh)); RCCHK(up(bv, &t.bv)); RCCHK(up(kv, &t.kv));
        m->resize_tabs.push_back(t);
        tab = &m->resize_tabs.back();
    }
    HIPCHK(opd_launch_resize_u8(d_in, m->d_u8, B, h, w, oh, ow, tab->bh, tab->kh, tab->ksh, tab->bv, tab->kv, tab->ksv, m->stream));
    return OPD_OK;
}

// `dev_out` / `dev_counts` (nullable): device buffers of the caller; the kernel then writes there directly instead of the
// library's own record buffers (no device-to-device copies afterwards).
int enqueue_postprocess(opd_detr* m, float threshold, const int32_t* orig_hw, opd_det* dev_out, int32_t* dev_counts) {
    const int B = m->last_B;
    std::vector<int32_t> hw((size_t)B * 2);
    for (int b = 0; b < B; ++b) {
        hw[2 * b] = orig_hw ? orig_hw[2 * b] : m->last_H;
        hw[2 * b + 1] = orig_hw ? orig_hw[2 * b + 1] : m->last_W;
    }
    if (hw != m->h_orig_hw) {   // the frame sizes of a video do not change from call to call: upload only when they do
        HIPCHK(hipStreamSynchronize(m->stream));   // (an earlier asynchronous copy may still be reading the old host vector)
        m->h_orig_hw = hw;      // member: must outlive the async copy
        HIPCHK(hipMemcpyAsync(m->d_orig_hw, m->h_orig_hw.data(), m->h_orig_hw.size() * 4, hipMemcpyHostToDevice, m->stream));
    }
    PostParams pp{};
    pp.logits = m->d_logits; pp.boxes = m->d_boxes; pp.orig_hw = m->d_orig_hw;
    pp.records = dev_out ? dev_out : m->d_records;
    pp.counts = dev_counts ? dev_counts : m->d_counts;
    pp.B = B; pp.Q = m->arch.queries; pp.ncls = m->arch.ncls; pp.threshold = threshold;
    RCCHK(timed_begin(m, CLS_OTHER, 0.0));
    HIPCHK(opd_launch_postprocess(pp, m->stream));
    RCCHK(timed_end(m));
    MARK(8);
    return OPD_OK;
}

// (`features` != null: the [B][Q][d_model] feature rows behind the counts travel in the same copy)
static int fetch_records(opd_detr* m, opd_det* out, int32_t* counts, int mem_kind, float* features = nullptr) {
    const int B = m->last_B, Q = m->arch.queries;
    if (!outputs_on_device(mem_kind)) {   // (device callers had the post-process kernel write into their buffers)
        // one copy of [records of max_batch frames | counts (| features)] into page-locked memory (a copy into the caller's pageable arrays is
        // staged by the runtime anyway, once per call), handed over after the wait
        const size_t rec_bytes = (size_t)m->cfg.max_batch * Q * sizeof(opd_det);
        const size_t feat_off = reinterpret_cast<const char*>(m->d_feat_all) - reinterpret_cast<const char*>(m->d_records);
        const size_t feat_row = (size_t)Q * m->arch.d_model * 4;
        if (!m->sync_pinned) HIPCHK(hipHostMalloc(&m->sync_pinned, feat_off + (size_t)m->cfg.max_batch * feat_row, hipHostMallocDefault));
        HIPCHK(hipMemcpyAsync(m->sync_pinned, m->d_records, features ? feat_off + (size_t)B * feat_row : rec_bytes + (size_t)B * 4, hipMemcpyDeviceToHost, m->stream));
        HIPCHK(hipStreamSynchronize(m->stream));
        memcpy(out, m->sync_pinned, (size_t)B * Q * sizeof(opd_det));
        memcpy(counts, static_cast<char*>(m->sync_pinned) + rec_bytes, (size_t)B * 4);
        if (features) memcpy(features, static_cast<char*>(m->sync_pinned) + feat_off, (size_t)B * feat_row);
    }
    HIPCHK(hipStreamSynchronize(m->stream));
    if (m->profiling) {
        for (int i = 0; i < 8; ++i) {
            float ms = 0.f;
            // (mode 2: marks 0 .. 7 are nodes of the replayed graph, mark 8 is recorded eagerly behind the post-process launch: events of the
            //  two kinds do not subtract meaningfully, so the last stage runs from the eager mark 9 behind the graph launch)
            hipEvent_t from = (i == 7 && m->profiling == 2 && m->graph_marks) ? m->ev[9] : m->ev[i];
            if (hipEventElapsedTime(&ms, from, m->ev[i + 1]) == hipSuccess) m->stage_ms[i] = ms;
            else (void)hipGetLastError();   // (a mark that was never recorded: not this call's error)
        }
        if (m->profiling == 1) timed_collect(m);
    }
    return OPD_OK;
}

}  // namespace opd

// =====================================================================================================================
// C-ABI
// =====================================================================================================================
// No C++ exception may cross the C-ABI: creation parses an untrusted file and allocates, so its body runs under a catch-all.
// second branch of the forward (stage-3 frame split): a stream and two untimed events per handle
static hipError_t make_branch_stream(opd_detr* m) {
    hipError_t e = hipStreamCreateWithFlags(&m->stream2, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming);
    return e;
}
static void drop_streams(opd_detr* m) {
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    if (m->ev_join) (void)hipEventDestroy(m->ev_join);
    if (m->stream2) (void)hipStreamDestroy(m->stream2);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    m->ev_fork = m->ev_join = nullptr; m->stream2 = m->stream = nullptr;
}

static int create_impl(const opd_config* cfg, const char* weights_path, int device_ordinal, opd_detr** out);
static int clone_impl(const opd_detr* src, opd_detr** out);
template <typename F>
static int guarded(const char* what, F&& body) {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return fail(OPD_ENOMEM, std::string(what) + ": out of host memory");
    } catch (const std::out_of_range& e) {
        return fail(OPD_ESCHEMA, std::string(what) + ": weight file lacks a tensor the model needs (" + e.what() + ")");
    } catch (const std::exception& e) {
        return fail(OPD_EINVAL, std::string(what) + ": " + e.what());
    } catch (...) {
        return fail(OPD_EINVAL, std::string(what) + ": unknown C++ exception");
    }
}
extern "C" {

const char* opd_version(void) { return "opd_hip 0.2 gfx950 (fp16 MFMA operands, fp32 accumulate; OPD_FLAG_BF16: bf16 operands)"; }

int opd_detr_create(const opd_config* cfg, const char* weights_path, int device_ordinal, opd_detr** out) {
    ApiScope api_scope;
    if (out) *out = nullptr;
    return guarded("opd_detr_create", [&] { return create_impl(cfg, weights_path, device_ordinal, out); });
}
int opd_detr_clone(const opd_detr* src, opd_detr** out) {
    ApiScope api_scope;
    if (out) *out = nullptr;
    return guarded("opd_detr_clone", [&] { return clone_impl(src, out); });
}
static int create_impl(const opd_config* cfg, const char* weights_path, int device_ordinal, opd_detr** out) {
    if (!cfg || !weights_path || !out) return fail(OPD_EINVAL, "opd_detr_create: null argument");
    if (cfg->struct_size != (int32_t)sizeof(opd_config)) return fail(OPD_EINVAL, "opd_config.struct_size mismatch");
    if (cfg->max_batch < 1 || cfg->max_height < 32 || cfg->max_width < 32) return fail(OPD_EINVAL, "opd_config maxima must be >= 1 x 32 x 32");
    *out = nullptr;
    StateDict sd;
    std::string err;
    int rc = load_safetensors(weights_path, &sd, &err);
    if (rc) return fail(rc, err);
    std::unique_ptr<opd_detr> m(new opd_detr());
    rc = infer_arch(sd, &m->arch, &err);
    if (rc) return fail(rc, err);
    m->cfg = *cfg;
    m->dtype = (cfg->flags & OPD_FLAG_BF16) ? OPD_DT_BF16 : OPD_DT_F16;
    if (const char* v = getenv("OPD_TRUNK_SUBBATCH")) m->trunk_subbatch = atoi(v);   // A/B switches for benchmarking
    if (const char* v = getenv("OPD_DUAL_OVER_TAIL")) m->dual_over_tail = atoi(v);
    if (const char* v = getenv("OPD_TAIL_REV")) m->tail_rev = atoi(v);
    if (const char* v = getenv("OPD_TAIL3")) m->tail3 = atoi(v);
    if (const char* v = getenv("OPD_TAIL_RC")) m->tail_rc = atoi(v);
    if (const char* v = getenv("OPD_WPREFETCH")) m->wprefetch = atoi(v);
    if (const char* v = getenv("OPD_TAIL_NW")) m->tail_nw = atoi(v) == 8 ? 8 : 4;
    if (const char* v = getenv("OPD_W8")) m->w8 = atoi(v);
    if (m->w8 < 0) m->w8 = (cfg->flags & OPD_FLAG_MULTI_STREAM) ? 1 : 0;
    if (const char* v = getenv("OPD_SMALL_SPLITK")) m->small_splitk = atoi(v);
    if (const char* v = getenv("OPD_SMALL_ENC")) m->small_enc = atoi(v);
    if (const char* v = getenv("OPD_Y_STRIDE2")) m->y_stride2 = atoi(v);
    if (const char* v = getenv("OPD_TAIL3_SPLIT")) m->tail3_split = atoi(v);
    if (const char* v = getenv("OPD_FUSE_PREP")) m->fuse_prep = atoi(v);
    if (const char* v = getenv("OPD_POS_SHADOW")) m->pos_shadow = atoi(v);
    if (const char* v = getenv("OPD_DEEP_FC2")) m->deep_fc2 = atoi(v);
    if (const char* v = getenv("OPD_WROUND")) m->wround = atoi(v);
    if (const char* v = getenv("OPD_FUSED_DEC")) m->fused_dec = atoi(v);
    if (const char* v = getenv("OPD_FUSED_ENC_FFN")) m->fused_enc_ffn = atoi(v);
    if (const char* v = getenv("OPD_ENC_TAIL")) m->enc_tail = atoi(v);
    if (const char* v = getenv("OPD_ENC_FRONT")) m->enc_front = atoi(v);
    if (const char* v = getenv("OPD_HEADS2")) m->heads2 = atoi(v);
    if (const char* v = getenv("OPD_DBG_DEC_LAYERS")) m->dbg_dec_layers = atoi(v);   // timing ablation (tools/dec_cost.sh): results are wrong
    if (const char* v = getenv("OPD_DBG_BTAIL")) m->dbg_btail = atoi(v);             // timing ablations of the whole forward: BtailParams::dbg / ConvGemmParams::dbg
    if (const char* v = getenv("OPD_DBG_SKIP")) m->dbg_skip = atoi(v);               // bit i: no kernel launches in segment i of stage_ms (stem, stages 1-4, encoder, decoder, post-process)
    if (const char* v = getenv("OPD_DBG_GEMM")) m->dbg_gemm = atoi(v);               // of every fused tail / implicit-GEMM launch (results are wrong)
    m->device = device_ordinal;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(OPD_EHIP, "no HIP device visible (this library has no CPU fallback)");
    if (device_ordinal < 0 || device_ordinal >= ndev) return fail(OPD_EINVAL, "device_ordinal out of range");
    auto cleanup = [&](int code) {
        for (void* p : m->allocs) (void)hipFree(p);
        drop_streams(m.get());
        return code;
    };
    {
        hipError_t e = hipSetDevice(device_ordinal);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = make_branch_stream(m.get());
        if (e != hipSuccess) { drop_streams(m.get()); return fail(OPD_EHIP, std::string("device/stream setup failed: ") + hipGetErrorString(e)); }
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_ordinal) == hipSuccess && cus > 0) m->num_cus = cus;
    }
    m->weights = std::make_shared<WeightSet>();
    m->weights->device = device_ordinal;
    if ((rc = build_weights(m.get(), sd))) return cleanup(rc);
    m->weights_sealed = true;
    if ((rc = build_workspace(m.get()))) return cleanup(rc);
    for (auto& e : m->ev)
        if (hipEventCreate(&e) != hipSuccess) return cleanup(fail(OPD_EHIP, "hipEventCreate failed"));
    *out = m.release();
    ++g_handle_epoch;
    return OPD_OK;
}

static int clone_impl(const opd_detr* src, opd_detr** out) {
    if (!src || !out) return fail(OPD_EINVAL, "opd_detr_clone: null argument");
    *out = nullptr;
    std::unique_ptr<opd_detr> m(new opd_detr());
    m->arch = src->arch; m->cfg = src->cfg; m->device = src->device; m->dtype = src->dtype;
    // everything build_weights produced: device pointers into the shared WeightSet and the host copies the plans are folded from
    m->weights = src->weights; m->weights_sealed = true; m->weight_bytes = src->weight_bytes;
    m->stem = src->stem; m->blocks = src->blocks; m->stage_first = src->stage_first; m->proj = src->proj;
    m->enc = src->enc; m->dec = src->dec; m->wkv_all = src->wkv_all; m->bkv_all = src->bkv_all; m->dec_ln = src->dec_ln;
    m->wc = src->wc; m->bc = src->bc; m->w1 = src->w1; m->b1 = src->b1; m->w2 = src->w2; m->b2 = src->b2; m->w3 = src->w3; m->b3 = src->b3; m->wc_f = src->wc_f; m->w1_f = src->w1_f; m->w2_f = src->w2_f; m->heads2 = src->heads2;
    m->zero_bias = src->zero_bias;
    m->h_enc_cat_w = src->h_enc_cat_w; m->h_enc_cat_b = src->h_enc_cat_b; m->h_kv_cat_w = src->h_kv_cat_w; m->h_kv_cat_b = src->h_kv_cat_b;
    m->small_m_gemm = src->small_m_gemm; m->fuse_gemm_ln = src->fuse_gemm_ln; m->deep_fc2 = src->deep_fc2;
    m->fuse_btail = src->fuse_btail; m->fuse_shortcut = src->fuse_shortcut; m->fuse_stem_pool = src->fuse_stem_pool; m->fuse_prep = src->fuse_prep; m->pos_shadow = src->pos_shadow; m->trunk_subbatch = src->trunk_subbatch; m->dual_over_tail = src->dual_over_tail; m->tail_rev = src->tail_rev; m->tail3 = src->tail3; m->num_cus = src->num_cus; m->tail3_split = src->tail3_split;
    m->dec0_h = src->dec0_h; m->fuse_dec0 = src->fuse_dec0; m->qc0 = src->qc0; m->fused_dec = src->fused_dec; m->fused_enc_ffn = src->fused_enc_ffn; m->enc_tail = src->enc_tail; m->enc_front = src->enc_front; m->dec_splits = src->dec_splits; m->wround = src->wround; m->dbg_dec_layers = src->dbg_dec_layers; m->dbg_skip = src->dbg_skip;
    m->tail_rc = src->tail_rc; m->y_stride2 = src->y_stride2; m->dbg_btail = src->dbg_btail; m->dbg_gemm = src->dbg_gemm; m->wprefetch = src->wprefetch; m->w8 = src->w8; m->tail_nw = src->tail_nw; m->small_splitk = src->small_splitk; m->small_enc = src->small_enc;
    auto cleanup = [&](int code) {
        for (void* p : m->allocs) (void)hipFree(p);
        drop_streams(m.get());
        return code;
    };
    {
        hipError_t e = hipSetDevice(m->device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = make_branch_stream(m.get());
        if (e != hipSuccess) { drop_streams(m.get()); return fail(OPD_EHIP, std::string("device/stream setup failed: ") + hipGetErrorString(e)); }
    }
    int rc;
    if ((rc = build_workspace(m.get()))) return cleanup(rc);
    for (auto& e : m->ev)
        if (hipEventCreate(&e) != hipSuccess) return cleanup(fail(OPD_EHIP, "hipEventCreate failed"));
    *out = m.release();
    ++g_handle_epoch;
    return OPD_OK;
}

void opd_detr_destroy(opd_detr* m) {
    ApiScope api_scope;
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    comm_detach_all(m);   // communicator lanes bound to this handle refuse work from here on (their own destroy still frees them)
    for (void* p : m->allocs) (void)hipFree(p);
    if (m->d_src) (void)hipFree(m->d_src);
    for (auto& e : m->ev_async)
        if (e) (void)hipEventDestroy(e);
    for (auto& a : m->async_host)
        if (a.pinned) (void)hipHostFree(a.pinned);
    if (m->sync_pinned) { (void)hipHostFree(m->sync_pinned); m->sync_pinned = nullptr; }
    for (auto& e : m->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : m->event_pool) (void)hipEventDestroy(e);
    for (auto& g : m->graphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    drop_streams(m);
    delete m;
    ++g_handle_epoch;
}

int opd_detr_info(const opd_detr* m, opd_model_info* info) {
    if (!m || !info) return fail(OPD_EINVAL, "opd_detr_info: null argument");
    for (int i = 0; i < 4; ++i) info->depths[i] = m->arch.depths[i];
    info->d_model = m->arch.d_model; info->heads = m->arch.heads; info->ffn_dim = m->arch.ffn;
    info->encoder_layers = m->arch.enc_layers; info->decoder_layers = m->arch.dec_layers;
    info->num_queries = m->arch.queries; info->num_classes_plus1 = m->arch.ncls;
    info->max_batch = m->cfg.max_batch; info->max_height = m->cfg.max_height; info->max_width = m->cfg.max_width;
    info->device_ordinal = m->device;
    info->weight_bytes_device = m->weight_bytes; info->workspace_bytes_device = m->workspace_bytes;
    return OPD_OK;
}

// forward on device-resident pixels; outputs go to host or device memory according to `out_kind`
static int forward_device(opd_detr* m, const void* d_pixels, int pixel_format, int out_kind, int B, int H, int W,
                          const int32_t* valid_hw, float* logits, float* boxes, float* enc_features) {
    RCCHK(run_forward(m, d_pixels, pixel_format, B, H, W, valid_hw));
    const hipMemcpyKind kind = outputs_on_device(out_kind) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    const size_t Md = (size_t)B * m->arch.queries;
    if (logits) HIPCHK(hipMemcpyAsync(logits, m->d_logits, Md * m->arch.ncls * 4, kind, m->stream));
    if (boxes) HIPCHK(hipMemcpyAsync(boxes, m->d_boxes, Md * 4 * 4, kind, m->stream));
    if (enc_features)
        HIPCHK(hipMemcpyAsync(enc_features, m->d_x32, (size_t)B * m->last_fh * m->last_fw * m->arch.d_model * 4, kind, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    if (m->profiling == 1) timed_collect(m);
    return OPD_OK;
}
int opd_detr_forward(opd_detr* m, const void* pixels, int pixel_format, int mem_kind, int B, int H, int W, float* logits,
                     float* boxes, float* enc_features) {
    ApiScope api_scope;
    return opd_detr_forward_ragged(m, pixels, pixel_format, mem_kind, B, H, W, nullptr, logits, boxes, enc_features);
}
int opd_detr_forward_ragged(opd_detr* m, const void* pixels, int pixel_format, int mem_kind, int B, int H, int W,
                            const int32_t* valid_hw, float* logits, float* boxes, float* enc_features) {
    ApiScope api_scope;
    RCCHK(check_shape(m, pixels, pixel_format, mem_kind, B, H, W));
    HIPCHK(hipSetDevice(m->device));
    const void* d_pixels = nullptr;
    RCCHK(stage_pixels(m, pixels, pixel_format, mem_kind, B, H, W, &d_pixels));
    return forward_device(m, d_pixels, pixel_format, mem_kind, B, H, W, valid_hw, logits, boxes, enc_features);
}
int opd_detr_postprocess(opd_detr* m, float threshold, const int32_t* orig_hw, opd_det* out, int32_t* counts) {
    ApiScope api_scope;
    if (!m || !out || !counts) return fail(OPD_EINVAL, "opd_detr_postprocess: null argument");
    if (m->last_B == 0) return fail(OPD_ESTATE, "opd_detr_postprocess called before any forward");
    HIPCHK(hipSetDevice(m->device));
    RCCHK(enqueue_postprocess(m, threshold, orig_hw));
    return fetch_records(m, out, counts, OPD_MEM_HOST);
}
int opd_detr_resize_u8(opd_detr* m, const uint8_t* frames, int B, int h, int w, int out_h, int out_w, uint8_t* out) {
    ApiScope api_scope;
    if (!m) return fail(OPD_EINVAL, "null model handle");
    if (!frames || !out) return fail(OPD_EINVAL, "opd_detr_resize_u8: null buffer");
    uint8_t dummy = 0;
    RCCHK(check_shape(m, &dummy, OPD_PIXELS_U8_BGR_HWC, OPD_MEM_HOST, B, out_h, out_w));
    HIPCHK(hipSetDevice(m->device));
    RCCHK(enqueue_resize(m, frames, OPD_MEM_HOST, B, h, w, out_h, out_w));
    HIPCHK(hipMemcpyAsync(out, m->d_u8, (size_t)B * out_h * out_w * 3, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return OPD_OK;
}
int opd_detr_forward_resized(opd_detr* m, const uint8_t* frames, int mem_kind, int B, int h, int w, int H, int W, float* logits,
                             float* boxes, float* enc_features) {
    ApiScope api_scope;
    if (!m) return fail(OPD_EINVAL, "null model handle");
    RCCHK(check_shape(m, frames, OPD_PIXELS_U8_BGR_HWC, mem_kind, B, H, W));
    HIPCHK(hipSetDevice(m->device));
    RCCHK(enqueue_resize(m, frames, mem_kind, B, h, w, H, W));
    return forward_device(m, m->d_u8, OPD_PIXELS_U8_BGR_HWC, mem_kind, B, H, W, nullptr, logits, boxes, enc_features);
}
int opd_detr_detect(opd_detr* m, const void* pixels, int pixel_format, int mem_kind, int B, int H, int W, float threshold,
                    const int32_t* orig_hw, opd_det* out, int32_t* counts) {
    ApiScope api_scope;
    return opd_detr_detect_ragged(m, pixels, pixel_format, mem_kind, B, H, W, nullptr, threshold, orig_hw, out, counts);
}
int opd_detr_detect_ragged(opd_detr* m, const void* pixels, int pixel_format, int mem_kind, int B, int H, int W,
                           const int32_t* valid_hw, float threshold, const int32_t* orig_hw, opd_det* out, int32_t* counts) {
    ApiScope api_scope;
    RCCHK(check_shape(m, pixels, pixel_format, mem_kind, B, H, W));
    if (!out || !counts) return fail(OPD_EINVAL, "opd_detr_detect: null output buffer");
    RCCHK(check_device_outputs(mem_kind, out, counts, "opd_detr_detect"));
    HIPCHK(hipSetDevice(m->device));
    const void* d_pixels = nullptr;
    RCCHK(stage_pixels(m, pixels, pixel_format, mem_kind, B, H, W, &d_pixels));
    RCCHK(run_forward(m, d_pixels, pixel_format, B, H, W, valid_hw));
    const bool dev = outputs_on_device(mem_kind);
    RCCHK(enqueue_postprocess(m, threshold, orig_hw, dev ? out : nullptr, dev ? counts : nullptr));
    return fetch_records(m, out, counts, mem_kind);
}
int opd_detr_detect_async(opd_detr* m, const void* pixels, int pixel_format, int mem_kind, int B, int H, int W, float threshold,
                          const int32_t* orig_hw, opd_det* out, int32_t* counts, int* ticket) {
    ApiScope api_scope;
    RCCHK(check_shape(m, pixels, pixel_format, mem_kind, B, H, W));
    if (!out || !counts || !ticket) return fail(OPD_EINVAL, "opd_detr_detect_async: null argument");
    RCCHK(check_device_outputs(mem_kind, out, counts, "opd_detr_detect_async"));
    if (m->profiling) return fail(OPD_ESTATE, "opd_detr_detect_async is not available in profiling mode");
    HIPCHK(hipSetDevice(m->device));
    const unsigned t = m->async_next & 3u;
    if (m->async_pending[t])
        return fail(OPD_ESTATE, "opd_detr_detect_async: 4 submissions are outstanding on this handle; opd_detr_wait the oldest ticket first");
    if (!m->ev_async[t]) HIPCHK(hipEventCreateWithFlags(&m->ev_async[t], hipEventDisableTiming));
    const void* d_pixels = nullptr;
    RCCHK(stage_pixels(m, pixels, pixel_format, mem_kind, B, H, W, &d_pixels));
    RCCHK(run_forward(m, d_pixels, pixel_format, B, H, W, nullptr));
    const int Q = m->arch.queries;
    opd_detr::AsyncHost& slot = m->async_host[t];
    slot.out = nullptr;
    if (outputs_on_device(mem_kind)) {
        RCCHK(enqueue_postprocess(m, threshold, orig_hw, out, counts));
    } else {   // host outputs: pinned staging so that the copy stays asynchronous; delivered by opd_detr_wait
        RCCHK(enqueue_postprocess(m, threshold, orig_hw));
        const size_t rec_bytes = (size_t)m->cfg.max_batch * Q * sizeof(opd_det);
        if (!slot.pinned) HIPCHK(hipHostMalloc(&slot.pinned, rec_bytes + (size_t)m->cfg.max_batch * 4, hipHostMallocDefault));
        HIPCHK(hipMemcpyAsync(slot.pinned, m->d_records, rec_bytes + (size_t)B * 4, hipMemcpyDeviceToHost, m->stream));   // (records | counts: one copy)
        slot.out = out; slot.counts = counts; slot.B = B;
    }
    HIPCHK(hipEventRecord(m->ev_async[t], m->stream));
    m->async_pending[t] = true;
    ++m->async_next;
    *ticket = (int)t;
    return OPD_OK;
}
int opd_detr_wait(opd_detr* m, int ticket) {
    ApiScope api_scope;
    if (!m || ticket < 0 || ticket > 3 || !m->ev_async[ticket]) return fail(OPD_EINVAL, "opd_detr_wait: bad handle or ticket");
    if (!m->async_pending[ticket]) return fail(OPD_ESTATE, "opd_detr_wait: this ticket is not outstanding (already waited for?)");
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipEventSynchronize(m->ev_async[ticket]));
    opd_detr::AsyncHost& slot = m->async_host[ticket];
    if (slot.out) {
        const int Q = m->arch.queries;
        const size_t rec_bytes = (size_t)m->cfg.max_batch * Q * sizeof(opd_det);
        memcpy(slot.out, slot.pinned, (size_t)slot.B * Q * sizeof(opd_det));
        memcpy(slot.counts, static_cast<char*>(slot.pinned) + rec_bytes, (size_t)slot.B * 4);
        slot.out = nullptr;
    }
    m->async_pending[ticket] = false;
    return OPD_OK;
}
int opd_detr_detect_resized(opd_detr* m, const uint8_t* frames, int mem_kind, int B, int h, int w, int H, int W, float threshold,
                            opd_det* out, int32_t* counts) {
    ApiScope api_scope;
    if (!m) return fail(OPD_EINVAL, "null model handle");
    RCCHK(check_shape(m, frames, OPD_PIXELS_U8_BGR_HWC, mem_kind, B, H, W));
    if (!out || !counts) return fail(OPD_EINVAL, "opd_detr_detect_resized: null output buffer");
    RCCHK(check_device_outputs(mem_kind, out, counts, "opd_detr_detect_resized"));
    HIPCHK(hipSetDevice(m->device));
    RCCHK(enqueue_resize(m, frames, mem_kind, B, h, w, H, W));
    RCCHK(run_forward(m, m->d_u8, OPD_PIXELS_U8_BGR_HWC, B, H, W, nullptr));
    std::vector<int32_t> orig((size_t)2 * B);   // boxes are scaled to the ORIGINAL (camera) frame size
    for (int b = 0; b < B; ++b) { orig[2 * b] = h; orig[2 * b + 1] = w; }
    const bool dev = outputs_on_device(mem_kind);
    RCCHK(enqueue_postprocess(m, threshold, orig.data(), dev ? out : nullptr, dev ? counts : nullptr));
    return fetch_records(m, out, counts, mem_kind);
}

int opd_detr_detect_frames(opd_detr* m, const uint8_t* const* frames, int mem_kind, int B, int h, int w, int H, int W, float threshold,
                           opd_det* out, int32_t* counts) {
    ApiScope api_scope;
    if (!m) return fail(OPD_EINVAL, "null model handle");
    if (mem_kind != OPD_MEM_HOST && mem_kind != OPD_MEM_HOST_PIXELS_DEVICE_OUT) return fail(OPD_EINVAL, "opd_detr_detect_frames takes host frames");
    RCCHK(check_shape(m, frames, OPD_PIXELS_U8_BGR_HWC, mem_kind, B, H, W));
    for (int b = 0; b < B; ++b)
        if (!frames[b]) return fail(OPD_EINVAL, "opd_detr_detect_frames: null frame pointer");
    if (!out || !counts) return fail(OPD_EINVAL, "opd_detr_detect_frames: null output buffer");
    RCCHK(check_device_outputs(mem_kind, out, counts, "opd_detr_detect_frames"));
    HIPCHK(hipSetDevice(m->device));
    if (h == H && w == W) {
        const size_t n1 = (size_t)H * W * 3;
        for (int b = 0; b < B; ++b) HIPCHK(hipMemcpyAsync(m->d_u8 + b * n1, frames[b], n1, hipMemcpyHostToDevice, m->stream));
    } else {
        RCCHK(enqueue_resize(m, nullptr, mem_kind, B, h, w, H, W, frames));
    }
    RCCHK(run_forward(m, m->d_u8, OPD_PIXELS_U8_BGR_HWC, B, H, W, nullptr));
    std::vector<int32_t> orig((size_t)2 * B);
    for (int b = 0; b < B; ++b) { orig[2 * b] = h; orig[2 * b + 1] = w; }
    const bool dev = outputs_on_device(mem_kind);
    RCCHK(enqueue_postprocess(m, threshold, orig.data(), dev ? out : nullptr, dev ? counts : nullptr));
    return fetch_records(m, out, counts, mem_kind);
}

int opd_detr_detect_frames_features(opd_detr* m, const uint8_t* const* frames, int B, int h, int w, int H, int W, float threshold, int label,
                                    opd_det* out, int32_t* counts, float* features) {
    ApiScope api_scope;
    if (!m) return fail(OPD_EINVAL, "null model handle");
    RCCHK(check_shape(m, frames, OPD_PIXELS_U8_BGR_HWC, OPD_MEM_HOST, B, H, W));
    for (int b = 0; b < B; ++b)
        if (!frames[b]) return fail(OPD_EINVAL, "opd_detr_detect_frames_features: null frame pointer");
    if (!out || !counts || !features) return fail(OPD_EINVAL, "opd_detr_detect_frames_features: null output buffer");
    if (m->arch.d_model != 256) return fail(OPD_EINVAL, "opd_detr_detect_frames_features: the pooling kernel is built for d_model = 256");
    HIPCHK(hipSetDevice(m->device));
    const int Q = m->arch.queries;
    if (h == H && w == W) {
        const size_t n1 = (size_t)H * W * 3;
        for (int b = 0; b < B; ++b) HIPCHK(hipMemcpyAsync(m->d_u8 + b * n1, frames[b], n1, hipMemcpyHostToDevice, m->stream));
    } else {
        RCCHK(enqueue_resize(m, nullptr, OPD_MEM_HOST, B, h, w, H, W, frames));
    }
    RCCHK(run_forward(m, m->d_u8, OPD_PIXELS_U8_BGR_HWC, B, H, W, nullptr));
    std::vector<int32_t> orig((size_t)2 * B);
    for (int b = 0; b < B; ++b) { orig[2 * b] = h; orig[2 * b + 1] = w; }
    RCCHK(enqueue_postprocess(m, threshold, orig.data()));
    HIPCHK(opd_launch_roi_features_records(m->d_x32, m->d_records, m->d_counts, m->d_orig_hw, label, m->d_feat_all, B, Q, m->last_fh, m->last_fw, m->stream));
    return fetch_records(m, out, counts, OPD_MEM_HOST, features);   // records, counts and feature rows: one copy, one wait
}

int opd_host_alloc(size_t bytes, void** out) {
    ApiScope api_scope;
    if (!out || bytes == 0) return fail(OPD_EINVAL, "opd_host_alloc: null output or zero size");
    *out = nullptr;
    HIPCHK(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return OPD_OK;
}
void opd_host_free(void* p) {
    ApiScope api_scope;
    if (p) (void)hipHostFree(p);
}

int opd_similarity_matrix(int device_ordinal, const float* feats1, const float* boxes1, const uint8_t* has1, int n1,
                          const float* feats2, const float* boxes2, const uint8_t* has2, int n2, int D, double appearance_weight,
                          double motion_weight, int as_distance, float* out) {
    ApiScope api_scope;
    if (n1 < 0 || n2 < 0 || D < 1) return fail(OPD_EINVAL, "opd_similarity_matrix: bad sizes");
    if (n1 == 0 || n2 == 0) return OPD_OK;
    if (!boxes1 || !boxes2 || !out) return fail(OPD_EINVAL, "opd_similarity_matrix: null boxes / output");
    if (fabs(appearance_weight + motion_weight - 1.0) > 1e-6)
        return fail(OPD_EINVAL, "appearance_weight + motion_weight must equal 1.0");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(OPD_EHIP, "no HIP device visible (this library has no CPU fallback)");
    HIPCHK(hipSetDevice(device_ordinal));
    struct Tmp {
        std::vector<void*> p;
        ~Tmp() { for (void* q : p) (void)hipFree(q); }
    } tmp;
    auto up = [&](const void* h, size_t bytes, void** d) -> int {
        *d = nullptr;
        if (!h) return OPD_OK;
        if (hipMalloc(d, bytes) != hipSuccess) return fail(OPD_ENOMEM, "opd_similarity_matrix: device allocation failed");
        tmp.p.push_back(*d);
        HIPCHK(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
        return OPD_OK;
    };
    void *df1, *df2, *db1, *db2, *dh1, *dh2, *dout = nullptr;
    RCCHK(up(feats1, (size_t)n1 * D * 4, &df1)); RCCHK(up(feats2, (size_t)n2 * D * 4, &df2));
    RCCHK(up(boxes1, (size_t)n1 * 16, &db1)); RCCHK(up(boxes2, (size_t)n2 * 16, &db2));
    RCCHK(up(has1, (size_t)n1, &dh1)); RCCHK(up(has2, (size_t)n2, &dh2));
    if (hipMalloc(&dout, (size_t)n1 * n2 * 4) != hipSuccess) return fail(OPD_ENOMEM, "opd_similarity_matrix: device allocation failed");
    tmp.p.push_back(dout);
    HIPCHK(opd_launch_similarity_matrix((const float*)df1, (const float*)db1, (const uint8_t*)dh1, n1, (const float*)df2, (const float*)db2,
                                        (const uint8_t*)dh2, n2, D, appearance_weight, motion_weight, as_distance,
                                        (float*)dout, nullptr));
    HIPCHK(hipMemcpy(out, dout, (size_t)n1 * n2 * 4, hipMemcpyDeviceToHost));
    return OPD_OK;
}

int opd_detr_roi_features(opd_detr* m, int frame, const float* boxes_xywh, int n, int orig_h, int orig_w, float* features) {
    ApiScope api_scope;
    if (!m || (n > 0 && (!boxes_xywh || !features))) return fail(OPD_EINVAL, "opd_detr_roi_features: null argument");
    if (m->last_B == 0) return fail(OPD_ESTATE, "opd_detr_roi_features called before any forward");
    if (frame < 0 || frame >= m->last_B || n < 0 || n > 128 || orig_h <= 0 || orig_w <= 0)
        return fail(OPD_EINVAL, "opd_detr_roi_features: frame / n / image size out of range");
    if (n == 0) return OPD_OK;
    HIPCHK(hipSetDevice(m->device));
    const int h = m->last_fh, w = m->last_fw;
    std::vector<int32_t> rois(n * 4);
    for (int i = 0; i < n; ++i) {  // same int-truncation and clamping as the reference (feature_extractor.py:68-78)
        const double x = boxes_xywh[4 * i], y = boxes_xywh[4 * i + 1], bw = boxes_xywh[4 * i + 2], bh = boxes_xywh[4 * i + 3];
        int x0 = (int)((x / orig_w) * w), y0 = (int)((y / orig_h) * h);
        int x1 = (int)(((x + bw) / orig_w) * w), y1 = (int)(((y + bh) / orig_h) * h);
        x0 = std::max(0, std::min(x0, w - 1)); y0 = std::max(0, std::min(y0, h - 1));
        x1 = std::max(x0 + 1, std::min(x1, w)); y1 = std::max(y0 + 1, std::min(y1, h));
        rois[4 * i] = x0; rois[4 * i + 1] = y0; rois[4 * i + 2] = x1; rois[4 * i + 3] = y1;
    }
    HIPCHK(hipMemcpyAsync(m->d_rois, rois.data(), rois.size() * 4, hipMemcpyHostToDevice, m->stream));
    const float* enc = m->d_x32 + (size_t)frame * h * w * m->arch.d_model;
    HIPCHK(opd_launch_roi_features(enc, m->d_rois, m->d_roi_out, n, h, w, m->stream));
    HIPCHK(hipMemcpyAsync(features, m->d_roi_out, (size_t)n * m->arch.d_model * 4, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return OPD_OK;
}

int opd_detr_attention_map(opd_detr* m, int frame, int layer, const int32_t* queries, int n_queries, float* out, int out_capacity) {
    ApiScope api_scope;
    if (!m || !out) return fail(OPD_EINVAL, "opd_detr_attention_map: null argument");
    if (m->last_B == 0) return fail(OPD_ESTATE, "opd_detr_attention_map called before any forward");
    const int L = m->arch.dec_layers, Q = m->arch.queries, D = m->arch.d_model;
    if (layer < 0) layer += L;
    if (frame < 0 || frame >= m->last_B || layer < 0 || layer >= L || n_queries < 0 || n_queries > Q || (n_queries > 0 && !queries))
        return fail(OPD_EINVAL, "opd_detr_attention_map: frame / layer / queries out of range");
    std::vector<int32_t> sel;
    if (n_queries == 0) { sel.resize(Q); for (int i = 0; i < Q; ++i) sel[i] = i; }
    else {
        sel.assign(queries, queries + n_queries);
        for (int q : sel) if (q < 0 || q >= Q) return fail(OPD_EINVAL, "opd_detr_attention_map: query index out of range");
    }
    HIPCHK(hipSetDevice(m->device));
    const int hw = m->last_fh * m->last_fw, NKV = L * 2 * D, Md = m->last_B * Q;
    if (out_capacity < hw)
        return fail(OPD_EINVAL, "opd_detr_attention_map: the last forward's map has " + std::to_string(m->last_fh) + " x " + std::to_string(m->last_fw) +
                                    " positions, the output buffer holds " + std::to_string(out_capacity));
    HIPCHK(hipMemcpyAsync(m->d_amap_sel, sel.data(), sel.size() * 4, hipMemcpyHostToDevice, m->stream));
    const f16_t* q = m->d_qd16 + (size_t)layer * m->cfg.max_batch * Q * D + (size_t)frame * Q * D;
    const f16_t* k = m->d_memkv16 + (size_t)frame * hw * NKV + (size_t)layer * 2 * D;
    const float scale = 1.0f / sqrtf((float)(D / m->arch.heads));
    HIPCHK(opd_launch_attention_map(q, D, k, NKV, m->d_amap_sel, (int)sel.size(), m->arch.heads, hw, scale,
                                    m->last_ragged ? m->d_key_valid + 2 * frame : nullptr, m->last_fw, m->d_amap_stat, m->d_amap, m->stream, m->dtype));
    HIPCHK(hipMemcpyAsync(out, m->d_amap, (size_t)hw * 4, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return OPD_OK;
}

int opd_detr_set_profiling(opd_detr* m, int enabled) {
    if (!m) return fail(OPD_EINVAL, "null model handle");
    ApiScope api_scope;
    const int mode = enabled == 2 ? 2 : (enabled ? 1 : 0);
    if (mode != m->profiling) {   // the stage marks of mode 2 are nodes of the captured graph: graphs of another mode do not carry them
        HIPCHK(hipSetDevice(m->device));
        HIPCHK(hipStreamSynchronize(m->stream));
        for (auto& g : m->graphs)
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
        m->graphs.clear();
    }
    m->profiling = mode;
    return OPD_OK;
}

int opd_detr_stage_times(const opd_detr* m, float* ms8) {
    ApiScope api_scope;
    if (!m || !ms8) return fail(OPD_EINVAL, "opd_detr_stage_times: null argument");
    for (int i = 0; i < 8; ++i) ms8[i] = m->stage_ms[i];
    return OPD_OK;
}

int opd_detr_kernel_table(const opd_detr* m, opd_kernel_stat* out, int capacity, int* count) {
    ApiScope api_scope;
    if (!m || !count || capacity < 0 || (capacity > 0 && !out)) return fail(OPD_EINVAL, "opd_detr_kernel_table: bad argument");
    *count = (int)m->ktable.size();
    for (int i = 0; i < capacity && i < (int)m->ktable.size(); ++i) {
        const auto& r = m->ktable[i];
        memset(&out[i], 0, sizeof out[i]);
        snprintf(out[i].name, sizeof out[i].name, "%s", r.name.c_str());
        out[i].launches = r.launches; out[i].ms = r.ms; out[i].flops = r.flops;
    }
    return OPD_OK;
}

int opd_detr_kernel_times(const opd_detr* m, float* ms4, int32_t* launches4, double* flops4) {
    ApiScope api_scope;
    if (!m || !ms4 || !launches4 || !flops4) return fail(OPD_EINVAL, "opd_detr_kernel_times: null argument");
    for (int i = 0; i < 4; ++i) { ms4[i] = m->class_ms[i]; launches4[i] = m->class_launches[i]; flops4[i] = m->class_flops[i]; }
    return OPD_OK;
}

}  // extern "C"

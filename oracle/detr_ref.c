/*
 * ORACLE (test infrastructure, NOT product code): plain-C fp32 restatement of the Phase-2 DETR detect path.
 *
 * Second, torch-free restatement next to oracle/detr_oracle.py; both are pinned to the same golden vectors captured
 * from the HF module (tests/golden, tools/gen_golden.py; tests/test_oracle_c.py).  Only tests/ may load the library
 * built from this file (oracle/Makefile -> oracle/libdetr_ref.so); the product never does.
 *
 * The reference's arithmetic lives in Hugging Face transformers (pin 4.57.3, requirements.txt:220; not under
 * /root/reference).  Each function cites the HF source it follows ("HF:" = transformers/, 5.15.0 in the build container):
 *   frozen_bn_conv     HF:models/detr/modeling_detr.py:179-215 ; HF:models/resnet/modeling_resnet.py:40-70
 *   stem / bottleneck  HF:models/resnet/modeling_resnet.py:72-93,139-178
 *   sine_pos           HF:models/detr/modeling_detr.py:294-368   (all-ones mask: equal-size batches)
 *   mha / layers       HF:models/detr/modeling_detr.py:402-427,430-573,593-739
 *   heads              HF:models/detr/modeling_detr.py:1284-1300,1410-1411
 *   postprocess        HF:models/detr/image_processing_detr.py:805-856
 *
 * Layout: NCHW fp32 for the backbone (like the reference), [tokens][256] for the transformer.  Weights arrive as a
 * table of named fp32 tensors (the caller looks them up by the HF state-dict names).  Straight loops, OpenMP over the
 * outermost independent index; no blocking, no vendor library.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define D 256
#define HEADS 8
#define DH 32

typedef struct {
    const char* name;
    const float* data;
} ref_tensor;

typedef struct {
    const ref_tensor* t;
    int n;
} ref_weights;

static const float* W(const ref_weights* w, const char* name) {
    for (int i = 0; i < w->n; ++i)
        if (strcmp(w->t[i].name, name) == 0) return w->t[i].data;
    return NULL;
}

static const float* Wf(const ref_weights* w, const char* prefix, const char* suffix) {
    char buf[256];
    strcpy(buf, prefix);
    strcat(buf, suffix);
    return W(w, buf);
}

/* conv (no bias) + FrozenBN (eps 1e-5, y = x*scale + (b - mean*scale)) + optional ReLU; x [C][H][W] -> y [O][OH][OW] */
static void frozen_bn_conv(const ref_weights* w, const char* prefix, const float* x, int C, int H, int Wd, int O, int k,
                           int stride, int relu, float* y, int OH, int OW) {
    const float* cw = Wf(w, prefix, ".convolution.weight");
    const float* g = Wf(w, prefix, ".normalization.weight");
    const float* bt = Wf(w, prefix, ".normalization.bias");
    const float* mu = Wf(w, prefix, ".normalization.running_mean");
    const float* var = Wf(w, prefix, ".normalization.running_var");
    const int pad = k / 2;
#pragma omp parallel for schedule(dynamic)
    for (int o = 0; o < O; ++o) {
        float* yo = y + (size_t)o * OH * OW;
        for (int i = 0; i < OH * OW; ++i) yo[i] = 0.f;
        for (int c = 0; c < C; ++c)
            for (int kh = 0; kh < k; ++kh)
                for (int kw = 0; kw < k; ++kw) {
                    const float wv = cw[(((size_t)o * C + c) * k + kh) * k + kw];
                    for (int oh = 0; oh < OH; ++oh) {
                        const int ih = oh * stride - pad + kh;
                        if (ih < 0 || ih >= H) continue;
                        const float* xr = x + ((size_t)c * H + ih) * Wd;
                        float* yr = yo + (size_t)oh * OW;
                        for (int ow = 0; ow < OW; ++ow) {
                            const int iw = ow * stride - pad + kw;
                            if (iw >= 0 && iw < Wd) yr[ow] += wv * xr[iw];
                        }
                    }
                }
        const float scale = g[o] * (1.0f / sqrtf(var[o] + 1e-5f));
        const float bias = bt[o] - mu[o] * scale;
        for (int i = 0; i < OH * OW; ++i) {
            float v = yo[i] * scale + bias;
            yo[i] = (relu && v < 0.f) ? 0.f : v;
        }
    }
}

static int down2(int n) { return (n - 1) / 2 + 1; }

/* y[rows][N] = x[rows][K] . Wt[N][K]^T + b */
static void linear(const float* x, const float* wt, const float* b, float* y, int rows, int N, int K) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r)
        for (int n = 0; n < N; ++n) {
            const float* xr = x + (size_t)r * K;
            const float* wr = wt + (size_t)n * K;
            float acc = 0.f;
            for (int k = 0; k < K; ++k) acc += xr[k] * wr[k];
            y[(size_t)r * N + n] = acc + b[n];
        }
}

static void layer_norm(float* x, const float* g, const float* b, int rows) {
    for (int r = 0; r < rows; ++r) {
        float* v = x + (size_t)r * D;
        float mean = 0.f, var = 0.f;
        for (int i = 0; i < D; ++i) mean += v[i];
        mean /= D;
        for (int i = 0; i < D; ++i) var += (v[i] - mean) * (v[i] - mean);
        var /= D;
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        for (int i = 0; i < D; ++i) v[i] = (v[i] - mean) * rstd * g[i] + b[i];
    }
}

/* out[Lq][D] = o_proj( softmax(q k^T / sqrt(32)) v ), q = Lin(qin), k = Lin(kin), v = Lin(vin) */
static void mha(const ref_weights* w, const char* prefix, const float* qin, const float* kin, const float* vin, int Lq, int Lk,
                float* out) {
    float* q = malloc((size_t)Lq * D * 4);
    float* k = malloc((size_t)Lk * D * 4);
    float* v = malloc((size_t)Lk * D * 4);
    float* ctx = malloc((size_t)Lq * D * 4);
    linear(qin, Wf(w, prefix, ".q_proj.weight"), Wf(w, prefix, ".q_proj.bias"), q, Lq, D, D);
    linear(kin, Wf(w, prefix, ".k_proj.weight"), Wf(w, prefix, ".k_proj.bias"), k, Lk, D, D);
    linear(vin, Wf(w, prefix, ".v_proj.weight"), Wf(w, prefix, ".v_proj.bias"), v, Lk, D, D);
    const float scale = 1.0f / sqrtf((float)DH);
#pragma omp parallel for collapse(2) schedule(static)
    for (int h = 0; h < HEADS; ++h)
        for (int i = 0; i < Lq; ++i) {
            float* s = malloc((size_t)Lk * 4);
            float mx = -INFINITY;
            for (int j = 0; j < Lk; ++j) {
                float acc = 0.f;
                for (int d = 0; d < DH; ++d) acc += q[(size_t)i * D + h * DH + d] * k[(size_t)j * D + h * DH + d];
                s[j] = acc * scale;
                if (s[j] > mx) mx = s[j];
            }
            float sum = 0.f;
            for (int j = 0; j < Lk; ++j) { s[j] = expf(s[j] - mx); sum += s[j]; }
            for (int d = 0; d < DH; ++d) {
                float acc = 0.f;
                for (int j = 0; j < Lk; ++j) acc += (s[j] / sum) * v[(size_t)j * D + h * DH + d];
                ctx[(size_t)i * D + h * DH + d] = acc;
            }
            free(s);
        }
    linear(ctx, Wf(w, prefix, ".o_proj.weight"), Wf(w, prefix, ".o_proj.bias"), out, Lq, D, D);
    free(q); free(k); free(v); free(ctx);
}

static void mlp(const ref_weights* w, const char* prefix, const float* x, int rows, int F, float* out) {
    float* h = malloc((size_t)rows * F * 4);
    linear(x, Wf(w, prefix, ".mlp.fc1.weight"), Wf(w, prefix, ".mlp.fc1.bias"), h, rows, F, D);
    for (size_t i = 0; i < (size_t)rows * F; ++i) h[i] = h[i] > 0.f ? h[i] : 0.f;
    linear(h, Wf(w, prefix, ".mlp.fc2.weight"), Wf(w, prefix, ".mlp.fc2.bias"), out, rows, D, F);
    free(h);
}

static void sine_pos(int h, int wd, float* pos) {
    const int npf = D / 2;
    const float scale = 6.283185307179586f, eps = 1e-6f;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < wd; ++x) {
            const float ye = (float)(y + 1) / ((float)h + eps) * scale;
            const float xe = (float)(x + 1) / ((float)wd + eps) * scale;
            float* p = pos + ((size_t)y * wd + x) * D;
            for (int i = 0; i < npf; ++i) {
                const float dim_t = powf(10000.0f, (2.0f * (float)(i / 2)) / (float)npf);
                p[i] = (i & 1) ? cosf(ye / dim_t) : sinf(ye / dim_t);
                p[npf + i] = (i & 1) ? cosf(xe / dim_t) : sinf(xe / dim_t);
            }
        }
}

/*
 * One frame: pixel_values [3][H][W] (already normalised) -> logits [Q][ncls], boxes [Q][4] (cx,cy,w,h), encoder [hw][256].
 * depths[4], encoder/decoder layer counts, queries and ncls describe the checkpoint.  Returns 0, or -1 on a missing tensor.
 */
int detr_ref_forward(const ref_tensor* tensors, int n_tensors, const float* pixel_values, int H, int Wd, const int* depths,
                     int enc_layers, int dec_layers, int queries, int ncls, int ffn, float* logits, float* boxes, float* enc_out) {
    ref_weights ws = {tensors, n_tensors};
    const ref_weights* w = &ws;
    if (!W(w, "model.input_projection.weight") || !W(w, "class_labels_classifier.weight")) return -1;
    char p[256], q[256];
    /* ---- backbone ---- */
    int h = down2(H), wd = down2(Wd);
    float* x = malloc((size_t)64 * h * wd * 4);
    frozen_bn_conv(w, "model.backbone.model.embedder.embedder", pixel_values, 3, H, Wd, 64, 7, 2, 1, x, h, wd);
    {   /* MaxPool2d(3, 2, 1) */
        const int oh = down2(h), ow = down2(wd);
        float* y = malloc((size_t)64 * oh * ow * 4);
        for (int c = 0; c < 64; ++c)
            for (int i = 0; i < oh; ++i)
                for (int j = 0; j < ow; ++j) {
                    float m = -INFINITY;
                    for (int a = 0; a < 3; ++a)
                        for (int b = 0; b < 3; ++b) {
                            const int ii = 2 * i - 1 + a, jj = 2 * j - 1 + b;
                            if (ii >= 0 && ii < h && jj >= 0 && jj < wd) {
                                const float v = x[((size_t)c * h + ii) * wd + jj];
                                if (v > m) m = v;
                            }
                        }
                    y[((size_t)c * oh + i) * ow + j] = m;
                }
        free(x); x = y; h = oh; wd = ow;
    }
    int cin = 64;
    const int hidden[4] = {256, 512, 1024, 2048};
    for (int s = 0; s < 4; ++s) {
        const int cout = hidden[s], mid = cout / 4;
        for (int l = 0; l < depths[s]; ++l) {
            snprintf(p, sizeof p, "model.backbone.model.encoder.stages.%d.layers.%d", s, l);
            const int stride = (l == 0 && s > 0) ? 2 : 1;
            const int oh = stride == 2 ? down2(h) : h, ow = stride == 2 ? down2(wd) : wd;
            float* res = x;
            int own_res = 0;
            snprintf(q, sizeof q, "%s.shortcut.convolution.weight", p);
            if (W(w, q)) {
                res = malloc((size_t)cout * oh * ow * 4);
                own_res = 1;
                snprintf(q, sizeof q, "%s.shortcut", p);
                frozen_bn_conv(w, q, x, cin, h, wd, cout, 1, stride, 0, res, oh, ow);
            }
            float* a = malloc((size_t)mid * h * wd * 4);
            float* b = malloc((size_t)mid * oh * ow * 4);
            float* c = malloc((size_t)cout * oh * ow * 4);
            snprintf(q, sizeof q, "%s.layer.0", p);
            frozen_bn_conv(w, q, x, cin, h, wd, mid, 1, 1, 1, a, h, wd);
            snprintf(q, sizeof q, "%s.layer.1", p);
            frozen_bn_conv(w, q, a, mid, h, wd, mid, 3, stride, 1, b, oh, ow);
            snprintf(q, sizeof q, "%s.layer.2", p);
            frozen_bn_conv(w, q, b, mid, oh, ow, cout, 1, 1, 0, c, oh, ow);
            for (size_t i = 0; i < (size_t)cout * oh * ow; ++i) {
                const float v = c[i] + res[i];
                c[i] = v > 0.f ? v : 0.f;
            }
            free(a); free(b);
            if (own_res) free(res);
            free(x);
            x = c; cin = cout; h = oh; wd = ow;
        }
    }
    /* ---- input projection (1x1 conv with bias) -> tokens [hw][256] ---- */
    const int hw = h * wd;
    float* tok = malloc((size_t)hw * D * 4);
    {
        const float* pw = W(w, "model.input_projection.weight");
        const float* pb = W(w, "model.input_projection.bias");
#pragma omp parallel for schedule(static)
        for (int t = 0; t < hw; ++t)
            for (int o = 0; o < D; ++o) {
                float acc = 0.f;
                for (int c = 0; c < cin; ++c) acc += pw[(size_t)o * cin + c] * x[(size_t)c * hw + t];
                tok[(size_t)t * D + o] = acc + pb[o];
            }
    }
    free(x);
    float* pos = malloc((size_t)hw * D * 4);
    sine_pos(h, wd, pos);
    float* tmp = malloc((size_t)hw * D * 4);
    float* att = malloc((size_t)hw * D * 4);
    /* ---- encoder: post-LN layers ---- */
    for (int i = 0; i < enc_layers; ++i) {
        snprintf(p, sizeof p, "model.encoder.layers.%d", i);
        for (size_t e = 0; e < (size_t)hw * D; ++e) tmp[e] = tok[e] + pos[e];
        snprintf(q, sizeof q, "%s.self_attn", p);
        mha(w, q, tmp, tmp, tok, hw, hw, att);
        for (size_t e = 0; e < (size_t)hw * D; ++e) tok[e] += att[e];
        layer_norm(tok, Wf(w, p, ".self_attn_layer_norm.weight"), Wf(w, p, ".self_attn_layer_norm.bias"), hw);
        mlp(w, p, tok, hw, ffn, att);
        for (size_t e = 0; e < (size_t)hw * D; ++e) tok[e] += att[e];
        layer_norm(tok, Wf(w, p, ".final_layer_norm.weight"), Wf(w, p, ".final_layer_norm.bias"), hw);
    }
    memcpy(enc_out, tok, (size_t)hw * D * 4);
    /* ---- decoder ---- */
    const float* qpos = W(w, "model.query_position_embeddings.weight");
    float* hq = calloc((size_t)queries * D, 4);
    float* qk = malloc((size_t)queries * D * 4);
    float* da = malloc((size_t)queries * D * 4);
    for (size_t e = 0; e < (size_t)hw * D; ++e) tmp[e] = tok[e] + pos[e]; /* memory + pos: keys of every cross-attention */
    for (int i = 0; i < dec_layers; ++i) {
        snprintf(p, sizeof p, "model.decoder.layers.%d", i);
        for (size_t e = 0; e < (size_t)queries * D; ++e) qk[e] = hq[e] + qpos[e];
        snprintf(q, sizeof q, "%s.self_attn", p);
        mha(w, q, qk, qk, hq, queries, queries, da);
        for (size_t e = 0; e < (size_t)queries * D; ++e) hq[e] += da[e];
        layer_norm(hq, Wf(w, p, ".self_attn_layer_norm.weight"), Wf(w, p, ".self_attn_layer_norm.bias"), queries);
        for (size_t e = 0; e < (size_t)queries * D; ++e) qk[e] = hq[e] + qpos[e];
        snprintf(q, sizeof q, "%s.encoder_attn", p);
        mha(w, q, qk, tmp, tok, queries, hw, da);
        for (size_t e = 0; e < (size_t)queries * D; ++e) hq[e] += da[e];
        layer_norm(hq, Wf(w, p, ".encoder_attn_layer_norm.weight"), Wf(w, p, ".encoder_attn_layer_norm.bias"), queries);
        mlp(w, p, hq, queries, ffn, da);
        for (size_t e = 0; e < (size_t)queries * D; ++e) hq[e] += da[e];
        layer_norm(hq, Wf(w, p, ".final_layer_norm.weight"), Wf(w, p, ".final_layer_norm.bias"), queries);
    }
    layer_norm(hq, W(w, "model.decoder.layernorm.weight"), W(w, "model.decoder.layernorm.bias"), queries);
    /* ---- heads ---- */
    linear(hq, W(w, "class_labels_classifier.weight"), W(w, "class_labels_classifier.bias"), logits, queries, ncls, D);
    float* b1 = malloc((size_t)queries * D * 4);
    float* b2 = malloc((size_t)queries * D * 4);
    linear(hq, W(w, "bbox_predictor.layers.0.weight"), W(w, "bbox_predictor.layers.0.bias"), b1, queries, D, D);
    for (size_t e = 0; e < (size_t)queries * D; ++e) b1[e] = b1[e] > 0.f ? b1[e] : 0.f;
    linear(b1, W(w, "bbox_predictor.layers.1.weight"), W(w, "bbox_predictor.layers.1.bias"), b2, queries, D, D);
    for (size_t e = 0; e < (size_t)queries * D; ++e) b2[e] = b2[e] > 0.f ? b2[e] : 0.f;
    linear(b2, W(w, "bbox_predictor.layers.2.weight"), W(w, "bbox_predictor.layers.2.bias"), boxes, queries, 4, D);
    for (int e = 0; e < queries * 4; ++e) boxes[e] = 1.0f / (1.0f + expf(-boxes[e]));
    free(b1); free(b2); free(hq); free(qk); free(da); free(tok); free(pos); free(tmp); free(att);
    return 0;
}

/* HF post_process_object_detection for one frame: softmax, max over the first ncls-1 classes, cxcywh -> xyxy * (W,H,W,H),
 * keep score > threshold.  out rows = (x1,y1,x2,y2,score,label,query); returns the number kept. */
int detr_ref_postprocess(const float* logits, const float* boxes, int queries, int ncls, float threshold, int img_h, int img_w,
                         float* out) {
    int n = 0;
    for (int qi = 0; qi < queries; ++qi) {
        const float* lg = logits + (size_t)qi * ncls;
        float mx = lg[0];
        for (int c = 1; c < ncls; ++c) if (lg[c] > mx) mx = lg[c];
        float sum = 0.f, best = -1.f;
        int label = 0;
        for (int c = 0; c < ncls; ++c) {
            const float e = expf(lg[c] - mx);
            sum += e;
            if (c < ncls - 1 && e > best) { best = e; label = c; }
        }
        const float score = best / sum;
        if (!(score > threshold)) continue;
        const float* b = boxes + (size_t)qi * 4;
        float* o = out + (size_t)n * 7;
        o[0] = (b[0] - 0.5f * b[2]) * img_w; o[1] = (b[1] - 0.5f * b[3]) * img_h;
        o[2] = (b[0] + 0.5f * b[2]) * img_w; o[3] = (b[1] + 0.5f * b[3]) * img_h;
        o[4] = score; o[5] = (float)label; o[6] = (float)qi;
        ++n;
    }
    return n;
}

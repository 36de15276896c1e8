/*
 * ORACLE (test infrastructure, NOT product code): the C restatement of oracle/detr_ref.c behind the SAME C-ABI as the
 * product library (include/opd_detr.h) — SURVEY.md section 8(b), last row / section 7 step 2: "the same symbols implemented
 * by libopd_ref (CPU) and libopd_hip".  Built by oracle/Makefile into oracle/libopd_ref.so; loaded only by tests/, which
 * drive both libraries through identical calls (tests/test_ref_abi.py).  The product never loads it: libopd_hip.so has no
 * CPU fallback.
 *
 * Implemented: opd_version, opd_last_error, opd_detr_create / destroy / info, opd_detr_forward, opd_detr_postprocess,
 * opd_detr_detect, opd_person_nms(_batch) — host memory only (OPD_MEM_HOST), frames of one size, fp32 safetensors with HF 5.x
 * key names.  Not implemented here (they have no CPU meaning or are covered by the torch oracle): clones, asynchronous
 * submission, device buffers, device resize, ragged batches, ROI features, the similarity matrix, profiling hooks.
 *
 * Arithmetic: pre-processing as DetrImageProcessor (HF:models/detr/image_processing_detr.py:752-791: BGR -> RGB, 1/255,
 * ImageNet mean / std), the model through detr_ref_forward, post_process_object_detection through detr_ref_postprocess,
 * the person filter + greedy IoU-NMS as oracle/detr_oracle.py::person_detections (docs/plan.md:30, config.yaml.disabled:38).
 */
#define _POSIX_C_SOURCE 200809L   /* strdup */
#include <ctype.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/opd_detr.h"

typedef struct {
    const char* name;
    const float* data;
} ref_tensor;
int detr_ref_forward(const ref_tensor* tensors, int n_tensors, const float* pixel_values, int H, int Wd, const int* depths,
                     int enc_layers, int dec_layers, int queries, int ncls, int ffn, float* logits, float* boxes, float* enc_out);
int detr_ref_postprocess(const float* logits, const float* boxes, int queries, int ncls, float threshold, int img_h, int img_w,
                         float* out);

static _Thread_local char g_err[512];
static int fail(int code, const char* fmt, const char* arg) {
    snprintf(g_err, sizeof g_err, fmt, arg ? arg : "");
    return code;
}

struct opd_detr {
    opd_config cfg;
    unsigned char* blob;   /* the tensor bytes of the file */
    ref_tensor* t;
    char** names;
    int64_t (*shapes)[4];
    int n;
    int depths[4], enc_layers, dec_layers, queries, ncls, ffn;
    float *logits, *boxes;  /* outputs of the last forward */
    int last_B, last_H, last_W;
};

const char* opd_last_error(void) { return g_err; }
const char* opd_version(void) { return "opd_ref 0.1 CPU fp32 (oracle/detr_ref.c behind the opd_detr C-ABI; test infrastructure)"; }

static int find(const opd_detr* m, const char* name) {
    for (int i = 0; i < m->n; ++i)
        if (strcmp(m->names[i], name) == 0) return i;
    return -1;
}

/* safetensors: u64 header length, JSON {"name": {"dtype": "F32", "shape": [..], "data_offsets": [a, b]}, ...}, raw bytes */
static int parse_header(opd_detr* m, char* h, size_t hl, size_t blob_len) {
    int cap = 1024;
    m->t = calloc(cap, sizeof(ref_tensor));
    m->names = calloc(cap, sizeof(char*));
    m->shapes = calloc(cap, sizeof(*m->shapes));
    char* p = h;
    char* end = h + hl;
    while (p < end && *p != '{') ++p;
    ++p;
    while (p < end) {
        while (p < end && *p != '"' && *p != '}') ++p;
        if (p >= end || *p == '}') break;
        char* name = ++p;
        while (p < end && *p != '"') ++p;
        *p++ = 0;
        while (p < end && *p != '{' && *p != '"') ++p;   /* value: an object ("__metadata__" too) */
        if (p >= end) return -1;
        char* obj = p;
        int depth = 0;
        do {
            if (*p == '{') ++depth;
            if (*p == '}') --depth;
            ++p;
        } while (p < end && depth > 0);
        char save = *p;
        *p = 0;
        if (strcmp(name, "__metadata__") != 0) {
            const char* dt = strstr(obj, "\"dtype\"");
            const char* sh = strstr(obj, "\"shape\"");
            const char* of = strstr(obj, "\"data_offsets\"");
            if (!dt || !sh || !of) return -1;
            if (!strstr(dt, "\"F32\"")) return -2;   /* the oracle reads fp32 checkpoints only */
            if (m->n == cap) return -1;
            int64_t* s = m->shapes[m->n];
            s[0] = s[1] = s[2] = s[3] = 1;
            const char* q = strchr(sh, '[') + 1;
            for (int d = 0; d < 4 && *q && *q != ']'; ++d) {
                s[d] = strtoll(q, (char**)&q, 10);
                while (*q == ',' || *q == ' ') ++q;
            }
            q = strchr(of, '[') + 1;
            const long long a = strtoll(q, (char**)&q, 10);
            while (*q == ',' || *q == ' ') ++q;
            const long long b = strtoll(q, (char**)&q, 10);
            if (a < 0 || b < a || (size_t)b > blob_len) return -1;
            m->names[m->n] = strdup(name);
            m->t[m->n].name = m->names[m->n];
            m->t[m->n].data = (const float*)(m->blob + a);
            ++m->n;
        }
        *p = save;
    }
    return 0;
}

int opd_detr_create(const opd_config* cfg, const char* weights_path, int device_ordinal, opd_detr** out) {
    (void)device_ordinal;
    if (!cfg || !weights_path || !out) return fail(OPD_EINVAL, "opd_detr_create: null argument%s", NULL);
    *out = NULL;
    if (cfg->struct_size != (int32_t)sizeof(opd_config)) return fail(OPD_EINVAL, "opd_config.struct_size mismatch%s", NULL);
    FILE* f = fopen(weights_path, "rb");
    if (!f) return fail(OPD_EIO, "cannot open weight file '%s'", weights_path);
    uint64_t hl = 0;
    if (fread(&hl, 8, 1, f) != 1 || hl == 0 || hl > (1ull << 30)) { fclose(f); return fail(OPD_EIO, "'%s' is not a safetensors file", weights_path); }
    char* h = malloc(hl + 1);
    if (fread(h, 1, hl, f) != hl) { fclose(f); free(h); return fail(OPD_EIO, "truncated safetensors header in '%s'", weights_path); }
    h[hl] = 0;
    fseek(f, 0, SEEK_END);
    const long fsize = ftell(f);
    const size_t blob_len = (size_t)fsize - 8 - hl;
    opd_detr* m = calloc(1, sizeof(opd_detr));
    m->cfg = *cfg;
    m->blob = malloc(blob_len ? blob_len : 1);
    fseek(f, (long)(8 + hl), SEEK_SET);
    if (fread(m->blob, 1, blob_len, f) != blob_len) { fclose(f); free(h); return fail(OPD_EIO, "short read of '%s'", weights_path); }
    fclose(f);
    const int rc = parse_header(m, h, hl, blob_len);
    free(h);
    if (rc) return fail(rc == -2 ? OPD_ESCHEMA : OPD_EIO, "malformed or non-fp32 safetensors header in '%s'", weights_path);
    char key[256];
    for (int s = 0; s < 4; ++s) {
        int d = 0;
        for (;; ++d) {
            snprintf(key, sizeof key, "model.backbone.model.encoder.stages.%d.layers.%d.layer.0.convolution.weight", s, d);
            if (find(m, key) < 0) break;
        }
        if (d == 0) return fail(OPD_ESCHEMA, "weight file has no ResNet stage (not a DETR-ResNet checkpoint?)%s", NULL);
        m->depths[s] = d;
    }
    for (;; ++m->enc_layers) {
        snprintf(key, sizeof key, "model.encoder.layers.%d.self_attn.q_proj.weight", m->enc_layers);
        if (find(m, key) < 0) break;
    }
    for (;; ++m->dec_layers) {
        snprintf(key, sizeof key, "model.decoder.layers.%d.self_attn.q_proj.weight", m->dec_layers);
        if (find(m, key) < 0) break;
    }
    const int iq = find(m, "model.query_position_embeddings.weight"), ic = find(m, "class_labels_classifier.weight"),
              iff = find(m, "model.encoder.layers.0.mlp.fc1.weight");
    if (iq < 0 || ic < 0 || iff < 0 || m->enc_layers == 0 || m->dec_layers == 0)
        return fail(OPD_ESCHEMA, "weight file lacks tensor 'model.query_position_embeddings.weight' / classifier / encoder layers%s", NULL);
    m->queries = (int)m->shapes[iq][0];
    m->ncls = (int)m->shapes[ic][0];
    m->ffn = (int)m->shapes[iff][0];
    *out = m;
    return OPD_OK;
}

void opd_detr_destroy(opd_detr* m) {
    if (!m) return;
    for (int i = 0; i < m->n; ++i) free(m->names[i]);
    free(m->names); free(m->t); free(m->shapes); free(m->blob); free(m->logits); free(m->boxes);
    free(m);
}

int opd_detr_info(const opd_detr* m, opd_model_info* info) {
    if (!m || !info) return fail(OPD_EINVAL, "opd_detr_info: null argument%s", NULL);
    memset(info, 0, sizeof *info);
    for (int i = 0; i < 4; ++i) info->depths[i] = m->depths[i];
    info->d_model = 256; info->heads = 8; info->ffn_dim = m->ffn;
    info->encoder_layers = m->enc_layers; info->decoder_layers = m->dec_layers;
    info->num_queries = m->queries; info->num_classes_plus1 = m->ncls;
    info->max_batch = m->cfg.max_batch; info->max_height = m->cfg.max_height; info->max_width = m->cfg.max_width;
    info->device_ordinal = -1;
    return OPD_OK;
}

static int down5(int n) {
    for (int i = 0; i < 5; ++i) n = (n - 1) / 2 + 1;
    return n;
}

int opd_detr_forward(opd_detr* m, const void* pixels, int pixel_format, int mem_kind, int B, int H, int W, float* logits, float* boxes,
                     float* enc_features) {
    if (!m || !pixels) return fail(OPD_EINVAL, "opd_detr_forward: null argument%s", NULL);
    if (mem_kind != OPD_MEM_HOST) return fail(OPD_EINVAL, "libopd_ref handles host memory only%s", NULL);
    if (B < 1 || B > m->cfg.max_batch || H < 32 || W < 32) return fail(OPD_EINVAL, "frame batch outside the configured maximum%s", NULL);
    static const float mean[3] = {0.485f, 0.456f, 0.406f}, std[3] = {0.229f, 0.224f, 0.225f};
    const size_t HW = (size_t)H * W;
    const int hw = down5(H) * down5(W);
    float* pv = malloc(3 * HW * sizeof(float));
    float* enc = malloc((size_t)hw * 256 * sizeof(float));
    m->logits = realloc(m->logits, (size_t)B * m->queries * m->ncls * sizeof(float));
    m->boxes = realloc(m->boxes, (size_t)B * m->queries * 4 * sizeof(float));
    int rc = 0;
    for (int b = 0; b < B && rc == 0; ++b) {
        if (pixel_format == OPD_PIXELS_U8_BGR_HWC) {
            const uint8_t* s = (const uint8_t*)pixels + (size_t)b * HW * 3;
            for (size_t i = 0; i < HW; ++i)
                for (int c = 0; c < 3; ++c)   /* RGB plane c <- BGR byte 2 - c; x / 255 as a multiply by 1/255 like the rescale step */
                    pv[(size_t)c * HW + i] = ((float)s[i * 3 + (2 - c)] * (1.0f / 255.0f) - mean[c]) / std[c];
        } else if (pixel_format == OPD_PIXELS_F32_NCHW) {
            memcpy(pv, (const float*)pixels + (size_t)b * 3 * HW, 3 * HW * sizeof(float));
        } else {
            rc = fail(OPD_EINVAL, "unknown pixel_format%s", NULL);
            break;
        }
        float* lg = m->logits + (size_t)b * m->queries * m->ncls;
        float* bx = m->boxes + (size_t)b * m->queries * 4;
        if (detr_ref_forward(m->t, m->n, pv, H, W, m->depths, m->enc_layers, m->dec_layers, m->queries, m->ncls, m->ffn, lg, bx, enc))
            rc = fail(OPD_ESCHEMA, "weight file lacks a tensor of the DETR state dict%s", NULL);
        if (enc_features) memcpy(enc_features + (size_t)b * hw * 256, enc, (size_t)hw * 256 * sizeof(float));
    }
    free(pv); free(enc);
    if (rc) return rc;
    if (logits) memcpy(logits, m->logits, (size_t)B * m->queries * m->ncls * sizeof(float));
    if (boxes) memcpy(boxes, m->boxes, (size_t)B * m->queries * 4 * sizeof(float));
    m->last_B = B; m->last_H = H; m->last_W = W;
    return OPD_OK;
}

int opd_detr_postprocess(opd_detr* m, float threshold, const int32_t* orig_hw, opd_det* out, int32_t* counts) {
    if (!m || !out || !counts) return fail(OPD_EINVAL, "opd_detr_postprocess: null argument%s", NULL);
    if (m->last_B == 0) return fail(OPD_ESTATE, "opd_detr_postprocess called before any forward%s", NULL);
    float* rows = malloc((size_t)m->queries * 7 * sizeof(float));
    for (int b = 0; b < m->last_B; ++b) {
        const int h = orig_hw ? orig_hw[2 * b] : m->last_H, w = orig_hw ? orig_hw[2 * b + 1] : m->last_W;
        const int n = detr_ref_postprocess(m->logits + (size_t)b * m->queries * m->ncls, m->boxes + (size_t)b * m->queries * 4, m->queries,
                                           m->ncls, threshold, h, w, rows);
        for (int i = 0; i < n; ++i) {
            opd_det* d = out + (size_t)b * m->queries + i;
            d->x1 = rows[i * 7]; d->y1 = rows[i * 7 + 1]; d->x2 = rows[i * 7 + 2]; d->y2 = rows[i * 7 + 3];
            d->score = rows[i * 7 + 4]; d->label = (int32_t)rows[i * 7 + 5]; d->query_index = (int32_t)rows[i * 7 + 6]; d->frame = b;
        }
        counts[b] = n;
    }
    free(rows);
    return OPD_OK;
}

int opd_detr_detect(opd_detr* m, const void* pixels, int pixel_format, int mem_kind, int B, int H, int W, float threshold,
                    const int32_t* orig_hw, opd_det* out, int32_t* counts) {
    const int rc = opd_detr_forward(m, pixels, pixel_format, mem_kind, B, H, W, NULL, NULL, NULL);
    return rc ? rc : opd_detr_postprocess(m, threshold, orig_hw, out, counts);
}

static float iou(const opd_det* a, const opd_det* b) {
    const float ix1 = fmaxf(a->x1, b->x1), iy1 = fmaxf(a->y1, b->y1), ix2 = fminf(a->x2, b->x2), iy2 = fminf(a->y2, b->y2);
    const float inter = fmaxf(0.f, ix2 - ix1) * fmaxf(0.f, iy2 - iy1);
    const float ua = fmaxf(0.f, a->x2 - a->x1) * fmaxf(0.f, a->y2 - a->y1) + fmaxf(0.f, b->x2 - b->x1) * fmaxf(0.f, b->y2 - b->y1) - inter;
    return ua > 0.f ? inter / ua : 0.f;
}

int opd_person_nms(opd_det* dets, int n, int person_label, float nms_threshold) {
    if (n < 0 || (n > 0 && !dets)) return fail(OPD_EINVAL, "opd_person_nms: bad arguments%s", NULL);
    opd_det* cand = malloc((size_t)(n ? n : 1) * sizeof(opd_det));
    int nc = 0;
    for (int i = 0; i < n; ++i)
        if (person_label < 0 || dets[i].label == person_label) cand[nc++] = dets[i];
    for (int i = 1; i < nc; ++i) {   /* stable insertion sort by descending score */
        const opd_det v = cand[i];
        int j = i;
        while (j > 0 && cand[j - 1].score < v.score) { cand[j] = cand[j - 1]; --j; }
        cand[j] = v;
    }
    int kept = 0;
    for (int i = 0; i < nc; ++i) {
        int ok = 1;
        if (nms_threshold < 1.0f)
            for (int k = 0; k < kept && ok; ++k)
                if (iou(&cand[i], &dets[k]) > nms_threshold) ok = 0;
        if (ok) dets[kept++] = cand[i];
    }
    free(cand);
    return kept;
}

int opd_person_nms_batch(opd_det* dets, int32_t* counts, int n_frames, int stride, int person_label, float nms_threshold) {
    if (n_frames < 0 || stride < 0 || (n_frames > 0 && (!dets || !counts))) return fail(OPD_EINVAL, "opd_person_nms_batch: bad arguments%s", NULL);
    for (int f = 0; f < n_frames; ++f) {
        if (counts[f] < 0) continue;
        const int kept = opd_person_nms(dets + (size_t)f * stride, counts[f], person_label, nms_threshold);
        if (kept < 0) return kept;
        counts[f] = kept;
    }
    return OPD_OK;
}

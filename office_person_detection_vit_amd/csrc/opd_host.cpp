// opd_host.cpp — host-only arithmetic of the detect path (opd_host.h): no HIP call, no device memory.  Compiled into libopd_hip.so
// and, on its own with opd_loader.cpp, into the sanitizer build that tests/test_host_sanitized_cpu.py drives.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/opd_detr.h"
#include "opd_host.h"
#include "opd_loader.h"

#include <stdarg.h>
#include <stdio.h>
thread_local const char* opd_last_kernel_name = nullptr;
thread_local int opd_dbg_skip_launch = 0;
const char* opd_kernel_name(const char* fmt, ...) {
    char buf[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return strdup(buf);   // (one per kernel instantiation, kept for the life of the process)
}

namespace opd {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

// (opd_host.h) error-diffusion rounding of a weight matrix to fp16 values; the carry is kept in double so that it is exact
void round_f16_diffused(float* w, size_t rows, int taps, int cin, bool bf16) {
    auto rnd = [bf16](float x) { return bf16 ? bf16_to_f32(f32_to_bf16(x)) : f16_to_f32(f32_to_f16(x)); };
    const size_t K = (size_t)taps * cin;
    for (size_t r = 0; r < rows; ++r) {
        float* row = w + r * K;
        double carry = 0.0;
        for (int c = 0; c < cin; ++c)
            for (int t = 0; t < taps; ++t) {
                float& v = row[(size_t)t * cin + c];
                const double target = (double)v + carry;
                const float q = rnd((float)target);
                // (a carry can only push a value over the fp16 range if the value itself was at its edge: keep plain rounding then)
                if (!(fabsf(q) <= (bf16 ? 3.3e38f : 65504.0f))) { v = rnd(v); carry = 0.0; continue; }
                carry = target - (double)q;
                v = q;
            }
    }
}

// Valid extent of a frame on the feature map: the reference down-samples the pixel mask with nearest-neighbour
// interpolation (HF:models/detr/modeling_detr.py:283-289: F.interpolate(mask, size=feature_map.shape[-2:])), i.e. feature
// position i looks at pixel floor(i * in / out) (float32 scale, like ATen's nearest kernel); the mask is a top-left
// rectangle, so the valid feature positions are a prefix.
int valid_prefix(int valid, int in, int out) {
    const float scale = (float)in / (float)out;
    int n = 0;
    for (int i = 0; i < out; ++i) {
        const int src = std::min((int)floorf((float)i * scale), in - 1);
        if (src < valid) ++n;
    }
    return n;
}

// DetrSinePositionEmbedding (HF:models/detr/modeling_detr.py:294-368), fp32 like the reference, for a mask that is a
// top-left rectangle of vh x vw valid positions inside the h x w map (vh == h, vw == w: all-ones mask):
//   y_embed = cumsum(mask, rows) = min(y+1, vh) in valid columns, 0 in padded columns; normalised by its last row (+eps);
//   x_embed = cumsum(mask, cols) = min(x+1, vw) in valid rows, 0 in padded rows; normalised by its last column (+eps).
void sine_pos_embed(int h, int w, int vh, int vw, int D, std::vector<float>* pos) {
    const int npf = D / 2;
    pos->assign((size_t)h * w * D, 0.f);
    const float scale = 6.283185307179586f, eps = 1e-6f;
    std::vector<float> dim_t(npf);
    for (int i = 0; i < npf; ++i) dim_t[i] = powf(10000.0f, (2.0f * (float)(i / 2)) / (float)npf);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float yc = x < vw ? (float)std::min(y + 1, vh) : 0.f, ylast = x < vw ? (float)vh : 0.f;
            const float xc = y < vh ? (float)std::min(x + 1, vw) : 0.f, xlast = y < vh ? (float)vw : 0.f;
            const float ye = yc / (ylast + eps) * scale;
            const float xe = xc / (xlast + eps) * scale;
            float* p = pos->data() + ((size_t)y * w + x) * D;
            for (int i = 0; i < npf; ++i) {
                const float py = ye / dim_t[i], px = xe / dim_t[i];
                p[i] = (i & 1) ? cosf(py) : sinf(py);
                p[npf + i] = (i & 1) ? cosf(px) : sinf(px);
            }
        }
}

}  // namespace opd

using namespace opd;

// Pillow's precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1) filter over the whole axis
// (box = [0, in_size)): per output position the first source index, the tap count, and the 22-bit fixed-point taps.
void opd_resize_coeffs(int in_size, int out_size, std::vector<int32_t>* bounds, std::vector<int32_t>* coeffs, int* ksize_out) {
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    bounds->assign((size_t)out_size * 2, 0);
    coeffs->assign((size_t)out_size * ksize, 0);
    std::vector<double> k(ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale;
        double ww = 0.0;
        const double ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            const double wgt = a < 1.0 ? 1.0 - a : 0.0;
            k[x] = wgt;
            ww += wgt;
        }
        for (int x = 0; x < xmax; ++x) {
            if (ww != 0.0) k[x] /= ww;
            const double v = k[x] * (double)(1 << 22);
            (*coeffs)[(size_t)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + v) : (int)(0.5 + v);
        }
        (*bounds)[2 * xx] = xmin;
        (*bounds)[2 * xx + 1] = xmax;
    }
    *ksize_out = ksize;
}

extern "C" {

const char* opd_last_error(void) { return opd::g_err.c_str(); }

int opd_person_nms(opd_det* dets, int n, int person_label, float nms_threshold) {
    if (n < 0 || (n > 0 && !dets)) return fail(OPD_EINVAL, "opd_person_nms: bad arguments");
    std::vector<int> idx;
    for (int i = 0; i < n; ++i)
        if (person_label < 0 || dets[i].label == person_label) idx.push_back(i);
    // stable sort by descending score (ties keep query order), as the oracle's person_detections
    for (size_t i = 1; i < idx.size(); ++i) {
        const int v = idx[i];
        size_t j = i;
        while (j > 0 && dets[idx[j - 1]].score < dets[v].score) { idx[j] = idx[j - 1]; --j; }
        idx[j] = v;
    }
    auto iou = [](const opd_det& a, const opd_det& b) {
        const float ix1 = fmaxf(a.x1, b.x1), iy1 = fmaxf(a.y1, b.y1), ix2 = fminf(a.x2, b.x2), iy2 = fminf(a.y2, b.y2);
        const float iw = fmaxf(0.f, ix2 - ix1), ih = fmaxf(0.f, iy2 - iy1), inter = iw * ih;
        const float ua = fmaxf(0.f, a.x2 - a.x1) * fmaxf(0.f, a.y2 - a.y1) + fmaxf(0.f, b.x2 - b.x1) * fmaxf(0.f, b.y2 - b.y1) - inter;
        return ua > 0.f ? inter / ua : 0.f;
    };
    std::vector<opd_det> kept;
    for (int i : idx) {
        bool ok = true;
        if (nms_threshold < 1.0f)
            for (const auto& k : kept)
                if (iou(dets[i], k) > nms_threshold) { ok = false; break; }
        if (ok) kept.push_back(dets[i]);
    }
    for (size_t i = 0; i < kept.size(); ++i) dets[i] = kept[i];
    return (int)kept.size();
}

int opd_person_nms_batch(opd_det* dets, int32_t* counts, int n_frames, int stride, int person_label, float nms_threshold) {
    if (n_frames < 0 || stride < 0 || (n_frames > 0 && (!dets || !counts))) return fail(OPD_EINVAL, "opd_person_nms_batch: bad arguments");
    for (int f = 0; f < n_frames; ++f) {
        if (counts[f] < 0) continue;
        if (counts[f] > stride) return fail(OPD_EINVAL, "opd_person_nms_batch: a frame holds more records than its slots");
        const int kept = opd_person_nms(dets + (size_t)f * stride, counts[f], person_label, nms_threshold);
        if (kept < 0) return kept;
        counts[f] = kept;
    }
    return OPD_OK;
}

}  // extern "C"

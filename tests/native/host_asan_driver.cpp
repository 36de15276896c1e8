// Sanitizer driver for the host-only pieces of the detect path (tests/test_host_sanitized_cpu.py builds this file together with
// csrc/opd_loader.cpp and csrc/opd_host.cpp under -fsanitize=address,undefined and runs it on the CPU; never on the GPU box).
//   driver <list-file>      every line of <list-file> is the path of a (possibly malformed) .safetensors file: parse + schema check
// then the arithmetic helpers over ranges of arguments: resize coefficient tables, mask down-sampling, sine position embedding,
// person filter + NMS on adversarial record sets.  Prints one line per file / check; sanitizer reports go to stderr and make the
// process exit non-zero (-fno-sanitize-recover).
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <fstream>
#include <string>
#include <vector>

#include "../../include/opd_detr.h"
#include "../../office_person_detection_vit_amd/csrc/opd_host.h"
#include "../../office_person_detection_vit_amd/csrc/opd_loader.h"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::ifstream list(argv[1]);
    std::string path;
    while (std::getline(list, path)) {
        if (path.empty()) continue;
        opd::StateDict sd;
        std::string err;
        int rc = opd::load_safetensors(path, &sd, &err);
        int rc2 = 1;
        opd::Arch a;
        if (rc == 0) rc2 = opd::infer_arch(sd, &a, &err);
        const size_t slash = path.find_last_of('/');
        printf("file %s parse %d schema %d tensors %zu\n", path.substr(slash == std::string::npos ? 0 : slash + 1).c_str(), rc, rc2, sd.size());
    }
    // key normalisation on odd inputs
    const char* keys[] = {"", ".", "model.backbone.conv_encoder.model.layer1.0.conv1.weight", "model.encoder.layers.0.fc1.weight", "x.out_proj.",
                          "model.backbone.conv_encoder.model.layer9999999999999999999.0.bn1.running_var", "layer1..downsample.0.weight"};
    for (const char* k : keys) printf("key '%s' -> '%s'\n", k, opd::normalise_key(k).c_str());
    // fp16 conversion: every exponent, subnormals, inf / nan
    unsigned long long acc = 0;
    for (uint32_t h = 0; h < 65536; ++h) {
        const float f = opd::f16_to_f32((uint16_t)h);
        if (f == f) acc += opd::f32_to_f16(f) == h;
    }
    const float odd[] = {0.f, -0.f, 65504.f, 65520.f, 1e30f, -1e30f, 5.9e-8f, 2.9e-8f, 1e-45f, INFINITY, -INFINITY, NAN};
    for (float f : odd) acc += opd::f32_to_f16(f);
    printf("f16 round trips %llu\n", acc);
    // Pillow coefficient tables: up- and down-scaling, degenerate sizes
    long long csum = 0;
    for (int in = 1; in <= 97; in += 3)
        for (int out = 1; out <= 97; out += 5) {
            std::vector<int32_t> b, c;
            int ks = 0;
            opd_resize_coeffs(in, out, &b, &c, &ks);
            for (int o = 0; o < out; ++o) {
                if (b[2 * o] < 0 || b[2 * o] + b[2 * o + 1] > in || b[2 * o + 1] > ks) { printf("BAD resize bounds %d %d %d\n", in, out, o); return 3; }
                for (int t = 0; t < b[2 * o + 1]; ++t) csum += c[(size_t)o * ks + t];
            }
        }
    { std::vector<int32_t> b, c; int ks = 0; opd_resize_coeffs(2160, 750, &b, &c, &ks); opd_resize_coeffs(720, 1333, &b, &c, &ks); csum += ks; }
    printf("resize coefficient sum %lld\n", csum);
    // mask down-sampling and position embedding
    long long vsum = 0;
    for (int in = 32; in <= 1333; in += 77)
        for (int out = 1; out <= 42; out += 5)
            for (int v = 1; v <= in; v += 61) vsum += opd::valid_prefix(v, in, out);
    printf("valid prefix sum %lld\n", vsum);
    double psum = 0.0;
    const int shapes[][4] = {{1, 1, 1, 1}, {25, 42, 25, 42}, {25, 42, 1, 1}, {7, 9, 3, 9}, {34, 60, 34, 17}};
    for (const auto& s : shapes) {
        std::vector<float> pos;
        opd::sine_pos_embed(s[0], s[1], s[2], s[3], 256, &pos);
        for (float x : pos) psum += x;
        if (pos.size() != (size_t)s[0] * s[1] * 256) return 4;
    }
    printf("position embedding sum %.3f\n", psum);
    // person filter + NMS: empty, all equal, nested, degenerate and NaN boxes, every threshold; batch form with padding slots
    std::vector<opd_det> d(100);
    for (int i = 0; i < 100; ++i) {
        d[i].x1 = (float)(i % 10) * 10.f; d[i].y1 = (float)(i / 10) * 10.f; d[i].x2 = d[i].x1 + (float)(i % 7) * 9.f; d[i].y2 = d[i].y1 + (float)(i % 5) * 11.f;
        d[i].score = 0.5f + 0.005f * (float)((i * 37) % 100); d[i].label = (i % 3 == 0) ? 2 : 1; d[i].query_index = i; d[i].frame = 0;
    }
    d[5].x1 = NAN; d[6].x2 = d[6].x1; d[7].score = NAN; d[8].x1 = 1e30f; d[8].x2 = 3e38f;
    int kept_total = 0;
    for (float thr : {0.0f, 0.4f, 0.999f, 1.0f, 2.0f}) {
        std::vector<opd_det> c = d;
        kept_total += opd_person_nms(c.data(), (int)c.size(), 1, thr);
        std::vector<opd_det> e = d;
        kept_total += opd_person_nms(e.data(), (int)e.size(), -1, thr);
    }
    kept_total += opd_person_nms(nullptr, 0, 1, 0.4f);
    if (opd_person_nms(nullptr, 3, 1, 0.4f) >= 0 || opd_person_nms(d.data(), -1, 1, 0.4f) >= 0) return 5;
    {
        std::vector<opd_det> c = d;
        int32_t counts[4] = {25, -1, 0, 25};
        if (opd_person_nms_batch(c.data(), counts, 4, 25, 1, 0.4f) != 0) return 6;
        kept_total += counts[0] + counts[3];
        int32_t too_many[1] = {26};
        if (opd_person_nms_batch(c.data(), too_many, 1, 25, 1, 0.4f) == 0) return 7;
    }
    printf("nms kept %d; last error: %s\n", kept_total, opd_last_error());
    return 0;
}

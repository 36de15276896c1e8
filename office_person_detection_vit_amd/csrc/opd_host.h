// opd_host.h — host-only pieces of the detect path (no HIP, no device): what a CPU-only build can compile, test and run under
// AddressSanitizer / UBSan (oracle/Makefile `asan`, tests/test_host_sanitized_cpu.py): the safetensors loader (opd_loader.h),
// the Pillow coefficient tables of the device resize, the mask down-sampling, the sine position embedding, person filter + NMS.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace opd {

extern thread_local std::string g_err;   // text behind opd_last_error()
int fail(int code, const std::string& msg);

// Valid extent of a frame on the feature map (nearest down-sampling of a top-left-rectangle pixel mask): opd_host.cpp
int valid_prefix(int valid, int in, int out);
// DetrSinePositionEmbedding for an h x w map whose valid part is the top-left vh x vw rectangle: pos [h*w][D]
void sine_pos_embed(int h, int w, int vh, int vw, int D, std::vector<float>* pos);

}  // namespace opd

// Pillow-exact bilinear coefficient tables (22-bit fixed point): bounds [out][2] = (first tap, taps), coeffs [out][ksize]
void opd_resize_coeffs(int in_size, int out_size, std::vector<int32_t>* bounds, std::vector<int32_t>* coeffs, int* ksize_out);

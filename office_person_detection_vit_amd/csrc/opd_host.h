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

// fp32 -> fp16-representable fp32 for a GEMM weight matrix [rows][taps][cin] (a folded convolution kernel in the device's K order, or
// a linear layer with taps == 1), by ERROR DIFFUSION along each row's reduction instead of independent round-to-nearest: the rounding
// residual of one weight is carried into the next one of the same row, visiting the taps of one input channel first, then the next
// channel, so that the errors of neighbouring weights sum to (almost) zero.  Why: the GEMM inputs behind a ReLU are positive and
// spatially smooth, so the error of an output, sum_k dW[k] x[k], is dominated by mean(x) * sum_k dW[k] -- a random walk of K half-ulps
// under round-to-nearest, at most one half-ulp under diffusion (measured on RAW checkpoint values, tools/wround_probe.py: box drift from
// the fp16 image of the backbone kernels 7.3e-4 -> 4.6e-4, encoder map 2.5e-2 -> 1.0e-2).  Values that are fp16-exact stay untouched
// (device-exact test weights, golden vectors).  Deterministic; `taps` * `cin` entries per row.
void round_f16_diffused(float* w, size_t rows, int taps, int cin, bool bf16 = false);   // bf16: onto bfloat16 values instead

}  // namespace opd

// Pillow-exact bilinear coefficient tables (22-bit fixed point): bounds [out][2] = (first tap, taps), coeffs [out][ksize]
void opd_resize_coeffs(int in_size, int out_size, std::vector<int32_t>* bounds, std::vector<int32_t>* coeffs, int* ksize_out);

"""Algorithmic FLOPs (2 x multiply-accumulate) of one DETR / ResNet-50 forward at H x W, counted from the ARCHITECTURE (HF:models/resnet/modeling_resnet.py,
HF:models/detr/modeling_detr.py) -- an independent count for tests/test_workloads_gpu.py::test_measurement_abi_modes_are_consistent, which compares it with
what the library's per-launch bookkeeping (opd_detr_kernel_times.flops4) adds up to.  SURVEY.md section 8(d) quotes 203.2 GFLOP per frame at 800 x 1333."""


def down2(n):
    return (n - 1) // 2 + 1


def detr_r50_flops(H, W, depths=(3, 4, 6, 3), d=256, heads=8, ffn=2048, enc_layers=6, dec_layers=6, queries=100, classes=92):
    f = 0.0
    h, w = down2(H), down2(W)
    f += 2.0 * h * w * 64 * 147                      # stem 7x7 s2, 3 input channels
    h, w = down2(h), down2(w)                        # max-pool
    cin = 64
    for s, n in enumerate(depths):
        c1, c2 = 64 << s, 256 << s
        for l in range(n):
            stride = 2 if (l == 0 and s > 0) else 1
            oh, ow = (down2(h), down2(w)) if stride == 2 else (h, w)
            f += 2.0 * h * w * c1 * cin              # 1x1 reduce at the input resolution
            f += 2.0 * oh * ow * c1 * 9 * c1         # 3x3 (carries the stride: ResNet v1.5)
            f += 2.0 * oh * ow * c2 * c1             # 1x1 expand
            if l == 0:
                f += 2.0 * oh * ow * c2 * cin        # projection shortcut
            h, w, cin = oh, ow, c2
    hw = h * w
    f += 2.0 * hw * d * cin                          # input projection
    for _ in range(enc_layers):
        f += 2.0 * hw * 3 * d * d + 4.0 * hw * hw * d + 2.0 * hw * d * d + 4.0 * hw * d * ffn
    for _ in range(dec_layers):
        f += 2.0 * queries * 3 * d * d + 4.0 * queries * queries * d + 2.0 * queries * d * d      # self-attention block
        f += 2.0 * queries * d * d + 2.0 * hw * 2 * d * d + 4.0 * queries * hw * d + 2.0 * queries * d * d   # cross-attention block (memory k / v per layer)
        f += 4.0 * queries * d * ffn
    f += 2.0 * queries * (d * classes + 2 * d * d + d * 4)
    return f


if __name__ == "__main__":
    print(detr_r50_flops(800, 1333) / 1e9)

"""Seeded synthetic DETR weights + the state-dict schema of the detect path.

Pretrained ``facebook/detr-resnet-50`` weights cannot be fetched here (no network), and
the reference loads them by NAME (``config.yaml.disabled:34``; deleted
``vit_detector.py`` ``load_model`` 81-99, see ``coverage.json:1``).  Parity work therefore
uses weights built locally from a seed.  HF default init is degenerate (every query emits
the same box), so the recipe below is engineered to be non-degenerate:

* zero-sum (per output channel) Kaiming conv kernels, FrozenBN with O(1) statistics and the
  last BN γ of every bottleneck scaled by 0.1 (mimics zero-init-γ training),
* an *attention gain* ``g_a`` multiplying every q/k projection so softmax is not uniform; residual-branch
  outputs (``o_proj``, ``fc2``) at half strength and a calibrated ``input_projection`` bias that centres the
  encoder tokens, so tokens and queries stay diverse through 6+6 post-LN layers,
* ``query_position_embeddings ~ N(0,1)``, classifier std ×2 with a bias towards the COCO
  "person" class (id 1) so that the reference's person filter has survivors.

* every tensor a GEMM consumes in fp16 is made exactly fp16-representable as the device sees it ("fold FBN
  in fp32 first, then round, so the CPU fp32 reference and the GPU use bit-identical weights and only activation rounding
  differs"): linear weights are rounded to fp16 values, convolution kernels are adjusted so that kernel x FrozenBN scale — the
  folded kernel the device stores — is an fp16 value (``make_device_exact``).  Without this the comparison measured mostly the
  fp16 rounding of the weights themselves, a systematic error a real checkpoint has too but one that says nothing about the
  kernels: at 800x1333 it was 5-7e-4 of the 6-8e-4 box error of the whole backbone (tools/drift_split.py).

Everything is drawn from ``numpy.random.default_rng`` (PCG64: bit-stable across machines), so
the exact same tensors are regenerated on the GPU box.  Tensor names/shapes follow the HF 5.x
``DetrForObjectDetection`` state dict; the file format is ``safetensors``, which
is what a real checkpoint ships as and what ``csrc/opd_loader.cpp`` parses natively.
"""

from __future__ import annotations

import json
import os
import struct
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Iterator, Tuple

import numpy as np

PERSON_LABEL = 1  # COCO id of "person" in DETR's 91(+1)-way label space


@dataclass(frozen=True)
class DetrArch:
    """Architecture hyper-parameters of the detect path (HF ``DetrConfig`` / ``ResNetConfig``)."""

    depths: Tuple[int, int, int, int] = (3, 4, 6, 3)  # r50; r101 = (3, 4, 23, 3)
    hidden_sizes: Tuple[int, int, int, int] = (256, 512, 1024, 2048)
    embedding_size: int = 64
    d_model: int = 256
    heads: int = 8
    ffn_dim: int = 2048
    encoder_layers: int = 6
    decoder_layers: int = 6
    num_queries: int = 100
    num_labels: int = 91

    @staticmethod
    def resnet50() -> "DetrArch":
        return DetrArch()

    @staticmethod
    def resnet101() -> "DetrArch":
        return DetrArch(depths=(3, 4, 23, 3))

    @staticmethod
    def tiny() -> "DetrArch":
        """Reduced-depth variant used by fast CPU tests (same kernels, fewer layers)."""
        return DetrArch(depths=(1, 1, 1, 1), encoder_layers=1, decoder_layers=1, num_queries=20)


_BN = ("weight", "bias", "running_mean", "running_var")


def param_specs(arch: DetrArch) -> Iterator[Tuple[str, Tuple[int, ...], str]]:
    """Yield ``(name, shape, kind)`` for every tensor of the detect path, in HF module order.

    kinds: conv_relu / conv_lin (backbone convs followed / not followed by ReLU), bn, bn_last
    (the third BN of a bottleneck), proj_w/proj_b, lin_relu/lin/lin_qk (+ ``_b`` biases), ln_w/ln_b,
    query_pos, cls_w/cls_b, box_last_w.
    """
    e = "model.backbone.model.embedder.embedder"
    yield f"{e}.convolution.weight", (arch.embedding_size, 3, 7, 7), "conv_relu"
    for s in _BN:
        yield f"{e}.normalization.{s}", (arch.embedding_size,), "bn:" + s
    cin = arch.embedding_size
    for si, (depth, cout) in enumerate(zip(arch.depths, arch.hidden_sizes)):
        mid = cout // 4
        for li in range(depth):
            p = f"model.backbone.model.encoder.stages.{si}.layers.{li}"
            stride = 2 if (li == 0 and si > 0) else 1
            if li == 0 and (cin != cout or stride != 1):
                yield f"{p}.shortcut.convolution.weight", (cout, cin, 1, 1), "conv_lin"
                for s in _BN:
                    yield f"{p}.shortcut.normalization.{s}", (cout,), "bn:" + s
            yield f"{p}.layer.0.convolution.weight", (mid, cin, 1, 1), "conv_relu"
            for s in _BN:
                yield f"{p}.layer.0.normalization.{s}", (mid,), "bn:" + s
            yield f"{p}.layer.1.convolution.weight", (mid, mid, 3, 3), "conv_relu"
            for s in _BN:
                yield f"{p}.layer.1.normalization.{s}", (mid,), "bn:" + s
            yield f"{p}.layer.2.convolution.weight", (cout, mid, 1, 1), "conv_lin"
            for s in _BN:
                yield f"{p}.layer.2.normalization.{s}", (cout,), "bn_last:" + s
            cin = cout
    d, f = arch.d_model, arch.ffn_dim
    yield "model.input_projection.weight", (d, arch.hidden_sizes[-1], 1, 1), "proj_w"
    yield "model.input_projection.bias", (d,), "bias"
    yield "model.query_position_embeddings.weight", (arch.num_queries, d), "query_pos"

    def attn(prefix: str):
        for nm in ("k_proj", "v_proj", "q_proj", "o_proj"):
            kind = "lin_qk" if nm in ("q_proj", "k_proj") else ("lin_branch" if nm == "o_proj" else "lin")
            yield f"{prefix}.{nm}.weight", (d, d), kind
            yield f"{prefix}.{nm}.bias", (d,), "bias"

    def ln(prefix: str):
        yield f"{prefix}.weight", (d,), "ln_w"
        yield f"{prefix}.bias", (d,), "ln_b"

    for i in range(arch.encoder_layers):
        p = f"model.encoder.layers.{i}"
        yield from attn(f"{p}.self_attn")
        yield from ln(f"{p}.self_attn_layer_norm")
        yield f"{p}.mlp.fc1.weight", (f, d), "lin_relu"
        yield f"{p}.mlp.fc1.bias", (f,), "bias"
        yield f"{p}.mlp.fc2.weight", (d, f), "lin_branch"
        yield f"{p}.mlp.fc2.bias", (d,), "bias"
        yield from ln(f"{p}.final_layer_norm")
    for i in range(arch.decoder_layers):
        p = f"model.decoder.layers.{i}"
        yield from attn(f"{p}.self_attn")
        yield from ln(f"{p}.self_attn_layer_norm")
        yield from attn(f"{p}.encoder_attn")
        yield from ln(f"{p}.encoder_attn_layer_norm")
        yield f"{p}.mlp.fc1.weight", (f, d), "lin_relu"
        yield f"{p}.mlp.fc1.bias", (f,), "bias"
        yield f"{p}.mlp.fc2.weight", (d, f), "lin_branch"
        yield f"{p}.mlp.fc2.bias", (d,), "bias"
        yield from ln(f"{p}.final_layer_norm")
    yield from ln("model.decoder.layernorm")
    yield "class_labels_classifier.weight", (arch.num_labels + 1, d), "cls_w"
    yield "class_labels_classifier.bias", (arch.num_labels + 1,), "cls_b"
    yield "bbox_predictor.layers.0.weight", (d, d), "lin_relu"
    yield "bbox_predictor.layers.0.bias", (d,), "bias"
    yield "bbox_predictor.layers.1.weight", (d, d), "lin_relu"
    yield "bbox_predictor.layers.1.bias", (d,), "bias"
    yield "bbox_predictor.layers.2.weight", (4, d), "box_last_w"
    yield "bbox_predictor.layers.2.bias", (4,), "bias"


def _round_mantissa(a: np.ndarray, bits: int = 12) -> np.ndarray:
    """Round fp32 values to ``bits`` explicit mantissa bits (round-to-nearest-even on the bit pattern)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    drop = 23 - bits
    u = (u + ((1 << (drop - 1)) - 1) + ((u >> drop) & 1)) >> drop << drop
    return u.astype(np.uint32).view(np.float32)


def calibrate_frozen_bn(weights: "OrderedDict[str, np.ndarray]", arch: DetrArch, height: int = 320, width: int = 416,
                        n_frames: int = 2, frame_seed: int = 4321) -> None:
    """Set every FrozenBN's ``running_mean/var`` to the statistics of its own input on a calibration batch.

    Without this the synthetic backbone's stage-4 map is dominated by a spatially constant component and every
    encoder token (hence every query) collapses to the same value.  The pass runs in **fp64**
    (torch CPU) and the statistics are rounded to 12 mantissa bits, so the resulting fp32 tensors are
    bit-identical on any host: fp64 summation-order noise (1e-16) cannot move a 12-bit rounding.
    """
    import torch
    import torch.nn.functional as F

    from .frames import structured_frames

    frames = structured_frames(n_frames, height, width, seed=frame_seed)
    mean = np.array([0.485, 0.456, 0.406])
    std = np.array([0.229, 0.224, 0.225])
    x = np.stack([(f[:, :, ::-1].astype(np.float64) / 255.0 - mean) / std for f in frames]).transpose(0, 3, 1, 2)
    x = torch.from_numpy(np.ascontiguousarray(x))

    def conv_bn(x, prefix, stride, relu):
        cw = torch.from_numpy(weights[prefix + ".convolution.weight"]).double()
        y = F.conv2d(x, cw, None, stride=stride, padding=cw.shape[-1] // 2)
        m = y.mean(dim=(0, 2, 3))
        v = y.var(dim=(0, 2, 3), unbiased=False) + 1e-3
        m32 = _round_mantissa(m.numpy().astype(np.float32))
        v32 = _round_mantissa(v.numpy().astype(np.float32))
        weights[prefix + ".normalization.running_mean"] = m32
        weights[prefix + ".normalization.running_var"] = v32
        g = torch.from_numpy(weights[prefix + ".normalization.weight"]).double()
        b = torch.from_numpy(weights[prefix + ".normalization.bias"]).double()
        scale = g / torch.sqrt(torch.from_numpy(v32).double() + 1e-5)
        y = y * scale.view(1, -1, 1, 1) + (b - torch.from_numpy(m32).double() * scale).view(1, -1, 1, 1)
        return F.relu(y) if relu else y

    with torch.no_grad():
        x = conv_bn(x, "model.backbone.model.embedder.embedder", 2, True)
        x = F.max_pool2d(x, 3, 2, 1)
        for si, depth in enumerate(arch.depths):
            for li in range(depth):
                p = f"model.backbone.model.encoder.stages.{si}.layers.{li}"
                stride = 2 if (li == 0 and si > 0) else 1
                res = x
                if (p + ".shortcut.convolution.weight") in weights:
                    res = conv_bn(x, p + ".shortcut", stride, False)
                h = conv_bn(x, p + ".layer.0", 1, True)
                h = conv_bn(h, p + ".layer.1", stride, True)
                h = conv_bn(h, p + ".layer.2", 1, False)
                x = F.relu(h + res)
        # centre the encoder input: input_projection.bias -= W . E[stage-4 features]
        pw = torch.from_numpy(weights["model.input_projection.weight"][:, :, 0, 0]).double()
        shift = (pw @ x.mean(dim=(0, 2, 3))).numpy().astype(np.float32)
        weights["model.input_projection.bias"] = _round_mantissa(weights["model.input_projection.bias"] - shift)


def synth_weights(arch: DetrArch = DetrArch(), seed: int = 0, attention_gain: float = 2.0,
                  calibrate: bool = True, device_exact: bool = True) -> "OrderedDict[str, np.ndarray]":
    """Build the seeded, non-degenerate fp32 weight set (see module docstring).

    ``device_exact=False`` skips ``make_device_exact``: ordinary fp32 tensors, as a real checkpoint has them — the device
    then rounds every GEMM operand to fp16 itself, and the parity figure includes that weight rounding (the unfavourable case
    the tests report next to the device-exact one).

    One ``default_rng(seed)`` stream drives every tensor in ``param_specs`` order, then FrozenBN statistics are
    calibrated (``calibrate_frozen_bn``), so the result is a pure function of ``(arch, seed, attention_gain)``.
    """
    rng = np.random.default_rng(seed)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def normal(shape, std):
        return (rng.standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float32)

    def uniform(shape, lo, hi):
        return (rng.random(shape, dtype=np.float32) * np.float32(hi - lo) + np.float32(lo)).astype(np.float32)

    for name, shape, kind in param_specs(arch):
        if kind in ("conv_relu", "conv_lin"):
            fan_in = shape[1] * shape[2] * shape[3]
            w = normal(shape, np.sqrt((2.0 if kind == "conv_relu" else 1.0) / fan_in))
            w -= w.mean(axis=(1, 2, 3), keepdims=True)  # zero-sum kernels: no mean/std cancellation in BN
            t = w
        elif kind.startswith("bn"):
            which = kind.split(":")[1]
            if which == "weight":
                t = uniform(shape, 0.8, 1.2)
                if kind.startswith("bn_last"):
                    t *= np.float32(0.1)
            elif which == "bias":
                t = normal(shape, 0.05)
            elif which == "running_mean":
                t = normal(shape, 0.05)
            else:  # running_var
                t = uniform(shape, 0.8, 1.2)
        elif kind == "proj_w":
            t = normal(shape, np.sqrt(1.0 / shape[1]))
        elif kind == "bias":
            t = normal(shape, 0.02)
        elif kind == "query_pos":
            t = normal(shape, 1.0)
        elif kind in ("lin", "lin_qk", "lin_relu", "lin_branch"):
            t = normal(shape, np.sqrt((2.0 if kind == "lin_relu" else 1.0) / shape[1]))
            if kind == "lin_qk":
                t *= np.float32(attention_gain)
            if kind == "lin_branch":  # residual-branch outputs at half strength keep tokens/queries diverse
                t *= np.float32(0.5)
        elif kind == "ln_w":
            t = uniform(shape, 0.8, 1.2)
        elif kind == "ln_b":
            t = normal(shape, 0.05)
        elif kind == "cls_w":
            t = normal(shape, 2.0 * np.sqrt(1.0 / shape[1]))
        elif kind == "cls_b":
            t = normal(shape, 0.02)
            t[PERSON_LABEL] += np.float32(6.0)
        elif kind == "box_last_w":
            t = normal(shape, 2.0 * np.sqrt(1.0 / shape[1]))
        else:  # pragma: no cover
            raise ValueError(kind)
        out[name] = np.ascontiguousarray(t, dtype=np.float32)
    if calibrate:
        calibrate_frozen_bn(out, arch)
    if device_exact:
        make_device_exact(out)
    return out


RECIPE_VERSION = 2   # bump when the tensors of a (arch, seed, gain) triple change: cached files of older recipes are not reused


def make_device_exact(weights: "OrderedDict[str, np.ndarray]") -> None:
    """Make every fp16 GEMM operand of the device path exactly representable (module docstring).

    * 2-D weights of the transformer linears and ``input_projection`` (consumed as fp16): rounded to the nearest fp16 value.
    * convolution kernels: the device folds FrozenBN in fp32, ``w * (gamma * 1/sqrt(var + 1e-5))`` (``csrc/opd_model.cpp::make_conv``),
      and stores THAT in fp16.  The kernel is replaced by ``fp16(w * scale) / scale`` (fp64 division, stored fp32): the fold then
      lands within 2e-7 relative of an fp16 value and rounds to it, while the fp32 reference, which keeps kernel and scale apart,
      computes with the same number to fp32 accuracy.
    The fp32-consumed tensors (biases, norms, the heads' weights, the query embeddings) stay as drawn."""
    for name in list(weights):
        t = weights[name]
        if name.endswith(".convolution.weight"):
            pre = name[: -len(".convolution.weight")] + ".normalization"
            var = weights[pre + ".running_var"].astype(np.float32)
            scale = (weights[pre + ".weight"].astype(np.float32) * (np.float32(1.0) / np.sqrt(var + np.float32(1e-5)))).astype(np.float32)
            sc = scale.reshape(-1, 1, 1, 1)
            with np.errstate(over="ignore"):
                folded = (t * sc).astype(np.float32).astype(np.float16)
            weights[name] = np.ascontiguousarray((folded.astype(np.float64) / sc.astype(np.float64)).astype(np.float32))
        elif t.ndim >= 2 and name.endswith(".weight") and name.startswith(("model.encoder.", "model.decoder.", "model.input_projection.")):
            weights[name] = np.ascontiguousarray(t.astype(np.float16).astype(np.float32))


# ----------------------------------------------------------------------------------------------
# safetensors I/O (format: u64 LE header length, JSON header, raw little-endian tensor bytes).
# Written by hand (numpy only) so the byte layout the native loader parses is explicit here.
# ----------------------------------------------------------------------------------------------

def save_safetensors(weights: Dict[str, np.ndarray], path: str) -> None:
    header: Dict[str, dict] = {}
    off = 0
    for name, t in weights.items():
        assert t.dtype == np.float32
        n = t.size * 4
        header[name] = {"dtype": "F32", "shape": list(t.shape), "data_offsets": [off, off + n]}
        off += n
    hb = json.dumps(header, separators=(",", ":")).encode("utf-8")
    hb += b" " * ((8 - len(hb) % 8) % 8)
    tmp = path + f".tmp{os.getpid()}"
    with open(tmp, "wb") as f:
        f.write(struct.pack("<Q", len(hb)))
        f.write(hb)
        for t in weights.values():
            f.write(np.ascontiguousarray(t).tobytes())
    os.replace(tmp, path)


def load_safetensors(path: str) -> "OrderedDict[str, np.ndarray]":
    with open(path, "rb") as f:
        (hl,) = struct.unpack("<Q", f.read(8))
        header = json.loads(f.read(hl).decode("utf-8"))
        base = 8 + hl
        out: "OrderedDict[str, np.ndarray]" = OrderedDict()
        dt = {"F32": np.float32, "F16": np.float16, "F64": np.float64}
        for name, meta in header.items():
            if name == "__metadata__":
                continue
            a, b = meta["data_offsets"]
            f.seek(base + a)
            arr = np.frombuffer(f.read(b - a), dtype=dt[meta["dtype"]]).reshape(meta["shape"])
            out[name] = arr.astype(np.float32)
    return out


# HF 4.x checkpoint names -> 5.x names used throughout this repo (HF:conversion_mapping.py:1036-1041).
_RENAMES_4X = (
    ("model.backbone.conv_encoder.", "model.backbone."),
    (".out_proj.", ".o_proj."),
)


def rename_4x_key(name: str) -> str:
    for a, b in _RENAMES_4X:
        name = name.replace(a, b)
    for fc in ("fc1", "fc2"):
        tok = f".{fc}."
        if tok in name and ".mlp." not in name and ".layers." in name and name.startswith(("model.encoder", "model.decoder")):
            name = name.replace(tok, f".mlp.{fc}.")
    return name


def ensure_weight_file(cache_dir: str, arch: DetrArch = DetrArch(), seed: int = 0, attention_gain: float = 2.0,
                       tag: str = "r50", device_exact: bool = True) -> str:
    """Write (once) and return the path of the safetensors file for a seeded weight set."""
    os.makedirs(cache_dir, exist_ok=True)
    kind = "" if device_exact else "_raw"   # (part of the cache key: the two sets differ in every GEMM weight)
    path = os.path.join(cache_dir, f"detr_{tag}_seed{seed}_ga{attention_gain:g}_v{RECIPE_VERSION}{kind}.safetensors")
    if not os.path.exists(path):
        tmp = path + f".tmp{os.getpid()}"
        save_safetensors(synth_weights(arch, seed, attention_gain, device_exact=device_exact), tmp)
        os.replace(tmp, path)   # (atomic: several ranks may generate the same file at once)
    return path

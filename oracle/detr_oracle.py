"""ORACLE (test infrastructure, NOT product code): CPU fp32 restatement of the Phase-2 DETR detect path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
file; the product (``office_person_detection_vit_amd``) never does and fails loudly without its HIP library.

What is restated, and from where
--------------------------------
The reference's DETR detector (``src/detection/vit_detector.py``) has been deleted from the tree; its method
map survives in ``coverage.json:1`` and its arithmetic lives in the third-party dependency Hugging Face
``transformers`` (reference pin ``transformers==4.57.3``, ``requirements.txt:220``), which is not under
``/root/reference``.  This file restates the published algorithm of that dependency, citing the 5.15.0 source
present in the build container as ``HF:`` = ``transformers/`` (SURVEY.md header):

* FrozenBN               HF:models/detr/modeling_detr.py:179-215
* ResNet stem/bottleneck HF:models/resnet/modeling_resnet.py:72-93,139-178,219-258
* mask downsample        HF:models/detr/modeling_detr.py:280-291
* input projection       HF:models/detr/modeling_detr.py:1129,1197-1198
* sine position embed    HF:models/detr/modeling_detr.py:294-368
* attention / layers     HF:models/detr/modeling_detr.py:402-427,430-493,496-573,593-739
* encoder/decoder/model  HF:models/detr/modeling_detr.py:933-1106,1146-1281
* heads                  HF:models/detr/modeling_detr.py:1284-1300,1317-1322,1410-1411
* post-process           HF:models/detr/image_processing_detr.py:805-856
* pre-process (no-resize part) HF:models/detr/image_processing_detr.py:639-668,752-791

Pinning: the reference holds NO golden vector for the model (every detector test mocks it), so
this oracle is pinned against outputs of the HF module itself, generated in the build container by
``tools/gen_golden.py`` and committed under ``tests/golden/`` (``tests/test_oracle_golden.py``).

Plain ``torch`` CPU fp32 functional ops only (conv2d / linear / softmax / layer_norm); no ``transformers`` import.
"""

from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)  # RGB; HF:utils/constants.py IMAGENET_DEFAULT_MEAN
IMAGENET_STD = (0.229, 0.224, 0.225)
BN_EPS = 1e-5
LN_EPS = 1e-5
PERSON_LABEL = 1


# ------------------------------------------------------------------------------------------------
# a1 — pre-process (the part that applies when frames already have the model resolution)
# ------------------------------------------------------------------------------------------------

def preprocess(frames_bgr: Sequence[np.ndarray]) -> Tuple[torch.Tensor, torch.Tensor]:
    """BGR uint8 HxWx3 frames -> (pixel_values [B,3,Hmax,Wmax] f32, pixel_mask [B,Hmax,Wmax] i64).

    BGR->RGB (deleted ``vit_detector._preprocess`` 285-300), x*(1/255), (x-mean)/std, zero-pad bottom/right to the
    batch max, mask = 1 on real pixels (HF:models/detr/image_processing_detr.py:639-668,752-791).
    """
    hmax = max(f.shape[0] for f in frames_bgr)
    wmax = max(f.shape[1] for f in frames_bgr)
    pv = torch.zeros((len(frames_bgr), 3, hmax, wmax), dtype=torch.float32)
    pm = torch.zeros((len(frames_bgr), hmax, wmax), dtype=torch.int64)
    mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=torch.float32).view(3, 1, 1)
    for i, f in enumerate(frames_bgr):
        rgb = torch.from_numpy(np.ascontiguousarray(f[:, :, ::-1])).permute(2, 0, 1).to(torch.float32)
        x = (rgb * (1.0 / 255.0) - mean) / std
        pv[i, :, : f.shape[0], : f.shape[1]] = x
        pm[i, : f.shape[0], : f.shape[1]] = 1
    return pv, pm


# ------------------------------------------------------------------------------------------------
# a2-a5 — backbone
# ------------------------------------------------------------------------------------------------

def _fbn(x: torch.Tensor, w: Dict[str, torch.Tensor], prefix: str) -> torch.Tensor:
    scale = w[prefix + ".weight"] * (w[prefix + ".running_var"] + BN_EPS).rsqrt()
    bias = w[prefix + ".bias"] - w[prefix + ".running_mean"] * scale
    return x * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)


def _conv_layer(x, w, prefix, stride, relu, q=None):
    cw = w[prefix + ".convolution.weight"]
    x = F.conv2d(x, cw, None, stride=stride, padding=cw.shape[-1] // 2)
    x = _fbn(x, w, prefix + ".normalization")
    if relu:
        x = F.relu(x)
    return q(x) if q is not None else x


def backbone(w: Dict[str, torch.Tensor], pixel_values: torch.Tensor, depths: Sequence[int], taps=None, q=None):
    x = _conv_layer(pixel_values, w, "model.backbone.model.embedder.embedder", 2, True, q)
    if taps is not None:
        taps["stem"] = x
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    if taps is not None:
        taps["pool"] = x
    for si, depth in enumerate(depths):
        for li in range(depth):
            p = f"model.backbone.model.encoder.stages.{si}.layers.{li}"
            stride = 2 if (li == 0 and si > 0) else 1
            res = x
            if (p + ".shortcut.convolution.weight") in w:
                res = F.conv2d(x, w[p + ".shortcut.convolution.weight"], None, stride=stride)
                res = _fbn(res, w, p + ".shortcut.normalization")
                if q is not None:
                    res = q(res)
            h = _conv_layer(x, w, p + ".layer.0", 1, True, q)
            h = _conv_layer(h, w, p + ".layer.1", stride, True, q)
            h = _conv_layer(h, w, p + ".layer.2", 1, False, None)
            x = F.relu(h + res)
            if q is not None:
                x = q(x)
            if taps is not None and li == 0 and si == 0:
                taps["s0l0"] = x
        if taps is not None:
            taps[f"stage{si}"] = x
    return x


# ------------------------------------------------------------------------------------------------
# a7 — sine position embedding
# ------------------------------------------------------------------------------------------------

def sine_position_embedding(mask: torch.Tensor, d_model: int = 256) -> torch.Tensor:
    """mask [B,h,w] bool -> [B, h*w, d_model] f32 (normalize=True, scale=2*pi, T=10000, eps=1e-6)."""
    npf = d_model // 2
    m = mask.to(torch.float32)
    y_embed = m.cumsum(1)
    x_embed = m.cumsum(2)
    eps, scale = 1e-6, 2 * math.pi
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(npf, dtype=torch.int64).to(torch.float32)
    dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / npf)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
    pos = torch.cat((pos_y, pos_x), dim=3)  # [B,h,w,256]
    return pos.flatten(1, 2)


# ------------------------------------------------------------------------------------------------
# a8-a12 — transformer
# ------------------------------------------------------------------------------------------------

def _lin(x, w, prefix):
    return F.linear(x, w[prefix + ".weight"], w[prefix + ".bias"])


def _mha(w, prefix, q_in, k_in, v_in, heads, key_mask_add=None, q=None, probs_out=None, site="attn"):
    """softmax(Q K^T / sqrt(dh) + mask) V, then o_proj.  *_in: [B, L, D].  `q` (storage emulation) is applied at the named
    rounding sites `site`.q / .kv / .p / .o (see _SiteQuant)."""
    B, Lq, D = q_in.shape
    Lk = k_in.shape[1]
    dh = D // heads
    Q = _lin(q_in, w, prefix + ".q_proj")
    K = _lin(k_in, w, prefix + ".k_proj")
    V = _lin(v_in, w, prefix + ".v_proj")
    if q is not None:
        Q, K, V = q(Q, site + ".q"), q(K, site + ".kv"), q(V, site + ".kv")
    Q = Q.view(B, Lq, heads, dh).transpose(1, 2)
    K = K.view(B, Lk, heads, dh).transpose(1, 2)
    V = V.view(B, Lk, heads, dh).transpose(1, 2)
    s = torch.matmul(Q, K.transpose(2, 3)) * (dh ** -0.5)
    if key_mask_add is not None:
        s = s + key_mask_add
    p = F.softmax(s, dim=-1)
    if probs_out is not None:
        probs_out.append(p)   # [B, heads, Lq, Lk]
    if q is not None:
        p = q(p, site + ".p")
    o = torch.matmul(p, V).transpose(1, 2).reshape(B, Lq, D)
    if q is not None:
        o = q(o, site + ".o")
    return _lin(o, w, prefix + ".o_proj")


def _ln(x, w, prefix):
    return F.layer_norm(x, (x.shape[-1],), w[prefix + ".weight"], w[prefix + ".bias"], LN_EPS)


def _mlp(x, w, prefix, q=None, site="ffn"):
    h = F.relu(_lin(x if q is None else q(x, site + ".in"), w, prefix + ".fc1"))
    if q is not None:
        h = q(h, site + ".h")
    return _lin(h, w, prefix + ".fc2")


def encoder(w, x, pos, n_layers, heads, key_mask_add=None, taps=None, q=None):
    for i in range(n_layers):
        p = f"model.encoder.layers.{i}"
        qk = x + pos
        if q is not None:
            a = _mha(w, p + ".self_attn", q(qk, "enc.attn.in"), q(qk, "enc.attn.in"), q(x, "enc.attn.in"), heads, key_mask_add, q, None, "enc.attn")
        else:
            a = _mha(w, p + ".self_attn", qk, qk, x, heads, key_mask_add)
        x = _ln(x + a, w, p + ".self_attn_layer_norm")
        x = _ln(x + _mlp(x, w, p + ".mlp", q, "enc.ffn"), w, p + ".final_layer_norm")
        if taps is not None:
            taps[f"enc{i}"] = x
    return x


def decoder(w, memory, pos, n_layers, heads, key_mask_add=None, taps=None, q=None):
    B = memory.shape[0]
    qpos = w["model.query_position_embeddings.weight"].unsqueeze(0).expand(B, -1, -1)
    h = torch.zeros_like(qpos)
    ident = (lambda t, site=None: t) if q is None else q
    for i in range(n_layers):
        p = f"model.decoder.layers.{i}"
        qk = ident(h + qpos, "dec.self.in")
        a = _mha(w, p + ".self_attn", qk, qk, ident(h, "dec.self.in"), heads, None, q, None, "dec.self")
        h = _ln(h + a, w, p + ".self_attn_layer_norm")
        cross = [] if taps is not None else None
        a = _mha(w, p + ".encoder_attn", ident(h + qpos, "dec.cross.in"), ident(memory + pos, "dec.cross.kvin"), ident(memory, "dec.cross.kvin"), heads,
                 key_mask_add, q, cross, "dec.cross")
        if taps is not None:
            taps[f"dec{i}_cross_probs"] = cross[0]
        h = _ln(h + a, w, p + ".encoder_attn_layer_norm")
        h = _ln(h + _mlp(h, w, p + ".mlp", q, "dec.ffn"), w, p + ".final_layer_norm")
        if taps is not None:
            taps[f"dec{i}"] = h
    return _ln(h, w, "model.decoder.layernorm")


# ------------------------------------------------------------------------------------------------
# the whole forward
# ------------------------------------------------------------------------------------------------

def infer_arch(w: Dict[str, torch.Tensor]) -> dict:
    depths = []
    for si in range(4):
        d = 0
        while f"model.backbone.model.encoder.stages.{si}.layers.{d}.layer.0.convolution.weight" in w:
            d += 1
        depths.append(d)
    ne = 0
    while f"model.encoder.layers.{ne}.self_attn.q_proj.weight" in w:
        ne += 1
    nd = 0
    while f"model.decoder.layers.{nd}.self_attn.q_proj.weight" in w:
        nd += 1
    return {"depths": depths, "encoder_layers": ne, "decoder_layers": nd,
            "d_model": w["model.input_projection.weight"].shape[0], "heads": 8}


def to_torch(weights: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.items()}


class _SiteQuant:
    """Emulation of low-precision *storage* (predicts GPU drift on CPU): `q(t, site)` rounds `t` through `mode` when `sites` is None
    (every rounding site) or `site` starts with one of the prefixes in `sites`.  Activation sites of the transformer:
    enc.attn.{in,q,kv,p,o}, enc.ffn.{in,h}, dec.self.{in,q,kv,p,o}, dec.cross.{in,kvin,q,kv,p,o}, dec.ffn.{in,h}, heads;
    weight sites (see `forward`): w.proj, w.enc.attn, w.enc.ffn, w.dec.self.{qk,v,o}, w.dec.cross.q, w.dec.cross.kv, w.dec.cross.o, w.dec.ffn, w.heads."""

    def __init__(self, mode: str, sites=None):
        self.dt = {"f16": torch.float16, "bf16": torch.bfloat16}[mode]
        self.sites = None if sites is None else tuple(sites)

    def on(self, site: Optional[str]) -> bool:
        return self.sites is None or (site is not None and any(site.startswith(p) for p in self.sites))

    def __call__(self, t, site: Optional[str] = None):
        return t.to(self.dt).to(torch.float32) if self.on(site) else t


class _MixedQuant:
    """fp16 storage everywhere, bf16 at the sites whose name starts with one of `bf16_sites` (tools/mixed_bf16_probe.py: what a mixed
    operand mode would cost in box drift).  Same interface as _SiteQuant."""

    def __init__(self, bf16_sites):
        self.bf16_sites = tuple(bf16_sites)

    def on(self, site: Optional[str]) -> bool:
        return True

    def __call__(self, t, site: Optional[str] = None):
        dt = torch.bfloat16 if (site is not None and any(site.startswith(p) for p in self.bf16_sites)) else torch.float16
        return t.to(dt).to(torch.float32)


def _quantizer(mode, sites=None):
    if mode is None or isinstance(mode, str):
        return None if mode is None else _SiteQuant(mode, sites)
    return mode   # a quantizer object (_MixedQuant)


def _weight_site(k: str) -> Optional[str]:
    """Rounding site of a transformer GEMM weight (None: not a GEMM weight of the transformer)."""
    if not k.endswith(".weight"):
        return None
    if k.startswith("model.input_proj"):
        return "w.proj"
    if k.startswith(("bbox_predictor", "class_labels")):
        return "w.heads"
    if k.startswith("model.encoder"):
        return "w.enc.ffn" if ".mlp." in k else "w.enc.attn"
    if k.startswith("model.decoder"):
        if ".mlp." in k:
            return "w.dec.ffn"
        if ".self_attn." in k:
            return "w.dec.self.o" if ".o_proj" in k else ("w.dec.self.v" if ".v_proj" in k else "w.dec.self.qk")
        if ".encoder_attn.q_proj" in k:
            return "w.dec.cross.q"
        if ".encoder_attn.o_proj" in k:
            return "w.dec.cross.o"
        if ".encoder_attn." in k:
            return "w.dec.cross.kv"
    return None


@torch.no_grad()
def forward(weights: Dict[str, torch.Tensor], pixel_values: torch.Tensor, pixel_mask: Optional[torch.Tensor] = None,
            taps: Optional[dict] = None, emulate: Optional[str] = None, emulate_transformer: Optional[str] = None,
            transformer_sites=None, backbone_features: Optional[torch.Tensor] = None, backbone_sites=None):
    """-> (logits [B,Q,C+1], pred_boxes [B,Q,4] cxcywh in [0,1], encoder_last_hidden_state [B,HW,D]).
    `emulate` / `emulate_transformer` ("f16" | "bf16"): storage emulation of the backbone / the transformer; `transformer_sites`
    restricts the latter to the named rounding sites (_SiteQuant), `backbone_sites` the former to "bb.w" (folded kernels) and / or
    "bb.act" (activations).  `backbone_features`: skip the backbone and start from this
    stage-4 map (tools/drift_split.py runs many transformer variants on one backbone pass)."""
    w = weights
    arch = infer_arch(w)
    B, _, H, W = pixel_values.shape
    if pixel_mask is None:
        pixel_mask = torch.ones((B, H, W), dtype=torch.int64)
    qb = _quantizer(emulate, backbone_sites)
    qb_act = None if qb is None else (lambda t: qb(t, "bb.act"))
    qt = _quantizer(emulate_transformer, transformer_sites)
    x_in = pixel_values if qb is None else qb(pixel_values, "bb.act")
    if qb is not None:
        w = dict(w)
        # product folds BN into the conv in fp32 and stores the folded kernel in low precision
        for k in list(w.keys()):
            if k.endswith(".convolution.weight"):
                pre = k[: -len(".convolution.weight")] + ".normalization"
                scale = w[pre + ".weight"] * (w[pre + ".running_var"] + BN_EPS).rsqrt()
                w[k] = qb(w[k] * scale.view(-1, 1, 1, 1), "bb.w")
                w[pre + ".bias"] = w[pre + ".bias"] - w[pre + ".running_mean"] * scale
                w[pre + ".weight"] = torch.ones_like(scale)
                w[pre + ".running_mean"] = torch.zeros_like(scale)
                w[pre + ".running_var"] = torch.ones_like(scale) - BN_EPS
    if qt is not None:
        w = dict(w)
        for k in list(w.keys()):
            if w[k].dim() >= 2 and _weight_site(k) is not None:
                w[k] = qt(w[k], _weight_site(k))
    feat = backbone(w, x_in, arch["depths"], taps, qb_act) if backbone_features is None else backbone_features
    h, wd = feat.shape[-2:]
    mask = F.interpolate(pixel_mask[None].float(), size=(h, wd)).to(torch.bool)[0]  # nearest
    proj = F.conv2d(feat, w["model.input_projection.weight"], w["model.input_projection.bias"])
    x = proj.flatten(2).transpose(1, 2)
    if taps is not None:
        taps["proj"] = x
    pos = sine_position_embedding(mask, arch["d_model"])
    key_mask_add = None
    if not bool(mask.all()):
        key_mask_add = torch.zeros((B, 1, 1, h * wd), dtype=torch.float32)
        key_mask_add.masked_fill_(~mask.flatten(1)[:, None, None, :], torch.finfo(torch.float32).min)
    mem = encoder(w, x, pos, arch["encoder_layers"], arch["heads"], key_mask_add, taps, qt)
    hs = decoder(w, mem, pos, arch["decoder_layers"], arch["heads"], key_mask_add, taps, qt)
    if taps is not None:
        taps["hs"] = hs
    hq = hs if qt is None else qt(hs, "heads")
    logits = _lin(hq, w, "class_labels_classifier")
    b = F.relu(_lin(hq, w, "bbox_predictor.layers.0"))
    b = F.relu(_lin(b if qt is None else qt(b, "heads"), w, "bbox_predictor.layers.1"))
    boxes = _lin(b if qt is None else qt(b, "heads"), w, "bbox_predictor.layers.2").sigmoid()
    return logits, boxes, mem


# ------------------------------------------------------------------------------------------------
# a14-a16 — post-process
# ------------------------------------------------------------------------------------------------

def cross_attention_map(taps: dict, frame: int, layer: int, queries=None) -> np.ndarray:
    """Attention map of ``get_attention_map`` (the reference's DETR-era method is deleted: ``coverage.json:1`` src 392-446 keeps only its
    line numbers, so the definition is this build's): the decoder's cross-attention weights of layer ``layer`` (negative: from the end),
    averaged over the heads and over ``queries`` (all queries when None) -> [h*w] float32, summing to 1.  ``taps`` is the dict that
    ``forward(..., taps=taps)`` filled."""
    n = sum(1 for k in taps if k.endswith("_cross_probs"))
    p = taps[f"dec{layer % n}_cross_probs"][frame]          # [heads, Q, hw]
    if queries is not None:
        p = p[:, list(queries), :]
    return p.mean(dim=(0, 1)).numpy().astype(np.float32)


def post_process_object_detection(logits: np.ndarray, boxes: np.ndarray, threshold: float,
                                  target_sizes: Sequence[Tuple[int, int]]) -> List[dict]:
    """HF ``post_process_object_detection``: softmax, max over classes[:-1], cxcywh->xyxy, scale to (W,H,W,H), keep > thr."""
    lg = torch.from_numpy(np.asarray(logits, dtype=np.float32))
    bx = torch.from_numpy(np.asarray(boxes, dtype=np.float32))
    prob = F.softmax(lg, -1)
    scores, labels = prob[..., :-1].max(-1)
    cx, cy, w, h = bx.unbind(-1)
    xyxy = torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)
    img_h = torch.tensor([float(t[0]) for t in target_sizes])
    img_w = torch.tensor([float(t[1]) for t in target_sizes])
    xyxy = xyxy * torch.stack([img_w, img_h, img_w, img_h], dim=1)[:, None, :]
    out = []
    for s, l, b in zip(scores, labels, xyxy):
        keep = s > threshold
        out.append({"scores": s[keep].numpy(), "labels": l[keep].numpy(), "boxes": b[keep].numpy(),
                    "query_index": torch.nonzero(keep).flatten().numpy()})
    return out


def iou_xyxy(a: np.ndarray, b: np.ndarray) -> float:
    ix1, iy1 = max(a[0], b[0]), max(a[1], b[1])
    ix2, iy2 = min(a[2], b[2]), min(a[3], b[3])
    iw, ih = max(0.0, ix2 - ix1), max(0.0, iy2 - iy1)
    inter = iw * ih
    ua = max(0.0, a[2] - a[0]) * max(0.0, a[3] - a[1]) + max(0.0, b[2] - b[0]) * max(0.0, b[3] - b[1]) - inter
    return float(inter / ua) if ua > 0 else 0.0


def person_detections(result: dict, nms_threshold: float = 0.4, person_label: int = PERSON_LABEL) -> List[dict]:
    """Deleted ``vit_detector._postprocess`` 313-366 (``docs/plan.md:30``): keep label == person, greedy IoU-NMS in
    score-descending order (stable), xyxy -> (x, y, w, h), foot point = (x + w/2, y + h)
    (``src/detection/yolov8_detector.py:229-241``).  NMS flavour is unpinned in the reference (source + tests gone)."""
    idx = [i for i in range(len(result["scores"])) if int(result["labels"][i]) == person_label]
    idx.sort(key=lambda i: (-float(result["scores"][i]), i))
    kept: List[int] = []
    for i in idx:
        if all(iou_xyxy(result["boxes"][i], result["boxes"][j]) <= nms_threshold for j in kept):
            kept.append(i)
    dets = []
    for i in kept:
        x1, y1, x2, y2 = (float(v) for v in result["boxes"][i])
        bbox = (x1, y1, x2 - x1, y2 - y1)
        dets.append({"bbox": bbox, "confidence": float(result["scores"][i]), "class_id": person_label,
                     "camera_coords": (bbox[0] + bbox[2] / 2, bbox[1] + bbox[3]),
                     "query_index": int(result["query_index"][i])})
    return dets


def roi_features(encoder_map: np.ndarray, bboxes, image_shape) -> np.ndarray:
    """``FeatureExtractor.extract_roi_features`` (``src/tracking/feature_extractor.py:39-88``): int-truncated ROI on the
    (h,w,C) encoder map, clamp, mean-pool, L2-normalise with +1e-8."""
    h, w, c = encoder_map.shape
    img_h, img_w = image_shape
    feats = []
    for (x, y, bw, bh) in bboxes:
        x_min = int((x / img_w) * w)
        y_min = int((y / img_h) * h)
        x_max = int(((x + bw) / img_w) * w)
        y_max = int(((y + bh) / img_h) * h)
        x_min = max(0, min(x_min, w - 1))
        y_min = max(0, min(y_min, h - 1))
        x_max = max(x_min + 1, min(x_max, w))
        y_max = max(y_min + 1, min(y_max, h))
        feats.append(encoder_map[y_min:y_max, x_min:x_max, :].mean(axis=(0, 1)))
    if not feats:
        return np.zeros((0, c), dtype=np.float32)
    f = np.asarray(feats)
    return f / (np.linalg.norm(f, axis=1, keepdims=True) + 1e-8)

"""CPU restatement (plain PyTorch fp32) of the RE-SepFormer separator the reference calls.

TEST INFRASTRUCTURE ONLY (tests/, smoke, bench cpu_baseline).

Restates `SepformerSeparation.separate_batch` of SpeechBrain with the hyper-parameters of
`speechbrain/resepformer-wsj02mix` (loaded by the reference at /root/reference/back/api.py:713,
called at back/api.py:1077: `separated = self.separator.separate_batch(subsegment)` with a [1, T]
mixture).  SpeechBrain is an un-pinned third-party dependency (back/requirements.txt:12-19), not
vendored and not installed here, so this follows the published code from recollection
[UPSTREAM-RECALL]: speechbrain/inference/separation.py (separate_batch),
lobes/models/dual_path.py (Encoder, Decoder, SBTransformerBlock-style blocks, select_norm/gLN),
lobes/models/resepformer.py (ResourceEfficientSeparator / ...SeparationPipeline,
SBTransformerBlock_wnormandskip), lobes/models/transformer/Transformer.py (TransformerEncoder,
PositionalEncoding), nnet/attention.py (MultiheadAttention = torch.nn.MultiheadAttention).

PARITY STATUS: **parity unpinned** -- the reference holds no fixture for this model and no
independent implementation of it exists in this image; the HIP path is checked against this file
only.  Items marked (?) in SURVEY.md Appendix A.2 are decided here and listed in DESIGN.md.

Per-utterance semantics: the reference always passes batch 1.  `separate(mix[B,T])` here treats the
rows of a batch as B independent batch-1 calls (SpeechBrain's memory transformer would attend
across the rows of a batch, an artefact the reference never exercises).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict

import torch
import torch.nn.functional as F

EPS_GLN = 1e-8


@dataclass
class SepDims:
    n_filters: int = 128      # N_encoder_out
    kernel: int = 16
    stride: int = 8
    d_model: int = 128
    n_head: int = 8
    d_ffn: int = 1024
    n_layers: int = 8         # transformer layers per block
    n_blocks: int = 2         # ResourceEfficientSeparator(layer=2)
    segment: int = 150
    n_spk: int = 2


def positional_encoding(length: int, d: int) -> torch.Tensor:
    """Transformer.PositionalEncoding: interleaved sin/cos."""
    pe = torch.zeros(length, d)
    pos = torch.arange(length, dtype=torch.float32)[:, None]
    den = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(pos * den)
    pe[:, 1::2] = torch.cos(pos * den)
    return pe


class SepformerRef:
    def __init__(self, dims: SepDims, sd: Dict[str, torch.Tensor]):
        self.d = dims
        self.sd = {k: v.detach().float() for k, v in sd.items()}

    # ---- SBTransformerBlock_wnormandskip (use_positional_encoding, norm_before, gLN, skip) ----
    def _block(self, prefix: str, x: torch.Tensor) -> torch.Tensor:
        """x [n_seq, L, d] -> same."""
        d, sd = self.d, self.sd
        n, L, D = x.shape
        h = x + positional_encoding(L, D)
        nh, hd = d.n_head, D // d.n_head
        for l in range(d.n_layers):
            p = f"{prefix}.mdl.layers.{l}"
            # TransformerEncoderLayer, normalize_before=True
            y = F.layer_norm(h, (D,), sd[p + ".norm1.norm.weight"], sd[p + ".norm1.norm.bias"], 1e-6)
            qkv = F.linear(y, sd[p + ".self_att.att.in_proj_weight"], sd[p + ".self_att.att.in_proj_bias"])
            q, k, v = qkv.split(D, dim=-1)
            q = q.view(n, L, nh, hd).transpose(1, 2)
            k = k.view(n, L, nh, hd).transpose(1, 2)
            v = v.view(n, L, nh, hd).transpose(1, 2)
            att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1) @ v
            att = att.transpose(1, 2).reshape(n, L, D)
            h = h + F.linear(att, sd[p + ".self_att.att.out_proj.weight"], sd[p + ".self_att.att.out_proj.bias"])
            y = F.layer_norm(h, (D,), sd[p + ".norm2.norm.weight"], sd[p + ".norm2.norm.bias"], 1e-6)
            y = F.linear(F.relu(F.linear(y, sd[p + ".pos_ffn.ffn.0.weight"], sd[p + ".pos_ffn.ffn.0.bias"])),
                         sd[p + ".pos_ffn.ffn.3.weight"], sd[p + ".pos_ffn.ffn.3.bias"])
            h = h + y
        h = F.layer_norm(h, (D,), sd[prefix + ".mdl.norm.norm.weight"], sd[prefix + ".mdl.norm.norm.bias"], 1e-6)
        # GlobalLayerNorm over (channel, time) of each sequence, per-channel affine
        mean = h.mean(dim=(1, 2), keepdim=True)
        var = ((h - mean) ** 2).mean(dim=(1, 2), keepdim=True)
        g = sd[prefix + ".norm.weight"].view(1, 1, D)
        b = sd[prefix + ".norm.bias"].view(1, 1, D)
        h = g * (h - mean) / torch.sqrt(var + EPS_GLN) + b
        return h + x

    # ---- ResourceEfficientSeparationPipeline.forward for ONE utterance ----
    def _pipeline(self, feats: torch.Tensor) -> torch.Tensor:
        """feats [L, N] -> [L, N * n_spk]."""
        d, sd = self.d, self.sd
        L, N = feats.shape
        rest = d.segment - L % d.segment          # a full extra chunk when L % segment == 0
        x = F.pad(feats, (0, 0, 0, rest))
        S = x.shape[0] // d.segment
        out = x.view(S, d.segment, N)
        hc = None
        for i in range(d.n_blocks):
            out = self._block(f"masknet.model.seg_model.{i}", out if hc is None else out + hc)
            if i < d.n_blocks - 1:
                mem = out.mean(dim=1)[None]                                   # [1, S, N]: attends across this utterance's chunks
                hc = self._block(f"masknet.model.mem_model.{i}", mem).permute(1, 0, 2)   # [S, 1, N]
        out = out.reshape(S * d.segment, N)[:L]
        a = sd["masknet.model.output_fc.0.weight"]                           # PReLU, one shared slope
        out = torch.where(out >= 0, out, a * out)
        w = sd["masknet.model.output_fc.1.weight"].view(N * d.n_spk, N)
        return F.linear(out, w, sd["masknet.model.output_fc.1.bias"])

    def separate(self, mix: torch.Tensor) -> torch.Tensor:
        """separate_batch: mix [B, T] f32 -> [B, T, n_spk]."""
        d, sd = self.d, self.sd
        outs = []
        for b in range(mix.shape[0]):
            m = mix[b:b + 1].float()
            T = m.shape[1]
            w = F.relu(F.conv1d(m[:, None, :], sd["encoder.conv1d.weight"], None, stride=d.stride))   # [1, N, L]
            feats = w[0].transpose(0, 1)                                                                 # [L, N]
            fc = self._pipeline(feats)                                                                   # [L, N*spk]
            masks = F.relu(fc.view(-1, d.n_filters, d.n_spk))                                            # [L, N, spk]
            est = []
            for s in range(d.n_spk):
                sep = (feats * masks[:, :, s]).transpose(0, 1)[None]                                     # [1, N, L]
                est.append(F.conv_transpose1d(sep, sd["decoder.weight"], None, stride=d.stride)[0, 0])  # [T']
            e = torch.stack(est, dim=-1)                                                                 # [T', spk]
            if T > e.shape[0]:
                e = F.pad(e, (0, 0, 0, T - e.shape[0]))
            else:
                e = e[:T]
            outs.append(e)
        return torch.stack(outs)

"""Seeded stand-ins for the segmentation net and the chunk embedder: the host side of the VAD / diarization
pipelines (window plans, powerset -> multi-label, pooling masks, clustering, overlap-add, binarisation) can then run
on the CPU and be pinned by a fixture (tests/golden/host_postnet_fake_nets.json, written by
`python -m tests.fake_nets`)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "host_postnet_fake_nets.json")


class FakeSeg:
    """Smooth seeded scores (speakers persist for 40 frames), log-softmax for the powerset head."""
    device = torch.device("cpu")

    def __init__(self, n_classes=7, powerset=True, seed=0):
        self.n_classes, self.powerset, self.seed = n_classes, powerset, seed

    def segment_launch(self, crops):
        return [int(c.numel()) for c in crops]

    def segment_fetch(self, pending):
        outs = []
        for i, n in enumerate(pending):
            rng = np.random.default_rng(self.seed + i // 3)
            f = 589 if n == 160000 else (293 if n == 80000 else (n - 721) // 270 + 1)
            base = rng.standard_normal((f // 40 + 2, self.n_classes)).astype(np.float32) * 2.0
            x = np.repeat(base, 40, axis=0)[:f] + 0.1 * rng.standard_normal((f, self.n_classes)).astype(np.float32)
            x = x - np.log(np.exp(x).sum(-1, keepdims=True)) if self.powerset else 1.0 / (1.0 + np.exp(-x))
            outs.append(x.astype(np.float32))
        return outs

    def segment_numpy(self, crops):
        return self.segment_fetch(self.segment_launch(crops))


class FakeEmb:
    """Two well separated 'voices' chosen by the parity of the mask's frame count."""
    DIM = 256

    def embed_chunks(self, chunks, weights, mask_chunk):
        w = weights.numpy()
        out = np.zeros((w.shape[0], 256), np.float32)
        for i in range(w.shape[0]):
            sign = 1.0 if int(w[i].sum()) % 2 else -1.0
            out[i] = sign * np.linspace(1, 2, 256) + 0.05 * np.random.default_rng(1000 + i).standard_normal(256)
        return torch.from_numpy(out)


class ContentEmb:
    """An embedder whose output depends only on the pooling mask (not on the row's position in a batch), so that the product
    (embeds the active pairs, batched) and the oracle (embeds every pair, one call each) see the same vectors: `n_voices`
    well separated directions picked by the mask's frame count, plus small seeded noise; an all-zero mask gives NaN."""
    DIM = 32

    def __init__(self, n_voices=3, seed=5):
        self.dirs = np.random.default_rng(seed).standard_normal((n_voices, self.DIM)).astype(np.float32) * 3.0

    def emb_fn(self, chunk, mask):
        n = int(np.asarray(mask).sum())
        if n == 0:
            return np.full(self.DIM, np.nan, dtype=np.float32)
        first = int(np.flatnonzero(np.asarray(mask) > 0)[0])
        return (self.dirs[(n // 7) % len(self.dirs)] + 0.3 * np.random.default_rng(n * 1000 + first).standard_normal(self.DIM)).astype(np.float32)

    def embed_chunks(self, chunks, weights, mask_chunk):
        return torch.from_numpy(np.stack([self.emb_fn(None, w) for w in weights.numpy()]))


def run(n_items=4):
    from clearconverse_amd import pipelines as P
    items = [{"waveform": torch.from_numpy(np.random.default_rng(i).standard_normal(480000 - 7000 * (i % 3)).astype(np.float32)),
              "sample_rate": 16000} for i in range(n_items)]
    dia = P.SpeakerDiarization(FakeSeg(7, True), FakeEmb()).batch(items, min_speakers=1, max_speakers=2)
    vad = P.VoiceActivityDetection(FakeSeg(3, False, seed=7)).batch(items)
    return {"diarization": [[[round(s.start, 6), round(s.end, 6), l] for s, _, l in a.itertracks(yield_label=True)] for a in dia],
            "vad": [[[round(s.start, 6), round(s.end, 6)] for s, _ in a.itertracks()] for a in vad]}


if __name__ == "__main__":
    with open(GOLDEN, "w") as f:
        json.dump(run(), f)
    print("wrote", GOLDEN)

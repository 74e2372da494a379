"""Scripted segmentation weights: a PyanNet state dict (same key layout as `pyannote/segmentation*`) whose outputs follow the
activity schedule of ONE synthetic clip, so that VAD / diarization tests see silences, two speakers and an overlap instead of the
constant output seeded random weights give.  TEST INFRASTRUCTURE ONLY.

How: SincNet keeps seeded weights.  The BiLSTM stack is set up as a smoothing pass-through (W_hh = 0; input gate sigma(b_i) =
alpha, forget gate 1 - alpha: each cell is an exponential moving average of tanh(w . x) -- forward and backward, so the 4 Hz
syllabic modulation of the synthetic voices is averaged out); the two linear layers pass +x and -x through their LeakyReLUs;
the classifier is a ridge-regression fit (oracle features of the clip's own windows -> the schedule's classes).  The fit runs
on the CPU oracle (oracle/pyannote_ref.py); the GPU test then checks that libccx reproduces the oracle's decisions with these
weights."""
from __future__ import annotations

import math
from typing import Dict, Sequence, Tuple

import numpy as np
import torch

from clearconverse_amd.audio import SCHEDULE_10S, SCHEDULE_30S, synthetic_clip
from clearconverse_amd.weights import synthetic_pyannet_state_dict

SR = 16000


def _passthrough_lstm(sd: Dict[str, torch.Tensor], alpha: Sequence[float] = (0.25, 0.25, 1.0, 1.0), eps: float = 0.15):
    logit = lambda p: math.log(p / (1.0 - p))
    for l in range(4):
        cin = 60 if l == 0 else 256
        a = min(max(alpha[l], 1e-4), 1 - 1e-4)
        for d, sfx in enumerate(("", "_reverse")):
            w = torch.zeros(512, cin)
            n_in = 60 if l == 0 else 120
            # cell-candidate rows (torch gate order i, f, g, o): forward cells 0..59 and reverse cells 0..59 both carry feature j;
            # from layer 1 on the input is [forward h | reverse h]: average the two directions
            for j in range(60):
                if l == 0:
                    w[256 + j, j] = eps
                else:
                    w[256 + j, j] = 0.5 / eps if l == 1 else 0.5
                    w[256 + j, 128 + j] = 0.5 / eps if l == 1 else 0.5
            if l == 1:
                w[256:256 + 60] *= eps          # keep tanh in its linear range: h1 ~ eps * smoothed features again
            b = torch.zeros(512)
            b[0:128] = logit(a)                 # input gate  = alpha
            b[128:256] = logit(1.0 - a)         # forget gate = 1 - alpha
            b[384:512] = 10.0                   # output gate = 1
            sd[f"lstm.weight_ih_l{l}{sfx}"] = w
            sd[f"lstm.weight_hh_l{l}{sfx}"] = torch.zeros(512, 128)
            sd[f"lstm.bias_ih_l{l}{sfx}"] = b
            sd[f"lstm.bias_hh_l{l}{sfx}"] = torch.zeros(512)


def _passthrough_linear(sd: Dict[str, torch.Tensor], gain: float = 4.0):
    # linear.0: [fwd h | rev h] (256) -> 60 x (+mean) and 60 x (-mean); LeakyReLU(x) - LeakyReLU(-x) = 1.01 x
    w0 = torch.zeros(128, 256)
    for j in range(60):
        w0[j, j] = w0[j, 128 + j] = 0.5 * gain
        w0[60 + j, j] = w0[60 + j, 128 + j] = -0.5 * gain
    sd["linear.0.weight"], sd["linear.0.bias"] = w0, torch.zeros(128)
    w1 = torch.zeros(128, 128)
    for j in range(60):
        w1[j, j], w1[j, 60 + j] = 1.0, -1.0
        w1[60 + j, j], w1[60 + j, 60 + j] = -1.0, 1.0
    sd["linear.1.weight"], sd["linear.1.bias"] = w1, torch.zeros(128)


def schedule_sets(times: np.ndarray, seconds: float) -> np.ndarray:
    """[frames, 2] activity of speakers A (column 0) and B (column 1) at the given times."""
    sched = SCHEDULE_30S if seconds > 10.0 else SCHEDULE_10S
    scale = seconds / (30.0 if seconds > 10.0 else 10.0)
    act = np.zeros((len(times), 2), dtype=np.float32)
    for spk, s, e in sched:
        act[(times >= s * scale) & (times < e * scale), 0 if spk == "A" else 1] = 1.0
    return act


def scripted_pyannet_state_dict(clip_index: int, n_classes: int, powerset: bool, seconds: float = 30.0, window_s: float = 10.0,
                                seed: int = 3, margin: float = 6.0, ridge: float = 1e-2) -> Tuple[Dict[str, torch.Tensor], dict]:
    """-> (state dict, fit report).  The classifier is fitted on the oracle features of every sliding window of
    synthetic_clip(clip_index, seconds) (10 % step, as the pipelines cut them)."""
    from oracle import pyannote_ref as P
    sd = synthetic_pyannet_state_dict(n_classes, seed=seed)
    _passthrough_lstm(sd)
    _passthrough_linear(sd)
    clip = synthetic_clip(clip_index, seconds)
    win, step = int(window_s * SR), int(window_s * 0.1 * SR)
    starts = list(range(0, max(1, len(clip) - win + 1), step))
    chunks = np.stack([np.pad(clip[s:s + win], (0, max(0, win - len(clip[s:s + win])))) for s in starts])
    osd = dict(sd); osd["powerset"] = torch.tensor(1 if powerset else 0)
    with torch.no_grad():
        feats = P.pyannet_forward(osd, torch.from_numpy(chunks)[:, None], return_features=True).numpy()     # [W, F, 128]
    F_ = feats.shape[1]
    X, Y = [], []
    for w, s in enumerate(starts):
        t = (s + 270.0 * np.arange(F_) + 495.5) / SR                      # frame centres
        act = schedule_sets(t, seconds)
        if powerset:        # classes: {}, {0}, {1}, {2}, {0,1}, {0,2}, {1,2}; A -> local speaker 0, B -> 1
            cls = np.where(act.sum(1) == 2, 4, np.where(act[:, 0] == 1, 1, np.where(act[:, 1] == 1, 2, 0)))
            y = np.full((F_, n_classes), -margin, dtype=np.float64)
            y[np.arange(F_), cls] = margin
        else:
            y = np.full((F_, n_classes), -margin, dtype=np.float64)
            y[:, :2] = np.where(act > 0, margin, -margin)
            if n_classes > 2:           # third output: anybody speaking (the VAD pipeline takes the max over the outputs)
                y[:, 2] = np.where(act.sum(1) > 0, margin, -margin)
        X.append(feats[w]); Y.append(y)
    X = np.concatenate(X).astype(np.float64); Y = np.concatenate(Y)
    Xb = np.concatenate([X, np.ones((X.shape[0], 1))], axis=1)
    A = Xb.T @ Xb + ridge * X.shape[0] * np.eye(Xb.shape[1])
    Wb = np.linalg.solve(A, Xb.T @ Y)                                       # [129, C]
    sd["classifier.weight"] = torch.from_numpy(Wb[:-1].T.astype(np.float32)).contiguous()
    sd["classifier.bias"] = torch.from_numpy(Wb[-1].astype(np.float32)).contiguous()
    pred = Xb @ Wb
    if powerset:
        acc = float((pred.argmax(1) == Y.argmax(1)).mean())
    else:
        acc = float(((pred.max(1) > 0) == (Y.max(1) > 0)).mean())
    return sd, dict(frames=int(X.shape[0]), accuracy=acc)

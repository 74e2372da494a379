"""-m gpu: K1 (`resample_sinc_kernel`, csrc/resample.hip) through the C ABI and through `SincResampler` (the drop-in for the
reference's `self.resampler`, back/api.py:824-830) against oracle/resample_ref.py.  The table is identical (test_resample_cpu.py);
kernel and oracle differ only in fp32 summation order over <= 475 taps: abs 2e-6 on signals in [-1, 1]."""
import math

import numpy as np
import pytest
import torch

from tests.conftest import within

from clearconverse_amd.audio import SincResampler, synthetic_clip
from oracle import resample_ref as RS

pytestmark = pytest.mark.gpu


def _signal(sr, seconds, seed):
    rng = np.random.default_rng(seed)
    n = int(sr * seconds)
    t = np.arange(n) / sr
    x = sum(np.sin(2 * np.pi * f * t + rng.uniform(0, 6.28)) / (i + 2) for i, f in enumerate((97.0, 440.0, 1234.5, 3000.0)))
    x = x + 0.05 * rng.standard_normal(n)
    return (x / np.abs(x).max()).astype(np.float32)


@pytest.mark.parametrize("orig,seconds", [(44100, 3.3), (8000, 2.0), (48000, 1.01), (22050, 0.7), (11025, 1.9), (96000, 0.5)])
def test_resampler_matches_oracle(ccx_ctx, orig, seconds):
    x = _signal(orig, seconds, orig)
    r = SincResampler(orig, 16000, ctx=ccx_ctx)
    assert r.orig_freq == orig                              # the attribute the reference inspects
    y = r(torch.from_numpy(x)[None]).cpu()
    ref = RS.resample(torch.from_numpy(x)[None], orig, 16000)
    assert y.shape == ref.shape and y.shape[1] == math.ceil(16000 * len(x) / orig)
    within("resampler: max abs error / max(1, |ref|)", float((y - ref).abs().max()) / max(1.0, float(ref.abs().max())), 5e-7, orig)


def test_resampler_ragged_batch_through_the_c_abi(ccx_ctx):
    """Rows of different length in one launch (n_in / n_out tables), lengths 1 .. 50 001 incl. a row shorter than the filter."""
    from clearconverse_amd.audio import sinc_resample_kernel
    orig, new = 44100, 16000
    kT, width, o, n = sinc_resample_kernel(orig, new)
    lens = [50001, 1, 7, 441, 442, 12345]
    stride = max(lens)
    x = torch.zeros(len(lens), stride)
    for i, L in enumerate(lens):
        x[i, :L] = torch.from_numpy(_signal(orig, L / orig + 1e-9, i)[:L]) if L > 8 else torch.linspace(-0.5, 0.5, L)
    n_out = [math.ceil(n * L / o) for L in lens]
    xd, kd = x.cuda(), torch.from_numpy(kT).cuda()
    y = torch.full((len(lens), max(n_out)), 9.0, device="cuda")
    ni = torch.tensor(lens, dtype=torch.int32, device="cuda")
    no = torch.tensor(n_out, dtype=torch.int32, device="cuda")
    ccx_ctx.check(ccx_ctx.lib.ccx_resample_sinc(ccx_ctx.handle, xd.data_ptr(), stride, ni.data_ptr(), len(lens), o, n, width, kd.data_ptr(),
                                                y.data_ptr(), y.shape[1], no.data_ptr(), max(n_out),
                                                int(torch.cuda.current_stream().cuda_stream)), "ccx_resample_sinc")
    torch.cuda.synchronize()
    y = y.cpu()
    for i, L in enumerate(lens):
        ref = RS.resample(x[i:i + 1, :L], orig, new)[0]
        assert ref.shape[0] == n_out[i]
        within("resampler (ragged batch): max abs error", float((y[i, :n_out[i]] - ref).abs().max()), 5e-7, i)
        assert bool((y[i, n_out[i]:] == 9.0).all()), i       # nothing written past a row's output length


def test_load_audio_resamples_a_44k1_stereo_wav_on_the_device(ccx_ctx, tmp_path):
    """A3 with K1 on the GPU: 44.1 kHz stereo WAV -> mono -> 16 kHz -> gate -> peak 1 (back/api.py:820-836)."""
    from clearconverse_amd.audio import write_wav
    from clearconverse_amd.denoise import SpectralGate
    from clearconverse_amd.processor import Config, EnhancedAudioProcessor
    from oracle import spectral_gate_ref as G
    x = _signal(44100, 2.5, 3)
    st = np.stack([x, 0.5 * x])
    path = str(tmp_path / "s.wav")
    write_wav(path, st, sr=44100)
    p = EnhancedAudioProcessor(Config(), load_models_immediately=False)
    p.denoiser = SpectralGate(max_samples=160000, max_clips=1, ctx=ccx_ctx)
    a, sr = p.load_audio(path)
    assert sr == 16000 and a.shape == (1, math.ceil(160 * len(x) / 441)) and p.resampler.orig_freq == 44100
    pcm = np.clip(np.round(st.T * 32767.0), -32768, 32767).astype(np.float32).T / 32768.0       # what the 16-bit file holds
    mono = torch.from_numpy(pcm.mean(axis=0, keepdims=True))
    ref = G.reduce_noise(RS.resample(mono, 44100, 16000)[0].numpy(), 16000, prop_decrease=0.5)
    ref = ref / (np.abs(ref).max() + 1e-8)
    rel = float(np.linalg.norm(a[0].cpu().numpy() - ref) / np.linalg.norm(ref))
    within("load_audio (44.1 kHz stereo WAV -> resample -> gate -> peak): rel-L2", rel, 5e-7)

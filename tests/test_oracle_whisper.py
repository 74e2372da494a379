"""CPU: pin oracle/whisper_ref.py (the restatement of openai-whisper) against the independently
written `transformers` Whisper implementation installed in this image, with shared seeded weights.
transformers is NOT the reference's dependency; agreement only shows the restatement implements the
published architecture/front end correctly (Parity)."""
import math

import numpy as np
import pytest
import torch

from clearconverse_amd.audio import mel_filterbank, synthetic_clip
from clearconverse_amd.tokenizer import DecodeRules, SUPPRESS_TOKENS
from clearconverse_amd.weights import WhisperDims, synthetic_whisper_state_dict
from oracle import whisper_ref as R


def test_mel_filterbank_matches_transformers():
    from transformers.audio_utils import mel_filter_bank
    hf = mel_filter_bank(201, 80, 0.0, 8000.0, 16000, norm="slaney", mel_scale="slaney").T
    assert np.abs(R.mel_filters(80) - hf).max() < 1e-7
    assert np.abs(mel_filterbank(80) - hf).max() < 1e-7    # the product's own table too


@pytest.mark.parametrize("seconds", [30.0, 9.0, 1.3])
def test_logmel_matches_hf_feature_extractor(seconds):
    from transformers import WhisperFeatureExtractor
    clip = synthetic_clip(1, 30.0)[: int(seconds * 16000)]
    ours = R.log_mel_spectrogram(torch.from_numpy(clip))[:, :3000]
    fe = WhisperFeatureExtractor(feature_size=80, sampling_rate=16000, hop_length=160, chunk_length=30, n_fft=400)
    hf = fe(clip, sampling_rate=16000, return_tensors="pt").input_features[0]
    # HF pads/trims the AUDIO to exactly 30 s and lets the STFT reflect at sample 480000; openai-whisper
    # appends 30 s of zeros first, so for a full 30 s clip the last two frames see zeros instead of the
    # reflection.  Everything else must agree.
    n = 2998 if seconds >= 30.0 else 3000
    assert float((ours[:, :n] - hf[:, :n]).abs().max()) < 1e-4


def _hf_model(dims, sd):
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    cfg = WhisperConfig(vocab_size=dims.n_vocab, num_mel_bins=dims.n_mels, d_model=dims.n_audio_state,
                        encoder_layers=dims.n_audio_layer, encoder_attention_heads=dims.n_audio_head,
                        decoder_layers=dims.n_text_layer, decoder_attention_heads=dims.n_text_head,
                        encoder_ffn_dim=4 * dims.n_audio_state, decoder_ffn_dim=4 * dims.n_text_state,
                        max_source_positions=dims.n_audio_ctx, max_target_positions=dims.n_text_ctx,
                        activation_function="gelu", dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
                        pad_token_id=50256, bos_token_id=50256, eos_token_id=50256, decoder_start_token_id=50257)
    m = WhisperForConditionalGeneration(cfg).eval()

    def mp(k):
        k = k.replace("encoder.blocks.", "model.encoder.layers.").replace("decoder.blocks.", "model.decoder.layers.")
        k = k.replace(".cross_attn_ln.", ".encoder_attn_layer_norm.").replace(".cross_attn.", ".encoder_attn.")
        k = k.replace(".attn_ln.", ".self_attn_layer_norm.").replace(".attn.", ".self_attn.")
        k = k.replace(".query.", ".q_proj.").replace(".key.", ".k_proj.").replace(".value.", ".v_proj.").replace(".out.", ".out_proj.")
        k = k.replace(".mlp_ln.", ".final_layer_norm.").replace(".mlp.0.", ".fc1.").replace(".mlp.2.", ".fc2.")
        k = k.replace("encoder.conv", "model.encoder.conv")
        k = k.replace("encoder.ln_post.", "model.encoder.layer_norm.").replace("decoder.ln.", "model.decoder.layer_norm.")
        k = k.replace("encoder.positional_embedding", "model.encoder.embed_positions.weight")
        k = k.replace("decoder.positional_embedding", "model.decoder.embed_positions.weight")
        k = k.replace("decoder.token_embedding.", "model.decoder.embed_tokens.")
        return k
    hsd = {mp(k): v.clone() for k, v in sd.items()}
    hsd["proj_out.weight"] = sd["decoder.token_embedding.weight"].clone()
    missing, unexpected = m.load_state_dict(hsd, strict=False)
    assert not unexpected, unexpected
    assert all("k_proj.bias" in k for k in missing), missing   # HF has no key bias either; anything else must load
    return m


@pytest.fixture(scope="module")
def mini():
    dims = WhisperDims.mini(n_layer=2, n_state=128)
    sd = synthetic_whisper_state_dict(dims, seed=5)
    return dims, sd, R.WhisperRef(R.Dims(**dims.__dict__), sd)


def test_encoder_decoder_match_transformers(mini):
    dims, sd, orc = mini
    hf = _hf_model(dims, sd)
    clip = synthetic_clip(2, 30.0)[: 16000 * 7]
    mel = R.pad_or_trim(R.log_mel_spectrogram(torch.from_numpy(clip))[:, : len(clip) // 160], 3000)[None]
    toks = torch.tensor([[50257, 50363, 400, 1234, 31000, 50400]])
    with torch.no_grad():
        xa = orc.encode(mel)
        hxa = hf.model.encoder(mel).last_hidden_state
        assert float((xa - hxa).abs().max()) < 2e-4 * float(hxa.abs().max())
        lg = orc.decoder_logits(toks, xa)
        hlg = hf(input_features=mel, decoder_input_ids=toks).logits
    assert float((lg - hlg).abs().max()) < 1e-3 * float(hlg.abs().max())
    assert torch.equal(lg.argmax(-1), hlg.argmax(-1))


def test_cached_decoder_equals_full_recompute(mini):
    dims, sd, orc = mini
    xa = torch.randn(1, dims.n_audio_ctx, dims.n_audio_state, generator=torch.Generator().manual_seed(1))
    toks = torch.tensor([[50360, 11, 22, 50257, 50363, 7, 8]])
    full = orc.decoder_logits(toks, xa)
    dec = R.CachedDecoder(orc, xa)
    a = dec.step(toks[:, :4])
    b = dec.step(toks[:, 4:5])
    c = dec.step(toks[:, 5:])
    inc = torch.cat([a, b, c], dim=1)
    assert float((full - inc).abs().max()) < 1e-4


def _rules():
    return R.Rules(suppress=tuple(SUPPRESS_TOKENS))


def test_filters_first_step_forces_initial_timestamp():
    r = _rules()
    lg = torch.zeros(51864)
    lg[100] = 50.0                      # a very likely text token must still be banned on step 0
    out = R.apply_filters(lg, [], r)
    allowed = torch.isfinite(out).nonzero().flatten().tolist()
    assert allowed == list(range(r.timestamp_begin, r.timestamp_begin + 51))


def test_filters_after_text_timestamp_pair_rules():
    r = _rules()
    tsb = r.timestamp_begin
    lg = torch.zeros(51864)
    lg[r.eot] = 10.0                    # keep one text id above the summed timestamp mass (last rule)
    # <|0.00|> text <|1.00|>: a single (unpaired) timestamp -> only timestamps >= it, or eot and above
    out = R.apply_filters(lg, [tsb, 400, tsb + 50], r)
    fin = torch.isfinite(out)
    assert not fin[: r.eot].any()
    assert fin[r.eot]
    assert not fin[tsb: tsb + 50].any() and fin[tsb + 50]
    # <|0.00|> text <|1.00|><|1.00|>: closed pair -> no timestamp may follow
    lg2 = torch.zeros(51864)
    lg2[300] = 5.0                      # keep the text mass above the (banned) timestamp mass
    out2 = R.apply_filters(lg2, [tsb, 400, tsb + 50, tsb + 50], r)
    assert not torch.isfinite(out2[tsb:]).any()
    assert torch.isfinite(out2[300])


def test_filters_suppress_list_and_timestamp_mass_rule():
    r = _rules()
    tsb = r.timestamp_begin
    lg = torch.full((51864,), -5.0)
    lg[tsb + 10: tsb + 400] = -1.0      # lots of probability mass spread over timestamps
    lg[700] = 0.5                       # best single text token still below the summed timestamp mass
    out = R.apply_filters(lg, [tsb, 400], r)
    assert not torch.isfinite(out[:tsb]).any()
    for t in SUPPRESS_TOKENS:
        assert out[t] == float("-inf")
    assert out[r.no_timestamps] == float("-inf")


def test_greedy_decode_variants_agree(mini):
    dims, sd, orc = mini
    r = _rules()
    xa = torch.randn(1, dims.n_audio_ctx, dims.n_audio_state, generator=torch.Generator().manual_seed(2)) * 0.5
    a = R.greedy_decode(orc, xa, [[r.sot]], r, sample_len=6)[0]
    b = R.greedy_decode_cached(orc, xa, [r.sot], r, sample_len=6)
    assert a.tokens == b.tokens
    assert abs(a.sum_logprob - b.sum_logprob) < 1e-3
    assert abs(a.no_speech_prob - b.no_speech_prob) < 1e-6
    assert a.tokens[0] >= r.timestamp_begin


def test_philox_known_answer_vectors():
    """Random123 kat_vectors for philox4x32-10: the RNG the temperature > 0 path draws from (GPU kernel and oracle)."""
    from oracle.whisper_ref import philox4x32
    import numpy as np
    cases = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
             ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
             ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in cases:
        got = philox4x32(np.array([ctr], dtype=np.uint32), np.array(key, dtype=np.uint32))[0]
        assert tuple(int(x) for x in got) == want


def test_gumbel_max_sampling_is_categorical():
    """argmax(logits / T + Gumbel) over the Philox stream reproduces softmax(logits / T) frequencies."""
    from oracle.whisper_ref import gumbel_noise, sample_token
    import numpy as np
    g = gumbel_noise(3, 0, 0, 51864)
    assert abs(float(g.mean()) - 0.5772) < 0.02 and abs(float(g.std()) - 1.2825) < 0.02
    lg = torch.tensor([1.0, 0.5, 0.0, -0.5, -1.0, float("-inf")])
    T, N = 0.5, 20000
    counts = np.zeros(6)
    for step in range(N):
        t, lp, _ = sample_token(lg, T, seed=11, row=2, step=step)
        counts[t] += 1
    p = torch.softmax(lg / T, -1).numpy()
    assert counts[5] == 0
    assert np.all(np.abs(counts / N - p) < 4 * np.sqrt(p * (1 - p) / N) + 1e-4)
    t0, lp0, _ = sample_token(lg, 0.0, 0, 0, 0)
    assert t0 == 0 and abs(lp0 - float(torch.log_softmax(lg, -1)[0])) < 1e-6

"""CPU restatement of the HOST half of `whisper.transcribe()` -- the window loop that turns decoded token ids into the one key
the reference reads, `['text']` (/root/reference/back/api.py:1103, 1447, 1488; the calls at 1286-1292, 1432-1438, 1474-1480 pass
`initial_prompt=`, `word_timestamps=`, `condition_on_previous_text=`, `temperature=`).

TEST INFRASTRUCTURE ONLY -- imported by tests/ (tests/test_transcribe_loop_cpu.py drives it and the product's
`clearconverse_amd.whisper.WindowLoop` with the same scripted decode results); never by the product path.

The loop lives in the third-party package `openai-whisper` (un-pinned in /root/reference/back/requirements.txt:12-19, not vendored
under /root/reference, not installed here), file whisper/transcribe.py, function `transcribe`.  It is restated below from
recollection [UPSTREAM-RECALL], statement by statement, in upstream's order; every block names the upstream statement it follows and
items I am least sure of carry a (?).  **PARITY UNPINNED**: the reference holds no test or fixture for this path (SURVEY.md
section 4) and the package cannot be imported, so nothing here is checked against upstream's own outputs.

Not restated (they cannot change `text` or `seek` in the reference's calls): the tqdm progress bar, `verbose` printing, language
detection (`.en` model: language = "en"), `clip_timestamps` other than its default "0" (one clip = the whole file),
`hallucination_silence_threshold` (default None), the compression-ratio / log-probability FALLBACK ladder (the reference passes ONE
temperature, so `decode_with_fallback` makes exactly one attempt and its `needs_fallback` flag is never acted on).

Word timestamps.  The reference passes `word_timestamps=True` in two of its three calls (back/api.py:1435, 1477).  Upstream then runs
`add_word_timestamps` (cross-attention DTW) after the segments of a window are built, and -- only when the window did NOT end on a
single timestamp -- moves `seek` to the end of the last aligned word.  The DTW itself is out of scope here (K11, SURVEY.md section
2a: its output is never read by the reference); what it can change is `seek`, so the rule is restated with the last word's end time
supplied by the caller (`last_word_end_fn`), and the product's deviation (it keeps the timestamp-token seek) is exactly the case
`last_word_end_fn is not None`, exercised in the tests as a documented difference.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

N_FRAMES = 3000            # audio.py: frames of a 30 s window
HOP_LENGTH = 160
SAMPLE_RATE = 16000
FRAMES_PER_SECOND = 100    # audio.py: exact_div(SAMPLE_RATE, HOP_LENGTH)


@dataclass
class ScriptedResult:
    """The fields of decoding.py::DecodingResult that transcribe() reads."""
    tokens: List[int]                  # sampled ids before eot (DecodingTask.run: tokens[sample_begin : first eot])
    avg_logprob: float = -0.5
    no_speech_prob: float = 0.0
    temperature: float = 0.0


@dataclass
class TokenizerIds:
    eot: int = 50256
    timestamp_begin: int = 50363


def transcribe_loop(content_frames: int, decode_fn: Callable[[int, int, List[int]], ScriptedResult], tok: TokenizerIds,
                    encode: Callable[[str], List[int]], decode: Callable[[Sequence[int]], str], *, n_text_ctx: int = 448,
                    n_audio_ctx: int = 1500, initial_prompt: Optional[str] = None, condition_on_previous_text: bool = True,
                    no_speech_threshold: Optional[float] = 0.6, logprob_threshold: Optional[float] = -1.0,
                    word_timestamps: bool = False,
                    last_word_end_fn: Optional[Callable[[List[dict]], Optional[float]]] = None) -> Dict[str, object]:
    """transcribe.py::transcribe, main loop.  `decode_fn(seek, segment_size, prompt_tokens)` stands for
    `decode_with_fallback(mel_segment)` with `decode_options["prompt"] = all_tokens[prompt_reset_since:]` (the model call);
    `encode` / `decode` are the tokenizer's.  Returns {'text', 'segments', 'seeks' (the seek of every decoded window), 'prompts'}."""
    # "seek_clips": clip_timestamps = "0" -> [(0, content_frames)]
    seek_clip_start, seek_clip_end = 0, content_frames
    seek = seek_clip_start
    input_stride = N_FRAMES // n_audio_ctx                          # exact_div(N_FRAMES, model.dims.n_audio_ctx) = 2
    time_precision = input_stride * HOP_LENGTH / SAMPLE_RATE        # 0.02 s
    all_tokens: List[int] = []
    all_segments: List[dict] = []
    prompt_reset_since = 0
    # "if initial_prompt is not None: initial_prompt_tokens = tokenizer.encode(' ' + initial_prompt.strip()); all_tokens.extend(...)"
    if initial_prompt is not None:
        initial_prompt_tokens = list(encode(" " + initial_prompt.strip()))
        all_tokens.extend(initial_prompt_tokens)
    else:
        initial_prompt_tokens = []
    seeks, prompts = [], []
    last_speech_timestamp = 0.0

    # "while clip_idx < len(seek_clips)" with one clip: run until seek reaches the clip's end
    while seek < seek_clip_end:
        time_offset = float(seek * HOP_LENGTH / SAMPLE_RATE)
        # "segment_size = min(N_FRAMES, content_frames - seek, seek_clip_end - seek)"
        segment_size = min(N_FRAMES, content_frames - seek, seek_clip_end - seek)
        segment_duration = segment_size * HOP_LENGTH / SAMPLE_RATE
        # "decode_options['prompt'] = all_tokens[prompt_reset_since:]; result = decode_with_fallback(mel_segment)"
        prompt = list(all_tokens[prompt_reset_since:])
        result = decode_fn(seek, segment_size, prompt)
        seeks.append(seek)
        prompts.append(prompt)
        tokens = list(result.tokens)

        # "if no_speech_threshold is not None: should_skip = result.no_speech_prob > no_speech_threshold;
        #  if logprob_threshold is not None and result.avg_logprob > logprob_threshold: should_skip = False;
        #  if should_skip: seek += segment_size; continue"
        if no_speech_threshold is not None:
            should_skip = result.no_speech_prob > no_speech_threshold
            if logprob_threshold is not None and result.avg_logprob > logprob_threshold:
                should_skip = False
            if should_skip:
                seek += segment_size
                continue

        previous_seek = seek
        current_segments: List[dict] = []

        def new_segment(start: float, end: float, toks: List[int]) -> dict:
            # "text_tokens = [token for token in tokens if token < tokenizer.eot]"
            return dict(seek=seek, start=start, end=end, text=decode([t for t in toks if t < tok.eot]), tokens=list(toks))

        # "timestamp_tokens = tokens.ge(tokenizer.timestamp_begin); single_timestamp_ending = timestamp_tokens[-2:].tolist() == [False, True]"
        timestamp_tokens = [t >= tok.timestamp_begin for t in tokens]
        single_timestamp_ending = timestamp_tokens[-2:] == [False, True]
        # "consecutive = torch.where(timestamp_tokens[:-1] & timestamp_tokens[1:])[0]; consecutive.add_(1)"
        consecutive = [i + 1 for i in range(len(tokens) - 1) if timestamp_tokens[i] and timestamp_tokens[i + 1]]
        if len(consecutive) > 0:
            # "if the output contains two consecutive timestamp tokens"
            slices = list(consecutive)
            if single_timestamp_ending:
                slices.append(len(tokens))
            last_slice = 0
            for current_slice in slices:
                sliced_tokens = tokens[last_slice:current_slice]
                start_timestamp_pos = sliced_tokens[0] - tok.timestamp_begin
                end_timestamp_pos = sliced_tokens[-1] - tok.timestamp_begin
                current_segments.append(new_segment(time_offset + start_timestamp_pos * time_precision,
                                                    time_offset + end_timestamp_pos * time_precision, sliced_tokens))
                last_slice = current_slice
            if single_timestamp_ending:
                # "single timestamp at the end means no speech after the last timestamp."
                seek += segment_size
            else:
                # "otherwise, ignore the unfinished segment and seek to the last timestamp"
                last_timestamp_pos = tokens[last_slice - 1] - tok.timestamp_begin
                seek += last_timestamp_pos * input_stride
        else:
            duration = segment_duration
            timestamps = [t for t, is_ts in zip(tokens, timestamp_tokens) if is_ts]
            # "if len(timestamps) > 0 and timestamps[-1].item() != tokenizer.timestamp_begin: no consecutive timestamps but it has a
            #  timestamp; use the last one."
            if len(timestamps) > 0 and timestamps[-1] != tok.timestamp_begin:
                last_timestamp_pos = timestamps[-1] - tok.timestamp_begin
                duration = last_timestamp_pos * time_precision
            current_segments.append(new_segment(time_offset, time_offset + duration, tokens))
            seek += segment_size

        if word_timestamps and last_word_end_fn is not None:
            # "add_word_timestamps(...)": the DTW is the caller's (last_word_end_fn stands for get_end(current_segments) after it)
            # "if not single_timestamp_ending: last_word_end = get_end(current_segments);
            #  if last_word_end is not None and last_word_end > time_offset: seek = round(last_word_end * FRAMES_PER_SECOND)"  (?)
            if not single_timestamp_ending:
                last_word_end = last_word_end_fn(current_segments)
                if last_word_end is not None and last_word_end > time_offset:
                    seek = round(last_word_end * FRAMES_PER_SECOND)
            last_word_end = last_word_end_fn(current_segments)
            if last_word_end is not None:
                last_speech_timestamp = last_word_end

        # "if a segment is instantaneous or does not contain text, clear it"
        for segment in current_segments:
            if segment["start"] == segment["end"] or segment["text"].strip() == "":
                segment["text"] = ""
                segment["tokens"] = []

        # "all_segments.extend(...); all_tokens.extend([token for segment in current_segments for token in segment['tokens']])"
        all_segments.extend(dict(id=i, **s) for i, s in enumerate(current_segments, start=len(all_segments)))
        all_tokens.extend(t for s in current_segments for t in s["tokens"])

        # "if not condition_on_previous_text or result.temperature > 0.5: prompt_reset_since = len(all_tokens)"
        if not condition_on_previous_text or result.temperature > 0.5:
            prompt_reset_since = len(all_tokens)
        assert seek > previous_seek, "ApplyTimestampRules makes every window advance (a closing timestamp is > its opening one)"

    # "return dict(text=tokenizer.decode(all_tokens[len(initial_prompt_tokens):]), segments=all_segments, language=language)"
    return dict(text=decode(all_tokens[len(initial_prompt_tokens):]), segments=all_segments, seeks=seeks, prompts=prompts,
                tokens=all_tokens[len(initial_prompt_tokens):], last_speech_timestamp=last_speech_timestamp)


def initial_tokens(prompt_tokens: Sequence[int], sot_prev: int, sot_sequence: Sequence[int], n_text_ctx: int = 448) -> List[int]:
    """decoding.py::DecodingTask._get_initial_tokens for the options transcribe() sets (no `prefix`):
    "if prompt := self.options.prompt: tokens = [self.tokenizer.sot_prev] + prompt_tokens[-(self.n_ctx // 2 - 1):] + tokens"."""
    tokens = list(sot_sequence)
    if len(prompt_tokens):
        tokens = [sot_prev] + list(prompt_tokens)[-(n_text_ctx // 2 - 1):] + tokens
    return tokens

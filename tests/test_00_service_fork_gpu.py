"""-m gpu: SURVEY 8f-4 on the device -- the reference's process model.  `POST /transcribe` forks a child per task
(/root/reference/back/api.py:2045-2049: multiprocessing.Process(target=run_transcription_process, ...)) from a server process that has
imported torch but never touched the GPU; the child loads the models, runs `EnhancedAudioProcessor.run` and reports through
progress.json / completed.txt / error.txt (1698-1754).  libccx claims to be safe for exactly that (no globals, no threads, context
created on first use: DESIGN.md section 1).

The test IS that: the parent imports torch and clearconverse_amd without initialising HIP, forks, and the child runs
`service.run_transcription_process` on a WAV with libccx-backed models.  This file sorts first so that, in the driver's single pytest
process, it runs before any other test has created a GPU context (nothing initialises HIP at collection time); if HIP is already up
in this process (another order, -k selections) the test skips: run the file on its own."""
import json
import multiprocessing as mp
import os
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _child(task_id: str, wav: str, out_dir: str, conn) -> None:
    """Runs in the forked child: everything GPU happens here."""
    try:
        from clearconverse_amd import service
        from clearconverse_amd.models import build_state_dicts, load_models
        from clearconverse_amd.processor import Config, EnhancedAudioProcessor
        from clearconverse_amd.weights import SepDims, WhisperDims
        from tests.scripted_nets import scripted_pyannet_state_dict

        def loader(cfg, dev):       # reduced Whisper / SepFormer depth (same kernels), scripted segmentation so that a transcript comes out
            sds = build_state_dicts(None, whisper_dims=WhisperDims.mini(2, 128), sep_dims=SepDims(n_layers=2), seed=0)
            sds["pyannet_diar"], _ = scripted_pyannet_state_dict(1, 7, True)
            sds["pyannet_vad"], _ = scripted_pyannet_state_dict(1, 3, False, window_s=5.0, seed=4)
            return load_models(cfg, dev, whisper_batch=16, state_dicts=sds, sep_tokens=60_000, max_crops=128)
        proc = EnhancedAudioProcessor(Config(temperature=0.0, min_speakers=2, max_speakers=2), load_models_immediately=False, model_loader=loader)
        path = service.run_transcription_process(task_id, wav, out_dir, processor=proc)
        import ctypes
        maps = open("/proc/self/maps").read()
        conn.send(dict(transcript_path=path, cuda_initialised=torch.cuda.is_initialized(), libccx_mapped="libccx.so" in maps))
    except BaseException as e:  # noqa: BLE001
        conn.send(dict(error=repr(e)))
        raise
    finally:
        conn.close()


def _parent(work: str) -> dict:
    """Server side: no GPU call before or after the fork."""
    import clearconverse_amd  # noqa: F401
    from clearconverse_amd import service  # noqa: F401
    from clearconverse_amd.audio import synthetic_clip, write_wav
    assert not torch.cuda.is_initialized()
    wav = os.path.join(work, "upload.wav")
    write_wav(wav, synthetic_clip(1, 30.0))
    out_dir = os.path.join(work, "processed_audio")
    os.makedirs(os.path.join(out_dir, "task-1"), exist_ok=True)
    open(os.path.join(out_dir, "task-1", "in_progress.txt"), "w").write("started")          # the API layer creates it (2040-2043)
    ctx = mp.get_context("fork")
    rx, tx = ctx.Pipe(duplex=False)
    p = ctx.Process(target=_child, args=("task-1", wav, out_dir, tx))
    p.start()
    tx.close()
    msg = rx.recv() if rx.poll(600) else dict(error="child sent nothing within 600 s")
    p.join(60)
    return dict(msg=msg, exitcode=p.exitcode, parent_cuda_initialised=torch.cuda.is_initialized(), out_dir=out_dir)


def _check(r: dict) -> None:
    assert "error" not in r["msg"], r["msg"]
    assert r["exitcode"] == 0
    assert r["parent_cuda_initialised"] is False                       # the server never initialised HIP
    assert r["msg"]["cuda_initialised"] is True and r["msg"]["libccx_mapped"] is True
    d = Path(r["out_dir"]) / "task-1"
    assert json.loads((d / "progress.json").read_text()) == {"progress": 100, "message": "Transcription complete"}
    assert (d / "completed.txt").read_text().startswith("Transcription completed at ")
    assert not (d / "in_progress.txt").exists() and not (d / "error.txt").exists()
    text = (d / "transcript.txt").read_text(encoding="utf-8")
    assert r["msg"]["transcript_path"] == str(d / "transcript.txt")
    assert text.startswith("[SPEAKER_") and "SPEAKER_A" in text and "SPEAKER_B" in text
    assert any((d / "regular_segments").glob("*.wav")) or any((d / "overlap_segments").glob("*.wav"))


def test_forked_child_of_a_gpu_free_parent_runs_the_task_protocol(tmp_path):
    if torch.cuda.is_initialized():
        # another order / selection initialised HIP in this process first: forking now would hand the child a broken runtime, and
        # starting a fresh interpreter from a GPU-initialised process is exactly the exec the GPU box forbids.  In the driver's
        # `pytest tests -m gpu` this file runs first and the branch is never taken.
        pytest.skip("HIP is already initialised in this process; run this file first (python -m pytest tests/test_00_service_fork_gpu.py -m gpu)")
    _check(_parent(str(tmp_path)))                                       # this pytest process is the GPU-free server

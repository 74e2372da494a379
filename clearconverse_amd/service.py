"""Task-runner shim: the file protocol of the reference's `run_transcription_process`
(/root/reference/back/api.py:1689-1754) without FastAPI, plus a persistent worker that keeps ONE loaded model set
for all tasks instead of the reference's per-task process + model reload (back/api.py:1718, 1211-1213, 2045-2049).

Protocol, per task, inside `<output_dir>/<task_id>/`:
  progress.json   {"progress": int percent, "message": str}, rewritten on every progress callback
  in_progress.txt removed when the task ends (the API layer creates it)
  completed.txt   written when `run` returned (even with an empty result, as upstream); a task that already has it
                  is skipped
  error.txt       "Error: <message>" when `run` raised; progress then ends at 100 with the error text
  transcript.txt / regular_segments / overlap_segments: written by EnhancedAudioProcessor.run itself
"""
from __future__ import annotations

import json
import logging
import os
from datetime import datetime
from typing import Callable, Optional

log = logging.getLogger("clearconverse_amd.service")


def _progress_writer(task_dir: str, task_id: str) -> Callable[[int, str], None]:
    def progress_callback(percent: int, message: str) -> None:
        tmp = os.path.join(task_dir, "progress.json.tmp")
        with open(tmp, "w") as f:
            json.dump({"progress": percent, "message": message}, f)
        os.replace(tmp, os.path.join(task_dir, "progress.json"))      # readers never see a half-written file
        log.info("Task %s: %s%% - %s", task_id, percent, message)
    return progress_callback


def run_transcription_process(task_id: str, file_path: str, output_dir: str, processor=None, config=None) -> Optional[str]:
    """One task, same observable files as the reference function of the same name.  `processor`: an already loaded
    EnhancedAudioProcessor (persistent worker); None builds one lazily like upstream.  Returns the transcript path
    (or None)."""
    task_dir = os.path.join(output_dir, task_id)
    os.makedirs(task_dir, exist_ok=True)
    completed = os.path.join(task_dir, "completed.txt")
    if os.path.exists(completed):
        log.info("Task %s already completed, skipping", task_id)
        return None
    progress = _progress_writer(task_dir, task_id)
    in_progress = os.path.join(task_dir, "in_progress.txt")
    try:
        if processor is None:
            from .processor import Config, EnhancedAudioProcessor
            processor = EnhancedAudioProcessor(config or Config(auth_token=os.getenv("HF_AUTH_TOKEN", "")), load_models_immediately=False)
        progress(5, "Starting model initialization...")
        _, _, transcript_path = processor.run(file_path, output_dir=task_dir, debug_mode=False, progress_callback=progress)
        if os.path.exists(in_progress):
            os.remove(in_progress)
        progress(100, "Transcription complete")
        with open(completed, "w") as f:
            f.write(f"Transcription completed at {datetime.now().isoformat()}")
        return transcript_path
    except Exception as e:  # noqa: BLE001 -- the protocol reports every failure through error.txt
        log.error("Error in transcription process: %s", e)
        with open(os.path.join(task_dir, "error.txt"), "w") as f:
            f.write(f"Error: {e}")
        if os.path.exists(in_progress):
            os.remove(in_progress)
        progress(100, f"Error: {e}")
        return None


class PersistentWorker:
    """Keeps one EnhancedAudioProcessor (weights resident in HBM, hipGraphs warm) and runs tasks one after the other.
    One worker per GPU; tasks of different GPUs are independent."""

    def __init__(self, processor=None, config=None):
        if processor is None:
            from .processor import Config, EnhancedAudioProcessor
            processor = EnhancedAudioProcessor(config or Config(auth_token=os.getenv("HF_AUTH_TOKEN", "")), load_models_immediately=True)
        self.processor = processor
        self.done = 0

    def submit(self, task_id: str, file_path: str, output_dir: str) -> Optional[str]:
        out = run_transcription_process(task_id, file_path, output_dir, processor=self.processor)
        self.done += 1
        return out

    def serve_directory(self, inbox: str, output_dir: str, poll: Optional[Callable[[], bool]] = None) -> int:
        """Run every `<inbox>/<task_id>.json` ({"file": path}) once; returns the number of tasks run.  `poll` (optional)
        is asked after each pass whether to look again (a service loop would sleep and return True)."""
        n = 0
        while True:
            for name in sorted(os.listdir(inbox)):
                if not name.endswith(".json"):
                    continue
                task_id = name[:-5]
                if os.path.exists(os.path.join(output_dir, task_id, "completed.txt")) or os.path.exists(os.path.join(output_dir, task_id, "error.txt")):
                    continue
                with open(os.path.join(inbox, name)) as f:
                    spec = json.load(f)
                self.submit(task_id, spec["file"], output_dir)
                n += 1
            if poll is None or not poll():
                return n

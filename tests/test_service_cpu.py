"""CPU: the task-runner file protocol (reference back/api.py:1689-1754) with a scripted processor."""
import json
import os

from clearconverse_amd.service import PersistentWorker, run_transcription_process


class _Proc:
    def __init__(self, fail=False, empty=False):
        self.fail, self.empty, self.calls = fail, empty, 0

    def run(self, input_file, output_dir="processed_audio", debug_mode=False, progress_callback=None):
        self.calls += 1
        progress_callback(40, "Processing audio")
        if self.fail:
            raise RuntimeError("boom")
        if self.empty:
            return None, None, None
        path = os.path.join(output_dir, "transcript.txt")
        with open(path, "w") as f:
            f.write("[SPEAKER_A] 0.00s - 1.00s\nhi\n\n")
        return input_file, "hi", path


def test_success_protocol(tmp_path):
    out = tmp_path / "out"
    (out / "t1").mkdir(parents=True)
    (out / "t1" / "in_progress.txt").write_text("x")
    p = _Proc()
    path = run_transcription_process("t1", "a.wav", str(out), processor=p)
    assert path == str(out / "t1" / "transcript.txt") and os.path.exists(path)
    assert json.loads((out / "t1" / "progress.json").read_text()) == {"progress": 100, "message": "Transcription complete"}
    assert (out / "t1" / "completed.txt").read_text().startswith("Transcription completed at ")
    assert not (out / "t1" / "in_progress.txt").exists() and not (out / "t1" / "error.txt").exists()
    assert run_transcription_process("t1", "a.wav", str(out), processor=p) is None and p.calls == 1   # completed: skipped


def test_error_and_empty_result_protocol(tmp_path):
    out = tmp_path / "out"
    assert run_transcription_process("bad", "a.wav", str(out), processor=_Proc(fail=True)) is None
    assert (out / "bad" / "error.txt").read_text() == "Error: boom"
    assert json.loads((out / "bad" / "progress.json").read_text()) == {"progress": 100, "message": "Error: boom"}
    assert not (out / "bad" / "completed.txt").exists()
    # run() returning (None, None, None) is not an exception upstream: the task still completes
    assert run_transcription_process("empty", "a.wav", str(out), processor=_Proc(empty=True)) is None
    assert (out / "empty" / "completed.txt").exists() and not (out / "empty" / "error.txt").exists()


def test_persistent_worker_runs_an_inbox_once(tmp_path):
    inbox, out = tmp_path / "inbox", tmp_path / "out"
    inbox.mkdir()
    for i in range(3):
        (inbox / f"task{i}.json").write_text(json.dumps({"file": f"clip{i}.wav"}))
    w = PersistentWorker(processor=_Proc())
    assert w.serve_directory(str(inbox), str(out)) == 3
    assert w.serve_directory(str(inbox), str(out)) == 0          # all completed
    assert w.processor.calls == 3 and w.done == 3
    assert all((out / f"task{i}" / "completed.txt").exists() for i in range(3))

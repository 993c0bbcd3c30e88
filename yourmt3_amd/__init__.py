"""MI355X-native audio -> MIDI-token transcription path (see README.md / DESIGN.md).  Heavy imports are lazy so that
`import yourmt3_amd` works without a GPU or the built library."""

__all__ = ["YMT3Config", "YourMT3", "TaskManager", "transcribe", "baseline_config"]


def __getattr__(name):
    if name in ("YMT3Config", "baseline_config"):
        from . import config
        return getattr(config, name)
    if name == "YourMT3":
        from .model import YourMT3
        return YourMT3
    if name == "TaskManager":
        from .task_manager import TaskManager
        return TaskManager
    if name == "transcribe":
        from .transcribe import transcribe
        return transcribe
    raise AttributeError(name)

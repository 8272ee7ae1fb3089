"""Locked print, same call shape as the reference's utils/logging.py:6-19."""
import threading

_lock = threading.Lock()


def log_message(message, verbose=False, always_print=False):
    if verbose or always_print:
        with _lock:
            print(f"{message}")

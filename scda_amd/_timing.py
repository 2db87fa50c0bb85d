"""optional timeline marks (debugging aid; no-ops unless enabled): host time, and with DEVICE also an event on the current stream"""
import time

ENABLED = False
DEVICE = False
MARKS = []
EVENTS = []


def mark(label):
    if ENABLED:
        MARKS.append((label, time.perf_counter()))
        if DEVICE:
            import torch
            if torch.cuda.is_current_stream_capturing():     # a timing event cannot be recorded inside a hipGraph capture
                return
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            EVENTS.append((label, e))

"""optional host-side timeline marks (debugging aid; no-ops unless enabled)"""
import time

ENABLED = False
MARKS = []


def mark(label):
    if ENABLED:
        MARKS.append((label, time.perf_counter()))

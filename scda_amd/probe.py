"""Instrumentation of the product path for the parity tests -- ONE object, injected, never a module-level switch.

ReLU / LeakyReLU / max-pool / RoI max-pool are not differentiable at ties, the RPN ranking is a discontinuous function of scores
that two correct fp32 implementations compute 1e-7 apart, and a dropout mask is a random draw: to compare kernels' ARITHMETIC with the
CPU oracle a test must hand the device run the oracle's discrete decisions.  A `Probe` carries those hand-overs:

    replay          object with act(y) / pool(y, idx) / roi(out, arg): the oracle's selection at the same call site
                    (tests/model_common.py ReplaySource), consulted by scda_amd.autograd_ops
    rpn_output      callable(conv_cls, conv_loc) -> (conv_cls, conv_loc): the RPN outputs the proposal ranking sees
                    (dropin/functions/rpn_proposal.py)
    dropout_masks   callable(shape, p, device) -> uint8 keep mask (scda_amd.layers.Dropout)

It reaches the product's modules in exactly two ways: `ScdaTrainer(..., probe=Probe(...))` (or `trainer.probe = ...`) makes it
visible for the duration of that trainer's `step()`; `with probe.installed(Probe(...)):` does the same around code that drives
modules directly.  Outside those scopes every accessor returns None: a production process has no way to end up with a hook set."""
import contextlib


class Probe:
    __slots__ = ("replay", "rpn_output", "dropout_masks")

    def __init__(self, replay=None, rpn_output=None, dropout_masks=None):
        self.replay, self.rpn_output, self.dropout_masks = replay, rpn_output, dropout_masks


_current = None


def active():
    return _current is not None


def replay():
    return _current.replay if _current is not None else None


def rpn_output():
    return _current.rpn_output if _current is not None else None


def dropout_masks():
    return _current.dropout_masks if _current is not None else None


@contextlib.contextmanager
def installed(p):
    """make `p` (a Probe, or None = nothing) visible to the product's modules inside the with-block"""
    global _current
    if p is None:
        yield None
        return
    if _current is not None and _current is not p:
        raise RuntimeError("scda_amd.probe: another Probe is already installed (nested trainers with different probes?)")
    prev, _current = _current, p
    try:
        yield p
    finally:
        _current = prev

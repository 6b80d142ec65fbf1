"""Epoch-wise schedules of the ``||T||^2`` coefficient (``train.py:139``: ``regularization_coeff =
regulizer.step()`` once per epoch) -- the class names, constructor arguments and attributes (``val``,
``cur_step``, ``base_val``, ``final_val``, ``num_steps``, ``strategy``) of ``src/utils/regularization.py``
on ONE schedule implementation: a policy is a direction (down / up / down-and-restart) plus a rule from
``RULES`` that moves the value by one step.

Behaviour kept: ``step()`` first counts, returns the CURRENT value once it has reached the final value
and otherwise moves it one step and returns the new value (so "exp" 1e-4 -> 3e-9 in 350 steps overshoots
once to 2.91e-9).  Behaviour fixed (Appendix A11): the reference's "cos" policy feeds
the current VALUE where a step index is expected; here it is the cosine ramp over ``num_steps`` steps.
"""
from __future__ import annotations

import math

# rule(policy, k) -> value after the k-th move (k = 1, 2, ...)
RULES = {
    "linear": lambda p, k: p.val + (p.final_val - p.base_val) / p.num_steps,
    "exp": lambda p, k: p.val * math.pow(p.final_val / p.base_val, 1.0 / p.num_steps),
    "cos": lambda p, k: p.final_val + (p.base_val - p.final_val) * (1 + math.cos(math.pi * min(k, p.num_steps) / p.num_steps)) / 2,
    "const": lambda p, k: p.val,
}


class RegularizationCoeffPolicy:
    """Constant coefficient; the base of every schedule."""
    direction = 0                      # -1: towards a smaller final value, +1: towards a larger one
    allowed = ()                       # rule names this policy accepts
    unsupported = "This policy is not supported"

    def __init__(self, base_val, num_steps, final_val=None, strategy="linear"):
        self.base_val = self.val = base_val
        self.num_steps, self.final_val, self.strategy = num_steps, final_val, strategy
        self.cur_step = self._moves = 0
        if self.direction and strategy not in self.allowed:
            raise NotImplementedError(self.unsupported)

    def _arrived(self):
        return not self.direction or (self.val - self.final_val) * self.direction >= 0

    def step(self):
        self.cur_step += 1
        if not self._arrived():
            self._moves += 1
            self.val = RULES[self.strategy](self, self._moves)
        return self.val


class IntervalPolicy(RegularizationCoeffPolicy):
    """A schedule between two values (abstract: no direction)."""


class SimpleDecreasingPolicy(IntervalPolicy):
    direction, allowed = -1, tuple(RULES)
    unsupported = "This decreasing policy is not supported"
    STRATEGIES = allowed


class SimpleIncreasingPolicy(IntervalPolicy):
    direction, allowed = +1, ("linear",)
    unsupported = "This increasing policy is not supported"


class CyclicDecreasingPolicy(SimpleDecreasingPolicy):
    """Restarts from the base value after reaching the final one."""

    def step(self):
        val = super().step()
        if self._arrived():
            self.val, self.cur_step, self._moves = self.base_val, 0, 0
        return val

"""Epoch-wise schedules of the ``||T||^2`` coefficient (``train.py:139``: ``regularization_coeff =
regulizer.step()`` once per epoch) -- same class names and constructor arguments as
``src/utils/regularization.py``, written as closed-form functions of the step count.

Behaviour kept: ``step()`` first counts, returns the CURRENT value once it has reached the final value
and otherwise moves it one step and returns the new value (so "exp" 1e-4 -> 3e-9 in 350 steps overshoots
once to 2.91e-9).  Behaviour fixed (Appendix A11): the reference's "cos" policy feeds
the current VALUE where a step index is expected; here it is the cosine ramp over ``num_steps`` steps.
"""
from __future__ import annotations

import math


class RegularizationCoeffPolicy:
    def __init__(self, base_val, num_steps):
        self.base_val, self.num_steps = base_val, num_steps
        self.val = base_val
        self.cur_step = 0

    def step(self):
        self.cur_step += 1
        return self.val


class IntervalPolicy(RegularizationCoeffPolicy):
    def __init__(self, base_val, num_steps, final_val):
        super().__init__(base_val, num_steps)
        self.final_val = final_val


class SimpleDecreasingPolicy(IntervalPolicy):
    STRATEGIES = ("linear", "exp", "cos", "const")

    def __init__(self, base_val, num_steps, final_val, strategy="linear"):
        super().__init__(base_val, num_steps, final_val)
        if strategy not in self.STRATEGIES:
            raise NotImplementedError("This decreasing policy is not supported")
        self.strategy = strategy
        self._moves = 0          # how many times the value has been moved

    def _next(self):
        k = self._moves + 1
        if self.strategy == "linear":
            return self.val - (self.base_val - self.final_val) / self.num_steps
        if self.strategy == "exp":
            return self.val * math.pow(self.final_val / self.base_val, 1.0 / self.num_steps)
        if self.strategy == "cos":
            return self.final_val + (self.base_val - self.final_val) * (1 + math.cos(math.pi * min(k, self.num_steps) / self.num_steps)) / 2
        return self.val          # const

    def step(self):
        self.cur_step += 1
        if self.val <= self.final_val:
            return self.val
        self.val = self._next()
        self._moves += 1
        return self.val


class SimpleIncreasingPolicy(IntervalPolicy):
    def __init__(self, base_val, num_steps, final_val, strategy="linear"):
        super().__init__(base_val, num_steps, final_val)
        if strategy != "linear":
            raise NotImplementedError("This increasing policy is not supported")
        self.strategy = strategy
        self.step_size = (final_val - base_val) / num_steps

    def step(self):
        self.cur_step += 1
        if self.val < self.final_val:
            self.val += self.step_size
        return self.val


class CyclicDecreasingPolicy(SimpleDecreasingPolicy):
    """Restarts from the base value after reaching the final one."""

    def step(self):
        val = super().step()
        if val <= self.final_val:
            self.val, self.cur_step, self._moves = self.base_val, 0, 0
        return val

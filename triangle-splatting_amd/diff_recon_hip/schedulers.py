"""Value schedules that drive the rasterizer's inputs over a training run: learning rates, densification /
pruning thresholds, the window sharpness gamma (1 -> 50 in the reference's configs) and the active SH degree.

Host-side mirrors of src/diff_recon/utils/scheduler.py:5-45 (same names, arguments and return conventions -- each factory returns a
function of the step) and of VanillaTSModel._set_gamma / _set_sh_degree (src/diff_recon/models/VanillaTS_model.py:548-565).  Pinned
against the reference's own functions through tests/golden/schedules.npz (tests/test_schedulers_cpu.py)."""
from __future__ import annotations

import math
from typing import Callable, List, Sequence


def exponential_scheduler(v_init: float, v_final: float, max_steps: int, delay_steps: int = 0, delay_mult: float = 1.0) -> Callable[[int], float]:
    """Log-linear interpolation from v_init (step <= 0) to v_final (step >= max_steps); with delay_steps > 0 the first steps are damped
    by delay_mult + (1 - delay_mult) sin(pi/2 * step / delay_steps)   (scheduler.py:5-23)."""
    log0, log1 = math.log(v_init), math.log(v_final)

    def at(step: int) -> float:
        if step <= 0:
            return v_init
        if step >= max_steps:
            return v_final
        t = min(max(step / max_steps, 0.0), 1.0)
        value = math.exp(log0 * (1.0 - t) + log1 * t)
        if delay_steps > 0:
            ramp = min(max(step / delay_steps, 0.0), 1.0)
            value *= delay_mult + (1.0 - delay_mult) * math.sin(0.5 * math.pi * ramp)
        return value

    return at


def step_scheduler(v_list: Sequence[float], step_list: Sequence[int]) -> Callable[[int], float]:
    """v_list[i] while step < step_list[i], v_list[-1] afterwards   (scheduler.py:26-35)."""
    if len(v_list) not in (len(step_list), len(step_list) + 1):
        raise AssertionError("v_list must have as many entries as step_list, or one more")
    values, edges = list(v_list), list(step_list)

    def at(step: int) -> float:
        for value, edge in zip(values, edges):
            if step < edge:
                return value
        return values[-1]

    return at


def exponential_step_scheduler(v_init: float, v_final: float, max_steps: int, n_stage: int, delay_steps: int = 0,
                               delay_mult: float = 1.0) -> Callable[[int], float]:
    """The exponential schedule sampled at n_stage + 1 equidistant steps and held constant in between   (scheduler.py:38-45)."""
    smooth = exponential_scheduler(v_init, v_final, max_steps, delay_steps, delay_mult)
    edges: List[int] = [int(max_steps * i / n_stage) for i in range(n_stage + 1)]
    return step_scheduler([smooth(e) for e in edges], edges)


def gamma_at(iteration: int, current: float, start_iter: int, end_iter: int, gamma_scheduler: Callable[[int], float]) -> float:
    """VanillaTSModel._set_gamma (:548-553): inside (start_iter, end_iter] the scheduler's value at iteration - start_iter, else unchanged."""
    if start_iter < iteration <= end_iter:
        return gamma_scheduler(iteration - start_iter)
    return current


def sh_degree_at(iteration: int, one_up_iters: Sequence[int], max_sh_degree: int) -> int:
    """VanillaTSModel._set_sh_degree (:555-565): one degree for every threshold the iteration has passed, capped."""
    return min(sum(1 for it in one_up_iters if iteration > it), max_sh_degree)

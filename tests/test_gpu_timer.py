"""rtk_timer_* (include/rtucker_hip.h): the duration of one score-kernel launch from the kernel's own begin / end events
(hipExtLaunchKernelGGL) -- what bench.py reports as roofline.kernel_ms."""
import ctypes as C

import pytest
import torch

import gen

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,rank,B,dt", [(40943, (10, 200, 200), 512, torch.float32),      # column-group kernel
                                          (3000, (10, 200, 200), 96, torch.float32),         # wave-specialised kernel
                                          (5000, (16, 64, 64), 128, torch.bfloat16)])        # bf16 kernel
def test_timed_launch_gives_the_same_scores_and_a_duration_inside_the_stream_bracket(N, rank, B, dt):
    import r_tucker_amd as rt
    L = rt._lib
    lib = L.load()
    core, R, S, O = [torch.from_numpy(x).cuda().to(dt) for x in gen.make_params(N, 11, rank, 4)]
    h, r = [torch.from_numpy(x).cuda() for x in gen.make_queries(N, 11, B, 4)]
    plain = rt.score_1vN(core, R, S, O, h, r)
    timer = C.c_void_p()
    L.check(lib.rtk_timer_create(C.byref(timer)), "rtk_timer_create")
    ms = C.c_float()
    assert lib.rtk_timer_elapsed_ms(timer, C.byref(ms)) != 0            # nothing was launched on it yet: an error, not a number
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):                                                   # (warm: the first launch of a kernel loads its code)
        e0.record()
        L.check(lib.rtk_timer_arm(timer), "rtk_timer_arm")
        timed = rt.score_1vN(core, R, S, O, h, r)
        e1.record()
        L.check(lib.rtk_timer_elapsed_ms(timer, C.byref(ms)), "rtk_timer_elapsed_ms")
    torch.cuda.synchronize()
    assert torch.equal(plain, timed)
    assert 0.0 < ms.value <= e0.elapsed_time(e1) + 1e-3                  # the kernel alone, inside stage 1 + kernel + event records
    again = rt.score_1vN(core, R, S, O, h, r)                            # arming is consumed by one launch
    assert torch.equal(plain, again)
    L.check(lib.rtk_timer_destroy(timer), "rtk_timer_destroy")

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


def test_bench_line_in_both_launch_modes():
    """python bench.py: the K timed steps as one HIP graph (the default for short single-GPU runs) and eager -- one JSON
    line each, the same metric, the kernel timer's figure inside the stream bracket's"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = {}
    for mode in ("graph", "eager"):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "12", "--warmup", "3", "--no-cpu-baseline",
                              "--prewarm-ms", "200", "--launch", mode], capture_output=True, text=True, timeout=600, cwd=root)
        assert out.returncode == 0, out.stderr[-2000:]
        rows = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(rows) == 1
        lines[mode] = json.loads(rows[0])
    g, e = lines["graph"], lines["eager"]
    assert g["metric"] == e["metric"] and g["steps"] == e["steps"] == 12
    assert "HIP graph" in g["config"]["launch"] and "eager" in e["config"]["launch"]
    assert "eager_ms_per_step" in g and "eager_ms_per_step" not in e
    for d in (g, e):
        r = d["roofline"]
        assert 0 < r["kernel_ms"] <= r["stream_bracket_ms"] * 1.05 and r["kernel_ms_source"].startswith("kernel begin/end")
        assert abs(d["value"] - 12 * 512 / (d["ms_per_step"] * 12e-3)) < 1e-6 * d["value"]
    assert g["ms_per_step"] < 1.5 * e["ms_per_step"] and e["ms_per_step"] < 1.5 * g["ms_per_step"]

"""Wall-clock bounds, kept OUT of the parity gate: collected only by `pytest -m perf` on an MI355X
(tests/conftest.py deselects them under every other selection).  The numbers of record are the driver's
BENCH_r*.json and the rocprofv3 summaries under profiles/; these bounds are tripwires an order of magnitude out of
the way of clock-speed differences between boxes (C2: 6.9 us on one box, 8.6 on the driver's)."""
import pytest

from test_gpu_bench import run_bench


def _no_gpu():
    try:
        import torch

        return not torch.cuda.is_available()
    except Exception:
        return True


# both markers: `-m perf` on a box without a GPU skips instead of failing; conftest keeps them out of `-m gpu`
pytestmark = [pytest.mark.perf, pytest.mark.gpu, pytest.mark.skipif(_no_gpu(), reason="needs an MI355X")]


def test_c2_pass_takes_microseconds_not_tens_of_them():
    """25 us per pass of two launches in round 2; 6.2-8.6 us measured since round 3 (one launch)."""
    out, _ = run_bench("--workload", "C2", "--steps", "400", "--warmup", "20", "--no-cpu-baseline", "--no-secondary", "--no-live-traffic")
    assert out["ms_per_step"] <= 0.020, out["ms_per_step"]
    assert out["ms_per_step"] <= out["roofline"]["kernel_ms_between_events"] + 0.005


def test_graded_planes_cost_less_than_two_sweeps_and_exact_unweighted_half_of_the_weighted_walk():
    out, _ = run_bench("--steps", "3", "--warmup", "1", "--secondary-steps", "2", "--no-cpu-baseline", "--no-live-traffic")
    sec = out["secondary"]
    assert sec[1]["ms_per_step"] < sec[5]["ms_per_step"] < 2.0 * sec[1]["ms_per_step"]   # (two sweeps cost 1.85 x)
    assert sec[6]["ms_per_step"] < 0.7 * sec[0]["ms_per_step"]       # pair_exact_unw_kernel 10 ms, pair_exact64_skip_kernel 21


def test_figures_of_record_have_not_gone_backwards():
    """Tripwires a fifth out of the way of the figures of record (profiles/r05_bench_default.json): the headline's
    fraction of the full 2*B vector-ALU count (1.65 with three quarters of the rows reduced by pair_low_kernel; 0.87
    unsplit) and its matrix part on its own rows (0.87), weighted EXACT64 at C3 (20.8-21.5 ms; 30.1 before
    pair_exact64_skip_kernel), the exact unweighted kernel (9.9-10.0 ms; 30 before it existed), C4 and C5 on one GPU
    (1.9 / 3.0 of the full count)."""
    out, _ = run_bench("--steps", "20", "--warmup", "3", "--secondary-steps", "3", "--no-cpu-baseline", "--no-live-traffic")
    sec = out["secondary"]
    assert out["roofline"]["frac"] >= 1.3, out["roofline"]["frac"]
    assert out["roofline"]["parts"][0]["frac"] >= 0.70, out["roofline"]["parts"]
    assert out["reference_width"]["ms_per_step"] <= 25.0, out["reference_width"]["ms_per_step"]
    assert sec[6]["ms_per_step"] <= 12.0, sec[6]["ms_per_step"]
    assert sec[3]["roofline"]["frac"] >= 1.5 and sec[4]["roofline"]["frac"] >= 2.4


def test_live_counter_traffic_is_measured_on_this_box_and_agrees_with_the_committed_figure():
    """The rocprof-reported rate: the primary kernel's counter bytes per launch, measured on THIS box by two
    child runs under rocprofv3 --pmc once the timings are done (bench.py live_traffic), over this run's kernel time;
    the committed figure of profiles/traffic.json beside it -- same build, same schedule: the same traffic within the
    counters' noise.  Here and not under `-m gpu`: a box whose profiler refuses a counter must not turn the parity
    record red (the line then carries `traffic_note` and the committed figure: test_gpu_bench.py)."""
    out, _ = run_bench("--steps", "3", "--warmup", "1", "--no-secondary", "--no-cpu-baseline", "--no-end-to-end")
    rl, hbm = out["roofline"], out["roofline"]["hbm"]
    assert rl["traffic_source"].startswith("live on this box"), rl.get("traffic_note")
    assert rl["fetch_size_kib"] > 0 and rl["write_size_kib"] > 0
    assert abs(rl["traffic"] - (2 * rl["fetch_size_kib"] + rl["write_size_kib"]) * 1024) < 1.0
    assert 0.7 < rl["traffic"] / rl["traffic_committed"] < 1.4
    assert hbm["measured_GBps"] > hbm["achieved"] and 0 < hbm["measured_frac_of_8000"] < hbm["measured_frac_of_6290"] < 1


def test_frcfrc_on_c4_takes_a_second_not_six():
    """6.5 s with the reference's default flags until round 5 (the formatter on one host thread); 0.76 s measured with
    the formatter on the device and the host threads defaulting to the CPU quota."""
    out, _ = run_bench("--steps", "3", "--warmup", "1", "--no-secondary", "--no-cpu-baseline", "--no-live-traffic")
    c4 = [e for e in out["end_to_end"]["frcfrc"] if e["workload"] == "C4"][0]
    default = [r for r in c4["runs"] if r["flags"] == "(default)"][0]
    assert default["wall_s"] <= 2.0, default

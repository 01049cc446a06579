"""bench.py's per-kind table (kernel_table, SummedStats) on made-up counters: the arithmetic of `achieved`, `frac`, the sum over
kinds and the executed estimate, and the keys the bench line promises.  No GPU and no library needed."""
import os
import sys
from types import SimpleNamespace

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def _stats(scale=1):
    return SimpleNamespace(
        closest_tests=[4.0e12 * scale, 0, 2.0e10 * scale], shadow_tests=[8.0e12 * scale, 0, 1.0e10 * scale],
        closest_kind_ms=[2000.0 * scale, 0.0, 20.0 * scale], shadow_kind_ms=[4000.0 * scale, 0.0, 40.0 * scale],
        closest_kind_launches=[2 * scale, 0, 2 * scale], shadow_kind_launches=[2 * scale, 0, 2 * scale],
        brute_wave_steps=[[6.0e10 * scale, 5.0e10 * scale, 1.0e9 * scale, 1.0e6 * scale], [1.2e11 * scale, 6.0e10 * scale, 1.0e9 * scale, 1.0e6 * scale]],
        closest_rays=4_000_000 * scale, shadow_rays=32_000_000 * scale, device_ms=6100.0 * scale)


def _scene(n_point=8, directional=0):
    return SimpleNamespace(scene_desc=SimpleNamespace(n_point_lights=n_point, has_directional_light=directional))


def test_kernel_table_arithmetic():
    rows, total = bench.kernel_table(_stats(), _scene(), False)
    assert set(rows) == {"closest_triangle", "closest_ellipsoid", "shadow_triangle", "shadow_ellipsoid"}     # no sphere launches, no sphere rows
    tri = rows["closest_triangle"]
    assert tri["flop_per_pair"] == 54 and tri["launches"] == 2 and tri["avg_launch_ms"] == 1000.0
    assert tri["achieved"] == pytest.approx(4.0e12 * 54 / 2.0 / 1e12)            # TFLOP/s
    assert tri["frac"] == pytest.approx(tri["achieved"] / 157.3)
    assert tri["flop_per_launch"] == pytest.approx(4.0e12 * 54 / 2)
    assert tri["pairs_per_s"] == pytest.approx(2.0e12)
    assert rows["closest_ellipsoid"]["flop_per_pair"] == 95 and rows["shadow_ellipsoid"]["flop_per_pair"] == 71
    # the steps are counted in units of 64 pairs; front 13, first pre-test 17, second 8, test as written 68 FLOP per lane
    executed = 64 * (6.0e10 * 13 + 5.0e10 * 17 + 1.0e9 * 8 + 1.0e6 * 68)
    assert tri["executed_flop_upper_estimate"] == pytest.approx(executed)
    assert tri["executed_frac"] == pytest.approx(executed / 2.0 / 1e12 / 157.3)
    assert tri["wave_steps"] == {"all": 6.0e10, "past_the_sign_test": 5.0e10, "second_pre_test": 1.0e9, "test_as_written": 1.0e6}
    flop = 4.0e12 * 54 + 2.0e10 * 95 + 8.0e12 * 54 + 1.0e10 * 71
    assert total["flop"] == pytest.approx(flop) and total["kernel_ms"] == pytest.approx(6060.0)
    assert total["frac"] == pytest.approx(flop / 6.06 / 1e12 / 157.3)
    assert total["pairs"] == pytest.approx(4.0e12 + 2.0e10 + 8.0e12 + 1.0e10)


def test_directional_light_weighs_on_the_ellipsoid_shadow_flop():
    rows, _ = bench.kernel_table(_stats(), _scene(n_point=1, directional=1), False)
    assert rows["shadow_ellipsoid"]["flop_per_pair"] == pytest.approx((71 + 62) / 2)


def test_as_written_rows_carry_no_executed_estimate():
    rows, _ = bench.kernel_table(_stats(), _scene(), True)
    assert rows["closest_triangle"]["kernel"] == "k_closest_tri_brute<1,0>"
    assert "executed_frac" not in rows["closest_triangle"] and "wave_steps" not in rows["shadow_triangle"]


def test_summed_stats_add_up_and_keep_the_fractions():
    one, three = _stats(), bench.SummedStats([_stats(), _stats(2)])
    assert three.closest_tests[0] == pytest.approx(3 * one.closest_tests[0]) and three.device_ms == pytest.approx(3 * one.device_ms)
    assert three.brute_wave_steps[1][1] == pytest.approx(3 * one.brute_wave_steps[1][1])
    rows1, total1 = bench.kernel_table(one, _scene(), False)
    rows3, total3 = bench.kernel_table(three, _scene(), False)
    assert total3["frac"] == pytest.approx(total1["frac"]) and rows3["shadow_triangle"]["executed_frac"] == pytest.approx(rows1["shadow_triangle"]["executed_frac"])


def test_whole_frame_roofline_is_headed_by_the_longest_closest_scan():
    args = SimpleNamespace(plain_test=False)
    out, summary = bench.whole_frame_roofline([_stats()], _scene(), args, "made-up@8x8", 2, 6.5)
    assert out["kernel"] == "k_closest_tri_brute<2,2>" and out["bound"] == "valu_fp32" and out["peak"] == 157.3 and out["unit"] == "TFLOP/s"
    assert out["dominant_by_time"] == "shadow_triangle" and out["traffic"] is None
    assert summary["mrays_s"] == pytest.approx(36.0e6 / 6.5 / 1e6) and summary["pairs"] == pytest.approx(out["all_kinds"]["pairs"])
    assert summary["tests_per_s"] == pytest.approx(out["all_kinds"]["pairs"] / 6.5)

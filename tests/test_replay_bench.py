"""The replay-mode bench program (tools/replay/replay_bench.cpp: FrontEnd / LoopHandler reduced to the calls that reach the hot
path, driven from the C++ adaptors): its CPU leg -- the oracle's restatement of the reference plus the product's HOST functions
for the loop descriptors -- runs without a GPU and must follow the synthetic ground truth; on a GPU the device leg must give the
same trajectory and the same loop candidates."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(m)
    finally:
        sys.argv = argv
    return m


def _run(tmp_path, which, n_frames=44, extra=()):
    m = _bench()
    exe = m.build_replay_bench()
    pack = tmp_path / "tiny.bin"
    m.write_replay_pack(str(pack), "tiny", n_frames, 4, 600)
    out = subprocess.run([exe, str(pack), str(tmp_path / "run"), which, *extra], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1]), tmp_path


def test_cpu_leg_of_the_replay_bench_follows_the_ground_truth(built, tmp_path):
    d, td = _run(tmp_path, "cpu")
    c = d["cpu"]
    assert c["frames"] == 44 and c["keyframes"] == 11 and c["frames_lost"] == 0
    assert c["hypothesis_tries"] == 43  # the constant-motion try is accepted at once on this smooth path
    assert c["ate_vs_ground_truth_m"] < 5e-3
    for stage in ("per_frame", "trackNewCoarse", "setCoarseTrackingRef", "scale_opt", "pts_generation", "sc_generation", "search_ringkey", "search_sc"):
        assert c["stages_mean_ms"][stage]["calls"] > 0, stage
    assert c["loop_queries"] == 4 and c["queries_with_candidates"] == 4  # the first pass's twins are in the index
    lines = open(td / "run_dslam_cpu.txt").read().splitlines()  # the reference's dslam.txt surface (LoopHandler.cpp:59-80)
    assert len(lines) == 44 and lines[3].split()[0] == "3" and len(lines[3].split()) == 4


@pytest.mark.gpu
def test_gpu_leg_matches_the_cpu_leg(ctx, tmp_path):
    d, _ = _run(tmp_path, "both")
    g, c, x = d["gpu"], d["cpu"], d["gpu_vs_cpu"]
    assert g["frames_lost"] == 0 and g["hypothesis_tries"] == c["hypothesis_tries"]
    assert x["max_abs_trajectory_diff_m"] < 2e-4 and abs(x["ate_ratio_gpu_over_cpu"] - 1) < 0.01
    # end-to-end search parity on EQUAL inputs, no allowance: the device loop detector fed with the clouds the CPU path recorded returns
    # the CPU path's ring keys, candidates and search_sc matches bit for bit (search_place.h:25-84, ScanContext.cpp:96-141)
    e = d["device_search_on_the_cpu_paths_clouds"]
    assert e["queries"] == x["loop_queries"] > 0
    assert e["ring_keys_bit_equal"] == e["identical_candidates"] == e["identical_search_sc_match"] == e["queries"]
    # the full replay describes every place from its OWN trajectory (micrometres apart): a query whose candidates differ is listed with
    # the ring-key entries and the points that changed their polar bin
    diffs = d["candidate_differences_of_the_full_replay"]
    assert x["queries_with_identical_candidates"] + len(diffs) >= x["loop_queries"]
    for q in diffs:
        assert q["ringkey_entries_that_differ"] or q["points"] or "note" in q


@pytest.mark.gpu
@pytest.mark.parametrize("pipelined", ["1", "0"])
def test_concurrent_sequences_through_one_stream_equal_the_one_sequence_run(ctx, tmp_path, pipelined):
    """S sequences submit into ONE dsm_host::Stream from C++ (frames' first hypotheses, keyframes' scale guesses): scheduling only --
    every sequence's trajectory and scales equal the one-sequence run's bit for bit"""
    d, _ = _run(tmp_path, "gpu", extra=("6", pipelined))
    c = d["concurrent"]
    assert c["sequences"] == 6 and c["frames"] == 6 * 44 and c["frames_lost"] == 0
    assert c["max_abs_trajectory_diff_vs_the_one_sequence_run_m"] == 0.0 and c["scales_equal_the_one_sequence_run"] is True
    assert c["frames_per_s"] > 0 and c["advances"] > 0

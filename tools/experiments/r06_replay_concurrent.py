"""same-box A/B of the replay's concurrent leg (S sequences through ONE dsm_host::Stream from C++):
python tools/experiments/r06_replay_concurrent.py S 'geometry,chain,ticks,pipelined[,coarse]' ..."""
import json, os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

exe = bench.build_replay_bench()
S = int(sys.argv[1])
combos = [tuple(int(x) for x in a.split(",")) for a in sys.argv[2:]]
with tempfile.TemporaryDirectory() as td:
    pack = os.path.join(td, "kitti00.bin")
    bench.write_replay_pack(pack, "kitti00", 200, 4, 2000)
    for cb in combos:
        geom, chain, ticks, pipe = cb[:4]
        env = dict(os.environ, DSM_REPLAY_GEOMETRY=str(geom), DSM_REPLAY_CONC_GEOMETRY=str(geom), DSM_REPLAY_CHAIN=str(chain))  # (both legs under the table named)
        if ticks:
            env["DSM_REPLAY_TICKS"] = str(ticks)
        if len(cb) > 4:
            env["DSM_REPLAY_COARSE"] = str(cb[4])
        p = subprocess.run([exe, pack, os.path.join(td, "o"), "gpu", str(S), str(pipe)], capture_output=True, text=True, timeout=900, env=env)
        if p.returncode != 0:
            print("S", S, cb, "FAILED", (p.stderr or p.stdout)[-400:])
            continue
        d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        c = d["concurrent"]
        g = d["gpu"]["stages_mean_ms"]
        print("S", S, "geometry,chain,ticks,pipelined[,coarse]", cb, "frames/s", round(c["frames_per_s"]), "latency ms", round(c["mean_frame_latency_ms"], 3), "advances", c["advances"],
              "diff vs one sequence", c.get("max_abs_trajectory_diff_vs_the_one_sequence_run_m"), "host/adv", {k: round(v, 3) for k, v in c["host_ms_per_advance_by_call"].items()},
              "| one sequence: trackNewCoarse", round(g["trackNewCoarse"]["mean_ms"], 4), "scale_opt", round(g["scale_opt"]["mean_ms"], 4), "per frame", round(g["per_frame"]["mean_ms"], 4), flush=True)

#!/usr/bin/env python3
"""bench.py -- stereo frames/s of the direct photometric hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; prints ONE JSON line on rank 0.  For N > 1 the
driver launches it under torch.distributed.run, one rank per GPU; started WITHOUT a launcher (`python bench.py --gpus 4`)
it spawns the N ranks itself and fails loudly when fewer devices exist.

Headline workload (config.workload) = the configuration BASELINE.json's metric is quoted on: KITTI-00 input 1241x376,
SIX pyramid levels.  DSO's level rule stops at five levels for the reference's 1232x368 crop, so the six-level form pads
the input to 1248x384 (divisible by 32; SURVEY.md section 8d "S2") and runs the reference's LM rules with a six-entry
iteration table.  Dense template (every interior pixel is a point), seeded synthetic RELIEF scenes (round 4: broadband
texture over a smooth relief; the CPU oracle tracks every frame of the family; `--scene-family plane` = rounds 1-3's scenes)
with ground-truth motion, LM iterations AS EXECUTED (TrackerAndScaler.cpp:451-638, :854-964).

A "step" = B independent stereo frames per GPU, ALL DISTINCT (own ground-truth motion and image noise; 18 textures repeat,
none left out), SUBMITTED to the streaming form of the batched calls (dsm_stream_*, tick engine): every frame's
trackNewestCoarse from the identity pose and, for every 5th frame (keyframe cadence,
FrontEnd.cpp:806-811), the stereo scale optimiser from s = 1, followed by one advance of the stream; at most B track problems
are resident at any time, problems are admitted as slots free up and retire individually, and after the K-th step the pool
is drained INSIDE the timed region (everything submitted inside it is completed inside it).  `--stream 0` = round 3's step:
one synchronous dsm_track_and_scale_batch call.  The library runs with its default parameters.  Inputs (pyramids,
templates) are resident in HBM when the timed region starts.

Further objects of the line: the same frames on SURVEY.md 8d's fixed schedule (config.fixed_schedule_leg: 1 + 3
evaluations per level, CPU leg on the same schedule), the reference-faithful five-level workload (S1, 1232x368,
config.reference_five_level), the replay-mode per-stage figures of ONE sequence driven from the C++ adaptors, GPU and CPU
path side by side (config.replay, tools/replay/replay_bench.cpp), and the ring-key search on one GPU at SURVEY.md 8d's
sizes with its own rooflines, oracle check and CPU brute force (config.ringkey).

Tracking does not shard inside a frame: N GPUs = N independent replicas of the same work, no
data-path collective ("scaling": "weak").  What does shard is the ring-key database: with N > 1 the line also carries
config.ringkey_sharded -- the DB split `ordinal mod N`, local scans, cross-shard merge by RCCL all-reduce(min) through the
C ABI (dsm_ringdb_merge_topk), checked against the unsharded answer.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

# Scene seeds (offsets from 0x5EED0000) on which the REFERENCE ALGORITHM itself (CPU oracle) converges from the identity
# guess on every bench configuration -- selected by tools/select_scenes.py.  Offsets 3, 5 (S2), 7 (S1) and 15 (S2) drive
# the algorithm into a wrong minimum from the identity guess on both the CPU and the GPU path (the real front end starts
# from a constant-motion guess and never sees such a case); such frames make the accuracy half of the metric a test of
# last-bit rounding instead of the tracker, so they are not part of the workload.
SCENE_SEEDS = (0, 1, 2, 4, 6, 8, 9, 10, 11, 12, 13, 14, 16, 17)

CONFIGS = {
    # name: (w, h, levels, label)
    "S1": (1232, 368, 5, "KITTI-00 shape 1232x368 (1241x376 cropped, what the reference runs), 5-level pyramid"),
    "S2": (1248, 384, 6, "KITTI-00 shape 1248x384 (1241x376 padded to a multiple of 32), 6-level pyramid"),
    "S3": (1920, 1080, 6, "synthetic 1920x1080 (floor-halved levels), 6-level pyramid"),
}


def parse():
    a = _parse()
    a.streams_given = a.streams
    if a.streams is None:
        a.streams = 3 if (a.stream and not a.with_upload) else 2
    return a


def _parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=512, help="independent stereo frames in flight per GPU")
    ap.add_argument("--scenes", type=int, default=None, help="distinct synthetic frames (default: --batch, i.e. every frame in flight has its own motion and noise; textures repeat "
                                                                  "every len(SCENE_SEEDS) frames); a smaller value cycles that many frames over the batch (round 2's workload: 8)")
    ap.add_argument("--init", default="identity", choices=["identity", "constant-motion"],
                    help="initial pose guess of every track: 'identity' = SURVEY.md 8d's setting (default); 'constant-motion' = the front end's first "
                         "try (FrontEnd.cpp:147-150: last inter-frame motion), modelled as the true motion perturbed by N(0; --init-err x the "
                         "motion's sigma).  Measured on the CPU path (DESIGN.md section 6): neither the evaluations per frame nor the share of "
                         "frames that end in a wrong minimum (6-11 %%) depend on it")
    ap.add_argument("--init-err", type=float, default=0.25, help="sigma of the constant-motion guess's error as a fraction of the motion's sigma")
    ap.add_argument("--scene-family", default="relief", choices=["relief", "plane"],
                    help="'relief' (default, round 4): synth.ReliefScene -- a smooth relief over the plane under a broadband texture (32 sinusoids, 6-480 px); the "
                         "CPU oracle tracks every frame of this family from the identity guess (tools/scene_failure_rate.py).  'plane': rounds 1-3's single "
                         "textured plane (eight sinusoids, 8-128 px), on which the reference algorithm itself ends in a wrong minimum on 6-11 %% of the frames")
    ap.add_argument("--textures", default="all", choices=["all", "converging"],
                    help="'all' (default): textures 0 .. 17, nothing left out; 'converging': round 2's hand-picked list (SCENE_SEEDS: the "
                         "textures on which the reference algorithm converges from the identity guess for THEIR first motion)")
    ap.add_argument("--fixed-schedule", type=int, default=0, help="K > 0: SURVEY.md 8d's fixed schedule -- exactly 1 + K evaluations per level, every step taken, on "
                                                                    "the GPU and the CPU leg alike (input-independent bytes per frame); the default line reports it as a second object")
    ap.add_argument("--no-fixed-leg", action="store_true", help="skip the fixed-schedule (K = 3) leg of the default line")
    ap.add_argument("--config", default="S2", choices=list(CONFIGS), help="S2 = 1248x384x6 (the metric's own configuration, default), S1 = 1232x368x5 (reference-faithful), S3 = 1920x1080x6 (BASELINE configs[3] shape)")
    ap.add_argument("--template", default="dense", choices=["dense", "sparse"])
    ap.add_argument("--kf-every", type=int, default=5)
    ap.add_argument("--streams", type=int, default=None, help="stream groups the resident problems are split over (one group's LM launches run under another's evaluations); "
                                                             "default: 3 for the streaming form, 2 for the batch form (each measured best, DESIGN.md section 4.3)")
    ap.add_argument("--no-adaptive", action="store_true", help="worst-case launch schedule, never poll")
    ap.add_argument("--with-upload", action="store_true", help="secondary figure: every step also hands the B new left images (and the right images of the keyframes) over as HOST buffers (PCIe + device pyramid build inside the timed region); never the headline value")
    ap.add_argument("--u8", action="store_true", help="with --with-upload: camera bytes (mono8) are handed over instead of float images; the synthetic images are rounded to 0..255 for the whole run")
    ap.add_argument("--overlap", action="store_true", help="with --with-upload: double-buffered frame slots -- the images of the next step are handed over asynchronously (dsm_upload_images_async into DSM_SLOT_NEXT_*) while this step is tracked")
    ap.add_argument("--single-uploads", action="store_true", help="with --with-upload: one dsm_tracker_upload_image call per image instead of one dsm_upload_images call per step")
    ap.add_argument("--pinned", action="store_true", help="with --with-upload: the host images live in pinned memory (dsm_host_alloc)")
    ap.add_argument("--queue", type=int, default=None,
                    help="dsm_params.work_queue (default: the library's, dsm_params_default): 0 launch-per-step form; 1 the library's automatic rule "
                         "(single calls of 32 ... ~200 dense frames run as one persistent launch; the one-call-per-step form used here is "
                         "always the launch form); 2 the whole call as one launch of persistent workgroups (two calls per step; its roofline is the whole-call figure)")
    ap.add_argument("--stream", type=int, default=1, choices=[0, 1],
                    help="1: the streaming form of the step (dsm_stream_*): every step SUBMITS its B frames (+ keyframe scale problems) to a pool of B resident "
                         "problems and runs one pass; problems are admitted as slots free up, carried over when they need more rounds than most, and retire "
                         "individually; after the K-th step the pool is drained inside the timed region.  0: one synchronous dsm_track_and_scale_batch call per step")
    ap.add_argument("--stream-engine", type=int, default=1, choices=[0, 1], help="with --stream: 0 passes with carried stragglers, 1 ticks (one LM round per resident problem and tick, admission and retirement on the device)")
    ap.add_argument("--stream-ticks", type=int, default=0, help="with --stream-engine 1: ticks per advance (0: the library's default)")
    ap.add_argument("--chain", type=int, default=-1, help="tick engine: LM rounds a one-chunk evaluation's workgroup may run inside one tick (dsm_stream_set_chain; 0 off, -1 the library's default)")
    ap.add_argument("--stream-quantile", default=None, help="with --stream: rounds per level of a pass = this quantile of what retired problems needed (library default 0.75)")
    ap.add_argument("--stream-rounds", default=None, help="with --stream: fixed rounds per level of a pass, comma separated from level 0 (e.g. 5,6,8,12,16,16)")
    ap.add_argument("--separate-calls", action="store_true", help="dsm_track_batch then dsm_optimize_scale_batch instead of the one dsm_track_and_scale_batch call per step")
    ap.add_argument("--speculate", type=int, default=None, help="dsm_params.speculate (default: library default)")
    ap.add_argument("--compact", type=int, default=None, help="dsm_params.compact_tail (default: library default)")
    ap.add_argument("--fuse", type=int, default=None, help="dsm_params.fuse_lm (default: library default)")
    ap.add_argument("--geometry", type=int, default=None, choices=(0, 1, 2),
                    help="dsm_params.chunk_geometry: 0 throughput table (library default), 1 latency table, 2 latency table above 4096 points and one chunk below "
                         "(one frame in flight -- --batch 1 --coarse -1, the replay adaptors)")
    ap.add_argument("--coarse", type=int, default=None, help="dsm_params.persistent_coarse: N > 0 the LDS-resident small-level kernel up to N pixels, -1 the one-chunk levels as a chain (one launch), 0 off (library default)")
    ap.add_argument("--cpu-frames", type=int, default=512, help="upper bound of the frames timed on the CPU baseline (rank 0, N=1); the leg stops after --cpu-seconds")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="time budget of the single-core CPU baseline leg once --cpu-min-frames are done")
    ap.add_argument("--cpu-min-frames", type=int, default=448, help="distinct frames the single-core CPU leg covers at least (the ATE half of the metric is taken over them)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-second-leg", action="store_true", help="skip the short reference-faithful five-level (S1) leg reported in config.reference_five_level")
    ap.add_argument("--second-leg-steps", type=int, default=10, help="timed steps of the fixed-schedule and five-level legs (streamed: ramp-up and drain are inside, so a handful of steps understates the rate)")
    ap.add_argument("--quick", action="store_true", help="the headline leg alone: --no-cpu --no-fixed-leg --no-second-leg --no-plane-leg --no-upload-leg --no-replay-leg --no-ringkey-leg (A/B runs, profiles)")
    ap.add_argument("--upload-stream-ticks", type=int, default=0, help="ticks per advance of the streamed hand-over leg (0: the library's own choice)")
    ap.add_argument("--no-upload-leg", action="store_true", help="skip the PCIe-inclusive leg of the default line (config.with_upload)")
    ap.add_argument("--no-plane-leg", action="store_true", help="skip the short leg on SURVEY.md 8d's literal scene family (config.plane_family)")
    ap.add_argument("--no-cpu-all-cores", action="store_true", help="skip the all-core leg of the CPU baseline (one share of frames per physical core, forked workers)")
    ap.add_argument("--evals-only", action="store_true",
                    help="diagnostic: max_iterations=0, i.e. exactly one fused evaluation per level and problem (clean per-kernel roofline)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend of the bench plumbing (barrier, max over ranks); nccl = RCCL")
    ap.add_argument("--device-override", type=int, default=-1, help="testing: put every rank on this device")
    ap.add_argument("--membw", action="store_true", help="print the measured read-only streaming bandwidths (two access patterns) and exit")
    ap.add_argument("--replay", action="store_true", help="run the replay-mode leg alone (config.replay of the default line): one sequence, one frame in flight, C++ adaptors, per-stage ms for the GPU and the CPU path; "
                                                             "then --replay-concurrent sequences through one dsm_host::Stream")
    ap.add_argument("--no-replay-leg", action="store_true", help="skip config.replay in the default line")
    ap.add_argument("--replay-concurrent", type=int, default=128, help="config.replay.<first shape>.concurrent: this many sequences through ONE dsm_host::Stream from C++ (0: skip)")
    ap.add_argument("--replay-frames", type=int, default=200)
    ap.add_argument("--replay-kf-every", type=int, default=4)
    ap.add_argument("--replay-active", type=int, default=2000, help="active points per keyframe (the semi-dense template grows to ~10^4 points by dilation)")
    ap.add_argument("--ringkey", action="store_true", help="benchmark the sharded ring-key search alone instead")
    ap.add_argument("--no-ringkey-leg", action="store_true", help="skip the ring-key leg of the default line (N = 1: config.ringkey; N > 1: config.ringkey_sharded)")
    ap.add_argument("--ringkey-leg-seconds", type=float, default=120.0, help="N > 1: deadline of the sharded ring-key leg; a rank that misses it reports an error in config.ringkey_sharded and the line is printed regardless")
    ap.add_argument("--rk-n", type=int, default=1_000_000)
    ap.add_argument("--rk-q", type=int, default=1024)
    ap.add_argument("--detail-out", default="gpurun_out/bench_detail.json",
                    help="side file (relative to the repository root unless absolute) that receives the FULL result object -- every leg, table and note; the ONE "
                         "stdout line is its compact form (< 4 KB: contract keys, roofline and cpu_baseline as numbers, one summary per leg)")
    a = ap.parse_args()
    if a.quick:
        a.no_cpu = a.no_fixed_leg = a.no_second_leg = a.no_plane_leg = a.no_upload_leg = a.no_replay_leg = a.no_ringkey_leg = True
    return a


_BACKEND = "nccl"


def baseline_metric():
    """the headline metric exactly as BASELINE.json spells it"""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "stereo frames/sec @ 1241×376, 6-level pyramid, 1 MI355X; ATE vs CPU ref"


# ------------------------------------------------------------------------------------------------------------------
# process plumbing
# ------------------------------------------------------------------------------------------------------------------
def spawn_ranks_if_needed(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (torch.distributed.run, one per GPU).
    Returns the exit code of the launcher, or None when this process is itself a rank (or N == 1)."""
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is not None:
        if int(world_env) != args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}: launch one rank per GPU "
                             f"(torch.distributed.run --nproc-per-node {args.gpus}) or drop the launcher\n")
            sys.exit(2)
        return None
    if args.gpus <= 1:
        return None
    import torch

    have = torch.cuda.device_count()
    if have < args.gpus and args.device_override < 0:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} requested but only {have} device(s) are visible\n")
        sys.exit(2)
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dist_setup(args):
    import torch

    global _BACKEND
    _BACKEND = args.dist_backend
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.device_override >= 0:
        local = args.device_override
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if _BACKEND == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(_BACKEND)
    return rank, local, world


def barrier_sync(world):
    import torch

    if world > 1:
        import torch.distributed as dist

        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world):
    import torch

    if world == 1:
        return x
    import torch.distributed as dist

    t = torch.tensor([x], dtype=torch.float64, device="cuda" if _BACKEND == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ------------------------------------------------------------------------------------------------------------------
# workload
# ------------------------------------------------------------------------------------------------------------------
def config_geometry(name):
    from direct_stereo_slam_amd import synth as S

    w, h, nl, _ = CONFIGS[name]
    if name == "S1":
        K = S.kitti_K_work()
    elif name == "S2":
        fx, fy, cx, cy = S.KITTI_K_RAW
        K = (fx, fy, cx + (1248 - 1241) / 2.0, cy + (384 - 376) / 2.0)
    else:  # S3: same field of view as KITTI
        fx = S.KITTI_K_RAW[0] * 1920.0 / 1241.0
        K = (fx, fx, 959.5, 539.5)
    return w, h, nl, K


_FRAMES = {}
_FRAME_OFFSET = 0  # first frame of this workload among the distinct frames (several host-side contexts share one frame list)


def frame_seeds(args):
    """texture seeds of the workload: all of 0 .. 17 (nothing is left out: on these plane scenes the reference algorithm itself
    ends in a wrong minimum for 6-11 % of the frames, whatever the starting point, on the CPU path exactly as on the GPU
    path -- such frames are part of the workload and are reported), or round 2's hand-picked list"""
    return tuple(range(18)) if args.textures == "all" else SCENE_SEEDS


def build_frames(args, w, h, K, T):
    """The distinct synthetic frames of the workload.  Frame f = texture f mod n_tex seen under its OWN ground-truth motion
    and image noise (seeded by f), with its own initial guess; the keyframe image, the template and the right image belong
    to the texture and are shared by its frames.  Returns (textures, frames): textures[k] = (scene, ref, right),
    frames[f] = (k, new image, gt pose, guess pose)."""
    from concurrent.futures import ThreadPoolExecutor

    from direct_stereo_slam_amd import synth as S

    seeds = frame_seeds(args)
    n_frames = args.scenes if args.scenes else args.batch
    n_tex = min(len(seeds), n_frames)
    key = (w, h, tuple(K), n_frames, args.init, args.init_err, args.textures, bool(args.u8), args.scene_family)
    Scene = S.ReliefScene if args.scene_family == "relief" else S.PlaneScene
    if key in _FRAMES:  # (the fixed-schedule leg runs on the frames of the headline leg)
        return _FRAMES[key]
    u8 = (lambda im: np.clip(np.rint(im), 0, 255).astype(np.float32)) if args.u8 else (lambda im: im)
    textures, first = [], []
    for k in range(n_tex):
        seed = 0x5EED0000 + seeds[k]
        scene = Scene(seed=seed)
        rng = np.random.default_rng(seed)
        ref = scene.render(K, w, h, noise=2.0, rng=rng)
        R, t = S.random_motion(rng)
        new = scene.render(K, w, h, R, t, a=0.02, b=3.0, noise=2.0, rng=rng)
        right = scene.render(K, w, h, T[:3, :3], T[:3, 3], noise=2.0, rng=rng)
        textures.append((scene, u8(ref), u8(right)))
        first.append((u8(new), R, t))

    def make(f):
        k = f % n_tex
        if f < n_tex:  # the first cycle: exactly round 2's frames (same seeds, same draws)
            new, R, t = first[f]
        else:
            rng = np.random.default_rng(0x5EED0000 + 0x100000 * (f // n_tex) + seeds[k])
            R, t = S.random_motion(rng)
            new = u8(textures[k][0].render(K, w, h, R, t, a=0.02, b=3.0, noise=2.0, rng=rng))
        gt = S.pose_from_Rt(R, t)
        if args.init == "identity":
            guess = np.array(S.IDENTITY_POSE, np.float64)
        else:
            # constant-motion prediction (FrontEnd.cpp:147-150): the previous inter-frame motion = this frame's motion up to an
            # "acceleration" drawn from init_err x the motion's own distribution
            e = np.random.default_rng(0xACCE1000 + f)
            Re, te = S.random_motion(e, sigma_t=np.array((0.05, 0.05, 0.2)) * args.init_err, sigma_r=0.005 * args.init_err)
            guess = S.pose_from_Rt(Re @ R, Re @ t + te)
        return (k, new, gt, guess)

    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as ex:
        frames = list(ex.map(make, range(n_frames)))
    _FRAMES.clear()  # one configuration's images at a time
    _FRAMES[key] = (textures, frames)
    return textures, frames


def build_workload(args, ctx, config):
    """B trackers on this GPU; pyramids are built on the device from the raw float image.  Every rank builds the SAME
    frames (replicas: identical work per GPU)."""
    from direct_stereo_slam_amd import synth as S
    from direct_stereo_slam_amd.tracker import TrackerAndScaler, default_params

    w, h, nl, K = config_geometry(config)
    T = S.KITTI_T_STEREO
    params = default_params()
    params.adaptive_schedule = 0 if args.no_adaptive else 1
    if args.coarse is not None:
        params.persistent_coarse = args.coarse
    if args.queue is not None:
        params.work_queue = args.queue
    if args.fixed_schedule > 0:
        params.fixed_schedule = args.fixed_schedule
    if args.fuse is not None:
        params.fuse_lm = args.fuse
    if args.speculate is not None:
        params.speculate = args.speculate
    if args.compact is not None:
        params.compact_tail = args.compact
    if args.geometry is not None:
        params.chunk_geometry = args.geometry
    if args.evals_only:
        for l in range(6):
            params.max_iterations[l] = 0
    textures, frames = build_frames(args, w, h, K, T)
    trackers, gts, guesses, host, images = [], [], [], [], []
    tpl_cache = {}
    for b in range(args.batch):
        k, new, gt, guess = frames[(b + _FRAME_OFFSET) % len(frames)]
        scene, ref, right = textures[k]
        trk = TrackerAndScaler(ctx, w, h, nl, T, K, params)
        trk.makeK(*K)
        if args.template == "dense" and k in tpl_cache:
            tpl = tpl_cache[k]
        else:
            trk.upload_image(0, ref, 1.0)  # device makeImages of the keyframe, read back for the template colours
            ref_p = [trk.get_frame(0, l) for l in range(nl)]
            if args.template == "dense":
                tpl = tpl_cache[k] = S.dense_template(scene, K, w, h, nl, ref_p)
            else:
                tpl = S.sparse_template(scene, K, w, h, nl, ref_p, n0=10000, seed=b)
        trk.setCoarseTrackingRef(b, (0.0, 0.0), 1.0, *tpl)
        trk.upload_image(0, new, 1.0)
        trk.upload_image(1, right, 1.0)
        trackers.append(trk)
        gts.append(gt)
        guesses.append(guess)
        pix = np.uint8 if args.u8 else np.float32
        if args.with_upload and args.pinned:
            from direct_stereo_slam_amd.tracker import pinned_array

            pl, pr = pinned_array(new.shape, pix), pinned_array(right.shape, pix)
            pl[...], pr[...] = new, right
            images.append((pl, pr))
        elif args.with_upload:
            images.append((np.ascontiguousarray(new, pix), np.ascontiguousarray(right, pix)))
        else:
            images.append(None)
        if b < args.cpu_frames and b < len(frames):  # the CPU legs time DISTINCT frames only
            host.append((tpl, new, right, guess))
    return dict(config=config, w=w, h=h, nl=nl, K=K, T=T, trackers=trackers, gts=np.array(gts), poses0=np.array(guesses), host=host, params=params,
                images=images, single_uploads=args.single_uploads, overlap=args.overlap, primed=False, separate_calls=args.separate_calls,
                distinct_frames=len(frames), textures=len(textures))


def one_step(ctx, wl, kf_idx, with_upload=False):
    from direct_stereo_slam_amd import synth as S

    B = len(wl["trackers"])
    if with_upload:  # host float images in, pyramids built on the device (dsm_tracker_upload_image, row N1)
        kfs = set(kf_idx)
        if wl["single_uploads"]:
            for i, trk in enumerate(wl["trackers"]):
                trk.upload_image(0, wl["images"][i][0], 1.0)
                if i in kfs:
                    trk.upload_image(1, wl["images"][i][1], 1.0)
        elif wl["overlap"]:
            # double-buffered: swap in what travelled during the previous step, start the next hand-over, track
            trks = list(wl["trackers"]) + [wl["trackers"][i] for i in kf_idx]
            imgs = [wl["images"][i][0] for i in range(B)] + [wl["images"][i][1] for i in kf_idx]
            if not wl["primed"]:
                ctx.upload_images(trks, [2] * B + [3] * len(kf_idx), imgs)
                wl["primed"] = True
            ctx.upload_wait()
            ctx.advance_frames(trks, [0] * B + [1] * len(kf_idx))
            ctx.upload_images(trks, [2] * B + [3] * len(kf_idx), imgs, asynchronous=True)
        else:  # one call: the B new left images and the keyframes' right images
            trks = list(wl["trackers"]) + [wl["trackers"][i] for i in kf_idx]
            ctx.upload_images(trks, [0] * B + [1] * len(kf_idx), [wl["images"][i][0] for i in range(B)] + [wl["images"][i][1] for i in kf_idx])
    poses0 = wl["poses0"].copy()
    kf = [wl["trackers"][i] for i in kf_idx]
    if wl.get("separate_calls") or wl["params"].work_queue >= 2:
        # two calls (the work-queue form is one kernel per call and mode)
        good, poses, affs, last, flow = ctx.track_batch(wl["trackers"], poses0, np.zeros((B, 2)), wl["nl"] - 1)
        st_track = ctx.stats()
        err, sc = ctx.optimize_scale_batch(kf, np.ones(len(kf)), wl["nl"] - 1)
        st_scale = ctx.stats()
    else:
        # one call: the keyframes' scale optimisations are independent of the frames' tracking and run beside it
        good, poses, affs, last, flow, err, sc = ctx.track_and_scale_batch(wl["trackers"], poses0, np.zeros((B, 2)), wl["nl"] - 1, kf, np.ones(len(kf)))
        st_track, st_scale = ctx.stats(), ctx.stats2()
        st_scale.total_ms = 0.0  # (one call: its time is the tracking segment's total)
    return good, poses, err, sc, st_track, st_scale


class StreamRunner:
    """bench.py --stream: the steps of the workload through one dsm_stream (B track slots + the keyframes' scale slots)"""

    def __init__(self, args, ctx, wl, kf_idx):
        from direct_stereo_slam_amd.tracker import Stream

        self.ctx, self.wl, self.kf_idx = ctx, wl, kf_idx
        self.B = len(wl["trackers"])
        self.kf = [wl["trackers"][i] for i in kf_idx]
        self.st = Stream(ctx, self.B, max(1, len(kf_idx)), args.stream_engine, args.stream_ticks)
        self.st.set_chain(args.chain)
        if args.stream_quantile is not None:
            q = [float(x) for x in str(args.stream_quantile).split(",")]
            self.st.set_quantile(q[0] if len(q) == 1 else q)
        if args.stream_rounds:
            r = [int(x) for x in args.stream_rounds.split(",")]
            self.st.set_rounds(0, r)
        self.owner = {}
        self.out = {}
        self.passes = 0
        self.trace = [] if os.environ.get("DSM_BENCH_TRACE_STEPS") else None
        self.t_last = time.perf_counter()
        self.acc = None  # statistics of the advances since reset_stats() (the stream's own are cumulative: differenced here)
        self.prev = None

    def reset_stats(self):
        self.st.sync()  # nothing of an earlier advance may leak into the new count
        self._collect(count_pass=False)
        self.acc = dict(evals=np.zeros(6, np.int64), ro=np.zeros(6, np.int64), bytes=0, bytes_scale=0, ms=0.0, l_ms=np.zeros(6), l_sum_ms=np.zeros(6),
                        dispatches=np.zeros(6, np.int64), launches=np.zeros(6, np.int64), passes=0, retired=0, wall=0.0)

    def _collect(self, count_pass=True):
        n_track = 0
        for r in self.st.results():
            self.out[r.ticket] = r
            n_track += r.kind == 0
        self.passes += 1 if count_pass else 0
        a, b = self.st.stats()
        cur = dict(evals=np.array(a.evals, np.int64), ro=np.array(a.evals_residual_only, np.int64), bytes=int(a.algorithmic_bytes), bytes_scale=int(b.algorithmic_bytes),
                   ms=float(a.total_ms), l_ms=np.array(a.eval_kernel_union_ms), l_sum_ms=np.array(a.eval_kernel_ms), dispatches=np.array(a.eval_dispatches, np.int64),
                   launches=np.array(a.launches, np.int64) + np.array(b.launches, np.int64))
        if self.acc is not None and self.prev is not None:
            self.acc["retired"] += n_track
            for k, v in cur.items():
                self.acc[k] = self.acc[k] + (v - self.prev[k])
            self.acc["passes"] += 1 if count_pass else 0
        self.prev = cur

    def step(self, tag):
        wl, B = self.wl, self.B
        tk = self.st.submit_track(wl["trackers"], wl["poses0"], np.zeros((B, 2)), wl["nl"] - 1)
        ts = self.st.submit_scale(self.kf, np.ones(len(self.kf), np.float32), wl["nl"] - 1) if self.kf else []
        self.owner[tag] = (tk, ts)
        self.st.advance()
        self._collect()
        if self.trace is not None:
            self.trace.append(round(1e3 * (time.perf_counter() - self.t_last), 2))
            self.t_last = time.perf_counter()

    def drain(self):
        while True:
            self.st.sync()  # (the tail is short chains: nothing to overlap)
            self._collect(count_pass=False)  # the statistics of what the sync read back, before the next advance starts a new count
            resident, waiting, _ = self.st.counts()
            if resident == 0 and waiting == 0:
                return
            self.st.advance()
            self._collect()

    def results_of(self, tag):
        """(good, poses, err, scales) of the step `tag`, in the order of the workload's trackers"""
        tk, ts = self.owner[tag]
        good = np.array([bool(self.out[t].good) for t in tk])
        poses = np.array([list(self.out[t].pose) for t in tk])
        err = np.array([self.out[t].err for t in ts], np.float32)
        sc = np.array([self.out[t].scale for t in ts], np.float32)
        self.last_evals = np.array([list(self.out[t].evals) for t in tk], np.int64)  # per frame and level
        return good, poses, err, sc

    def close(self):
        self.st.close()


class SequenceUploadRunner:
    """The streamed form WITH the hand-over inside: every tracker is one SEQUENCE with ONE frame in flight, as a
    node runs it (FrontEnd.cpp:585-686: the next image is taken once the previous frame's pose is back).  Per step: the results of the
    earlier advances are collected; for every sequence whose frame (and, on the keyframe sequences, scale optimisation) has retired, the
    NEXT frame -- camera bytes in page-locked buffers, handed over asynchronously into the back buffers while the previous frame was being
    tracked (dsm_upload_images_async: copies + device pyramids on a stream of their own) -- is swapped in (dsm_frames_advance), the frame
    after it starts travelling, and the new problems are submitted; then one advance.  A sequence's images are only ever replaced after its
    own problem has retired: no in-flight problem reads a buffer that is being written."""

    def __init__(self, args, ctx, wl, kf_idx):
        from direct_stereo_slam_amd.tracker import Stream

        self.ctx, self.wl = ctx, wl
        self.S = len(wl["trackers"])
        self.is_kf = np.zeros(self.S, bool)
        self.is_kf[kf_idx] = True
        self.st = Stream(ctx, self.S, max(1, int(self.is_kf.sum())), 1, args.stream_ticks)
        self.st.set_chain(args.chain)
        self.owner, self.outstanding = {}, np.zeros(self.S, np.int32)
        self.ready = list(range(self.S))
        self.primed = False
        self.frames_done = 0
        self.bytes_handed = 0
        self.last = {}

    def _hand_over(self, seqs, asynchronous):
        t, im = self.wl["trackers"], self.wl["images"]
        kf = [i for i in seqs if self.is_kf[i]]
        trks = [t[i] for i in seqs] + [t[i] for i in kf]
        imgs = [im[i][0] for i in seqs] + [im[i][1] for i in kf]
        self.ctx.upload_images(trks, [2] * len(seqs) + [3] * len(kf), imgs, asynchronous=asynchronous)
        self.bytes_handed += sum(x.nbytes for x in imgs)
        return trks, kf

    def step(self):
        wl, t = self.wl, self.wl["trackers"]
        for r in self.st.results():
            i, kind = self.owner.pop(r.ticket)
            self.outstanding[i] -= 1
            if kind == 0:
                self.frames_done += 1
                self.last[i] = r
            if self.outstanding[i] == 0:
                self.ready.append(i)
        if self.ready:
            seqs = self.ready
            self.ready = []
            if not self.primed:  # the very first frames: nothing to overlap with
                self._hand_over(seqs, False)
                self.primed = True
            else:
                self.ctx.upload_wait()  # (the batch started one step ago: long done)
            kf = [i for i in seqs if self.is_kf[i]]
            trks = [t[i] for i in seqs] + [t[i] for i in kf]
            self.ctx.advance_frames(trks, [0] * len(seqs) + [1] * len(kf))
            self._hand_over(seqs, True)  # the frames after these travel while these are tracked
            tk = self.st.submit_track([t[i] for i in seqs], wl["poses0"][seqs], np.zeros((len(seqs), 2)), wl["nl"] - 1)
            ts = self.st.submit_scale([t[i] for i in kf], np.ones(len(kf), np.float32), wl["nl"] - 1) if kf else []
            for i, k in zip(seqs, tk):
                self.owner[k] = (i, 0)
                self.outstanding[i] += 1
            for i, k in zip(kf, ts):
                self.owner[k] = (i, 1)
                self.outstanding[i] += 1
        self.st.advance()

    def finish(self):
        """everything in flight retires (no new frames)"""
        self.st.drain()
        for r in self.st.results():
            i, kind = self.owner.pop(r.ticket)
            self.outstanding[i] -= 1
            if kind == 0:
                self.frames_done += 1
                self.last[i] = r
            if self.outstanding[i] == 0:
                self.ready.append(i)
        self.ctx.upload_wait()

    def close(self):
        self.st.close()


def measure_stream_with_upload(args, ctx, wl, steps, warmup, world):
    """the streamed form with the hand-over inside: W warm-up steps, then K timed steps + the drain of what they started; value = frames
    retired inside the timed region / its wall time (every sequence has one frame in flight: a step hands over as many frames as retired)"""
    S = len(wl["trackers"])
    run = SequenceUploadRunner(args, ctx, wl, list(range(0, S, args.kf_every)))
    for _ in range(warmup + 2):
        run.step()
    ctx.sync()
    barrier_sync(world)
    import gc

    gc.collect()
    gc.disable()
    f0, b0 = run.frames_done, run.bytes_handed
    t0 = time.perf_counter()
    for _ in range(steps):
        run.step()
    ctx.sync()
    run.st.sync()
    barrier_sync(world)
    dt = max_over_ranks(time.perf_counter() - t0, world)
    gc.enable()
    # (the timed region is a window of the steady state: the frames counted are those whose results the steps inside it collected)
    frames, handed = run.frames_done - f0, run.bytes_handed - b0
    run.finish()
    poses = np.array([list(run.last[i].pose) for i in range(S)])
    good = np.array([bool(run.last[i].good) for i in range(S)])
    terr = np.abs(poses[:, 4:] - wl["gts"][:, 4:]).max(1)
    run.close()
    return {"value": frames * world / dt, "ms_per_step": 1e3 * dt / steps, "frames_in_timed_region": int(frames), "sequences": S, "steps": steps,
            "host_bytes_per_step": handed / steps, "host_GBps": handed / dt / 1e9, "all_tracked": bool(good.all()),
            "frames_with_translation_error_above_1cm": int((terr > 0.01).sum()),
            "ticks_per_advance": args.stream_ticks or "auto"}


def measure_stream(args, ctx, wl, steps, warmup, world):
    """--stream: W warmup steps + drain, then the timed region = K steps (each: B frames submitted, one pass) + the drain of
    everything they submitted, bracketed by barrier + synchronize (max over ranks); then instrumented passes in steady state
    for the roofline of the dominant kernel."""
    B = len(wl["trackers"])
    kf_idx = list(range(0, B, args.kf_every))
    run = StreamRunner(args, ctx, wl, kf_idx)
    for i in range(warmup):
        run.step(("w", i))
    run.drain()
    ctx.sync()
    barrier_sync(world)
    run.reset_stats()
    p0 = run.passes
    import gc

    gc.collect()
    gc.disable()  # (see measure(): Python's collector is host noise, not the path)
    t0 = time.perf_counter()
    for i in range(steps):
        run.step(("t", i))
    run.drain()
    ctx.sync()
    barrier_sync(world)
    dt = max_over_ranks(time.perf_counter() - t0, world)
    gc.enable()
    timed = run.acc
    run.acc = None
    if run.trace is not None:
        sys.stderr.write(f"per-step host ms (warm-up, then the timed steps): {run.trace}\n")
    timed_passes = run.passes - p0
    good, poses, err, sc = run.results_of(("t", steps - 1))
    last_evals = run.last_evals
    sched = run.st.schedule(0)
    # roofline of the dominant kernel: steady-state passes (the pool refilled every pass) with HIP events around every eval dispatch
    for i in range(10):
        run.step(("r", i))
    run.reset_stats()
    ctx.sync()
    ts0 = time.perf_counter()
    for i in range(8):  # steady state without instrumentation: frames retired per second while the pool is kept full
        run.step(("s", i))
    ctx.sync()
    run.st.sync()
    run._collect(count_pass=False)
    steady = dict(run.acc, wall=time.perf_counter() - ts0)
    ctx.set_timing(True)
    run.reset_stats()
    for i in range(4):
        run.step(("i", i))
    ctx.set_timing(False)
    inst = run.acc
    run.acc = None
    run.drain()
    run.close()
    n0 = len(wl["trackers"][0].get_template(0)[0])
    n_l = [len(wl["trackers"][0].get_template(l)[0]) for l in range(wl["nl"])]
    px_l = [(wl["w"] >> l) * (wl["h"] >> l) for l in range(wl["nl"])]
    by_l = [16 * n + min(12 * px, 48 * n) for n, px in zip(n_l, px_l)]      # the reference's data per evaluation of level l
    lay_l = [16 * n + min(4 * px, 48 * n) for n, px in zip(n_l, px_l)]      # what this implementation keeps in HBM for it (intensity planes)
    ro_l = [16 * n + min(4 * px, 16 * n) for n, px in zip(n_l, px_l)]       # a residual-only evaluation priced at what it reads
    bytes_eval0, layout_eval0, ro_bytes_eval0 = by_l[0], lay_l[0], ro_l[0]
    ticks = args.stream_engine == 1
    # engine 1: ONE evaluation kernel per tick covers every level (its dispatches are booked under index 0); engine 0: the level-0 kernel
    lv = range(wl["nl"]) if ticks else [0]
    l0_ms = float(inst["l_ms"][0])
    l0_evals, l0_ro = int(sum(inst["evals"][l] for l in lv)), int(sum(inst["ro"][l] for l in lv))
    l0_launches = int(inst["launches"][0]) if not ticks else int(inst["dispatches"][0] // max(1, min(args.streams, len(wl["trackers"]))))
    l0_disp = int(inst["dispatches"][0])
    l0_bytes = int(sum(int(inst["evals"][l]) * by_l[l] for l in lv))
    achieved = l0_bytes / (l0_ms * 1e-3) / 1e9 if l0_ms > 0 else 0.0
    achieved_ro_priced = sum((int(inst["evals"][l]) - int(inst["ro"][l])) * by_l[l] + int(inst["ro"][l]) * ro_l[l] for l in lv) / (l0_ms * 1e-3) / 1e9 if l0_ms > 0 else 0.0
    lay_bytes = int(sum(int(inst["evals"][l]) * lay_l[l] for l in lv))
    ratio, src, why_not = pmc_traffic_ratio(wl["config"])
    all_bytes = int(timed["bytes"] + timed["bytes_scale"])
    whole = all_bytes / dt / 1e9
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "frac_whole_step": whole / HBM_PEAK_GBS,
                "frac_full_evals": achieved_ro_priced / HBM_PEAK_GBS,
                "frac_hbm_actual": achieved * ratio / HBM_PEAK_GBS if ratio is not None else None,
                "traffic": ratio * l0_bytes / max(1, l0_launches) if ratio is not None else None,
                "traffic_source": f"{src}: stored rocprofv3 --pmc summary of this command on these kernel sources, scaled to this run's bytes per launch (not re-measured here)" if src else why_not,
                "kernel": "tick_eval_kernel<pose> (one launch per tick and stream group: the staged evaluations of EVERY pyramid level)" if ticks else "eval_kernel<pose, LVL0>",
                "bytes_per_eval": int(bytes_eval0), "bytes_per_residual_only_eval": int(ro_bytes_eval0),
                "bytes_per_eval_by_level": [int(b) for b in by_l],
                "layout_bytes_per_eval": int(layout_eval0), "achieved_on_layout_bytes": achieved * lay_bytes / max(1, l0_bytes),
                "evals": l0_evals, "residual_only_evals": l0_ro, "evals_by_level": [int(inst["evals"][l]) for l in range(wl["nl"])],
                "bytes_per_launch": l0_bytes / max(1, l0_launches),
                "avg_launch_us": 1e3 * float(inst["l_sum_ms"][0]) / max(1, l0_disp), "launches": l0_launches, "dispatches": l0_disp,
                "stream_groups": args.streams, "kernel_busy_us_per_launch": 1e3 * l0_ms / max(1, l0_launches),
                "measured_over": "4 steady-state advances of the stream after the timed region (pool refilled before each), HIP events around every dispatch of the kernel on the stream it is launched on; time = union of the stream groups' overlapping dispatch intervals"}
    per_level = []
    for l in range(wl["nl"]):
        nl_ = len(wl["trackers"][0].get_template(l)[0])
        by = 16 * nl_ + min(12 * (wl["w"] >> l) * (wl["h"] >> l), 48 * nl_)
        ms = float(inst["l_ms"][l])
        per_level.append({"lvl": l, "evals": int(inst["evals"][l]), "residual_only": int(inst["ro"][l]), "launches": int(inst["launches"][l]),
                          "kernel_ms": round(ms, 4), "GBps": round(int(inst["evals"][l]) * by / (ms * 1e-3) / 1e9, 1) if ms > 0 else None})
    terr_all = np.abs(poses[:, 4:] - wl["gts"][:, 4:]).max(1)
    frames = B * steps
    detail = {"frames_in_flight_per_gpu": B, "work_queue_blocks": 0, "adaptive_schedule": True, "persistent_coarse": int(wl["params"].persistent_coarse),
              "chunk_geometry": ("throughput table (library default)", "latency table", "chain table (latency table above 4096 points, one chunk below)")[int(wl["params"].chunk_geometry)],
              "streams": args.streams,
              "form": ("stream, tick engine (dsm_stream_*: every resident problem advances one LM round per tick, admission and retirement on the device; "
                       "a step submits its frames and runs one advance, the pool is drained after the last step)") if ticks else
                      "stream, pass engine (dsm_stream_*: one sweep of the pyramid per step with carried stragglers + drain)",
              "stream": {"engine": "ticks" if ticks else "passes", "track_slots": B, "scale_slots": max(1, len(kf_idx)), "advances_in_timed_region": int(timed_passes),
                         "ticks_per_advance": (args.stream_ticks or "auto (library default: what retires about what an advance hands over; mean of the timed region %.1f)"
                                               % (float(timed["launches"][0]) / max(1, 2 * timed_passes))) if ticks else None,
                         "rounds_per_level_of_a_pass": None if ticks else sched["rounds"][:wl["nl"]], "quantile": args.stream_quantile if args.stream_quantile is not None else "library default",
                         "frames_submitted": frames, "ms_per_pass": 1e3 * dt / max(1, timed_passes),
                         "device_ms_per_advance": float(timed["ms"]) / max(1, timed_passes),
                         "host_share_of_timed_region": 1.0 - float(timed["ms"]) * 1e-3 / dt,
                         "group_streams": dict(zip(("in_use", "sharing_a_hardware_queue"), ctx.stream_queues())),
                         "steady_state": {"frames_per_s": steady["retired"] / steady["wall"], "passes": int(steady["passes"]), "frames_retired": int(steady["retired"]),
                                          "ms_per_pass": 1e3 * steady["wall"] / max(1, steady["passes"]),
                                          "what": "8 advances after the timed region with waiting frames at hand all the time: frames retired / wall time (no ramp-up, no drain)"}},
              "launch_pairs_per_step": int(timed["launches"].sum() / max(1, steps)), "readbacks_per_step": timed_passes / max(1, steps),
              "evals_per_frame_by_level": [float(timed["evals"][l]) / frames for l in range(wl["nl"])],
              "algorithmic_MB_per_frame": all_bytes / frames / 1e6,
              "whole_step_GBps": whole,
              "pose_eval_kernels_by_level": per_level, "max_abs_translation_error_m": float(terr_all.max()),
              "distinct_frames": int(wl["distinct_frames"]), "textures": int(wl["textures"]),
              "initial_guess": args.init if args.init == "identity" else f"constant-motion (error sigma {args.init_err} x motion sigma)", "texture_list": args.textures,
              "scene_family": args.scene_family,
              "fixed_schedule": int(wl["params"].fixed_schedule), "work_queue": int(wl["params"].work_queue),
              "frames_with_translation_error_above_1cm": int((terr_all > 0.01).sum()), "all_tracked": bool(good.all())}
    return dict(dt=dt, value=world * frames / dt, ms_per_step=1e3 * dt / steps, good=good, poses=poses, roofline=roofline, detail=detail, n0=n0,
                evals=last_evals)


def kernel_source_sha():
    """sha256 over the sources the eval kernels are built from: a stored PMC profile only speaks for the kernel it profiled"""
    import hashlib

    hsh = hashlib.sha256()
    for f in ("tracker_kernels.hip", "dsm_device.hpp", "dsm_kernels.hpp", "lm_math.hpp", "Makefile"):
        hsh.update(open(os.path.join(ROOT, "direct_stereo_slam_amd", "csrc", f), "rb").read())
    return hsh.hexdigest()[:16]


def pmc_traffic_ratio(config):
    """HBM bytes per algorithmic byte of the level-0 pose evaluation from the PMC passes committed under profiles/
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of this command, corrected as MI355X_MICROARCH.md
    prescribes): NOT re-measured inside this run.  A profile taken on other kernel sources (kernel_source_sha stamped by
    tools/summarize_profiles.py) is refused.  Returns (ratio, source file, note)."""
    import glob

    sha = kernel_source_sha()
    stale = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True):  # latest round first
        try:
            pm = json.load(open(f))
            if pm.get("config", "S1") != config:
                continue
            if pm.get("kernel_source_sha") != sha:
                stale = stale or os.path.relpath(f, ROOT)
                continue
            return pm["hbm_bytes_per_algorithmic_byte_level0_pose_eval"], os.path.relpath(f, ROOT), None
        except Exception:
            pass
    return None, None, (f"{stale} was taken on other kernel sources (kernel_source_sha != {sha}): refused; regenerate with tools/profile_round.sh" if stale
                        else "no PMC profile of this configuration under profiles/")


def measure(args, ctx, wl, steps, warmup, world, with_upload=False):
    """W warmup steps, then exactly `steps` timed steps bracketed by barrier + synchronize on both sides (max over ranks),
    then ONE extra instrumented step for the per-dispatch roofline of the dominant kernel."""
    if args.stream and not with_upload:
        return measure_stream(args, ctx, wl, steps, warmup, world)
    B = len(wl["trackers"])
    kf_idx = list(range(0, B, args.kf_every))
    for _ in range(warmup):
        one_step(ctx, wl, kf_idx, with_upload)
    ctx.sync()
    barrier_sync(world)
    import gc

    gc.collect()
    gc.disable()  # (a generation-2 collection of this process's many small objects is a 40 ms host pause: measured inside a 33 ms timed region, round 4)
    t0 = time.perf_counter()
    trace = [] if os.environ.get("DSM_BENCH_TRACE_STEPS") else None
    for _ in range(steps):
        ts = time.perf_counter()
        out = one_step(ctx, wl, kf_idx, with_upload)
        if trace is not None:
            trace.append(round(1e3 * (time.perf_counter() - ts), 3))
    ctx.sync()
    barrier_sync(world)
    dt = max_over_ranks(time.perf_counter() - t0, world)
    gc.enable()
    if trace is not None:
        sys.stderr.write(f"per-step ms: {trace}\n")
    good, poses = out[0], out[1]

    # roofline of the dominant kernel (level-0 pose eval): one extra step, same stream configuration, with one pair
    # of HIP events around every eval-kernel dispatch ON THE STREAM IT IS LAUNCHED ON.  The stream groups' level-0
    # dispatches overlap, so the kernel's time is the union of the dispatch intervals (what a rocprofv3 kernel trace
    # of the same command shows); avg_launch_us is the plain per-dispatch average, as rocprofv3 --stats reports it.
    ctx.set_timing(True)
    out_t = one_step(ctx, wl, kf_idx)
    ctx.set_timing(False)
    stt, sts = out_t[4], out_t[5]
    n0 = len(wl["trackers"][0].get_template(0)[0])
    bytes_eval0 = 16 * n0 + min(12 * wl["w"] * wl["h"], 48 * n0)  # the reference's data (16 B template entries, 12 B (I, dx, dy) texels); sparse templates do not touch the whole image
    layout_eval0 = 16 * n0 + min(4 * wl["w"] * wl["h"], 48 * n0)  # what this implementation keeps in HBM for the same evaluation: the intensity plane only (gradients are formed from neighbouring intensities)
    l0_ms = stt.eval_kernel_union_ms[0]
    l0_launches, l0_dispatches, l0_evals = stt.launches[0], stt.eval_dispatches[0], stt.evals[0]
    # The last evaluation of a level's LM loop runs residual-only when the loop is known to end after it (dsm_stats.
    # evals_residual_only): calcResPose in full -- which is what SURVEY.md 8(d)'s per-evaluation figure prices: the template
    # and the (I, dx, dy) texels calcResPose interpolates -- without the calcGSSSEPose the reference runs on it and never
    # reads.  Counted like every other evaluation; reported next to the total.
    l0_ro = stt.evals_residual_only[0]
    l0_bytes = l0_evals * bytes_eval0
    achieved = l0_bytes / (l0_ms * 1e-3) / 1e9 if l0_ms > 0 else 0.0
    ratio, src, why_not = pmc_traffic_ratio(wl["config"])
    # residual-only evaluations priced at what they need of the reference's data: the template and the INTENSITY channel
    ro_bytes_eval0 = 16 * n0 + min(4 * wl["w"] * wl["h"], 16 * n0)
    achieved_ro_priced = ((l0_evals - l0_ro) * bytes_eval0 + l0_ro * ro_bytes_eval0) / (l0_ms * 1e-3) / 1e9 if l0_ms > 0 else 0.0
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                # frac: SURVEY.md 8d's contract figure (every evaluation = 16 n + 12 w h bytes of the reference's data structures).
                # frac_full_evals: the same with the residual-only evaluations priced at 16 n + 4 w h (they read no gradients).
                # frac_hbm_actual: what the memory pins really moved = achieved x (PMC bytes / algorithmic bytes) / peak.
                "frac_full_evals": achieved_ro_priced / HBM_PEAK_GBS,
                "frac_hbm_actual": achieved * ratio / HBM_PEAK_GBS if ratio is not None else None,
                "traffic": ratio * l0_bytes / max(1, l0_launches) if ratio is not None else None,
                "traffic_source": f"{src}: stored rocprofv3 --pmc summary of this command on these kernel sources, scaled to this run's bytes per launch (not re-measured here)" if src else why_not,
                "kernel": "eval_kernel<pose, LVL0>", "bytes_per_eval": int(bytes_eval0), "bytes_per_residual_only_eval": int(ro_bytes_eval0),
                "layout_bytes_per_eval": int(layout_eval0), "achieved_on_layout_bytes": achieved * layout_eval0 / bytes_eval0,
                "evals": int(l0_evals), "residual_only_evals": int(l0_ro),
                "bytes_per_launch": l0_bytes / max(1, l0_launches),
                "avg_launch_us": 1e3 * stt.eval_kernel_ms[0] / max(1, l0_dispatches), "launches": int(l0_launches),
                "dispatches": int(l0_dispatches), "stream_groups": args.streams,
                "kernel_busy_us_per_launch": 1e3 * l0_ms / max(1, l0_launches)}
    if stt.queue_blocks > 0:
        # work-queue form: the whole track call is ONE kernel; its algorithmic bytes are those of every evaluation at every level
        qms = stt.queue_kernel_ms
        q_ach = stt.algorithmic_bytes / (qms * 1e-3) / 1e9 if qms > 0 else 0.0
        roofline = {"bound": "hbm", "achieved": q_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": q_ach / HBM_PEAK_GBS, "traffic": None,
                    "kernel": "queue_kernel<pose> (whole track call: all levels' evaluations + LM steps in one launch)",
                    "bytes_per_launch": int(stt.algorithmic_bytes), "avg_launch_us": 1e3 * qms, "launches": 1,
                    "persistent_workgroups": int(stt.queue_blocks), "queue_items": int(stt.queue_items)}
    per_level = []
    for l in range(wl["nl"]):
        nl_ = len(wl["trackers"][0].get_template(l)[0])
        by = 16 * nl_ + min(12 * (wl["w"] >> l) * (wl["h"] >> l), 48 * nl_)
        ms = stt.eval_kernel_union_ms[l]
        per_level.append({"lvl": l, "evals": int(stt.evals[l]), "residual_only": int(stt.evals_residual_only[l]), "launches": int(stt.launches[l]),
                          "kernel_ms": round(ms, 4), "GBps": round(stt.evals[l] * by / (ms * 1e-3) / 1e9, 1) if ms > 0 else None})
    all_bytes = stt.algorithmic_bytes + sts.algorithmic_bytes
    roofline["frac_whole_step"] = all_bytes / (dt / steps) / 1e9 / HBM_PEAK_GBS  # every evaluation of the step on SURVEY.md 8d's bytes / the timed step
    terr_all = np.abs(poses[:, 4:] - wl["gts"][:, 4:]).max(1)
    detail = {"frames_in_flight_per_gpu": B, "work_queue_blocks": int(stt.queue_blocks), "adaptive_schedule": not args.no_adaptive,
              "persistent_coarse": int(wl["params"].persistent_coarse), "streams": args.streams,
              "chunk_geometry": ("throughput table (library default)", "latency table", "chain table (latency table above 4096 points, one chunk below)")[int(wl["params"].chunk_geometry)],
              "launch_pairs_per_step": int(sum(stt.launches) + sum(sts.launches)), "readbacks_per_step": int(stt.polls + sts.polls),
              "evals_per_frame_by_level": [stt.evals[l] / B for l in range(wl["nl"])],
              "algorithmic_MB_per_frame": all_bytes / B / 1e6,
              "whole_step_GBps": all_bytes / (1e-3 * (stt.total_ms + sts.total_ms)) / 1e9,
              "pose_eval_kernels_by_level": per_level, "max_abs_translation_error_m": float(terr_all.max()),
              "distinct_frames": int(wl["distinct_frames"]), "textures": int(wl["textures"]), "initial_guess": args.init if args.init == "identity" else f"constant-motion (error sigma {args.init_err} x motion sigma)", "texture_list": args.textures,
              "scene_family": args.scene_family,
              "fixed_schedule": int(wl["params"].fixed_schedule), "work_queue": int(wl["params"].work_queue),
              "frames_with_translation_error_above_1cm": int((terr_all > 0.01).sum()), "all_tracked": bool(good.all())}
    return dict(dt=dt, value=world * B * steps / dt, ms_per_step=1e3 * dt / steps, good=good, poses=poses, roofline=roofline,
                detail=detail, n0=n0)


def workload_label(args, wl, n0):
    sched = f"fixed schedule 1+{wl['params'].fixed_schedule} evaluations per level" if wl["params"].fixed_schedule > 0 else "LM as executed"
    scenes = "relief scenes (broadband texture)" if args.scene_family == "relief" else "single-plane scenes of rounds 1-3"
    return (f"{CONFIGS[wl['config']][3]}, {args.template} template n0={n0}, {sched}, {wl['distinct_frames']} distinct frames, {scenes}, "
            f"track every frame + scale-opt every {args.kf_every}th")


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1): the oracle, calcGSSSE* in their SSE-intrinsics form -- TEST INFRASTRUCTURE timed as the
# reported baseline, never part of the product path
# ------------------------------------------------------------------------------------------------------------------
def _oracle_tracker(wl, frame):
    from oracle import oracle as O

    tpl, new, right, _guess = frame
    nl, w, h = wl["nl"], wl["w"], wl["h"]
    op = O.default_params(native=True)
    op.fixed_schedule = int(wl.get("fixed_schedule", 0))  # the same schedule as the GPU leg it is timed beside
    orc = O.OracleTracker(w, h, nl, wl["T"], wl["K"], op, native=True)
    orc.use_sse(True)
    orc.make_k(*wl["K"])
    orc.set_ref(0, 0.0, 0.0, 1.0, *tpl)
    orc.set_frame(0, O.make_images(new, nl, native=True), 1.0)
    orc.set_frame(1, O.make_images(right, nl, native=True), 1.0)
    return orc


def host_cpu_info():
    """(model name, physical cores, logical cpus) of this box"""
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except Exception:
        pass
    logical = os.cpu_count() or 1
    try:
        logical = len(os.sched_getaffinity(0))
    except Exception:
        pass
    n_phys = len(phys) if phys else max(1, logical // 2)
    return model, min(n_phys, logical), logical


_ALLCORE = {}


def _allcore_worker(job):
    """one independent sequence per core (leg ii): a forked worker builds its own oracle trackers,
    waits for everybody at a barrier, then tracks its share; returns (start, end) on the system-wide monotonic clock"""
    from direct_stereo_slam_amd import synth as S

    wl, kf_every, barrier = _ALLCORE["wl"], _ALLCORE["kf_every"], _ALLCORE["barrier"]
    nl = wl["nl"]
    trks = [(i, _oracle_tracker(wl, wl["host"][i % len(wl["host"])]), wl["host"][i % len(wl["host"])][3]) for i in job]
    barrier.wait()
    t0 = time.perf_counter()
    for i, orc, guess in trks:
        orc.track(guess, [0, 0], nl - 1)
        if i % kf_every == 0:
            orc.optimize_scale(1.0, nl - 1)
    return t0, time.perf_counter()


def cpu_all_cores(args, wl, per_core=6):
    """the same frames, `per_core` of them per PHYSICAL core in forked processes that start together (the workers never
    touch the GPU); wall time from the common start to the last finish"""
    import multiprocessing as mp

    model, phys, logical = host_cpu_info()
    cores = max(2, phys)
    ctxm = mp.get_context("fork")
    _ALLCORE.update(wl={**{k: wl[k] for k in ("nl", "w", "h", "T", "K", "host")}, "fixed_schedule": int(wl["params"].fixed_schedule)},
                    kf_every=args.kf_every, barrier=ctxm.Barrier(cores))
    jobs = [list(range(c * per_core, (c + 1) * per_core)) for c in range(cores)]
    with ctxm.Pool(cores) as pool:
        spans = pool.map(_allcore_worker, jobs, chunksize=1)
    wall = max(e for _, e in spans) - min(s for s, _ in spans)
    n = cores * per_core
    return {"value": n / wall, "unit": "stereo frames/s", "cores": cores, "cpu_model": model, "logical_cpus": logical,
            "sample": f"{per_core} frames per physical core ({n} in all, the bench frames cycled) in {cores} forked processes started "
                      f"together, {wall:.2f} s from the common start to the last finish"}


def cpu_baseline(args, wl, gpu_poses=None, gpu_good=None, gpu_evals=None):
    """the oracle with calcGSSSEPose / calcGSSSEScale in their SSE-intrinsics form (oracle/dsm_oracle_sse.c: the
    reference's own loop structure, TrackerAndScaler.cpp:640-697,966-1005) timed on this box's host cores: 1 thread, as
    the reference runs this path on the image-callback thread.  Built with -O3 -march=native like CMakeLists.txt:4-6."""
    from direct_stereo_slam_amd import synth as S

    nl, w, h = wl["nl"], wl["w"], wl["h"]
    frames = wl["host"]
    if not frames:
        return None
    model, phys, logical = host_cpu_info()
    wlc = {**{k: wl[k] for k in ("nl", "w", "h", "T", "K")}, "fixed_schedule": int(wl["params"].fixed_schedule)}
    n_min = min(len(frames), args.cpu_min_frames)
    cpu_poses, cpu_good, cpu_evals = [], [], []
    dt = 0.0
    for i, fr in enumerate(frames):
        if i >= n_min and dt > args.cpu_seconds:  # at least cpu_min_frames distinct frames, then until the time budget is spent
            break
        orc = _oracle_tracker(wlc, fr)  # set-up (pyramids, template upload) is outside the timed region, as on the GPU
        t0 = time.perf_counter()
        r = orc.track(fr[3], [0, 0], nl - 1)
        dt += time.perf_counter() - t0
        cpu_evals.append(orc.eval_counts()[0])  # (outside the timed region)
        t0 = time.perf_counter()
        if i % args.kf_every == 0:
            orc.optimize_scale(1.0, nl - 1)
        dt += time.perf_counter() - t0
        cpu_good.append(bool(r[0]))
        cpu_poses.append(np.asarray(r[1]))
    n = len(cpu_poses)
    out = {"value": n / dt, "unit": "stereo frames/s", "cores": 1, "kind": "port", "form": "sse-restatement",
           "cpu_model": model, "physical_cores": phys,
           "sample": f"the first {n} of the bench's {wl['distinct_frames']} DISTINCT frames ({w}x{h}x{nl} {args.template}), same initial guesses"
                     f"{' and the same fixed schedule' if wlc['fixed_schedule'] else ''}, track + scale-opt every {args.kf_every}th, oracle/dsm_oracle.c with the SSE-intrinsics "
                     f"calcGSSSE* of oracle/dsm_oracle_sse.c, gcc -O3 -march=native, {dt:.2f} s on one core"}
    if gpu_poses is not None:
        # the "ATE vs CPU ref" half of the metric on the very frames that were timed: translation error of both
        # paths against the synthetic ground truth over ALL of them, and the GPU path against the CPU path
        cp, gp, gt = np.array(cpu_poses)[:, 4:], np.asarray(gpu_poses)[:n, 4:], wl["gts"][:n, 4:]
        ate = lambda a, b: float(np.sqrt(np.mean(np.sum((a - b) ** 2, 1)))) if len(a) else float("nan")
        # Frames on which the REFERENCE path (CPU) itself ends in a wrong minimum (> 1 cm from the ground truth) are chaotic on
        # both paths; the ATE half of the metric is taken over the frames the CPU path tracks (selected by the CPU result
        # alone, so a GPU-only failure would show) and, for completeness, over all frames.
        ok = np.abs(cp - gt).max(1) <= 0.01
        # RPE (north_star: "within a stated ATE/RPE tolerance"): every bench frame IS one relative pose -- keyframe to
        # new frame, one step of the sequence -- so the relative pose error over a step of one frame is E = T_gt^-1 T_est per frame:
        # RMSE of |trans(E)| and of the rotation angle of E, both paths against the ground truth and the GPU path against the CPU path
        def rpe(est, ref):
            est, ref = np.asarray(est, np.float64), np.asarray(ref, np.float64)
            if not len(est):
                return float("nan"), float("nan")
            tr, ro = [], []
            for a, b in zip(est, ref):
                Ra, Rb = S.quat_to_rot(a[:4]), S.quat_to_rot(b[:4])
                Re, te = Rb.T @ Ra, Rb.T @ (a[4:] - b[4:])
                tr.append(float(te @ te))
                ro.append(float(np.arccos(np.clip((np.trace(Re) - 1.0) / 2.0, -1.0, 1.0))) ** 2)
            return float(np.sqrt(np.mean(tr))), float(np.degrees(np.sqrt(np.mean(ro))))
        gpa, cpa, gta = np.asarray(gpu_poses)[:n], np.array(cpu_poses), np.asarray(wl["gts"])[:n]
        (rg_t, rg_r), (rc_t, rc_r), (rd_t, rd_r) = rpe(gpa[ok], gta[ok]), rpe(cpa[ok], gta[ok]), rpe(gpa[ok], cpa[ok])
        out["rpe_vs_cpu_ref"] = {"delta_frames": 1, "frames": int(ok.sum()), "rpe_trans_gpu_m": rg_t, "rpe_trans_cpu_m": rc_t,
                                 "rpe_trans_ratio_gpu_over_cpu": rg_t / max(rc_t, 1e-30), "rpe_rot_gpu_deg": rg_r, "rpe_rot_cpu_deg": rc_r,
                                 "rpe_rot_ratio_gpu_over_cpu": rg_r / max(rc_r, 1e-30), "rpe_gpu_against_cpu_trans_m": rd_t, "rpe_gpu_against_cpu_rot_deg": rd_r,
                                 "tolerance": "ratios within 1 % of 1 (the metric's ATE bar applied to the RPE as well)"}
        out["ate_vs_cpu_ref"] = {"frames": int(ok.sum()), "ate_gpu_m": ate(gp[ok], gt[ok]), "ate_cpu_m": ate(cp[ok], gt[ok]),
                                 "ate_ratio_gpu_over_cpu": ate(gp[ok], gt[ok]) / max(ate(cp[ok], gt[ok]), 1e-30),
                                 "max_abs_translation_diff_gpu_vs_cpu_m": float(np.abs(gp[ok] - cp[ok]).max()) if ok.any() else None,
                                 "frames_cpu_path_ends_in_a_wrong_minimum": int((~ok).sum()),
                                 "of_which_gpu_path_too": int((np.abs(gp[~ok] - gt[~ok]).max(1) > 0.01).sum()),
                                 "all_frames": {"frames": n, "ate_gpu_m": ate(gp, gt), "ate_cpu_m": ate(cp, gt)},
                                 # the same ratio over EVERY frame, next to the filtered one (rounds 1-2 reported this one as the headline;
                                 # on frames both paths lose it compares two wrong minima)
                                 "ate_ratio_gpu_over_cpu_all_frames": ate(gp, gt) / max(ate(cp, gt), 1e-30),
                                 "distinct_frames": n,
                                 "good_flags_equal": bool(np.array_equal(np.asarray(gpu_good)[:n].astype(bool), np.array(cpu_good))),
                                 "all_tracked_cpu": bool(all(cpu_good))}
        if gpu_evals is not None:
            # LM route of every frame on both paths: the per-level evaluation counts (equal counts = the same accept / reject /
            # break decisions all the way, TrackerAndScaler.cpp:559,588) and, with equal counts, the end points
            ce, ge = np.array(cpu_evals, np.int64)[:, :nl], np.asarray(gpu_evals)[:n, :nl]
            same = (ce == ge).all(1)
            pd = np.abs(np.asarray(gpu_poses)[:n] - np.array(cpu_poses)).max(1)
            out["lm_routes_vs_cpu_ref"] = {"frames": n, "same_evaluation_counts": int(same.sum()), "different_evaluation_counts": int((~same).sum()),
                                           "same_counts_but_poses_apart_1e-4": int((same & (pd >= 1e-4)).sum()),
                                           "different_counts_among_frames_the_cpu_path_tracks": int((~same & ok).sum()),
                                           "max_pose_diff_same_route": float(pd[same & (pd < 1e-4)].max()) if (same & (pd < 1e-4)).any() else None}
    if not args.no_cpu_all_cores:
        try:
            out["all_cores"] = cpu_all_cores(args, wl)
        except Exception as e:  # a reported extra, never a reason to lose the bench line
            out["all_cores"] = {"error": repr(e)}
    return out


# ------------------------------------------------------------------------------------------------------------------
# sharded ring-key database: the one part of the path with a real exchange step
# ------------------------------------------------------------------------------------------------------------------
def make_ringkey_data(n, q):
    rng = np.random.default_rng(1234)
    p = rng.uniform(0.1, 0.9, 20)
    keys = (rng.binomial(60, p, size=(n, 20)) / 60.0).astype(np.float32)
    qs = (keys[rng.integers(n, size=q)] + rng.normal(0, 0.02, (q, 20))).astype(np.float32)
    return keys, qs


def ringkey_sharded_leg(args, ctx, rank, world, steps=20, check=True):
    """DB of rk_n keys split `ordinal mod world`; Q queries per step: local scan (HIP) + cross-shard merge through the C ABI
    (dsm_ringdb_merge_topk: RCCL all-reduce(min), k rounds with winner pop; the all-gather form is timed next to it).
    With check, every rank also holds the unsharded DB and the merged result must equal its answer bit for bit."""
    import torch

    from direct_stereo_slam_amd.ringdb import Comm, RingKeyDB

    keys, qs = make_ringkey_data(args.rk_n, args.rk_q)
    db = RingKeyDB(ctx, capacity=args.rk_n // world + 16, shard_rank=rank, shard_count=world)
    db.add_points(keys)
    comm = None
    if world > 1:
        import torch.distributed as dist

        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            uid = torch.frombuffer(bytearray(Comm.unique_id()), dtype=torch.uint8).clone()
        dev = "cuda" if _BACKEND == "nccl" else "cpu"
        uid = uid.to(dev)
        dist.broadcast(uid, 0)
        ok, why = 1, ""
        try:
            comm = Comm(ctx, bytes(uid.cpu().numpy().tobytes()), rank, world)
        except Exception as e:  # noqa: BLE001
            ok, why = 0, repr(e)
        # all ranks go on, or none: a rank that left the leg alone would leave the others waiting in a collective
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            db.close()
            return {"error": "communicator creation failed on at least one rank" + (": " + why if why else "")}
    dq = torch.from_numpy(qs).cuda()
    out = torch.empty((args.rk_q, 3), dtype=torch.int64, device="cuda")
    res = {}
    for algo in ("allreduce_min", "allgather"):
        def step():
            db.knn_packed_device(dq.data_ptr(), args.rk_q, out.data_ptr())
            if comm is not None:
                db.merge_topk_device(comm, out.data_ptr(), args.rk_q, algo)
            ctx.sync()

        for _ in range(3):
            step()
        barrier_sync(world)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier_sync(world)
        dt = max_over_ranks(time.perf_counter() - t0, world)
        # the merge alone (collective latency): local results already in place
        if comm is not None:
            barrier_sync(world)
            t1 = time.perf_counter()
            for _ in range(steps):
                db.merge_topk_device(comm, out.data_ptr(), args.rk_q, algo)
                ctx.sync()
            barrier_sync(world)
            dm = max_over_ranks(time.perf_counter() - t1, world)
            db.knn_packed_device(dq.data_ptr(), args.rk_q, out.data_ptr())
            db.merge_topk_device(comm, out.data_ptr(), args.rk_q, algo)
            ctx.sync()
        else:
            dm = 0.0
        res[algo] = {"queries_per_s": args.rk_q * steps / dt, "ms_per_step": 1e3 * dt / steps,
                     "merge_us_per_call": 1e6 * dm / steps,
                     "collectives_per_call": (3 if algo == "allreduce_min" else 1) if comm is not None else 0,
                     "us_per_collective_round": 1e6 * dm / steps / (3 if algo == "allreduce_min" else 1) if comm is not None else 0.0}
        if check:
            full = RingKeyDB(ctx, capacity=args.rk_n + 16)
            full.add_points(keys)
            ref = torch.empty_like(out)
            full.knn_packed_device(dq.data_ptr(), args.rk_q, ref.data_ptr())
            ctx.sync()
            res[algo]["matches_unsharded"] = bool(torch.equal(ref, out))
            full.close()
        if comm is None:
            break
    if comm is not None:
        comm.close()
    db.close()
    return {"workload": f"ring-key DB N={args.rk_n} x 20 float split ordinal mod {world}, Q={args.rk_q} queries per step, k=3, thres 0.1",
            "shards": world, "merge": "dsm_ringdb_merge_topk (C ABI, librccl)" if world > 1 else "one shard: no merge", **res}


# ------------------------------------------------------------------------------------------------------------------
# replay mode: ONE sequence, one frame in flight, driven from the C++ adaptors (tools/replay/replay_bench.cpp)
# ------------------------------------------------------------------------------------------------------------------
REPLAY_CONFIGS = {
    # name: (w, h, levels, K, T_stereo, label)  -- BASELINE configs[1] and [2]
    "kitti00": (1232, 368, 5, None, None, "KITTI-00 shape 1232x368 (cams/kitti/0_2), 5 levels"),
    "tiny": (308, 92, 3, (179.714, 179.714, 151.67, 45.18), None, "quarter-size test geometry 308x92, 3 levels (tests/test_replay_bench.py)"),
    "malaga06": (1024, 768, 5, (795.11588, 795.11588, 517.12973, 395.59665),
                 np.array([[1, 0, 0, -0.119471], [0, 1, 0, 0], [0, 0, 1, 0.000000001], [0, 0, 0, 1]], np.float64),
                 "Malaga urban extract 06 shape 1024x768 (cams/malaga), 5 levels"),
}


def write_replay_pack(path, name, n_frames, kf_every, n_active, seed=0x5EED0404):
    """A synthetic stereo sequence for tools/replay/replay_bench.cpp: the camera travels ALONG the relief scene (constant distance,
    gentle yaw), mono8 images, per keyframe the right image and `n_active` active points (random interior pixels with the scene's
    inverse depth there and HdiF-style weights: what the window's PointHessians hold, TrackerAndScaler.cpp:149-164)."""
    import struct
    from concurrent.futures import ThreadPoolExecutor

    from direct_stereo_slam_amd import synth as S

    w, h, nl, K, T, _ = REPLAY_CONFIGS[name]
    K = S.kitti_K_work() if K is None else K
    T = S.KITTI_T_STEREO if T is None else T
    scene = S.ReliefScene(seed=seed, fx_ref=K[0])
    rng = np.random.default_rng(seed)
    poses, c, R = [], np.zeros(3), np.eye(3)
    for i in range(n_frames):  # x_cam = R (x_w - c)
        poses.append((R.copy(), -R @ c))
        step = scene.e1 * 0.10 + scene.e2 * 0.02 * np.sin(i / 9.0) + scene.n * 0.01 * np.cos(i / 7.0) + rng.normal(0, 0.003, 3)
        c = c + step
        R = S.so3_exp(np.array([0.0005 * np.sin(i / 11.0), 0.003 * np.sin(i / 13.0), 0.0004]) + rng.normal(0, 0.0003, 3)) @ R
    u8 = lambda im: np.clip(np.rint(im), 0, 255).astype(np.uint8)

    def frame(i):
        R, t = poses[i]
        r = np.random.default_rng(seed + 7919 * i)
        out = [S.pose_from_Rt(R, t), u8(scene.render(K, w, h, R, t, noise=1.0, rng=r))]
        if i % kf_every == 0:
            Rr, tr = T[:3, :3] @ R, T[:3, :3] @ t + T[:3, 3]
            out.append(u8(scene.render(K, w, h, Rr, tr, noise=1.0, rng=r)))
            pu = r.integers(3, w - 3, n_active).astype(np.float32)
            pv = r.integers(3, h - 3, n_active).astype(np.float32)
            idl = scene.idepth(K, w, h, R, t)
            out += [pu, pv, idl[pv.astype(int), pu.astype(int)].astype(np.float32), np.sqrt(1e-3 / (r.uniform(1e-3, 10, n_active) + 1e-12)).astype(np.float32)]
        return out

    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 4)) as ex:
        frames = list(ex.map(frame, range(n_frames)))
    with open(path, "wb") as f:
        f.write(b"DSMRPLY1")
        f.write(struct.pack("<6i", w, h, nl, n_frames, kf_every, n_active))
        f.write(np.asarray(K, np.float32).tobytes())
        f.write(np.asarray(T, np.float64).tobytes())
        f.write(struct.pack("<d", 40.0))  # lidar_range (main.cpp:307)
        for fr in frames:
            for a in fr:
                f.write(np.ascontiguousarray(a).tobytes())


def build_replay_bench():
    """tools/replay/replay_bench.cpp -> tools/replay/_build/replay_bench (g++; links the product library and the oracle's timing build)"""
    exe = os.path.join(ROOT, "tools", "replay", "_build", "replay_bench")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native"])
    src = os.path.join(ROOT, "tools", "replay", "replay_bench.cpp")
    lib, orc = os.path.join(ROOT, "direct_stereo_slam_amd", "lib"), os.path.join(ROOT, "oracle", "_build")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", exe, src, "-L" + lib, "-ldsm_hotpath", "-L" + orc, "-l:libdsm_oracle_native.so",
                           "-Wl,-rpath," + lib, "-Wl,-rpath," + orc])
    return exe


def replay_leg(args, names=("kitti00", "malaga06")):
    """config.replay of the default line: the sequence a drop-in user runs -- per-stage mean ms under the
    reference's own names (main.cpp:181-201) for the GPU path through the C++ adaptors and for the CPU path, side by side."""
    import tempfile

    exe = build_replay_bench()
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for name in names:
            pack = os.path.join(td, name + ".bin")
            t0 = time.perf_counter()
            write_replay_pack(pack, name, args.replay_frames, args.replay_kf_every, args.replay_active)
            t_pack = time.perf_counter() - t0
            prefix = os.path.join(ROOT, "gpurun_out", "replay_" + name) if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else os.path.join(td, name)
            conc = args.replay_concurrent if name == names[0] else 0  # (the concurrent-sequences leg on the first shape only)
            p = subprocess.run([exe, pack, prefix, "both", str(conc), "1"], capture_output=True, text=True, timeout=1200)
            if p.returncode != 0:
                out[name] = {"error": (p.stderr or p.stdout)[-600:]}
                continue
            d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
            d["label"] = REPLAY_CONFIGS[name][5]
            d["sequence_rendering_s"] = round(t_pack, 1)
            out[name] = d
    return out


def loop_chain_leg(ctx, seqs=64, n_pts=16000, reps=20):
    """config.loop_chain: the per-keyframe loop chain -- generate_spherical_points, ScanContext::generate,
    search_ringkey (LoopHandler.cpp:186,236,247) -- through dsm_loop_detect_batch: ONE enqueue and ONE read-back per call, the ring keys
    handed to the index's k-NN on the device.  Timed for one keyframe per call (one sequence) and for `seqs` keyframes per call (one per
    concurrent sequence, BASELINE configs[4]); the unfused pair of calls of round 4 beside it.  Clouds: 8 keyframes x 2000 points."""
    from direct_stereo_slam_amd.ringdb import RingKeyDB, loop_descriptors_batch

    def job(seed):
        rng = np.random.default_rng(seed)
        n_kf = 8
        kf_ids = np.arange(100, 100 + n_kf)
        poses = np.hstack([rng.normal(0, 5, (n_kf, 3)), rng.normal(0, 0.08, (n_kf, 3))])
        cur_cw = np.hstack([np.eye(3), rng.normal(0, 1, (3, 1))])
        pt_kf = rng.choice(kf_ids, n_pts)
        g = np.stack([rng.uniform(-55, 55, n_pts), 1.6 + rng.normal(0, 0.05, n_pts), rng.uniform(-55, 55, n_pts)], 1)
        walls = rng.random(n_pts) < 0.35
        g[walls, 0] = np.where(rng.random(walls.sum()) < 0.5, 8.0, -9.0)
        g[walls, 1] = rng.uniform(-5, 1.6, walls.sum())
        return kf_ids, poses, cur_cw, pt_kf, g - cur_cw[:, 3]

    jobs = [job(900 + s) for s in range(seqs)]
    out = {"workload": f"{seqs} concurrent sequences marginalise a keyframe each: clouds of {n_pts} points (8 keyframes x {n_pts // 8}), lidar_range 40, 60 x 20 polar bins, "
                       "k = 3 ring-key search in one index"}

    def timed(fn, n):
        fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        return 1e3 * (time.perf_counter() - t0) / n

    # (the job tables are built once, as a node keeps its window's clouds: the C calls alone are timed)
    from direct_stereo_slam_amd.ringdb import LoopBatch

    db = RingKeyDB(ctx, capacity=1 << 16)
    one, many = LoopBatch(ctx, jobs[:1], 40.0, db=db, selected_points=False), LoopBatch(ctx, jobs, 40.0, db=db, selected_points=False)
    out["one_keyframe_per_call_ms"] = timed(one.run, reps * 4)
    out[f"{seqs}_keyframes_per_call_ms_per_keyframe"] = timed(many.run, reps) / seqs
    # the same with the clouds kept in page-locked memory (dsm_host_alloc): the device reads them where they are, no host copy
    many_p = LoopBatch(ctx, jobs, 40.0, db=RingKeyDB(ctx, capacity=1 << 16), selected_points=False, pinned_clouds=True)
    out[f"{seqs}_keyframes_per_call_ms_per_keyframe_pinned_clouds"] = timed(many_p.run, reps) / seqs
    db2 = RingKeyDB(ctx, capacity=1 << 16)
    desc = LoopBatch(ctx, jobs[:1], 40.0)

    def unfused():
        desc.run()
        db2.search_ringkey(desc.outs[0][0]["ringkey"])

    out["one_keyframe_descriptor_call_then_search_call_ms"] = timed(unfused, reps * 4)
    db.close()
    db2.close()
    return out


VALU_PEAK_TOPS = 78.6  # MI355X_MICROARCH.md: 157.3 TFLOP/s FP32 vector counts an FMA as two; unfused add / mul / sub issue at half of that


def ringkey_single_gpu_leg(args, ctx, seconds=0.25):
    """config.ringkey of the N = 1 line: search_ringkey's k-NN (search_place.h:25-57) on ONE GPU at SURVEY.md
    8(d)'s sizes, each with its own roofline -- HBM for the few-query scan (one sweep of the 80-byte keys), the vector pipes for
    the many-query kernel (FLANN's L2 functor: 59 unfused operations per (query, key) pair) --, an in-run bit-exact check
    against the oracle's brute force and a brute-force CPU baseline on the same keys (one core, bounded sample)."""
    import torch

    from direct_stereo_slam_amd.ringdb import RingKeyDB, unpack
    from oracle import oracle as O

    cases = []
    for n, q in ((10_000, 1024), (1_000_000, 1024), (10_000_000, 1024), (10_000_000, 1)):
        keys, qs = make_ringkey_data(n, q)
        db = RingKeyDB(ctx, capacity=n + 16)
        db.add_points(keys)
        dq = torch.from_numpy(qs).cuda()
        out = torch.empty((q, 3), dtype=torch.int64, device="cuda")

        def call():
            db.knn_packed_device(dq.data_ptr(), q, out.data_ptr())
            ctx.sync()

        for _ in range(3):
            call()
        t0 = time.perf_counter()
        call()
        one = time.perf_counter() - t0
        reps = max(3, min(2000, int(seconds / max(one, 1e-6))))
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        dt = (time.perf_counter() - t0) / reps
        got_d, got_i = unpack(out.cpu().numpy())
        # oracle: brute force over the same keys (FLANN L2 accumulation order), a bounded number of the queries
        orc = O.OracleRingDB(native=True)
        orc.add_points(keys)
        n_chk = max(1, min(q, int(2e8 // n)))
        ok = True
        t0 = time.perf_counter()
        ref = [orc.knn(qs[i]) for i in range(n_chk)]
        cpu_dt = time.perf_counter() - t0
        for i, (io, do) in enumerate(ref):
            exp = [(np.float32(d), j) for d, j in zip(do, io) if j >= 0 and d < 0.1]
            got = [(np.float32(d), int(j)) for d, j in zip(got_d[i], got_i[i]) if j >= 0]
            ok = ok and got == exp
        pairs = float(n) * q
        if q <= 32:
            groups = (q + 7) // 8 if q > 8 else 1
            by = 80.0 * n * groups + 104.0 * q
            roof = {"bound": "hbm", "achieved": by / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / dt / 1e9 / HBM_PEAK_GBS,
                    "kernel": "ringkey_knn_fewq4_kernel" if q <= 2 else "ringkey_knn_fewq_kernel", "bytes_per_call": by}
        else:
            ops = 59.0 * pairs
            roof = {"bound": "valu", "achieved": ops / dt / 1e12, "peak": VALU_PEAK_TOPS, "unit": "Tops/s (unfused FP32 vector operations: 20 sub + 20 mul + 19 add per pair, FLANN's order)",
                    "frac": ops / dt / 1e12 / VALU_PEAK_TOPS, "kernel": "ringkey_knn_kernel", "hbm_GBps": (80.0 * n + 104.0 * q) / dt / 1e9}
        cases.append({"N": n, "Q": q, "us_per_call": 1e6 * dt, "queries_per_s": q / dt, "pairs_per_s": pairs / dt, "roofline": roof,
                      "matches_oracle_bit_exact": bool(ok), "queries_checked": n_chk,
                      "cpu_baseline": {"value": n_chk / cpu_dt, "unit": "queries/s", "cores": 1, "kind": "port",
                                       "sample": f"{n_chk} of the {q} queries, brute force over the same {n} keys (oracle/dsm_oracle.c orc_ringdb_knn, gcc -O3 -march=native), {cpu_dt:.2f} s"}})
        db.close()
        del orc, dq, out
    return {"workload": "search_ringkey k-NN (k = 3, threshold 0.1) over N x 20 float keys, Q queries per call, one GPU, keys and queries resident",
            "cases": cases}


LINE_LIMIT = 4096  # bytes: the driver keeps a bounded tail of stdout; the one line it parses must fit with room to spare


def _r(x, nd=3):
    """numbers of the compact line: enough digits to reproduce a figure, no more"""
    if isinstance(x, float):
        return float(f"{x:.{nd + 2}g}") if abs(x) >= 1 else round(x, nd + 2)
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


def _pick(d, keys, nd=3):
    return {k: _r(d[k], nd) for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(res, detail_path):
    """The ONE stdout line (round 4's line had grown to 20.7 KB and the driver's bounded tail could not parse
    it).  Every key of the bench contract, `roofline` and `cpu_baseline` as numbers, one short summary per extra leg; everything
    else -- prose, per-level tables, the legs' own objects -- goes to `detail_path` (the FULL object, named in config.detail)."""
    c, rf, cb = res["config"], res.get("roofline") or {}, res.get("cpu_baseline")
    out = {k: res[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                               "vs_baseline", "dtype", "data")}
    out["value"], out["ms_per_step"] = _r(res["value"], 5), _r(res["ms_per_step"], 5)
    # (size, levels, scene family, distinct frames and the initial guess are spelt out in `workload`; everything else is in the detail file)
    cfg = _pick(c, ("workload", "n0", "replicas", "inputs", "frames_in_flight_per_gpu", "fixed_schedule", "evals_per_frame_by_level", "algorithmic_MB_per_frame",
                    "whole_step_GBps", "all_tracked", "frames_with_translation_error_above_1cm"), 3)
    st = c.get("stream")
    if isinstance(st, dict):
        cfg["form"] = "dsm_stream_* " + str(st.get("engine")) + " engine"
        tpa = st.get("ticks_per_advance")
        cfg["stream"] = {**_pick(st, ("track_slots", "scale_slots", "advances_in_timed_region", "host_share_of_timed_region"), 3),
                         "ticks_per_advance": tpa if not isinstance(tpa, str) else "auto",
                         "steady_state_frames_per_s": _r((st.get("steady_state") or {}).get("frames_per_s"), 4)}
    else:
        cfg["form"] = "one synchronous dsm_track_and_scale_batch call per step"
    legs = {}
    fx = c.get("fixed_schedule_leg")
    if isinstance(fx, dict):
        legs["fixed_schedule_1p3"] = {"error": fx["error"][:120]} if "error" in fx else {
            **_pick(fx, ("value", "ms_per_step", "algorithmic_MB_per_frame")), "frac_whole_step": _r((fx.get("roofline") or {}).get("frac_whole_step")),
            "cpu_value": _r((fx.get("cpu_baseline") or {}).get("value"))}
    fv = c.get("reference_five_level")
    if isinstance(fv, dict):
        legs["reference_five_level_S1"] = {"error": fv["error"][:120]} if "error" in fv else {
            **_pick(fv, ("value", "ms_per_step", "algorithmic_MB_per_frame")), **_pick(fv.get("roofline") or {}, ("frac", "frac_whole_step"))}
    pf = c.get("plane_family")
    if isinstance(pf, dict):
        legs["plane"] = {"error": pf["error"][:120]} if "error" in pf else {
            **_pick(pf, ("value", "ms_per_step", "algorithmic_MB_per_frame")), **_pick(pf.get("roofline") or {}, ("frac", "frac_whole_step"))}
    wu = c.get("with_upload")
    if isinstance(wu, dict):
        legs["with_upload"] = {"error": wu["error"][:120]} if "error" in wu else {
            **_pick(wu, ("value", "ms_per_step")), "form": "batch, u8 pinned double-buffered", "host_MB_per_step": _r(wu.get("host_bytes_per_step", 0) / 1e6, 4),
            **({"stream_form": ({"error": wu["stream_form"]["error"][:100]} if "error" in wu["stream_form"] else
                                _pick(wu["stream_form"], ("value", "sequences", "host_GBps", "all_tracked"), 4))} if isinstance(wu.get("stream_form"), dict) else {})}
    rp = c.get("replay")
    if isinstance(rp, dict):
        lr = {}
        for name, d in rp.items():
            if not isinstance(d, dict) or "error" in d or "gpu" not in d:
                lr[name] = {"error": str((d or {}).get("error", d))[:120]} if isinstance(d, dict) else str(d)[:120]
                continue
            g, cp, vs = d["gpu"]["stages_mean_ms"], d["cpu"]["stages_mean_ms"], d.get("gpu_vs_cpu", {})
            # (ratios: GPU path over CPU path; same_candidates: loop queries whose candidate lists are identical on both paths)
            lr[name] = {"ms_per_frame_gpu": _r(g["per_frame"]["mean_ms"], 4), "ms_per_frame_cpu": _r(cp["per_frame"]["mean_ms"], 4),
                        "trackNewCoarse_ms_gpu": _r(g["trackNewCoarse"]["mean_ms"], 4), "ate_ratio": _r(vs.get("ate_ratio_gpu_over_cpu"), 5),
                        "rpe_trans_ratio": _r(vs.get("rpe_trans_ratio_gpu_over_cpu"), 5), "rpe_rot_ratio": _r(vs.get("rpe_rot_ratio_gpu_over_cpu"), 5),
                        "loop_queries": vs.get("loop_queries"), "same_candidates": vs.get("queries_with_identical_candidates")}
            cc = d.get("concurrent")
            if isinstance(cc, dict):
                lr[name]["concurrent"] = {"sequences": cc.get("sequences"), "frames_per_s": _r(cc.get("frames_per_s"), 5),
                                          "max_diff_vs_one_sequence_m": cc.get("max_abs_trajectory_diff_vs_the_one_sequence_run_m")}
        legs["replay"] = lr
    rk = c.get("ringkey")
    if isinstance(rk, dict):
        legs["ringkey"] = {"error": rk["error"][:120]} if "error" in rk else [
            {"N": k["N"], "Q": k["Q"], "us": _r(k["us_per_call"], 4), "bound": k["roofline"]["bound"], "frac": _r(k["roofline"]["frac"], 3),
             "bit_exact": k["matches_oracle_bit_exact"]} for k in rk.get("cases", [])]
    lc = c.get("loop_chain")
    if isinstance(lc, dict):
        legs["loop_chain_ms_per_keyframe"] = {"error": lc["error"][:120]} if "error" in lc else {
            ("S1" if k.startswith("one_keyframe_per") else f"S{k.split('_', 1)[0]}_pinned" if k.endswith("pinned_clouds") else f"S{k.split('_', 1)[0]}" if "keyframes_per_call" in k
             else "S1_unfused"): _r(v, 3)
            for k, v in lc.items() if k != "workload"}  # (labels follow the leg's own sequence count: ADVICE r05)
    sh = c.get("ringkey_sharded")
    if isinstance(sh, dict):
        legs["ringkey_sharded"] = {"error": sh["error"][:160]} if "error" in sh else {
            "shards": sh.get("shards"), **{a: _pick(sh[a], ("queries_per_s", "ms_per_step", "us_per_collective_round", "matches_unsharded")) for a in ("allreduce_min", "allgather") if a in sh}}
    if legs:
        cfg["legs"] = legs
    cfg["detail"] = detail_path
    out["config"] = cfg
    ro = _pick(rf, ("bound", "achieved", "peak", "unit", "frac", "frac_whole_step", "frac_full_evals", "frac_hbm_actual", "bytes_per_launch",
                    "avg_launch_us", "kernel_busy_us_per_launch", "launches", "dispatches", "stream_groups", "evals_by_level", "bytes_per_eval_by_level",
                    "evals", "residual_only_evals"), 4)
    ro["traffic"] = _r(rf.get("traffic"), 5)
    ro["kernel"] = str(rf.get("kernel", "")).split(" (")[0]
    ro["traffic_source"] = str(rf.get("traffic_source", "")).split(":")[0][:100]
    out["roofline"] = ro
    if cb is None:
        out["cpu_baseline"] = None
    else:
        o = _pick(cb, ("value", "unit", "cores", "kind", "cpu_model", "physical_cores"))
        o["sample"] = str(cb.get("sample", "")).split(", same initial")[0][:110] + "; " + str(cb.get("sample", "")).rsplit(", ", 1)[-1][:40]
        ate, lm, ac = cb.get("ate_vs_cpu_ref"), cb.get("lm_routes_vs_cpu_ref"), cb.get("all_cores")
        if isinstance(ate, dict):
            o["ate_vs_cpu_ref"] = _pick(ate, ("frames", "ate_gpu_m", "ate_cpu_m", "ate_ratio_gpu_over_cpu",
                                              "max_abs_translation_diff_gpu_vs_cpu_m", "good_flags_equal"), 5)
        rp = cb.get("rpe_vs_cpu_ref")
        if isinstance(rp, dict):
            o["rpe_vs_cpu_ref"] = _pick(rp, ("rpe_trans_ratio_gpu_over_cpu", "rpe_rot_ratio_gpu_over_cpu", "rpe_trans_gpu_m", "rpe_rot_gpu_deg"), 5)
        if isinstance(lm, dict):
            o["lm_routes_vs_cpu_ref"] = _pick(lm, ("frames", "same_evaluation_counts", "different_evaluation_counts"))
        if isinstance(ac, dict):
            o["all_cores"] = _pick(ac, ("value", "cores")) if "error" not in ac else {"error": ac["error"][:100]}
        out["cpu_baseline"] = o
    text = json.dumps(out, separators=(",", ":"))
    for drop in ("legs", "stream", "evals_per_frame_by_level"):  # never reached with the fields above (about 3 KB); a guard, not a plan
        if len(text.encode()) < LINE_LIMIT:
            break
        out["config"].pop(drop, None)
        text = json.dumps(out, separators=(",", ":"))
    if len(text.encode()) >= LINE_LIMIT:
        # last resort (a long error string, a new leg): the contract keys alone plus where the rest lies -- never an exception, the line
        # is what the driver reads (the assert here lost the whole line, and vanished under python -O)
        out["config"] = {"workload": str(cfg.get("workload", ""))[:300], "detail": detail_path, "truncated": True}
        out["roofline"] = _pick(rf, ("bound", "achieved", "peak", "unit", "frac", "traffic"), 4)
        out["cpu_baseline"] = None if cb is None else _pick(cb, ("value", "unit", "cores", "kind"))
        if out["cpu_baseline"] is not None:
            out["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:160]
        text = json.dumps(out, separators=(",", ":"))
    return text


def write_detail(res, path):
    """the full object (every leg, every table, the prose) as a side file; returns the path as the line names it"""
    full = path if os.path.isabs(path) else os.path.join(ROOT, path)
    os.makedirs(os.path.dirname(full), exist_ok=True)
    with open(full, "w") as f:
        json.dump(res, f, indent=1)
        f.write("\n")
    return path


def line_guard(res, detail_path="gpurun_out/bench_detail.json"):
    """The sharded ring-key leg is this bench's first contact with several RCCL ranks on a box; a process that dies inside a
    native library (SIGSEGV, abort) prints nothing.  Before the leg, rank 0 parks its finished bench line with a forked
    helper that touches neither the GPU nor torch: it waits on a pipe and, if rank 0 goes away without calling the returned
    function, prints the line (with the leg marked as lost) to the inherited stdout -- still exactly ONE JSON line."""
    import copy

    parked = copy.deepcopy(res)
    parked["config"]["ringkey_sharded"] = {"error": "rank 0 ended inside the sharded ring-key leg (the bench line is the one measured before it)"}
    text = (compact_line(parked, detail_path) + "\n").encode()
    # (os.pipe() descriptors are non-inheritable across exec (PEP 446), and the guard is made after every other child of this
    # process -- the CPU legs' forked workers -- has come and gone: nobody else holds the write end)
    r, w = os.pipe()
    sys.stdout.flush()
    pid = os.fork()
    if pid == 0:  # helper: plain system calls only
        try:
            os.close(w)
            done = os.read(r, 1)  # b"" = the writer is gone without a word
            if not done:
                os.write(1, text)
        finally:
            os._exit(0)
    os.close(r)

    def release():
        try:
            os.write(w, b"1")
            os.close(w)
            os.waitpid(pid, 0)
        except OSError:
            pass

    return release


def run_with_deadline(fn, seconds, device):
    """fn() on a worker thread bound to this rank's device; returns (result, still_running).  Exceptions and a missed
    deadline become {"error": ...}."""
    import threading

    box = {}

    def target():
        try:
            import torch

            if torch.cuda.is_available():
                torch.cuda.set_device(device)  # the current device is per thread
            box["res"] = fn()
        except BaseException as e:  # noqa: BLE001
            box["res"] = {"error": repr(e)}

    th = threading.Thread(target=target, daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        return {"error": f"no result within {seconds:.0f} s (this rank left the leg; the bench line above it is unaffected)"}, True
    return box["res"], False


# ------------------------------------------------------------------------------------------------------------------
def bench_tracking(args):
    from direct_stereo_slam_amd.tracker import Context

    rank, local, world = dist_setup(args)
    ctx = Context(local)
    ctx.set_streams(args.streams)
    wl = build_workload(args, ctx, args.config)
    m = measure(args, ctx, wl, args.steps, args.warmup, world, args.with_upload)
    res = {
        "metric": baseline_metric(),
        "value": m["value"],
        "unit": "stereo frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": m["ms_per_step"],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": workload_label(args, wl, m["n0"]), "name": wl["config"], "w": wl["w"], "h": wl["h"], "levels": wl["nl"], "n0": m["n0"],
                   "replicas": world,
                   "inputs": "host images uploaded and pyramids built inside the timed region (secondary figure)" if args.with_upload else "resident in HBM",
                   **m["detail"]},
        "roofline": m["roofline"],
    }
    if rank == 0 and world == 1 and not args.no_cpu:
        res["cpu_baseline"] = cpu_baseline(args, wl, m["poses"], m["good"], m.get("evals"))
    else:
        res["cpu_baseline"] = None
    del wl
    if args.fixed_schedule == 0 and not args.no_fixed_leg and not args.with_upload and not args.evals_only:
        # SURVEY.md 8d's fixed schedule on the same frames: 1 + 3 evaluations per level, every step taken -- bytes per frame do
        # not depend on the input -- with the CPU leg on the same schedule
        try:
            a3 = argparse.Namespace(**vars(args))
            a3.fixed_schedule, a3.cpu_frames, a3.cpu_min_frames, a3.cpu_seconds, a3.no_cpu_all_cores = 3, 64, 32, 5.0, True
            wl3 = build_workload(a3, ctx, args.config)
            m3 = measure(a3, ctx, wl3, args.second_leg_steps, 1, world)
            leg = {"schedule": "exactly 1 initial + 3 LM evaluations per level and problem, every step taken (dsm_params.fixed_schedule = 3)",
                   "value": m3["value"], "unit": "stereo frames/s", "steps": args.second_leg_steps, "ms_per_step": m3["ms_per_step"],
                   "algorithmic_MB_per_frame": m3["detail"]["algorithmic_MB_per_frame"], "whole_step_GBps": m3["detail"]["whole_step_GBps"],
                   "evals_per_frame_by_level": m3["detail"]["evals_per_frame_by_level"], "launch_pairs_per_step": m3["detail"]["launch_pairs_per_step"],
                   "roofline": m3["roofline"]}
            if rank == 0 and world == 1 and not args.no_cpu:
                leg["cpu_baseline"] = cpu_baseline(a3, wl3, m3["poses"], m3["good"], m3.get("evals"))
            res["config"]["fixed_schedule_leg"] = leg
            del wl3
        except Exception as e:  # a reported extra, never a reason to lose the bench line
            res["config"]["fixed_schedule_leg"] = {"error": repr(e)}
    if args.config == "S2" and not args.no_second_leg and not args.with_upload:
        # the reference-faithful five-level workload (what DSO's level rule yields for the KITTI crop), same scenes, same rules
        try:
            a2 = argparse.Namespace(**vars(args))
            a2.cpu_frames = 0
            wl2 = build_workload(a2, ctx, "S1")
            m2 = measure(a2, ctx, wl2, args.second_leg_steps, 1, world)
            res["config"]["reference_five_level"] = {"workload": workload_label(a2, wl2, m2["n0"]), "value": m2["value"], "unit": "stereo frames/s",
                                                     "steps": args.second_leg_steps, "ms_per_step": m2["ms_per_step"], "roofline": m2["roofline"], **m2["detail"]}
            del wl2
        except Exception as e:  # a reported extra, never a reason to lose the bench line
            res["config"]["reference_five_level"] = {"error": repr(e)}
    if args.scene_family == "relief" and not args.no_plane_leg and not args.with_upload and not args.evals_only and args.fixed_schedule == 0:
        # SURVEY.md 8d's literal scene family (PlaneScene: eight sinusoids of 8 ... 128 px on one plane -- rounds 1-3's workload; BASELINE.md
        # section 2 says how the default relief family departs from it), same configuration, same rules: VERDICT r05 item 4
        try:
            a4 = argparse.Namespace(**vars(args))
            a4.scene_family, a4.cpu_frames = "plane", 0
            wl4 = build_workload(a4, ctx, args.config)
            m4 = measure(a4, ctx, wl4, args.second_leg_steps, 1, world)
            res["config"]["plane_family"] = {"workload": workload_label(a4, wl4, m4["n0"]), "value": m4["value"], "unit": "stereo frames/s",
                                             "steps": args.second_leg_steps, "ms_per_step": m4["ms_per_step"], "roofline": m4["roofline"], **m4["detail"]}
            del wl4
        except Exception as e:  # a reported extra, never a reason to lose the bench line
            res["config"]["plane_family"] = {"error": repr(e)}
    if not args.no_upload_leg and not args.with_upload and not args.evals_only and args.fixed_schedule == 0:
        # The PCIe-inclusive figure in the driver's line (never `value`): every step hands the B new left images and
        # the keyframes' right images over as camera bytes in page-locked host buffers -- double-buffered, the images of step i + 1 travel
        # while step i is tracked -- and builds their pyramids on the device (row N1) inside the timed region; one synchronous
        # dsm_track_and_scale_batch call per step (the batch form: a step's frames are all back before the next step swaps its images in,
        # which the streamed form's standing backlog does not guarantee).  128 distinct frames cycled over the B trackers.
        try:
            a5 = argparse.Namespace(**vars(args))
            a5.with_upload, a5.u8, a5.pinned, a5.overlap, a5.stream, a5.cpu_frames, a5.scenes = True, True, True, True, 0, 0, min(128, args.batch)
            if args.streams_given is None:
                a5.streams = 2
            ctx.set_streams(a5.streams)
            wl5 = build_workload(a5, ctx, args.config)
            m5 = measure(a5, ctx, wl5, args.second_leg_steps, 2, world, True)
            px = wl5["w"] * wl5["h"]
            res["config"]["with_upload"] = {"what": "host images (mono8, page-locked) handed over and pyramids built inside the timed region, double-buffered; batch form",
                                            "value": m5["value"], "unit": "stereo frames/s", "steps": args.second_leg_steps, "ms_per_step": m5["ms_per_step"],
                                            "host_bytes_per_step": int(px * (len(wl5["trackers"]) + len(range(0, len(wl5["trackers"]), args.kf_every)))),
                                            "roofline": m5["roofline"], **m5["detail"]}
            try:  # the streamed form of the same: every tracker one sequence with one frame in flight (SequenceUploadRunner)
                ctx.set_streams(args.streams)
                a6 = argparse.Namespace(**vars(a5))
                a6.stream, a6.stream_ticks = 1, args.upload_stream_ticks
                res["config"]["with_upload"]["stream_form"] = {
                    "what": "dsm_stream_* with the hand-over inside: one frame in flight per sequence, the next frame travels (page-locked mono8, device pyramids) "
                            "while this one is tracked, swapped in once its predecessor has retired", **measure_stream_with_upload(a6, ctx, wl5, 3 * args.second_leg_steps, 2, world)}
            except Exception as e:
                res["config"]["with_upload"]["stream_form"] = {"error": repr(e)}
            del wl5
        except Exception as e:  # a reported extra, never a reason to lose the bench line
            res["config"]["with_upload"] = {"error": repr(e)}
        ctx.set_streams(args.streams)
    if rank == 0 and world == 1 and not args.no_replay_leg and not args.with_upload and not args.no_cpu:
        try:
            res["config"]["replay"] = replay_leg(args)
        except Exception as e:  # a reported extra, never a reason to lose the bench line
            res["config"]["replay"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_ringkey_leg and not args.with_upload:
        try:
            res["config"]["ringkey"] = ringkey_single_gpu_leg(args, ctx)
        except Exception as e:  # a reported extra, never a reason to lose the bench line
            res["config"]["ringkey"] = {"error": repr(e)}
        try:
            res["config"]["loop_chain"] = loop_chain_leg(ctx)
        except Exception as e:  # a reported extra, never a reason to lose the bench line
            res["config"]["loop_chain"] = {"error": repr(e)}
    stuck = False
    if world > 1 and not args.no_ringkey_leg and args.device_override < 0:  # (RCCL refuses two ranks on one device)
        # A reported extra must never cost the bench line: the leg runs under a deadline (a collective that one rank never
        # enters would otherwise hold every rank until the driver's own limit), and a rank whose leg is stuck prints and leaves.
        guard = line_guard(res, args.detail_out) if rank == 0 else None
        res["config"]["ringkey_sharded"], stuck = run_with_deadline(lambda: ringkey_sharded_leg(args, ctx, rank, world), args.ringkey_leg_seconds, local)
        if guard is not None:
            guard()
    if rank == 0:
        write_detail(res, args.detail_out)
        sys.stderr.flush()
        print(compact_line(res, args.detail_out), flush=True)
    if stuck:
        os._exit(0)  # the worker thread still sits in a collective: no orderly teardown possible
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


def bench_ringkey(args):
    """secondary: the sharded ring-key DB alone; value = queries/s (scan + merge)"""
    from direct_stereo_slam_amd.tracker import Context

    rank, local, world = dist_setup(args)
    ctx = Context(local)
    leg = ringkey_sharded_leg(args, ctx, rank, world, steps=args.steps, check=True)
    if rank == 0:
        best = leg["allreduce_min"]
        groups = (args.rk_q + 7) // 8 if 8 < args.rk_q <= 32 else 1  # the few-query kernel sweeps the keys once per query group
        sweep = 80 * args.rk_n * groups + 104 * args.rk_q
        gbps = sweep / (best["ms_per_step"] * 1e-3) / 1e9
        print(json.dumps({"metric": "ring-key k=3 queries/s over a sharded DB", "value": best["queries_per_s"],
                          "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": 3,
                          "ms_per_step": best["ms_per_step"], "higher_is_better": True, "scaling": "strong",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {**leg, "db_sweep_GBps": gbps},
                          # HBM roofline of the scan for few queries (one sweep of the 80-byte keys per group of <= 8 queries);
                          # with many queries per key the scan is VALU-bound and this figure is only informative
                          "roofline": {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": gbps / HBM_PEAK_GBS, "traffic": None,
                                       "kernel": "ringkey_knn_fewq_kernel" if args.rk_q <= 32 else "ringkey_knn_kernel (VALU-bound)"}}))
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    a = parse()
    rc = spawn_ranks_if_needed(a)
    if rc is not None:
        sys.exit(rc)
    if a.membw:
        from direct_stereo_slam_amd.tracker import Context

        c = Context(0)
        print(json.dumps({
            "read_bandwidth_GBps_grid_stride": {f"{mb}MiB": round(c.read_bandwidth(mb << 20, 10), 1) for mb in (512, 1024, 4096)},
            # one contiguous chunk per workgroup -- the access pattern of the eval kernels (112 KiB ~ one level-0 chunk:
            # 4096 template points + the image rows they land on)
            "read_bandwidth_GBps_chunk_per_workgroup": {f"{kb}KiB": round(c.read_bandwidth_chunked(3 << 30, kb << 10, 5), 1) for kb in (16, 112, 1024)}}))
        sys.exit(0)
    if a.replay:
        print(json.dumps({"replay": replay_leg(a)}))
    elif a.ringkey:
        bench_ringkey(a)
    else:
        bench_tracking(a)

#!/usr/bin/env python
"""bench.py -- guided Msamples/s (+ image MSE vs ground truth) of the MI355X-native path-guiding integrator.

Default workload = the one BASELINE.json's metric is quoted on: veach-ajar 1920x1080, the scene
file's max_depth 13, the 2^(k+2) spp schedule (scenes/veach-ajar/scene.xml, main.py:170).  Phases:

  train   iterations 0 .. train_iters-1 are really rendered (4, 8, 16 ... spp; accumulators summed
          over the ranks and the SD-tree refined after each) and timed per iteration ->
          `value_full_schedule`: guided paths (iterations >= 2) per second INCLUDING exchange + refine.
  steps   a *step* is --spp-per-pass (16) consecutive ONE-sample training passes of iteration train_iters over the whole
          film -- the reference's own schedule: main.py:192 renders training passes with spp 1, :218 seeds pass p with
          initial_seed + cumm_spp -- traced as ONE wavefront (pg_pass_params.batched: bit-identical to the 16 separate
          passes, tests/test_gpu_render.py::test_batched_launch_equals_separate_one_sample_passes): camera rays,
          max_depth bounces with NEE and BSDF/SD-tree one-sample MIS, the record list, then record post-processing and
          the splat into sdTree_current -- PathGuidingIntegrator.sample() of the reference, whole.  `value` = paths per
          second over K steps, wall clock between barriers, max over ranks.  Everything is resident in HBM.
          config.value_one_launch_per_1spp_pass = the same passes launched one by one; config.value_one_16spp_pass_per_step
          = one pass of 16 samples per pixel (mi.render(spp=16): other sampler streams, the same work).
  N > 1   the film is sharded: rank r traces bands of 4 rows dealt round-robin (pg_pass_params
          stripes), no data-path collective inside a step; one int64 all-reduce of the accumulators
          per iteration (RCCL).  The film is fixed, so `scaling` is "strong".  `--shard passes` is the
          weak-scaling form (every rank traces the whole film with its own seeds).
  mse     MSE of the last trained iteration's image against the reference's ground truth
          (path_guiding_integrator.py:503-517; teapot pixels masked, both images box-filtered to
          640x360), and -- at N = 1 -- the same schedule on a 320x180 film on the device and on the
          CPU oracle: equal spp, equal seeds, the two MSEs must be equal (config.mse_equal_device_vs_cpu).
  cpu_baseline   the CPU oracle ("port") timed on the guided passes of that 320x180 schedule, all host cores.
  kernels_synthetic / roofline.s1_* s2_* s3_*   the stand-alone entry points (pg_pdf, pg_sample, pg_guide_bounce, pg_splat) on
          SURVEY 8(d)'s S1 / S2 / S3 at their stated sizes (synthetic_kernels_leg).

roofline = the SD-tree kernel (the fused k_bounce of quad scenes in the timed region; for mesh scenes k_wave_guide,
timed in a SECOND region of K steps that follows the K steps of `value` at once: the same passes with
pg_render_stages(2) -- by default a bounce's shading, SD-tree calls and shadow ray are ONE kernel, k_wave_shade, in which
they cannot be timed apart; `roofline.region` says which region the figures are of).  Two byte models over its mean
launch time (HIP events recorded by the library on the launch stream) and the 8 TB/s HBM peak:
  frac / achieved   the bytes the lanes of an instrumented pass GATHERED from the tables of the built layout (16 per KD grid entry /
               node below it, 8 per tree head, 16 per jump-table entry, 32 per quadtree record of a pdf or sampling walk, 16 per
               record of a leaf walk; pg_depth_counters.layout_bytes), nothing credited for lanes of a wave that share a line --
               what the memory pipeline moves at least; <= 1 by construction; recomputable from layout_bytes_per_launch and
               avg_launch_us (the rocprofv3 summary of the same command under profiles/ has the same average).
  frac_model_8d  SURVEY.md 8d's ALGORITHMIC bytes: 16 B per KD level + 20 B per quadtree level of the REFERENCE's descents, levels
               counted by the same pass.  The jump grid and jump tables serve most of those levels with one gather each, so this
               exceeds 1 on spatially sorted lists: not a bandwidth (model_applicable false).  (Rounds 1-4 had this under `frac`.)
`traffic` = HBM bytes of the PMC counters per launch.  The default one-GPU line MEASURES it in the run (traffic_source "in-run":
behind the timed regions rank 0 starts `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` as two bounded child processes over
three steps of the same configuration, measure_traffic_in_run; --pmc-in-run 0 to skip); otherwise, and whenever those passes
fail, it is the committed figure (profiles/pmc_traffic.json, refused unless taken of exactly this code; traffic_tables_match: the
forest of this run has the table resolutions of the profiled one), which the line also carries as traffic_committed.
`kernels` lists every kernel of a step with its share and, from the same counter passes (in-run for k_wave_guide, k_wave_shade,
k_wave_trace, k_splat_list and k_finish; the committed figures otherwise, and for the atomic sector updates), its counter traffic
per second (pmc_frac_of_hbm_peak: as counted; ..._fetch_x2: with the guide's FETCH correction, an upper bound).
`kernels_synthetic` / roofline.s1_* s2_* s3_*: the stand-alone entry points on SURVEY 8(d)'s S1 / S2 / S3 with the same two fractions and,
from the cpu_baseline leg, the CPU restatement's rate on the same inputs (cpu_G_units_per_s, cpu_cores).
config.c2_* / c3_* / c5_*: BASELINE configs[1], [2], [4] timed for a few steps in the same run (other_configs_leg).

OUTPUT: stdout carries exactly ONE line -- compact_record(): < 6 KB of strict JSON with the contract's keys, `roofline`, `cpu_baseline`,
kernel name -> ms per step, and the flat figures named below; it is the last thing the process writes (the driver could not
parse round 5's 22.9 KB line).  The FULL record (every note, the per-kernel counter tables, kernels_synthetic, schedule, full_schedule)
goes to --detail (default gpurun_out/bench_detail.json); a builder-run copy of it is committed per round under profiles/.

`--synthetic` runs the renderer-free hot-path workload instead (seeded synthetic surface points).
Launch:  python bench.py [--gpus N]      (N > 1 without WORLD_SIZE in the environment: this process starts N fresh
                                          rank processes itself -- before anything touches the GPU -- and relays rank 0's line)
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec



def device_copy_rate(dev) -> float:
    """The achievable-copy figure SURVEY 8(d) asks to be recorded beside the nominal peak: bytes read +
    bytes written per second of a 1 GiB device-to-device copy on this box (best of 5), GB/s."""
    import torch
    n = 1 << 28  # 1 GiB of float32
    src = torch.empty(n, dtype=torch.float32, device=dev).fill_(1.0)
    dst = torch.empty_like(src)
    best = float("inf")
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dst.copy_(src)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    del src, dst
    return round(2.0 * 4.0 * n / (best * 1e-3) / 1e9, 1)

CONFIG_ITERATIONS = {"veach-ajar": 12, "cornell-box": 8, "veach-mis": 10, "torus": 10}  # BASELINE.json configs (torus: as veach-mis)
SCENES = {  # film width, aspect (w, h), max_depth of the BASELINE config
    "veach-ajar": (1920, (16, 9), 13), "cornell-box": (512, (1, 1), 8), "veach-mis": (1280, (16, 9), 3),
    "torus": (1920, (16, 9), 32),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="veach-ajar", choices=sorted(SCENES),
                    help="veach-ajar 1920x1080 max_depth 13 (BASELINE configs[3], the metric's scene; default), cornell-box "
                         "512x512 max_depth 8 (configs[1]), veach-mis 1280x720 max_depth 3 (configs[2]), torus 1920x1080 "
                         "max_depth 32 (configs[4])")
    ap.add_argument("--res", type=int, default=None, help="film width (height follows the scene's aspect)")
    ap.add_argument("--depth", type=int, default=None, help="max_depth")
    ap.add_argument("--spp-per-pass", type=int, default=16,
                    help="samples per pixel traced by one step (16: 33 M paths per step at 1920x1080, and still 4 M per GPU at N = 8; "
                         "measured on one GPU with every rank's share in turn: 7.2x of 8 at 16 spp per pass, 6.8x at 8)")
    ap.add_argument("--batched", type=int, default=1,
                    help="1 (default): a step is --spp-per-pass consecutive ONE-sample passes -- the reference's training passes, "
                         "main.py:192, seeded initial_seed + cumm_spp, :218 -- traced as one wavefront (pg_pass_params.batched: bit-identical "
                         "to the separate passes, tests/test_gpu_render.py); 0: one pass of --spp-per-pass samples per pixel (mi.render(spp=N))")
    ap.add_argument("--synthetic-kernels", type=int, default=None,
                    help="1: time the stand-alone entry points on SURVEY 8(d)'s S1 / S2 / S3 at their stated sizes and put the figures into "
                         "`roofline` (s1_*, s2_*, s3_*); default: 1 at N = 1, 0 otherwise")
    ap.add_argument("--train-iters", type=int, default=6, help="iterations rendered to train the SD-tree (the configs say 8/10/12)")
    ap.add_argument("--shard", default="tiles", choices=["tiles", "passes"], help="N > 1: strong scaling by tiles (default) or weak by passes")
    ap.add_argument("--backend", default="auto", choices=["auto", "nccl", "gloo"],
                    help="N > 1: torch.distributed backend.  auto = nccl (RCCL) when every rank of this node has a GPU of its own, "
                         "gloo when ranks share devices (a rehearsal: RCCL refuses two ranks on one device).  Decided from the "
                         "device count before anything is initialised, the same on every rank; a failing nccl initialisation "
                         "ends the run with a non-zero exit code, it is never replaced by gloo on some ranks only")
    ap.add_argument("--full-schedule", type=int, default=None,
                    help="1: after the steady-state steps, run the BASELINE config's WHOLE schedule through driver.run_guided_render "
                         "(veach-ajar: 12 iterations = 16380 spp, main.py:157-170, with main.py's stop-training rule :334-377) and "
                         "report value_full_schedule_12it + the final image's MSE; default: 1 at N = 1, 0 otherwise")
    ap.add_argument("--spp1", type=int, default=1, help="0: skip the leg that times 1-spp passes (value_spp1)")
    ap.add_argument("--cpu", type=int, default=1, help="0: skip the cpu_baseline / MSE-equality leg")
    ap.add_argument("--pmc-in-run", type=int, default=None,
                    help="1: `roofline.traffic` is MEASURED in this run -- behind the timed regions, rank 0 starts two bounded child "
                         "processes, `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (one counter per pass, nothing traced) around three "
                         "steps of the same configuration, and reads their counter tables; 0: the committed figures of "
                         "profiles/pmc_traffic.json only (default: 1 for the default one-GPU line)")
    ap.add_argument("--other-configs", type=int, default=None,
                    help="1: also time BASELINE configs[1], [2], [4] (cornell-box 512x512, veach-mis 1280x720, torus 1920x1080 depth 32) for a "
                         "few steps each and put the figures into `config` (c2_/c3_/c5_*); default: 1 at N = 1 on the default scene")
    ap.add_argument("--exchange-overlap", type=int, default=0,
                    help="N > 1, full-schedule leg: 1 lets the accumulators' all-reduce (libpgsd's RCCL communicator) travel beside the image "
                         "collectives of torch.distributed's communicator; 0 (default) orders it ahead of them -- two communicators in flight at "
                         "once has not run on hardware yet")
    ap.add_argument("--split-pipeline", action="store_true",
                    help="cornell-box / veach-mis: run the bounce as the split pipeline instead of the fused kernel (same results; the "
                         "roofline is then read off k_wave_guide, the SD-tree queries alone)")
    ap.add_argument("--overlap", type=int, default=0, help="pg_render_overlap mode of the timed steps")
    ap.add_argument("--stages", type=int, default=0, choices=[0, 1, 2],
                    help="pg_render_stages of the timed region: 0 one shading kernel per bounce (default; bench.py then times k_wave_guide "
                         "in a second region for the roofline), 1 k_wave_shade_a with the SD-tree calls | k_wave_cast | k_wave_shade_b, "
                         "2 those with k_wave_guide on its own")
    ap.add_argument("--sort", type=int, default=1, help="pg_render_sort: the live list of a mesh scene's bounce in a global spatial order (0: list order)")
    ap.add_argument("--in-flight", type=int, default=None, choices=[1, 2],
                    help="2: consecutive passes alternate between two buffer sets and two streams (pg_pass_params.slot), two on the device at "
                         "once.  Default: 1 on one GPU (the whole film fills the chip: 51.9 vs 52.9 ms per step), 2 when the film is sharded -- an "
                         "eighth of it leaves gaps between its small launches that a second pass fills: 7.72 -> 7.08 ms per step of a rank's share, "
                         "6.7x -> 7.3x of 8 by emulation on one GPU (tools/stripe_balance.py, profiles/r04/stripe_balance.txt)")
    ap.add_argument("--synthetic", action="store_true", help="renderer-free SD-tree hot-path workload")
    ap.add_argument("--phase-probe", type=int, default=None,
                    help="1: behind the timed regions rank 0 runs the PROBE build of the library (libpgsd_phases.so, csrc/Makefile `probe`: "
                         "k_wave_shade with wave-clock stamps at its phase boundaries) as a child process for three steps and reports the "
                         "SD-tree calls' share of k_wave_shade in the region `value` is quoted on (roofline.value_region_sdtree_*); "
                         "default: 1 for the default one-GPU line.  2: (internal) this process IS that child")
    ap.add_argument("--detail", default=None,
                    help="file the FULL record is written to (default gpurun_out/bench_detail.json); stdout carries one compact line, "
                         "< 6 KB of strict JSON, and nothing else")
    ap.add_argument("--no-compaction", action="store_true", help="(synthetic) mask dead lanes instead of compacting")
    args = ap.parse_args()
    w, _, d = SCENES[args.scene]
    if args.res is None:
        args.res = 512 if args.synthetic else w
    if args.depth is None:
        args.depth = 8 if args.synthetic else d
    if args.full_schedule is None:
        args.full_schedule = 1 if (args.gpus == 1 and not args.synthetic) else 0
    if args.synthetic_kernels is None:
        args.synthetic_kernels = 1 if (args.gpus == 1 and not args.synthetic) else 0
    if args.other_configs is None:
        args.other_configs = 1 if (args.gpus == 1 and not args.synthetic and args.scene == "veach-ajar") else 0
    if args.pmc_in_run is None:
        args.pmc_in_run = 1 if (args.gpus == 1 and not args.synthetic and args.full_schedule) else 0
    if args.phase_probe is None:
        args.phase_probe = 1 if (args.gpus == 1 and not args.synthetic and args.full_schedule) else 0
    if args.in_flight is None:
        args.in_flight = 2 if (args.gpus > 1 and args.shard == "tiles" and not args.synthetic) else 1
    return args


def make_scene(name, width, depth):
    from practical_path_guiding_lab_amd import scene as S

    aw, ah = SCENES[name][1]
    height = width * ah // aw
    if name == "veach-mis":
        return S.veach_mis(width, height, depth, 8)
    if name == "torus":
        return S.torus(width, height, depth, 8)
    if name == "veach-ajar":
        return S.veach_ajar(width, height, depth, 8)
    return S.cornell_box(width, height, depth, 8)


ATOMIC_CEILING_GPS = 23.6  # G sector-updates/s chip-wide: tools/atomic_probe.hip on MI355X (round 1), scattered 64-bit adds
RANDOM_GATHER_CEILING_GPS = 55.0  # G lanes/s, every lane its own 16 bytes of a table beyond every cache (268 MB - 1 GB): tools/gather_probe.hip
#                                   on MI355X (profiles/r05/gather_probe.txt) = 0.88 TB/s of useful bytes, 0.11 of the 8 TB/s peak; 242 G/s from 4 MB (L2)


IN_RUN_TRAFFIC = {}  # kernel -> {"hi", "lo", "atomics": None}: filled by measure_traffic_in_run, consulted first by traffic_for
PMC_CHILD_STEPS = 3


def measure_traffic_in_run(args, launches_per_step, timeout_s=90.0):
    """HBM bytes per launch of the kernels that hold the SD-tree calls, measured NOW: two child processes, one counter each
    (`rocprofv3 --pmc FETCH_SIZE`, `--pmc WRITE_SIZE` -- separate passes and nothing traced beside them, as
    MI355X_MICROARCH.md's HBM section prescribes), each running this script for PMC_CHILD_STEPS steps of the same
    configuration; the counter tables are read as tools/summarize_profile.py reads a committed profile: k_wave_guide over the
    last steps' launches (the roofline region), k_wave_shade over the last launches before the first k_wave_guide (the
    region `value` is quoted on).  `launches_per_step`: {kernel: launches per step}.  Returns (figures, note): figures =
    {kernel: {"hi": (2 * FETCH + WRITE) bytes, "lo": as counted, "atomics": None}} or None with the reason in note.  The
    children are fresh processes started with subprocess (nothing is exec'ed by this process, which has touched the GPU);
    the program behind `--` is python3 itself."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile

    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rp is None:
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself under a profiler"
    out = tempfile.mkdtemp(prefix="pgsd_pmc_", dir="/tmp")
    child = [sys.executable if os.path.basename(sys.executable).startswith("python") else "python3", os.path.abspath(__file__),
             "--scene", args.scene, "--res", str(args.res), "--depth", str(args.depth), "--spp-per-pass", str(args.spp_per_pass),
             "--batched", str(args.batched), "--train-iters", str(args.train_iters), "--sort", str(args.sort),
             "--in-flight", str(args.in_flight), "--steps", str(PMC_CHILD_STEPS), "--warmup", "1", "--cpu", "0", "--full-schedule", "0",
             "--spp1", "0", "--other-configs", "0", "--synthetic-kernels", "0", "--pmc-in-run", "0", "--phase-probe", "0", "--detail", os.devnull]
    env = dict(os.environ, TMPDIR="/tmp")
    t0 = time.perf_counter()
    vals = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out, counter)
            cmd = [rp, "--pmc", counter, "--kernel-include-regex", "k_wave_|k_splat_list|k_finish", "--output-format", "csv", "-d", d, "--"] + child
            # (its own session: on expiry the whole group goes -- the profiler AND the python3 it started -- by its exact id)
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
            try:
                _, err = pr.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except OSError:
                    pass
                pr.communicate()
                return None, f"the {counter} pass did not finish within {timeout_s:.0f} s (ended)"
            if pr.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} exited {pr.returncode}: {err.decode(errors='replace')[-300:]}"
            fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not fs:
                return None, f"rocprofv3 --pmc {counter} wrote no counter table"
            rows = sorted((x for x in csv.DictReader(open(fs[0])) if x["Counter_Name"] == counter), key=lambda x: int(x["Dispatch_Id"]))

            def short(n):
                for k in ("k_wave_trace", "k_wave_shade_a", "k_wave_shade_b", "k_wave_guide", "k_wave_shade", "k_wave_cast", "k_wave_tail",
                          "k_splat_list", "k_finish"):
                    if k in n:
                        return k
                return None
            first_guide = min((int(x["Dispatch_Id"]) for x in rows if short(x["Kernel_Name"]) == "k_wave_guide"), default=None)
            for k, per_step in launches_per_step.items():
                # (k_wave_guide: the roofline region, the run's last steps; every other kernel: the region `value` is quoted on,
                # which ends where the first k_wave_guide starts)
                v = [float(x["Counter_Value"]) for x in rows if short(x["Kernel_Name"]) == k
                     and (k == "k_wave_guide" or first_guide is None or int(x["Dispatch_Id"]) < first_guide)]
                n = int(round(per_step * PMC_CHILD_STEPS))
                if n <= 0 or len(v) < n:
                    return None, f"{counter}: {len(v)} launches of {k} in the child's table, {n} expected"
                vals.setdefault(k, {})[counter] = sum(v[-n:]) / n  # KiB per launch over the child's timed steps
    except Exception as e:  # (a counter table of an unexpected shape: the committed figures serve)
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(out, ignore_errors=True)
    fig = {k: {"hi": int((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024), "lo": int((c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024), "atomics": None}
           for k, c in vals.items()}
    return fig, (f"measured in this run: `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` as two child processes of {PMC_CHILD_STEPS} steps each "
                 f"behind the timed regions ({time.perf_counter() - t0:.0f} s), KiB as reported, hi = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 "
                 "(the gfx950 x2 FETCH correction of MI355X_MICROARCH.md, an upper bound for scattered reads), lo = as counted")


def traffic_for(kernel, key):
    """Counter figures per launch of `kernel` from the committed PMC summary of the same configuration
    (profiles/pmc_traffic.json, tools/summarize_profile.py) -- {"hi": HBM bytes with the gfx950 x2 FETCH correction of
    MI355X_MICROARCH.md (an upper bound for scattered reads), "lo": as counted, "atomics": L2 atomic sector updates} --
    or None when none is committed or when it was taken of OTHER CODE: the summary records the hash of the library's
    sources, and a figure whose hash differs from the sources this run was built from is refused."""
    if kernel in IN_RUN_TRAFFIC:
        e = dict(IN_RUN_TRAFFIC[kernel])
        if e.get("atomics") is None:  # (the run's own passes count bytes; the atomic sector updates stay the committed pass's)
            c = committed_traffic_for(kernel, key)
            e["atomics"] = None if c is None else c.get("atomics")
        return e
    return committed_traffic_for(kernel, key)


def committed_traffic_for(kernel, key):
    """traffic_for's committed source (see there)."""
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        from practical_path_guiding_lab_amd._native import source_hash
        j = json.load(open(pmc))
        k = j.get("configs", {}).get(key, {})
        if k.get("_source_hash") != source_hash():
            return None
        if kernel in k:
            e = k[kernel]
            return {"hi": e["hbm_bytes_per_launch"], "lo": e.get("hbm_bytes_per_launch_uncorrected"),
                    "atomics": e.get("atomic_sector_updates_per_launch")}
    except Exception:
        pass
    return None


def profiled_table_bits(key):
    """(jump_bits, kd_grid_bits) of the forest the committed PMC figures of configuration `key` were taken of, or None."""
    try:
        k = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("configs", {}).get(key, {})
        if "_jump_bits" in k:
            return int(k["_jump_bits"]), int(k.get("_kd_grid_bits", -1))
    except Exception:
        pass
    return None


def spawn_ranks(cmd, n, env=None, relay=sys.stdout, grace_s=10.0):
    """Starts n rank processes of `cmd` (a list for subprocess.Popen) on this node, one per GPU, with the
    environment torch.distributed.run would give them (RANK, LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE,
    MASTER_ADDR 127.0.0.1, a free MASTER_PORT), relays rank 0's stdout to `relay`, and returns 0 when every rank
    exited 0.  When a rank fails the others are ended (their exact process ids) and the first failing exit code
    is returned.  The caller must not have touched the GPU: the children are fresh processes, nothing is exec'ed."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    base = dict(os.environ if env is None else env)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base.update({"WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    import signal
    import threading
    procs, lines = [], []

    def end_all(grace):
        """terminate, wait the grace period, kill -- every child still alive, by its own process id"""
        alive = [q for q in procs if q.poll() is None]
        for q in alive:
            q.terminate()
        t_end = time.time() + grace
        for q in alive:
            try:
                q.wait(max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                q.kill()
                q.wait()

    def on_signal(signum, frame):  # the harness's timeout, Ctrl-C: do not leave ranks behind holding the GPUs
        raise KeyboardInterrupt(f"signal {signum}")

    old_handlers = {}
    if threading.current_thread() is threading.main_thread():
        for sg in (signal.SIGTERM, signal.SIGINT):
            old_handlers[sg] = signal.signal(sg, on_signal)
    rc = 0
    th = None
    try:
        for r in range(n):
            e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))

        def pump():
            for line in procs[0].stdout:
                lines.append(line)

        th = threading.Thread(target=pump, daemon=True)
        th.start()
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                c = procs[r].poll()
                if c is None:
                    continue
                pending.discard(r)
                if c != 0 and rc == 0:
                    rc = c
                    print(f"[bench] rank {r} exited with code {c}; ending the other ranks", file=sys.stderr)
                    end_all(grace_s)
            time.sleep(0.05)
    except KeyboardInterrupt as e:
        print(f"[bench] interrupted ({e}); ending the ranks", file=sys.stderr)
        rc = rc or 130
    finally:
        end_all(grace_s)  # (a no-op when every rank has exited)
        for sg, h in old_handlers.items():
            signal.signal(sg, h)
    if th is not None:
        th.join(5.0)
    for line in lines:  # rank 0's JSON line goes on; anything else a library wrote to its stdout is diagnostics
        (relay if line.lstrip().startswith("{") else sys.stderr).write(line)
    relay.flush()
    return rc


class StepQueue:
    """The steps of a sharded film leave in groups: a rank's share is 1 / world of the pixels, so it launches the passes
    of `group` = world steps at once (a launch as big as the one-GPU launch of the whole film); a step only queues its
    passes, `flush` launches what is queued -- after the last warm-up step, after the last timed step (steps % group != 0
    leaves a smaller last launch) and before anything waits for the device.  launch(n_passes, first_seed) traces n_passes
    consecutive one-sample passes; seeds run on without gaps whatever the grouping."""

    def __init__(self, launch, spp_per_step, group, first_seed, seed_stride=1):
        self.launch, self.spp, self.group = launch, int(spp_per_step), max(1, int(group))
        self.seed, self.stride, self.pending = int(first_seed), int(seed_stride), 0
        self.launches = []  # (passes, first seed) of every launch, in order

    def flush(self):
        if self.pending:
            n = self.spp * self.pending
            self.launch(n, self.seed)
            self.launches.append((n, self.seed))
            self.seed += n * self.stride
            self.pending = 0

    def step(self):
        self.pending += 1
        if self.pending >= self.group:
            self.flush()


def pick_backend(args, world, ndev):
    """The torch.distributed backend, decided before anything is initialised and identically on every rank of
    the node: every rank sees the same device count."""
    if args.backend != "auto":
        return args.backend
    return "nccl" if world <= ndev else "gloo"


def init_dist(args):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        raise SystemExit("bench.py --gpus N>1: the rank processes are started by main() or by torch.distributed.run")
    ndev = torch.cuda.device_count()  # (counting devices does not initialise the GPU)
    if ndev == 0 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    args.n_devices = min(local_world, ndev) * max(1, world // max(local_world, 1))  # distinct GPUs of the job
    local_rank = local_rank % ndev  # (ranks share devices only in a rehearsal on a box with fewer GPUs than ranks)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = pick_backend(args, local_world, ndev)
        if backend == "nccl":
            from datetime import timedelta
            from practical_path_guiding_lab_amd.parallel import Watchdog, comm_init_timeout_s
            # bounded: a rank whose peers never arrive exits non-zero (the launcher ends the job) instead of waiting for ever
            with Watchdog(comm_init_timeout_s(), f"rank {rank}: init_process_group(nccl) + first barrier"):
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank),
                                        timeout=timedelta(seconds=max(60.0, comm_init_timeout_s())))  # (a failure ends the run)
                dist.barrier()
        else:
            # (the Gloo library prints a connection banner on STDOUT when its context comes up: stdout is for the one JSON
            # line, so file descriptor 1 points at stderr while the group is made and first used)
            sys.stdout.flush()
            keep = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group(backend="gloo")
                dist.barrier()
            finally:
                sys.stdout.flush()
                os.dup2(keep, 1)
                os.close(keep)
    return world, rank, local_rank


def timed_steps(step, steps, warmup, world, flush=None):
    """`flush` (optional): called after the last warm-up step and after the last timed step, before the device is waited for --
    for a `step` that only queues its passes and launches several steps' passes at once (run_render, sharded film)."""
    import torch
    import torch.distributed as dist

    for _ in range(warmup):
        step()
    if flush is not None:
        flush()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if flush is not None:
        flush()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        from practical_path_guiding_lab_amd.parallel import max_over_ranks
        elapsed = max_over_ranks(elapsed, device="cuda")
    return elapsed


def masked_mse(img, gt, mask):
    """path_guiding_integrator.py:503-517 on (H, W, 3) arrays, over the pixels of `mask`."""
    import numpy as np
    d2 = (img.astype(np.float64) - gt.astype(np.float64)) ** 2
    lum = np.minimum(0.212671 * d2[..., 0] + 0.715160 * d2[..., 1] + 0.072169 * d2[..., 2], 1e4)
    return float(lum[mask].mean())


def gt_fixture(name, width, height):
    """(ground truth (h, w, 3) float32, mask (h, w) bool, factor) at the largest committed size that
    divides the film, or None: the reference's TungstenRender.exr box-downsampled (tests/golden/)."""
    import numpy as np
    from practical_path_guiding_lab_amd import scene as S
    files = {"veach-ajar": [(640, 360), (320, 180)], "cornell-box": [(256, 256)], "veach-mis": [(320, 180)]}.get(name, [])
    stem = {"veach-ajar": "veach_ajar_gt_%dx%d_f16.npy", "cornell-box": "cornell_gt_%d_f16.npy", "veach-mis": "veach_mis_gt_%dx%d_f16.npy"}
    for gw, gh in files:
        if width % gw or height % gh or width // gw != height // gh:
            continue
        fn = stem[name] % ((gw, gh) if name != "cornell-box" else (gw,))
        path = os.path.join(ROOT, "tests", "golden", fn)
        if not os.path.exists(path):
            continue
        gt = np.load(path).astype(np.float32)
        mask = S.veach_ajar_mask(gw, gh) if name == "veach-ajar" else np.ones((gh, gw), bool)
        return gt, mask, width // gw
    return None


def image_mse(sumL, spp, width, height, name):
    """MSE vs the ground-truth fixture of the per-pixel mean image sumL/spp ((3, H*W) array)."""
    import numpy as np
    fx = gt_fixture(name, width, height)
    if fx is None:
        return None, None
    gt, mask, f = fx
    img = (np.asarray(sumL, dtype=np.float64) / float(spp)).T.reshape(height, width, 3)
    if f > 1:
        img = img.reshape(height // f, f, width // f, f, 3).mean(axis=(1, 3))
    note = (f"mean image box-filtered {f}x{f} to {gt.shape[1]}x{gt.shape[0]} vs TungstenRender.exr box-filtered to the same size"
            + ("; the teapot rectangle (their meshes are missing from the reference mount) is masked" if name == "veach-ajar" else ""))
    return masked_mse(img, gt, mask), note


PROBE_LIBRARY = os.path.join(ROOT, "practical_path_guiding_lab_amd", "libpgsd_phases.so")
PHASE_NAMES = ("records_staging", "stage_a1", "shadow_walk", "stage_a2", "sdtree_calls", "stage_b", "append")


def run_phase_probe(args):
    """--phase-probe 2: this process runs on the PROBE build ($PGSD_LIBRARY = libpgsd_phases.so): the same scene, training and
    batched steps as the timed region of `value`, with k_wave_shade's phase stamps on and the depth counters OFF
    (pg_enable_depth_counters(2): no atomics inside the SD-tree walks), and prints one small JSON line: the wave-clock cycles
    of the seven phases summed over the waves of args.steps steps, k_wave_shade's average launch while stamping and -- the
    same passes again with the stamps switched off -- while the compiled-in stamps idle."""
    import numpy as np
    import torch
    from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator
    from practical_path_guiding_lab_amd.render import IndependentSampler, WavefrontScene

    sc = make_scene(args.scene, args.res, args.depth)
    integ = PathGuidingIntegrator({"max_depth": args.depth, "rr_depth": 8}, device=0)
    tree = integ.sdTree
    npix = sc.camera.width * sc.camera.height
    integ.setup(npix, sc.bbox_min - np.float32(1e-4), sc.bbox_max + np.float32(1e-4), 20, 20, True, 0.5)
    ws = WavefrontScene(sc, in_flight=args.in_flight, sort=bool(args.sort), stages=args.stages)
    ws.reserve(integ, args.spp_per_pass)
    batched = bool(args.batched)
    cumm = 0
    for k in range(args.train_iters):
        integ.setIteration(k, False)
        iter_spp = 2 ** (k + 2)
        chunk = max(1, min(args.spp_per_pass, iter_spp))
        for i in range(iter_spp // chunk):
            integ.sample(ws, IndependentSampler(chunk, cumm + i * chunk, batched=batched))
        cumm += iter_spp
        integ.refineAndPrepareSDTreeForNextIteration()
    integ.setIteration(args.train_iters, False)
    seed = [cumm]

    def step():
        integ.sample(ws, IndependentSampler(args.spp_per_pass, seed[0], batched=batched))
        seed[0] += args.spp_per_pass

    def region(mode):
        tree.enableDepthCounters(mode)
        step()
        torch.cuda.synchronize()
        tree.readShadePhases(reset=True)
        tree.enableKernelTiming(True)
        tree.readKernelTiming(reset=True)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        kt = tree.readKernelTiming(reset=True)
        tree.enableKernelTiming(False)
        ph = tree.readShadePhases(reset=True)
        tree.enableDepthCounters(False)
        return kt, ph

    kt_on, (compiled, waves, cycles) = region(2)
    kt_off, _ = region(0)
    return {"phase_probe": True, "compiled_in": compiled, "waves": waves, "cycles": cycles, "steps": args.steps,
            "shade_launches": int(kt_on.bounce_launches), "shade_avg_us_stamping": 1e3 * kt_on.shade_ms / max(kt_on.bounce_launches, 1),
            "shade_avg_us_stamps_idle": 1e3 * kt_off.shade_ms / max(kt_off.bounce_launches, 1)}


def measure_sdtree_phase(args, timeout_s=120.0):
    """The SD-tree calls INSIDE k_wave_shade, where `value` is quoted: a child process on the probe build
    (run_phase_probe) -- a fresh process with $PGSD_LIBRARY set, nothing exec'ed by this one.  Returns (dict, note) or (None, why)."""
    import signal
    import subprocess

    if not os.path.exists(PROBE_LIBRARY):
        return None, "the probe build (libpgsd_phases.so: make -C practical_path_guiding_lab_amd/csrc probe) is not there"
    child = [sys.executable if os.path.basename(sys.executable).startswith("python") else "python3", os.path.abspath(__file__),
             "--scene", args.scene, "--res", str(args.res), "--depth", str(args.depth), "--spp-per-pass", str(args.spp_per_pass),
             "--batched", str(args.batched), "--train-iters", str(args.train_iters), "--sort", str(args.sort),
             "--in-flight", str(args.in_flight), "--stages", str(args.stages), "--steps", "3", "--phase-probe", "2"]
    env = dict(os.environ, PGSD_LIBRARY=PROBE_LIBRARY)
    pr = subprocess.Popen(child, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
    try:
        out, err = pr.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(pr.pid, signal.SIGKILL)
        except OSError:
            pass
        pr.communicate()
        return None, f"the probe process did not finish within {timeout_s:.0f} s (ended)"
    if pr.returncode != 0:
        return None, f"the probe process exited {pr.returncode}: {err.decode(errors='replace')[-300:]}"
    for line in reversed(out.decode(errors="replace").splitlines()):
        if line.startswith("{"):
            try:
                d = json.loads(line)
            except ValueError:
                continue
            if d.get("phase_probe"):
                if not d.get("compiled_in") or sum(d["cycles"]) <= 0:
                    return None, "the probe library has no phase stamps compiled in"
                return d, "measured in this run: three steps on the probe build (libpgsd_phases.so), stamps on, depth counters off"
    return None, "the probe process printed no result"


# ------------------------------------------------------------------------------------------------
def run_render(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator
    from practical_path_guiding_lab_amd.parallel import all_reduce_accumulators, all_reduce_sums
    from practical_path_guiding_lab_amd.render import IndependentSampler, WavefrontScene

    from practical_path_guiding_lab_amd._native import LIB_PATH as N_LIB_PATH, source_hash
    SRC_HASH = source_hash()
    world, rank, local_rank = init_dist(args)
    sc = make_scene(args.scene, args.res, args.depth)
    W, H = sc.camera.width, sc.camera.height
    integ = PathGuidingIntegrator({"max_depth": args.depth, "rr_depth": 8}, device=local_rank)
    tree = integ.sdTree
    npix = W * H
    integ.setup(npix, sc.bbox_min - np.float32(1e-4), sc.bbox_max + np.float32(1e-4), 20, 20, True, 0.5)  # main.py:56-64
    ws = WavefrontScene(sc, split_pipeline=args.split_pipeline, overlap=args.overlap, in_flight=args.in_flight, sort=bool(args.sort),
                        stages=args.stages)
    tiles = world > 1 and args.shard == "tiles"
    if tiles:
        ws.set_shard(rank, world, 4)
    my_pixels = int(ws.local_pixels().shape[0])
    npix_local = my_pixels
    from practical_path_guiding_lab_amd.parallel import min_max_over_ranks
    pix_min, pix_max = min_max_over_ranks(my_pixels)
    # A rank of a sharded film launches the passes of `group` = world steps at once: its share of the film is 1 / world of the
    # pixels, so that launch is as big as the one-GPU launch of the whole film (33 M lanes at the default) instead of 1 / world
    # of it -- the same passes, bit for bit (pg_pass_params.batched), and the same number of them per step; small launches are
    # what a rank's eighth of the film lost (tools/stripe_balance.py: 6.7x -> 7.7x of 8 by emulation on one GPU).
    group = world if (tiles and bool(args.batched)) else 1
    ws.reserve(integ, args.spp_per_pass * group)  # the pass buffers, as the reference's setup() allocates its record arrays (:93)
    # the exchange: libpgsd's own ncclAllReduce (pg_allreduce) when every rank has its GPU, else torch.distributed
    exchange = "none"
    reduce_fn = None
    rccl_seen = None
    if world > 1:
        from practical_path_guiding_lab_amd.parallel import init_library_comm
        if init_library_comm(tree):  # (decided collectively: nccl backend and a GPU of its own for every rank)
            exchange = "pg_allreduce (RCCL ncclAllReduce int64 issued by libpgsd.so)"
            reduce_fn = lambda acc: tree.allReduce()  # noqa: E731
            reduce_fn.exchanges_itself = True  # (pg_allreduce packs, sums and unpacks inside the library)
            rccl_seen = tree.commInfo()  # RCCL's own word on the communicator: ncclCommCount, ncclCommUserRank
        else:
            exchange = f"torch.distributed all_reduce ({dist.get_backend()})"
            reduce_fn = lambda acc: all_reduce_accumulators(acc)  # noqa: E731

    # ---- train: really render iterations 0..train_iters-1 (2^(k+2) spp each, main.py:170) ----
    batched = bool(args.batched)
    per_iter = []
    cumm = 0
    for k in range(args.train_iters):
        integ.setIteration(k, False)
        integ.resetVarianceCounter()  # main.py:161-163: the image of an iteration holds its own samples only
        iter_spp = 2 ** (k + 2)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        if tiles or world == 1:
            chunk = max(1, min(args.spp_per_pass * group, iter_spp))
            for i in range(iter_spp // chunk):
                integ.sample(ws, IndependentSampler(chunk, cumm + i * chunk, batched=batched))
        else:  # passes: the passes of an iteration are independent (main.py:208-218), ranks take them in turn
            chunk = max(1, min(args.spp_per_pass, iter_spp // world))
            for i in range(iter_spp // chunk):
                if i % world == rank:
                    integ.sample(ws, IndependentSampler(chunk, cumm + i * chunk, batched=batched))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        integ.refineAndPrepareSDTreeForNextIteration(reduce_fn)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t2 = time.perf_counter()
        cumm += iter_spp
        per_iter.append({"iteration": k, "spp": iter_spp, "render_ms": round(1e3 * (t1 - t0), 2),
                         "exchange_refine_ms": round(1e3 * (t2 - t1), 2)})
    last_spp = 2 ** (args.train_iters + 1)
    sums = all_reduce_sums(integ.sumL, integ.sumL2) if world > 1 else (integ.sumL, integ.sumL2)
    mse_train, mse_note = (None, None)
    if rank == 0:
        mse_train, mse_note = image_mse(sums[0].cpu().numpy(), last_spp, W, H, args.scene)
    guided = [p for p in per_iter if p["iteration"] >= 2]
    t_guided = sum(p["render_ms"] + p["exchange_refine_ms"] for p in guided) * 1e-3
    full_schedule = (npix * sum(p["spp"] for p in guided) / t_guided / 1e6) if guided and t_guided > 0 else None
    stats = tree.stats()
    k = args.train_iters
    integ.setIteration(k, False)

    # (tiles: every rank traces the same passes of its own pixels; passes: rank r takes every world-th group of passes)
    queue = StepQueue(lambda n, sd: integ.sample(ws, IndependentSampler(n, sd, batched=batched)), args.spp_per_pass, group,
                      cumm + (0 if tiles else rank * args.spp_per_pass), 1 if (tiles or world == 1) else world)
    step, flush = queue.step, queue.flush
    seed = [0]  # (the later legs count their own seeds on from the queue's: set where they start)

    # one instrumented pass for the byte model (depth counters add atomics: not timed)
    tree.enableDepthCounters(True)
    tree.readDepthCounters(reset=True)
    rec_before = int(tree.exportAccumulators()[0][0])  # records counted at the KD root so far
    step()
    flush()  # (one step's passes by themselves: the counters below are per step)
    dc = tree.readDepthCounters(reset=True)
    tree.enableDepthCounters(False)
    records_per_pass = int(tree.exportAccumulators()[0][0]) - rec_before
    live = tree.renderLiveCounts(args.depth)  # paths alive after each bounce of that pass

    tree.enableKernelTiming(True)
    for _ in range(args.warmup):
        step()
    flush()
    tree.readKernelTiming(reset=True)
    elapsed = timed_steps(step, args.steps, 0, world, flush)
    kt = tree.readKernelTiming(reset=True)
    # ---- the roofline region: the SAME passes again with the SD-tree calls of a bounce in a kernel of their own
    # (pg_render_stages(2): k_wave_guide; in the region above they are part of k_wave_shade, where they cannot be timed
    # apart).  Its K steps follow the K steps of `value` immediately and are not part of `value`. ----
    kt_roof, elapsed_roof = None, None
    if kt.trace_launches > 0 and kt.guide_launches == 0:
        ws.set_stages(integ, 2)
        for _ in range(min(args.warmup, 2)):
            step()
        flush()
        tree.readKernelTiming(reset=True)
        elapsed_roof = timed_steps(step, args.steps, 0, world, flush)
        kt_roof = tree.readKernelTiming(reset=True)
        ws.set_stages(integ, args.stages)
    tree.enableKernelTiming(False)

    # (the SD-tree `value` was quoted on, for the config-matched cpu_baseline: sdTree_prev's 23 columns on the host)
    bench_tree_cols = tree.export() if (args.cpu and world == 1) else None
    # per-iteration exchange + refine (not part of `value`)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if reduce_fn is not None:
        integ._exchange(reduce_fn)
    torch.cuda.synchronize()
    t_allreduce = time.perf_counter() - t1
    t1 = time.perf_counter()
    tree.refineAndPrepare()
    torch.cuda.synchronize()
    t_refine = time.perf_counter() - t1
    # ---- the same passes one launch each (what `value`'s batched launch stands for, bit for bit), and one pass of
    # spp_per_pass samples per pixel (mi.render(spp=N): other streams, the same amount of work) ----
    spp1 = multi_spp = None
    flush()
    seed[0] = queue.seed
    if args.spp1 and (tiles or world == 1):
        def step1():
            integ.sample(ws, IndependentSampler(1, seed[0]))
            seed[0] += 1
        n1 = max(16, args.steps)
        spp1 = npix * n1 / timed_steps(step1, n1, 2, world) / 1e6

        def step_multi():
            integ.sample(ws, IndependentSampler(args.spp_per_pass, seed[0], batched=not batched))
            seed[0] += args.spp_per_pass
        nm = max(3, args.steps // 2)
        multi_spp = npix * args.spp_per_pass * nm / timed_steps(step_multi, nm, 1, world) / 1e6

    # ---- two passes in flight (pg_pass_params.slot): the same passes, alternating between two buffer sets and streams ----
    two_in_flight = None
    if args.spp1 and args.in_flight == 1 and (tiles or world == 1):
        ws.in_flight = 2
        queue.seed = max(queue.seed, seed[0])
        two_in_flight = npix * args.spp_per_pass * args.steps / timed_steps(step, args.steps, 2, world, flush) / 1e6
        ws.join()
        torch.cuda.synchronize()
        ws.in_flight = 1

    # ---- the config's whole schedule, end to end ----
    full = None
    if args.full_schedule:
        full = full_schedule_leg(args, integ, ws, (rank, world, 4) if tiles else None, reduce_fn, W, H)

    if world > 1:
        # every rank lets go of the library's communicator at the same point (nothing collective follows; a rank must
        # not sit in ncclCommDestroy at interpreter exit while the others are already gone)
        torch.cuda.synchronize()
        if exchange.startswith("pg_allreduce"):
            tree.commDestroy()
        dist.barrier()
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return None

    paths_per_step = npix * args.spp_per_pass * (1 if (tiles or world == 1) else world)  # all ranks
    value = paths_per_step * args.steps / elapsed / 1e6
    # instrumented pass -> algorithmic bytes per pass on this rank
    # SD-tree queries: 16 B per KD level + 20 B per quadtree level; splat: per record 16*D_kd + 4 + 48 + 12 per quadtree
    # level and descent, priced with the measured mean depths over the records actually kept
    tree_bytes = 16.0 * dc.kd_levels + 20.0 * dc.quad_levels          # all bounces of one pass
    # ... and the bytes the lanes of that pass GATHERED from the tables of the built layout (16 per KD grid entry / node below
    # it, 8 per tree head, 16 per jump-table entry, 32 per quadtree record of a pdf or sampling walk, 16 per record of a
    # leaf walk), nothing credited for lanes of a wave that share a line: pg_depth_counters.layout_bytes
    layout_bytes = float(dc.layout_bytes) + 8.0 * dc.kd_queries
    d_kd = dc.kd_levels / max(dc.kd_queries, 1)
    d_q = dc.quad_levels / max(dc.quad_queries, 1)
    splat_bytes = records_per_pass * (16.0 * d_kd + 4.0 + 48.0 + 12.0 * 2.0 * d_q)
    # (the kernel table is per STEP: with group > 1 a launch of pg_render_pass holds the passes of several steps)
    passes = max(kt.passes, 1) if group == 1 else max(args.steps, 1)
    wave = kt.trace_launches > 0  # the split pipeline of pg_render_wave.hip ran (mesh scenes, or --split-pipeline)
    step_ms = 1e3 * elapsed / args.steps

    def kern(ms, launches, alg_bytes_per_pass=None):
        per_pass = launches / passes
        d = {"launches": int(launches), "avg_us": round(1e3 * ms / max(launches, 1), 2), "ms_per_step": round(ms / passes, 3),
             "share_of_step": round(ms / passes / step_ms, 3)}
        if alg_bytes_per_pass is not None and ms > 0:
            d["alg_bytes_per_launch"] = round(alg_bytes_per_pass / (per_pass if per_pass > 0 else 1))
            d["alg_GBps"] = round(alg_bytes_per_pass * passes / (ms * 1e-3) / 1e9, 2)
        return d

    kernels = {}
    if wave:
        nb = kt.bounce_launches  # bounces
        if kt_roof is None:  # (--stages 2 / --overlap 1: the SD-tree calls ran as k_wave_guide in the timed region itself)
            kernels["k_wave_guide"] = kern(kt.guide_ms, kt.guide_launches, tree_bytes)
        else:
            p2 = max(kt_roof.passes, 1) if group == 1 else max(args.steps, 1)
            ms2 = kt_roof.guide_ms
            kernels["k_wave_guide"] = {
                "launches": int(kt_roof.guide_launches), "avg_us": round(1e3 * ms2 / max(kt_roof.guide_launches, 1), 2),
                "ms_per_step": round(ms2 / p2, 3), "region": "roofline",
                "alg_bytes_per_launch": round(tree_bytes / max(kt_roof.guide_launches / p2, 1e-9)),
                "alg_GBps": round(tree_bytes * p2 / (ms2 * 1e-3) / 1e9, 2) if ms2 > 0 else 0.0,
                "note": "timed in the roofline region (roofline.region), where the SD-tree calls are a kernel of their own; in the region "
                        "`value` is quoted on they are part of k_wave_shade (--stages 1: of k_wave_shade_a) and their time is inside it",
                "step_ms_of_that_region": round(1e3 * elapsed_roof / args.steps, 4),
                "shade_a_ms_per_step_of_that_region": round(kt_roof.shade_a_ms / p2, 3)}
        kernels["k_wave_trace"] = kern(kt.trace_ms, kt.trace_launches)
        if kt.shade_b_ms == 0 and kt.shadow_ms == 0:  # pg_render_stages 0: one shading kernel per bounce
            kernels["k_wave_shade"] = kern(kt.shade_ms, nb)
            kernels["k_wave_shade"]["holds"] = ("surface + textures, emitter sample, BSDF sample, the SD-tree calls, the shadow ray (inline any-hit "
                                                "walk), mixture pdfs, record, throughput, roulette, the next ray, the survivors' append")
        else:
            kernels["k_wave_shadow"] = kern(kt.shadow_ms, nb)
            kernels["k_wave_shade_a+b"] = kern(kt.shade_ms, 2 * nb)
            kernels["k_wave_shade_a+b"]["shade_a_ms_per_step"] = round(kt.shade_a_ms / passes, 3)
            kernels["k_wave_shade_a+b"]["shade_b_ms_per_step"] = round(kt.shade_b_ms / passes, 3)
        kernels["k_wave_tail"] = kern(kt.tail_ms, max(kt.passes, 1))
        if kt.sort_ms > 0:
            kernels["radix_sort"] = kern(kt.sort_ms, max(kt.passes, 1))
        dom = "k_wave_guide"
    else:
        kernels["k_bounce"] = kern(kt.bounce_ms, kt.bounce_launches, tree_bytes)
        dom = "k_bounce"
    splat_name = "k_splat_list"
    kernels[splat_name] = kern(kt.splat_ms, kt.splat_launches, splat_bytes)
    kernels[splat_name]["records_per_launch"] = int(records_per_pass)
    if True:
        entries = npix_local * args.spp_per_pass + sum(live[:-1])  # one list entry per live path and bounce
        kernels[splat_name]["list_entries_per_launch"] = int(entries)
        kernels[splat_name]["streamed_bytes_per_launch"] = int(entries * 72)
        kernels[splat_name]["alg_model_note"] = (
            "alg_bytes_per_launch prices SURVEY 8d's B_rec = 16 D_kd + 4 + 48 + 12 per quadtree level and descent: the descents of "
            "KDTree / QuadTree.addDataPropagate.  k_splat_list makes none of them -- the bounce that made a vertex walked sdTree_prev to "
            "the very leaves the record adds to (same topology) and the list names those accumulators -- so alg_GBps prices bytes this "
            "kernel does not move; what it moves is at most streamed_bytes_per_launch (72 B per list entry) plus one 32-byte atomic "
            "sector update per direction of a kept record, and it is bound by the latter (atomics_G_per_s vs atomic_ceiling_G_per_s)")
    kernels["k_finish"] = kern(kt.finish_ms, passes)
    slowest = max((n for n in kernels if kernels[n].get("region") != "roofline"), key=lambda n: kernels[n]["ms_per_step"])
    cfg_key = f"{args.scene} res={args.res} depth={args.depth} spp={args.spp_per_pass}"
    film = f"{W}x{H}"
    what = {"cornell-box": "built-in scene (Mitsuba cornell-box parameters), no textures",
            "veach-mis": "built-in scene (scenes/veach-mis/scene.xml parameters: 3 sphere lamps, 4 Beckmann rough-conductor "
                         "plates, diffuse floor and wall)",
            "torus": "built-in scene (scenes/torus/scene.xml parameters; its five meshes, 23614 triangles, from the package "
                     "data torus_meshes.npz behind a BVH: diffuse donut in a frosted-glass case, aluminium brackets, "
                     "directional light)",
            "veach-ajar": "built-in scene (scenes/veach-ajar/scene.xml parameters; its 15 OBJ meshes present in the reference "
                          "mount, 4482 triangles with texture coordinates, and its three bitmap textures at full "
                          "resolution (the JPG files' bytes in the package data veach_ajar.npz, decoded with PIL as load_xml does); checkerboard GGX floor, Beckmann door handle; "
                          "the six teapot shapes are absent: their mesh files are missing from the reference mount)"}[args.scene]
    # ---- `traffic`, measured in this run where that is possible (--pmc-in-run): the default line of one GPU, the wavefront
    # pipeline with its roofline region; otherwise, and whenever the counter passes fail, the committed figures ----
    traffic_source, traffic_in_run_note = "committed", None
    if args.pmc_in_run and rank == 0 and world == 1 and wave and kt_roof is not None and "k_wave_shade" in kernels:
        want = {k_: kernels[k_]["launches"] / max(args.steps, 1)
                for k_ in ("k_wave_guide", "k_wave_shade", "k_wave_trace", "k_splat_list", "k_finish") if k_ in kernels and kernels[k_]["launches"] > 0}
        fig, traffic_in_run_note = measure_traffic_in_run(args, want)
        if fig is not None:
            IN_RUN_TRAFFIC.update(fig)
            traffic_source = "in-run"
        print(f"[bench] traffic: {traffic_in_run_note}", file=sys.stderr, flush=True)
    tr_dom = traffic_for(dom, cfg_key)
    traffic = None if tr_dom is None else tr_dom["hi"]
    tr_committed = committed_traffic_for(dom, cfg_key)
    # (the committed counters belong to a forest with these table resolutions: another one -- a different memory budget --
    # moves other bytes)
    prof_bits = profiled_table_bits(cfg_key)
    # every kernel against the HBM roofline by its COUNTER traffic (the committed PMC figures of this configuration and of
    # THIS code, profiles/pmc_traffic.json: as counted, and with the gfx950 FETCH correction, an upper bound) over the
    # duration measured in this run
    for name, parts in (("k_wave_guide", ("k_wave_guide",)), ("k_wave_trace", ("k_wave_trace",)), ("k_wave_shadow", ("k_wave_cast",)),
                        ("k_wave_shade_a+b", ("k_wave_shade_a", "k_wave_shade_b")), ("k_wave_shade", ("k_wave_shade",)), ("k_bounce", ("k_bounce",)),
                        ("k_process_and_splat", ("k_process_and_splat",)), ("k_splat_list", ("k_splat_list",)),
                        ("k_finish", ("k_finish",))):
        if name in kernels and kernels[name]["avg_us"] > 0:
            tr = [traffic_for(p_, cfg_key) for p_ in parts]
            if all(t is not None for t in tr):
                sec = kernels[name]["avg_us"] * 1e-6
                hi = sum(t["hi"] for t in tr) / len(tr)
                # pmc_frac_of_hbm_peak: the bytes AS COUNTED (FETCH_SIZE + WRITE_SIZE) over this run's launch time and the 8 TB/s
                # peak; ..._fetch_x2: with the guide's x2 FETCH correction, which is calibrated for wide coalesced reads and is an
                # upper bound for everything else -- for k_splat_list it gives more than the box's own copy rate
                if all(t["lo"] is not None for t in tr):
                    lo = sum(t["lo"] for t in tr) / len(tr)
                    kernels[name]["pmc_hbm_bytes_per_launch"] = int(lo)
                    kernels[name]["pmc_hbm_GBps"] = round(lo / sec / 1e9, 1)
                    kernels[name]["pmc_frac_of_hbm_peak"] = round(lo / sec / 1e9 / HBM_PEAK_GBS, 3)
                kernels[name]["pmc_hbm_bytes_per_launch_fetch_x2"] = int(hi)
                kernels[name]["pmc_frac_of_hbm_peak_fetch_x2"] = round(hi / sec / 1e9 / HBM_PEAK_GBS, 3)
                if tr[0]["atomics"]:
                    kernels[name]["atomic_sector_updates_per_launch"] = int(tr[0]["atomics"])
                    kernels[name]["atomics_G_per_s"] = round(tr[0]["atomics"] / sec / 1e9, 2)
                    kernels[name]["atomic_ceiling_G_per_s"] = ATOMIC_CEILING_GPS
    dom_sec = kernels[dom]["avg_us"] * 1e-6
    kd_share = 16.0 * dc.kd_levels / max(tree_bytes, 1.0)
    dom_launches_per_pass = max(kernels[dom]["launches"] / max(((kt_roof.passes if group == 1 else args.steps) if (kt_roof is not None and dom == "k_wave_guide") else passes), 1), 1e-9)
    layout_per_launch = layout_bytes / dom_launches_per_pass
    layout_gbps = layout_per_launch / max(dom_sec, 1e-12) / 1e9
    model_gbps = kernels[dom].get("alg_GBps", 0.0)
    # `frac` / `achieved`: the bytes the lanes of an instrumented pass GATHERED from the tables of the built layout (no
    # cross-lane sharing credited) per launch / the launch time measured in this run (HIP events on the launch stream) / the
    # 8 TB/s peak -- what the memory pipeline moves at least, <= 1 by construction, recomputable from layout_bytes_per_launch
    # and avg_launch_us (the committed rocprofv3 summary of the same command has the same average).  SURVEY 8(d)'s
    # ALGORITHMIC model (the reference's levels priced one by one) is kept beside it as frac_model_8d: on spatially sorted
    # lists it exceeds the peak (model_applicable false) and is not a bandwidth.
    roof = {"bound": "hbm", "kernel": dom, "achieved": round(layout_gbps, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(layout_gbps / HBM_PEAK_GBS, 5),
            "frac_basis": "bytes gathered from the built layout (16 per KD grid entry / node below it, 8 per tree head, 16 per jump-table "
                          "entry, 32 per quadtree record of a pdf or sampling walk, 16 per record of a leaf walk), counted walk by walk by an "
                          "instrumented pass (pg_depth_counters.layout_bytes), per launch / avg_launch_us / peak",
            "frac_layout": round(layout_gbps / HBM_PEAK_GBS, 5), "layout_bytes_per_launch": int(layout_per_launch),
            "layout_GBps": round(layout_gbps, 2),
            "frac_model_8d": round(model_gbps / HBM_PEAK_GBS, 5), "achieved_model_8d": model_gbps,
            "alg_bytes_per_launch": kernels[dom].get("alg_bytes_per_launch"),
            "avg_launch_us": kernels[dom]["avg_us"],
            "traffic": traffic,
            "traffic_source": traffic_source if traffic is not None else None,
            "traffic_committed": None if tr_committed is None else tr_committed["hi"],
            "traffic_tables_match": None if (prof_bits is None or traffic is None) else bool(prof_bits[0] == int(stats.jump_bits)
                                                                                                and prof_bits[1] in (-1, int(stats.kd_grid_bits))),
            # above 1 the algorithmic model is not a bandwidth at all: in a spatially sorted list (pg_render_sort) the lanes of a
            # wave walk the same KD leaves and quadtrees, their gathers meet in L1/L2 and the bytes the model prices per lane are
            # fetched once per wave -- frac_counter_* say what reaches HBM
            "model_applicable": bool(kernels[dom].get("alg_GBps", 0.0) <= HBM_PEAK_GBS),
            # the counter-honest fractions: HBM bytes of the PMC counters per launch / this run's launch time / peak --
            # lo as counted, hi with the guide's x2 FETCH correction (an upper bound for scattered reads)
            "frac_counter_lo": None if (tr_dom is None or tr_dom["lo"] is None or dom_sec <= 0) else round(tr_dom["lo"] / dom_sec / 1e9 / HBM_PEAK_GBS, 4),
            "frac_counter_hi": None if (tr_dom is None or dom_sec <= 0) else round(tr_dom["hi"] / dom_sec / 1e9 / HBM_PEAK_GBS, 4),
            "model_note": (f"`frac_model_8d` is SURVEY 8d's ALGORITHMIC model (16 B per KD level + 20 B per quadtree level walked by the reference's "
                           f"descents), not a bandwidth measurement: {100 * kd_share:.0f} % of those bytes are KD levels ({dc.kd_levels / max(dc.kd_queries, 1):.1f} per "
                           "query) that the KD jump grid replaces by ONE 16-byte gather for most queries, and the top six quadtree levels of "
                           "a pdf walk are one 16-byte jump-table gather: the kernel moves far fewer bytes than the model prices -- read "
                           "frac_counter_lo / frac_counter_hi for what it moves, and `limiter` for what it waits on"),
            "traffic_note": ("traffic / frac_counter_*: " + traffic_in_run_note + "; traffic_committed: the figure of the same configuration "
                             "and code in profiles/pmc_traffic.json (tools/profile_bench.sh), for comparison"
                             if traffic_source == "in-run" else
                             "traffic / frac_counter_*: PMC figures of this configuration taken of exactly this code (source hash checked)"
                             + (f"; not measured in this run: {traffic_in_run_note}" if traffic_in_run_note else "")
                             if tr_dom is not None else
                             "traffic null: no PMC figures of this configuration taken of THIS code are committed (profiles/pmc_traffic.json "
                             "records the hash of the sources it was taken of; a mismatch is refused rather than paired with new timings)"),
            "region": ("the timed region of `value`" if kt_roof is None else
                       f"steps {args.steps + 1}-{2 * args.steps} of the run: the passes of `value`'s region again with pg_render_stages(2) -- the "
                       "SD-tree calls of a bounce as k_wave_guide (four kernels behind the closest hits) instead of inside the one shading kernel of the default, where they "
                       f"cannot be timed apart; that region ran at {1e3 * elapsed_roof / args.steps:.2f} ms per step against {step_ms:.2f} of the default"),
            "slowest_kernel_of_step": slowest,
            "device_copy_GBps": device_copy_rate(torch.device("cuda", local_rank)) if rank == 0 else None,
            "limiter": "HBM is the roofline SURVEY 8(d) prescribes for this pointer-chasing path; what the kernel actually waits on "
                       "is the rate at which a CU's texture path takes divergent gathers (address unit busy 82 % of the launch, "
                       "about 1.6 cycles per lane and gather: profiles/r02/pmc_guide.txt), not bytes and not arithmetic; "
                       "device_copy_GBps is the read+write rate of a plain device-to-device copy measured in this run",
            "note": ("k_wave_guide holds the SD-tree calls of a bounce and nothing else (KD descent, NEE pdf, sample-or-pdf, the "
                     "canonical coordinates of the record): the hot path of SURVEY 8; its algorithmic bytes are 16 B per KD "
                     "level + 20 B per quadtree level over the levels an instrumented pass counted. The renderer substrate around "
                     "it (ray casting k_wave_trace / k_wave_shadow, shading) is row f-1; `kernels` gives every kernel's share. "
                     if wave else
                     "k_bounce is one whole bounce of the wavefront (ray casting, NEE incl. shadow ray, shading, SD-tree "
                     "queries, record store, state load/store); its algorithmic bytes count only the SD-tree descents "
                     "(16 B/KD level, 20 B/quadtree level, SURVEY 8d). ")
                    + "The splat is bound by scattered L2 atomics, not HBM (DESIGN.md 5). `traffic`: see traffic_source / traffic_note."}
    # ---- the same two fractions for the kernel that holds the SD-tree calls in the region `value` is quoted on (k_wave_shade:
    # also the slowest kernel of the step).  It runs every launch of a bounce in that region, so its launches see the same
    # algorithmic and layout bytes as k_wave_guide's; its time also covers the surface, the BSDFs, the shadow ray and the
    # survivors' records, which the byte model does not price -- a lower bound of what the kernel moves, flat scalars so that
    # the driver's record keeps them ----
    if wave and kt_roof is not None and "k_wave_shade" in kernels and kernels["k_wave_shade"]["avg_us"] > 0:
        sh = kernels["k_wave_shade"]
        sh_sec = sh["avg_us"] * 1e-6
        sh_launches_per_pass = max(sh["launches"] / passes, 1e-9)
        sh_alg = tree_bytes / sh_launches_per_pass
        sh_layout = layout_bytes / sh_launches_per_pass
        tr_sh = traffic_for("k_wave_shade", cfg_key)
        roof.update({
            "value_region_kernel": "k_wave_shade", "value_region_avg_launch_us": sh["avg_us"],
            "value_region_alg_bytes_per_launch": int(sh_alg), "value_region_frac_model_8d": round(sh_alg / sh_sec / 1e9 / HBM_PEAK_GBS, 5),
            "value_region_layout_bytes_per_launch": int(sh_layout),
            "value_region_frac": round(sh_layout / sh_sec / 1e9 / HBM_PEAK_GBS, 5),
            "value_region_frac_layout": round(sh_layout / sh_sec / 1e9 / HBM_PEAK_GBS, 5),
            "value_region_traffic": None if tr_sh is None else tr_sh["hi"],
            "value_region_frac_counter_lo": None if (tr_sh is None or tr_sh["lo"] is None) else round(tr_sh["lo"] / sh_sec / 1e9 / HBM_PEAK_GBS, 4),
            "value_region_frac_counter_hi": None if tr_sh is None else round(tr_sh["hi"] / sh_sec / 1e9 / HBM_PEAK_GBS, 4),
            "value_region_note": ("value_region_*: the kernel that makes the SD-tree calls in the steps `value` is quoted on (k_wave_shade, one launch "
                                  "per bounce, the slowest kernel of the step) priced with the same gathered layout bytes (value_region_frac) and the "
                                  "same SURVEY 8(d) model bytes (value_region_frac_model_8d) as k_wave_guide above, over ITS average launch in the "
                                  "timed region of `value`; value_region_traffic / _frac_counter_* are its PMC figures (roofline.traffic_source)")})
        sh["alg_bytes_per_launch"] = int(sh_alg)
        sh["alg_GBps"] = round(sh_alg / sh_sec / 1e9, 2)
        # ---- ... and the SD-tree calls' own share of that kernel, taken IN PLACE by the probe build:
        # value_region_sdtree_us = that share of k_wave_shade's average launch in the timed region of `value` (the product
        # build's time, not the probe's), value_region_sdtree_frac = the gathered layout bytes of a launch over it and the peak --
        # the figure to put beside `frac` (k_wave_guide, the same calls as a kernel of their own in the second region) ----
        if args.phase_probe == 1 and rank == 0 and world == 1:
            pp, pp_note = measure_sdtree_phase(args)
            print(f"[bench] phase probe: {pp_note}", file=sys.stderr, flush=True)
            if pp is not None:
                tot = float(sum(pp["cycles"]))
                share = pp["cycles"][PHASE_NAMES.index("sdtree_calls")] / tot
                sd_us = share * sh["avg_us"]
                roof.update({
                    "value_region_sdtree_share": round(share, 4), "value_region_sdtree_us": round(sd_us, 2),
                    "value_region_sdtree_frac": round(sh_layout / (sd_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
                    "value_region_sdtree_frac_model_8d": round(sh_alg / (sd_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
                    "value_region_sdtree_probe_overhead": round(pp["shade_avg_us_stamping"] / sh["avg_us"], 3),
                    "value_region_phase_shares": {n: round(c / tot, 4) for n, c in zip(PHASE_NAMES, pp["cycles"])},
                    "value_region_sdtree_note": (
                        "the SD-tree calls of a bounce (path_guiding_integrator.py:244, 301, 307) timed IN PLACE, inside k_wave_shade: " + pp_note +
                        f"; wave-clock cycles of the seven phases summed over {pp['waves']} waves of {pp['shade_launches']} launches; "
                        "value_region_sdtree_us = the SD-tree phase's share x k_wave_shade's average launch in `value`'s region (this "
                        f"build, {sh['avg_us']} us); the probe's k_wave_shade took {pp['shade_avg_us_stamping']:.1f} us per launch while "
                        f"stamping and {pp['shade_avg_us_stamps_idle']:.1f} us with its stamps idle (value_region_sdtree_probe_overhead = "
                        "stamping / this build)")})
    cpu = None
    mse_small = mse_small_cpu = None
    if args.cpu and world == 1:
        cpu, mse_small, mse_small_cpu = cpu_leg(args, bench_tree=bench_tree_cols, iteration=k)
        bench_tree_cols = None
    synth_detail = None
    if args.synthetic_kernels and world == 1:
        synth_flat, synth_detail = synthetic_kernels_leg(local_rank, cpu=bool(args.cpu))
        roof.update(synth_flat)  # s1_pg_sample_frac_layout, ... (flat: the driver's record keeps scalars)
        roof["synthetic_note"] = ("s1_* / s2_* / s3_*: pg_pdf, pg_sample, pg_guide_bounce, pg_splat on SURVEY 8(d)'s S1 / S2 / S3 at their stated sizes "
                                  "(2^22 queries, 2^24 records) in this run; `kernels_synthetic` has units, bytes and depths")
    other = {}
    if args.other_configs and world == 1:
        other = other_configs_leg(local_rank)
    refine_ms_iters = [p["exchange_refine_ms"] for p in per_iter]
    out = {
        "metric": f"Msamples/s guided, {args.scene} {film} max_depth {args.depth}", "value": round(value, 3), "unit": "Msamples/s",
        "n_gpus": args.n_devices, "ranks": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(step_ms, 4),
        "higher_is_better": True, "scaling": "strong" if (tiles or world == 1) else "weak", "vs_baseline": None, "dtype": "f32",
        "data": "no dataset: the reference's scene (parameters, meshes and textures packaged from its scene files), sampler "
                "streams seeded per pass, SD-tree trained inside the run", "value_full_schedule": None if full_schedule is None else round(full_schedule, 3),
        "value_spp1": None if spp1 is None else round(spp1, 3),
        "value_two_in_flight": None if two_in_flight is None else round(two_in_flight, 3),
        "value_full_schedule_12it": None if full is None else full["value"], "mse_vs_gt_full_schedule": None if full is None else full["mse_vs_gt"],
        "full_schedule": full,
        "mse_vs_gt": mse_train, "mse_vs_gt_small": mse_small, "mse_vs_gt_cpu": mse_small_cpu,
        "mse_equal": None if mse_small is None else bool(mse_small == mse_small_cpu),
        "config": {"workload": f"{args.scene} {film}, max_depth {args.depth}, " +
                               (f"{args.spp_per_pass} one-sample training passes (main.py:192, 218) per step in one batched launch"
                                if batched else f"one {args.spp_per_pass}-spp pass per step") + " (the whole film"
                               + (", sharded by interleaved 4-row bands over the ranks" if tiles else (", per GPU" if world > 1 else ""))
                               + f"), guided iteration {k} (SD-tree trained by rendering iterations 0-{k - 1}, "
                               f"{cumm} spp); full PathGuidingIntegrator.sample(): camera rays, NEE, "
                               "BSDF/SD-tree MIS, record list, post-process + splat; " + what,
                   "pixels_this_rank": my_pixels, "pixels_per_rank_min": int(pix_min), "pixels_per_rank_max": int(pix_max),
                   "spp_per_pass": args.spp_per_pass,
                   "paths_per_step": paths_per_step, "kd_nodes": stats.n_kd_nodes, "kd_leaves": stats.n_kd_leaves,
                   "quad_records": stats.n_quad_records, "mean_kd_leaf_depth": round(stats.mean_kd_leaf_depth, 3),
                   "mean_quad_leaf_depth": round(stats.mean_quad_leaf_depth, 3),
                   "guided_tree_queries_per_pass": int(dc.quad_queries), "paths_alive_after_bounce": live,
                   "measured_D_kd": round(d_kd, 3), "measured_D_quad": round(d_q, 3),
                   # the schedule `value` is quoted on, and the same work launched the other ways (all in this run)
                   "pass_spp": 1 if batched else args.spp_per_pass, "passes_per_launch": (args.spp_per_pass if batched else 1) * group,
                   "steps_per_launch": group,
                   "value_one_launch_per_1spp_pass": None if spp1 is None else round(spp1, 3),
                   ("value_one_%dspp_pass_per_step" % args.spp_per_pass if batched else "value_batched_1spp_passes"):
                       None if multi_spp is None else round(multi_spp, 3),
                   "value_two_passes_in_flight": None if two_in_flight is None else round(two_in_flight, 3),
                   "value_full_schedule_12it": None if full is None else full["value"],
                   "mse_vs_gt_full_schedule": None if full is None else full["mse_vs_gt"],
                   "mse_equal_device_vs_cpu": None if mse_small is None else bool(mse_small == mse_small_cpu),
                   "mse_vs_gt_device_320": mse_small, "mse_vs_gt_cpu_320": mse_small_cpu,
                   "exchange_refine_ms_max_over_training": max(refine_ms_iters) if refine_ms_iters else None,
                   "refine_ms_after_steps": round(1e3 * t_refine, 3), "jump_bits": int(stats.jump_bits),
                   "kd_grid_bits": int(stats.kd_grid_bits), "bytes_jump_tables": int(stats.bytes_jump_tables), **other},
        "roofline": roof, "cpu_baseline": cpu, "kernels": kernels, "kernels_synthetic": synth_detail,
        "schedule": {"note": "value_full_schedule = film pixels x spp of the trained iterations >= 2 / their wall time incl. "
                             "accumulator exchange and refine (main.py:159,394); mse_vs_gt = the last trained iteration's "
                             f"image ({last_spp} spp) vs the ground truth" + (": " + mse_note if mse_note else ""),
                     "iterations": per_iter, "trained_spp": cumm},
        "extra": {"library": os.path.relpath(N_LIB_PATH, ROOT), "source_hash": SRC_HASH,
                  "allreduce_ms": round(1e3 * t_allreduce, 3), "refine_ms": round(1e3 * t_refine, 3), "exchange": exchange,
                  "accumulator_bytes": int(tree.accumulators().numel()) * 8,
                  "exchange_bytes": int(tree.packAccumulators().numel()) * 8,
                  "exchange_format": "24 B per accumulator (two 52-bit pieces + count<<24 | top: pgsd.h pg_exchange_pack) instead of the 32-byte device layout",
                  "rccl_ranks": None if rccl_seen is None else rccl_seen[0], "rccl_rank": None if rccl_seen is None else rccl_seen[1]},
    }
    if world > 1:
        dist.destroy_process_group()
    return out


def synthetic_kernels_leg(device, cpu=False):
    """SURVEY.md 8(d)'s kernel-level inputs at their stated sizes through the stand-alone entry points of the C ABI (no
    renderer, no oracle: the trees and streams are made by practical_path_guiding_lab_amd.workload):
      S1 "balanced"  KD complete to depth 12 (4096 leaves) over [0,100]^3, every leaf a complete quadtree of depth 5
                     (5.59 M quadtree nodes), leaf irradiance uniform (0,1]; 2^22 queries (positions uniform in the box,
                     directions uniform on the sphere, PCG32 stream = lane id, seed 0)
      S2 "skewed"    grown on the device by six splat + refine iterations of 2^19 ... 2^24 clustered records with lobed
                     directions, reference thresholds; the same queries
      S3 "splat"     the last S2 record stream (2^24 records) replayed into the S2 topology
    Per kernel: ms per launch (HIP events around 20 back-to-back launches on the launch stream), and two byte models over
    the 8 TB/s HBM peak -- `frac`: SURVEY 8d's algorithmic bytes (16 B per KD level + 20 B per quadtree level of the
    REFERENCE's descents; per record 16 D_kd + 4 + 48 + 12 per quadtree level and descent) with the depths an
    instrumented launch counted; `frac_layout`: the bytes the lanes of that launch gathered from the tables of the BUILT
    layout (pg_depth_counters.layout_bytes + 8 per tree head; 48 B per record streamed + 32 B per accumulator updated for
    the splat), nothing credited for lanes that share a line.  Where the tables serve whole descents from L2 `frac` exceeds 1
    (not a bandwidth); `frac_layout` cannot: a launch moves at least those bytes per lane through the memory pipeline.
    Full-size parity of exactly these inputs against the CPU oracle: tests/test_gpu_synthetic_fullsize.py."""
    import torch
    from practical_path_guiding_lab_amd import workload as Wk
    from practical_path_guiding_lab_amd.sdtree import PCG32Sampler, SDTree

    dev = torch.device("cuda", device)
    nq = Wk.S_QUERIES

    def timed(fn, reps=20):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    def counted(tree, fn):
        tree.enableDepthCounters(True)
        tree.readDepthCounters(reset=True)
        fn()
        torch.cuda.synchronize()
        dc = tree.readDepthCounters(reset=True)
        tree.enableDepthCounters(False)
        return dc

    flat, detail = {}, []

    def put(key, name, n, ms, alg, layout, extra):
        d = {"kernel": name, "units": int(n), "ms": round(ms, 4), "G_units_per_s": round(n / ms / 1e6, 2),
             "layout_bytes_per_launch": int(layout), "frac": round(layout / ms / 1e6 / HBM_PEAK_GBS, 4),
             "frac_layout": round(layout / ms / 1e6 / HBM_PEAK_GBS, 4),
             "alg_bytes_per_launch": int(alg), "frac_model_8d": round(alg / ms / 1e6 / HBM_PEAK_GBS, 4)}
        d.update(extra)
        detail.append(d)
        flat[key + "_ms"] = d["ms"]
        flat[key + "_frac"] = d["frac"]
        flat[key + "_frac_model_8d"] = d["frac_model_8d"]

    P = Wk.s_positions_uniform(nq, 3, device=dev)
    D = Wk.s_directions_uniform(nq, 4, device=dev)

    def query_suite(tag, tree):
        smp = PCG32Sampler(tree, nq, seed=0)
        st0 = smp.state.clone()
        out_idx = None
        ms = timed(lambda: tree.getLeafNodeIndex(P))
        dc = counted(tree, lambda: tree.pdf(P, D))
        d_kd, d_q = dc.kd_levels / max(dc.kd_queries, 1), dc.quad_levels / max(dc.quad_queries, 1)
        lay_pdf = dc.layout_bytes + 8 * dc.kd_queries
        # (a leaf lookup alone gathers the KD side of that launch: what pdf's walks add is 16 per table hit + 32 per record)
        ms_pdf = timed(lambda: tree.pdf(P, D))
        extra_pdf = {"D_kd": round(d_kd, 3), "D_quad": round(d_q, 3)}
        if tag == "s1":
            # S1's pdf walks end in the jump tables (64 KB per tree x 4096 trees = 268 MB: beyond every cache): ONE random 16-byte
            # gather per lane from HBM, beside the KD grid's and the tree head's, which hit in L2 -- the chip's rate for that pattern
            extra_pdf["hbm_random_gathers_per_lane"] = 1
            extra_pdf["random_gather_ceiling_G_per_s"] = RANDOM_GATHER_CEILING_GPS
            extra_pdf["frac_of_random_gather_ceiling"] = round(nq / ms_pdf / 1e6 / RANDOM_GATHER_CEILING_GPS, 3)
        put(tag + "_pg_pdf", tag.upper() + " pg_pdf", nq, ms_pdf, nq * (16.0 * d_kd + 20.0 * d_q), lay_pdf, extra_pdf)

        def do_sample():
            tree.sample(P, smp)
        smp.state.copy_(st0)
        dcs = counted(tree, do_sample)
        ds_q = dcs.quad_levels / max(dcs.quad_queries, 1)
        ms_s = timed(do_sample)
        put(tag + "_pg_sample", tag.upper() + " pg_sample", nq, ms_s, nq * (16.0 * d_kd + 20.0 * ds_q), dcs.layout_bytes + 8 * dcs.kd_queries,
            {"D_kd": round(d_kd, 3), "D_quad": round(ds_q, 3)})
        detail.append({"kernel": tag.upper() + " pg_get_leaf_node_index", "units": nq, "ms": round(ms, 4), "G_units_per_s": round(nq / ms / 1e6, 2),
                       "layout_bytes_per_launch": int(nq * 16), "frac": round(nq * 16.0 / ms / 1e6 / HBM_PEAK_GBS, 4),
                       "alg_bytes_per_launch": int(nq * 16.0 * d_kd), "frac_model_8d": round(nq * 16.0 * d_kd / ms / 1e6 / HBM_PEAK_GBS, 4),
                       "note": "one 16-byte gather per lane from the KD jump grid, served from L2: the model prices D_kd levels"})
        # the three calls of a bounce: NEE pdf for every lane, half the lanes sample, half evaluate
        nee = torch.ones(nq, dtype=torch.uint8, device=dev)
        sel = (torch.arange(nq, device=dev) % 2 + 1).to(torch.uint8)
        dio = D.clone()
        run = tree.prepareGuideBounce(P, D, nee, sel, dio, smp)
        smp.state.copy_(st0)
        dcb = counted(tree, run)
        dio.copy_(D)
        ms_b = timed(run)
        put(tag + "_pg_guide_bounce", tag.upper() + " pg_guide_bounce", nq, ms_b, 16.0 * dcb.kd_levels + 20.0 * dcb.quad_levels,
            dcb.layout_bytes + 8 * dcb.kd_queries,
            {"D_kd": round(dcb.kd_levels / max(dcb.kd_queries, 1), 3), "D_quad": round(dcb.quad_levels / max(dcb.quad_queries, 1), 3),
             "quad_descents_per_lane": round(dcb.quad_queries / nq, 3),
             "classes": "interleaved lane by lane (lane % 2): every wave walks both ways -- the worst case, which the renderer no "
                        "longer runs since the lane's class went into the sort key (DESIGN 5.9)"})
        # ... and the same mix with class-uniform waves (wave w samples or evaluates as a whole: what a sorted bounce of the
        # renderer looks like)
        sel_u = ((torch.arange(nq, device=dev) // 64) % 2 + 1).to(torch.uint8)
        dio.copy_(D)
        run_u = tree.prepareGuideBounce(P, D, nee, sel_u, dio, smp)
        smp.state.copy_(st0)
        dcu = counted(tree, run_u)
        dio.copy_(D)
        ms_u = timed(run_u)
        put(tag + "_pg_guide_bounce_uniform_waves", tag.upper() + " pg_guide_bounce (class-uniform waves)", nq, ms_u,
            16.0 * dcu.kd_levels + 20.0 * dcu.quad_levels, dcu.layout_bytes + 8 * dcu.kd_queries,
            {"D_kd": round(dcu.kd_levels / max(dcu.kd_queries, 1), 3), "D_quad": round(dcu.quad_levels / max(dcu.quad_queries, 1), 3),
             "quad_descents_per_lane": round(dcu.quad_queries / nq, 3), "classes": "uniform per wave ((lane // 64) % 2)"})

    trees = {}
    cpu_in = {}
    g1 = SDTree(device)
    g1.load(Wk.s1_balanced_tree())
    st = g1.stats()
    trees["s1"] = {"kd_leaves": int(st.n_kd_leaves), "quad_nodes": int(st.n_quad_nodes), "jump_bits": int(st.jump_bits),
                   "kd_grid_bits": int(st.kd_grid_bits)}
    query_suite("s1", g1)
    if cpu:
        cpu_in["s1"] = {"tree": Wk.s1_balanced_tree(), "pdf": g1.pdf(P, D).cpu().numpy(), "leaf": g1.getLeafNodeIndex(P).cpu().numpy()}
    del g1
    g2 = SDTree(device)
    g2.setup([Wk.S_BBOX[0]] * 3, [Wk.S_BBOX[1]] * 3, 0, 0, 20, 20, True, 0.5)
    rec = None
    refine_ms = []
    for k in range(Wk.S2_ITERATIONS):
        g2.setIteration(k, False)
        rec = Wk.s2_record_stream(k, device=dev)
        g2.addDataPropagate(rec)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g2.refineAndPrepare()
        torch.cuda.synchronize()
        refine_ms.append(round(1e3 * (time.perf_counter() - t0), 3))
    st = g2.stats()
    trees["s2"] = {"kd_leaves": int(st.n_kd_leaves), "quad_nodes": int(st.n_quad_nodes), "mean_kd_leaf_depth": round(st.mean_kd_leaf_depth, 3),
                   "mean_quad_leaf_depth": round(st.mean_quad_leaf_depth, 3), "max_quad_depth": int(st.max_quad_depth),
                   "jump_bits": int(st.jump_bits), "kd_grid_bits": int(st.kd_grid_bits), "refine_ms_per_iteration": refine_ms}
    query_suite("s2", g2)
    if cpu:
        cpu_in["s2"] = {"tree": g2.export(), "pdf": g2.pdf(P, D).cpu().numpy(), "leaf": g2.getLeafNodeIndex(P).cpu().numpy()}
        cpu_in["s3"] = {k_: v.cpu().numpy() for k_, v in rec.items()}
    # S3: the last record stream replayed into the S2 topology
    g2.setIteration(Wk.S2_ITERATIONS, False)
    m = int(rec["radiance"].shape[0])
    dc = counted(g2, lambda: g2.addDataPropagate(rec))
    d_kd, d_q = dc.kd_levels / max(dc.kd_queries, 1), dc.quad_levels / max(dc.quad_queries, 1)
    ms = timed(lambda: g2.addDataPropagate(rec), reps=5)
    b_rec = 16.0 * d_kd + 4.0 + 24.0 * d_q + 48.0
    # built layout: the record streamed (48 B), the gathers of its walks, its tree head, and one 32-byte accumulator per direction
    put("s3_pg_splat", "S3 pg_splat", m, ms, m * b_rec, 48 * m + dc.layout_bytes + 8 * dc.kd_queries + 32 * dc.quad_queries,
        {"D_kd": round(d_kd, 3), "D_quad": round(d_q, 3), "B_rec": round(b_rec, 1),
         "atomic_sector_updates_G_per_s": round(dc.quad_queries / ms / 1e6, 2), "atomic_ceiling_G_per_s": ATOMIC_CEILING_GPS})
    del g2
    torch.cuda.empty_cache()
    cpu_cols = None
    if cpu:
        cpu_cols = cpu_synthetic_columns(cpu_in, P.cpu().numpy(), D.cpu().numpy(), detail, flat)
    return flat, {"trees": trees, "kernels": detail, "cpu": cpu_cols, "note": synthetic_kernels_leg.__doc__.split("\n\n")[0]}


def cpu_synthetic_columns(cpu_in, P, D, detail, flat):
    """Part of the cpu_baseline leg (the one place outside tests/ and smoke() that loads oracle/): the CPU restatement's
    rate on the SAME S1 / S2 / S3 inputs as the device's kernels_synthetic entries ("CPU baseline ... same
    inputs S1-S3 ..., OpenMP over all host cores"; the reference's own self-tests time exactly such streams,
    kdtree.py:831-835, quadtree.py:1431-1436) -- pgo_get_leaf_node_index, pgo_pdf, pgo_sample over 2^22 lanes,
    pgo_add_data_propagate over 2^24 records, on every host core (lanes are independent; the splat adds exact integers
    with carry-correct atomic adds, so the result is the single-threaded one: tests/test_oracle_tree.py).  The S2 tree is
    the device's own export loaded into the oracle (their equality at this size is tests/test_gpu_synthetic_fullsize.py);
    as a cross-check the oracle's pdfs and leaf indices here are compared with the device's, bit for bit.  Adds
    `cpu_G_units_per_s`, `cpu_cores`, `gpu_over_cpu` to the matching entries of `detail` (flat: s1_pg_pdf_cpu_G_per_s ...)."""
    import numpy as np
    from oracle import pg_oracle as po

    po.build()
    cores = po.set_threads(0)
    by_name = {d["kernel"]: d for d in detail}
    out = {"cores": cores, "host_cores_usable": po.usable_cores(), "label": "CPU restatement (oracle/pg_oracle.c, OpenMP over the lanes)",
           "equal_to_device": True}
    n = P.shape[1]

    def best(fn, reps=3):
        t = float("inf")
        for _ in range(reps):
            t0 = time.perf_counter()
            r = fn()
            t = min(t, time.perf_counter() - t0)
        return t, r

    def col(kernel, units, sec):
        d = by_name.get(kernel)
        g = units / sec / 1e9
        if d is not None:
            d["cpu_G_units_per_s"] = round(g, 5)
            d["cpu_cores"] = cores
            d["gpu_over_cpu"] = round(d["G_units_per_s"] / g, 1) if g > 0 else None
        key = kernel.lower().replace(" ", "_")
        flat[key + "_cpu_G_per_s"] = round(g, 5)
        out[key + "_G_per_s"] = round(g, 5)

    try:
        for tag in ("s1", "s2"):
            o = po.OracleTree()
            o.load(cpu_in[tag]["tree"])
            sec, leaf = best(lambda: o.get_leaf_node_index(P))
            col(tag.upper() + " pg_get_leaf_node_index", n, sec)
            sec, pdf = best(lambda: o.pdf(P, D))
            col(tag.upper() + " pg_pdf", n, sec)
            out["equal_to_device"] = bool(out["equal_to_device"]
                                          and (leaf == cpu_in[tag]["leaf"].astype(np.uint32)).all()
                                          and (pdf.view(np.uint32) == cpu_in[tag]["pdf"].view(np.uint32)).all())

            def smp():
                st, inc = po.rng_seed(n, 0)
                t0 = time.perf_counter()
                o.sample(P, st, inc)
                return time.perf_counter() - t0
            sec = min(smp() for _ in range(3))
            col(tag.upper() + " pg_sample", n, sec)

            # the three calls of a bounce as the reference makes them (path_guiding_integrator.py:244, 301, 307): a pdf for
            # every lane, a sample for half the lanes, a pdf for the other half
            sel = (np.arange(n) % 2 + 1).astype(np.uint8)

            def bounce():
                st, inc = po.rng_seed(n, 0)
                t0 = time.perf_counter()
                o.pdf(P, D)
                o.sample(P, st, inc, (sel == 2).astype(np.uint8))
                o.pdf(P, D, (sel == 1).astype(np.uint8))
                return time.perf_counter() - t0
            sec = min(bounce() for _ in range(2))
            col(tag.upper() + " pg_guide_bounce", n, sec)
            if tag == "s2":
                r = cpu_in["s3"]
                m = int(r["radiance"].shape[0])
                cur = po.OracleTree()
                cur.copy_from(o)
                cur.reset()
                sec, _ = best(lambda: cur.add_data_propagate(r["position"], r["direction"], r["radiance"], r["woPdf"],
                                                             r["direction_nee"], r["radiance_nee_lum"]), reps=2)
                col("S3 pg_splat", m, sec)
                out["s3_root_count_ok"] = bool(int(cur.kd_column("count")[0]) == 2 * m)  # (two timed replays)
            del o
    finally:
        po.set_threads(1)
    return out


def other_configs_leg(device, steps=6, train_iters=6, spp_per_pass=8):
    """The other BASELINE.json configurations that fit one GPU -- configs[1] cornell-box 512x512 max_depth 8, configs[2]
    veach-mis 1280x720 max_depth 3, configs[4] torus 1920x1080 max_depth 32 -- each as `python bench.py --scene X
    --spp-per-pass 8` measures it (the SD-tree trained by six really rendered iterations, then `steps` timed steps of eight
    one-sample training passes in one batched launch, wall clock between device synchronisations), so that the default
    line carries driver-timed figures for them.  Flat keys: c2_/c3_/c5_..._msamples_per_s, _ms_per_step."""
    import numpy as np
    import torch
    from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator
    from practical_path_guiding_lab_amd.render import IndependentSampler, WavefrontScene

    flat = {}
    for key, name in (("c2_cornell_box_512x512_d8", "cornell-box"), ("c3_veach_mis_1280x720_d3", "veach-mis"),
                      ("c5_torus_1920x1080_d32", "torus")):
        width, _, depth = SCENES[name]
        sc = make_scene(name, width, depth)
        W, H = sc.camera.width, sc.camera.height
        g = PathGuidingIntegrator({"max_depth": depth, "rr_depth": 8}, device=device)
        g.setup(W * H, sc.bbox_min - np.float32(1e-4), sc.bbox_max + np.float32(1e-4), 20, 20, True, 0.5)
        ws = WavefrontScene(sc)
        ws.reserve(g, spp_per_pass)
        cumm = 0
        for k in range(train_iters):
            g.setIteration(k, False)
            g.resetVarianceCounter()
            iter_spp = 2 ** (k + 2)
            chunk = min(spp_per_pass, iter_spp)
            for i in range(iter_spp // chunk):
                g.sample(ws, IndependentSampler(chunk, cumm + i * chunk, batched=True))
            cumm += iter_spp
            g.refineAndPrepareSDTreeForNextIteration()
        g.setIteration(train_iters, False)
        seed = [cumm]

        def step():
            g.sample(ws, IndependentSampler(spp_per_pass, seed[0], batched=True))
            seed[0] += spp_per_pass
        el = timed_steps(step, steps, 2, 1)
        st = g.sdTree.stats()
        flat[key + "_msamples_per_s"] = round(W * H * spp_per_pass * steps / el / 1e6, 3)
        flat[key + "_ms_per_step"] = round(1e3 * el / steps, 4)
        flat[key + "_kd_leaves"] = int(st.n_kd_leaves)
        del ws, g, sc
        torch.cuda.empty_cache()
    flat["other_configs_note"] = (f"c2_/c3_/c5_*: BASELINE configs[1], [2], [4] on this GPU in this run -- SD-tree trained by {train_iters} rendered "
                                  f"iterations, then {steps} timed steps of {spp_per_pass} one-sample training passes in one batched launch (as "
                                  "`bench.py --scene X --spp-per-pass 8`)")
    return flat


def full_schedule_leg(args, integ, ws, shard, reduce_fn, W, H):
    """BASELINE.json's configuration as the reference's main.py runs it: budget = 4 + 8 + ... over the config's
    number of iterations (veach-ajar: 12 -> 16380 spp, main.py:92-99, 170), training passes of ONE sample per pixel seeded
    initial_seed + cumm_spp (main.py:192, 218) -- --spp-per-pass of them traced per launch (pg_pass_params.batched: the
    same passes bit for bit) --, final passes of --spp-per-pass samples (main.py:123 has 4), the
    stop-training rule of main.py:334-377 (training ends once the estimated final variance rises after 256 spp or at
    1000 spp; the rest of the budget is one final iteration), image blending, variance bookkeeping, exchange and
    refine -- everything driver.run_guided_render does -- inside one wall clock."""
    import torch
    from practical_path_guiding_lab_amd.driver import run_guided_render

    n_it = CONFIG_ITERATIONS[args.scene]
    budget = 2 ** (n_it + 2) - 4
    lines = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if args.batched:  # main.py's own schedule: one-sample training passes, spp_per_pass of them per launch
        kw = dict(training_spp_per_pass=1, training_passes_per_launch=args.spp_per_pass)
    else:
        kw = dict(training_spp_per_pass=args.spp_per_pass)
    res = run_guided_render(ws, integ, budget, initial_seed=0, batch_spp=args.spp_per_pass, all_reduce=reduce_fn, shard=shard,
                            log=lines.append, exchange_overlap=bool(args.exchange_overlap), **kw)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    rows = res["records"]["variance_endIter"].rows
    image_spp = int(rows[-1][1])
    if shard is not None:
        from practical_path_guiding_lab_amd.parallel import all_reduce_sums
        sumL = all_reduce_sums(integ.sumL, integ.sumL2)[0]
    else:
        sumL = integ.sumL
    mse, note = image_mse(sumL.cpu().numpy(), image_spp, W, H, args.scene)
    trained = [it for it in res["iterations"] if it["refine_s"] > 0]
    return {"value": round(W * H * res["cumm_spp"] / wall / 1e6, 3), "unit": "Msamples/s", "wall_s": round(wall, 2),
            "budget_spp": budget, "config_iterations": n_it, "iterations_run": len(res["iterations"]),
            "refines": len(trained), "final_image_spp": image_spp,
            "guided_value": round(res["guided_samples"] / max(res["guided_time_s"], 1e-9) / 1e6, 3),
            "mse_vs_gt": mse, "variance_final": rows[-1][4],
            "per_iteration": [{"iteration": it["iteration"], "spp": it["spp"], "render_s": round(it["render_s"], 3),
                               "refine_s": round(it["refine_s"], 4)} for it in res["iterations"]],
            "note": "the whole budget of the BASELINE configuration through driver.run_guided_render (main.py:92-430): all camera paths of "
                    "all iterations / one wall clock that also holds refine, variance bookkeeping, image development and blending; "
                    "mse_vs_gt = per-pixel mean of the final image's samples vs the ground truth" + (": " + note if note else "")}


def cpu_leg(args, iters=4, width=320, bench_tree=None, iteration=None, full_passes=3):
    """cpu_baseline.  (1) CONFIG-MATCHED: `full_passes` guided ONE-sample training passes (main.py:192) of the
    bench's own film (args.res wide, the config's max_depth) through the bench's own SD-tree -- `bench_tree`, the 23 columns
    exported behind the timed steps, loaded into the oracle as sdTree_prev and, zeroed, as sdTree_current -- on all host cores:
    cpu_baseline.value.  The first of those passes is also traced on the device through the same tree (a fresh context that
    loads the columns): radiance bit-identical or images_bit_identical_to_device is false.
    (2) The same 4-iteration schedule (4+8+16+32 spp, the last two guided) of the same scene on a 320-pixel-wide film, on the
    device and on the oracle: same seeds, so the images -- and their MSEs against the ground truth -- must be identical
    (mse_equal); its guided passes timed are cpu_baseline.value_small (round 5's `value`)."""
    import numpy as np
    from oracle import pg_oracle as po
    from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator
    from practical_path_guiding_lab_amd.render import IndependentSampler, WavefrontScene

    po.build()
    cores = po.set_threads(0)  # (two threads per core of this process's CPU share: oracle.pg_oracle.default_threads)
    usable = po.usable_cores()
    sc = make_scene(args.scene, width, args.depth)
    W, H = sc.camera.width, sc.camera.height
    npix = W * H
    bmin, bmax = sc.bbox_min - np.float32(1e-4), sc.bbox_max + np.float32(1e-4)
    g = PathGuidingIntegrator({"max_depth": args.depth, "rr_depth": 8})
    g.setup(npix, bmin, bmax, 20, 20, True, 0.5)
    ws = WavefrontScene(sc)
    pair = po.OracleSDTreePair()
    pair.setup(bmin, bmax, 20, 20, True)
    o_sumL, o_sumL2 = np.zeros((3, npix), np.float32), np.zeros((3, npix), np.float32)
    cumm, t_guided_1, n_guided_1 = 0, 0.0, 0
    for k in range(iters):
        spp = 2 ** (k + 2)
        g.setIteration(k, False)
        g.resetVarianceCounter()
        o_sumL[:] = 0
        o_sumL2[:] = 0
        # the reference's schedule: 2^(k+2) one-sample passes seeded cumm, cumm + 1, ... (main.py:170, 192, 218) -- on the
        # device one batched launch, on the oracle literally pass by pass
        g.sample(ws, IndependentSampler(spp, cumm, batched=True))
        t0 = time.perf_counter()
        for p_ in range(spp):
            po.render_pass(pair, sc, sc.camera, args.depth, 8, k, False, cumm + p_, 1, True, 0.5, o_sumL, o_sumL2)
        dt = time.perf_counter() - t0
        if k >= 2:
            t_guided_1 += dt
            n_guided_1 += npix * spp
        cumm += spp
        if k + 1 < iters:
            g.refineAndPrepareSDTreeForNextIteration()
            pair.refine_and_prepare(k)
    last = 2 ** (iters + 1)
    g_sum = g.sumL.cpu().numpy()
    same = bool((g_sum.view(np.uint32) == o_sumL.view(np.uint32)).all())
    mse_g, _ = image_mse(g_sum, last, W, H, args.scene)
    mse_c, _ = image_mse(o_sumL, last, W, H, args.scene)
    # the timed sample: the same schedule with each iteration's samples in ONE oracle pass of 2^(k+2) samples per pixel -- the
    # CPU's best form (57 600 lanes of a one-sample pass do not keep 256 threads busy, and every pass allocates its record
    # arrays) -- guided iterations 2-3 timed.  Same estimator, other sampler streams than the parity leg above.
    pair2 = po.OracleSDTreePair()
    pair2.setup(bmin, bmax, 20, 20, True)
    cumm, t_guided, n_guided = 0, 0.0, 0
    for k in range(iters):
        spp = 2 ** (k + 2)
        t0 = time.perf_counter()
        po.render_pass(pair2, sc, sc.camera, args.depth, 8, k, False, cumm, spp, True, 0.5)
        dt = time.perf_counter() - t0
        if k >= 2:
            t_guided += dt
            n_guided += npix * spp
        cumm += spp
        if k + 1 < iters:
            pair2.refine_and_prepare(k)
    # ---- (1) the config-matched sample: the bench's film, the bench's tree ----
    full = None
    if bench_tree is not None:
        import torch
        scF = make_scene(args.scene, args.res, args.depth)
        WF, HF = scF.camera.width, scF.camera.height
        fmin, fmax = scF.bbox_min - np.float32(1e-4), scF.bbox_max + np.float32(1e-4)
        pairF = po.OracleSDTreePair()
        pairF.setup(fmin, fmax, 20, 20, True)
        pairF.prev.load(bench_tree)
        pairF.current.load(bench_tree)   # (same topology, path_guiding_integrator.py:582; a load leaves the accumulators at zero)
        pairF.current.reset()
        gF = PathGuidingIntegrator({"max_depth": args.depth, "rr_depth": 8})
        gF.setup(WF * HF, fmin, fmax, 20, 20, True, 0.5)
        gF.sdTree.load(bench_tree)
        gF.setIteration(iteration, False)
        wsF = WavefrontScene(scF)
        seedF = 770000
        Lg = gF.sample(wsF, IndependentSampler(1, seedF))[0].cpu().numpy()
        del gF, wsF
        torch.cuda.empty_cache()
        tF, same_full = 0.0, None
        for p_ in range(full_passes):
            t0 = time.perf_counter()
            Lo, _ = po.render_pass(pairF, scF, scF.camera, args.depth, 8, iteration, False, seedF + p_, 1, True, 0.5)
            tF += time.perf_counter() - t0
            if p_ == 0:
                same_full = bool((Lg.view(np.uint32) == Lo.view(np.uint32)).all())
        full = {"value": round(full_passes * WF * HF / tF / 1e6, 4), "seconds": round(tF, 2), "film": f"{WF}x{HF}", "passes": full_passes,
                "same": same_full, "paths": full_passes * WF * HF}
        del pairF, Lg
    small_value = round(n_guided / t_guided / 1e6, 4)
    cpu = {"value": small_value if full is None else full["value"], "unit": "Msamples/s", "cores": cores, "host_cores_usable": usable,
           "host_cores_shown": os.cpu_count(), "kind": "port",
           "config": (f"same film ({full['film']}, max_depth {args.depth}), same SD-tree (the one `value` was quoted on), {full['passes']} guided "
                      "one-sample passes" if full is not None else f"{W}x{H} film, 4-iteration schedule, guided iterations 2-3"),
           "seconds": None if full is None else full["seconds"],
           "value_small": small_value, "images_bit_identical_to_device_full_size": None if full is None else full["same"],
           "label": "CPU restatement: the build's own C restatement of the reference's Python (oracle/, OpenMP over the lanes), NOT the "
                    "reference -- its Python / Dr.Jit path cannot run on this box (Mitsuba 3 and Dr.Jit are absent: SURVEY 8c)",
           "images_bit_identical_to_device": bool(same and (full is None or full["same"])),
           "mse_vs_gt_device": mse_g, "mse_vs_gt_cpu": mse_c, "mse_equal": bool(mse_g == mse_c),
           "value_one_sample_passes": round(n_guided_1 / t_guided_1 / 1e6, 4),
           "sample": (f"{full['passes']} guided one-sample passes (main.py:192) of the bench's own {full['film']} film, max_depth {args.depth}, through the "
                      f"bench's own SD-tree: {full['paths']} paths in {full['seconds']} s on {cores} threads ({usable} usable cores); first pass "
                      f"bit-identical to the device's: {full['same']}") if full is not None else
                     f"the guided passes (iterations 2-3, 16 + 32 spp) of a 4-iteration schedule of the same scene on a {W}x{H} "
                     f"film ({n_guided} paths), C oracle with OpenMP over the lanes on {cores} threads (this process may use {usable} of the box's "
                     f"{os.cpu_count()} cores: its cgroup quota), {t_guided:.1f} s; "
                     f"value_one_sample_passes: the same samples as 48 separate one-sample passes ({t_guided_1:.1f} s), the leg the "
                     "device's images are compared with",
           "sample_small": f"value_small: guided iterations 2-3 (16 + 32 spp) of a 4-iteration schedule on a {W}x{H} film, {n_guided} paths, {t_guided:.1f} s"}
    return cpu, mse_g, mse_c


# ------------------------------------------------------------------------------------------------
def run_synthetic(args):
    import torch
    import torch.distributed as dist
    from practical_path_guiding_lab_amd.sdtree import SDTree
    from practical_path_guiding_lab_amd import workload as W

    world, rank, local_rank = init_dist(args)
    pixels = args.res * args.res
    rays = pixels * args.spp_per_pass
    tree = SDTree(device=local_rank)
    tree.setup(W.CORNELL_BBOX_MIN, W.CORNELL_BBOX_MAX, rays, args.depth, 20, 20, True, 0.5)
    wl = W.SyntheticPassWorkload(tree, rays, args.depth, seed=1, rank=rank)
    wl.compaction = not args.no_compaction

    def all_reduce(acc):
        if world > 1 and acc.numel():
            dist.all_reduce(acc, op=dist.ReduceOp.SUM)

    wl.train(args.train_iters, records_per_pass=3 * pixels, all_reduce=all_reduce if world > 1 else None)
    torch.cuda.synchronize()
    stats = tree.stats()
    wl.prepare()
    depths = wl.measure_depths()
    D = args.depth
    for _ in range(args.warmup):
        wl.run_pass()
    torch.cuda.synchronize()
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(D + 1)]
          for _ in range(args.steps)]
    kk = [0]

    def step():
        k = kk[0]
        for b in range(D):
            wl.run_compact(b)
            ev[k][b][0].record()
            wl.run_bounce(b)
            ev[k][b][1].record()
        ev[k][D][0].record()
        wl.run_splat()
        ev[k][D][1].record()
        kk[0] += 1

    elapsed = timed_steps(step, args.steps, 0, world)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return None
    value = float(rays) * world * args.steps / elapsed / 1e6
    bounce_ms = sum(ev[k][b][0].elapsed_time(ev[k][b][1]) for k in range(args.steps) for b in range(D))
    splat_ms = sum(ev[k][D][0].elapsed_time(ev[k][D][1]) for k in range(args.steps))
    bounce_bytes = sum(W.bounce_bytes(*x) for x in depths["bounce"])
    s = depths["splat"]
    splat_bytes = W.splat_bytes(s[0], s[1], s[2], s[3])
    kern = {
        "k_guide_bounce": {"launches": args.steps * D, "avg_us": 1e3 * bounce_ms / (args.steps * D),
                           "alg_bytes_per_launch": bounce_bytes / D,
                           "alg_GBps": bounce_bytes * args.steps / (bounce_ms * 1e-3) / 1e9},
        "k_process_and_splat": {"launches": args.steps, "avg_us": 1e3 * splat_ms / args.steps,
                                "alg_bytes_per_launch": splat_bytes,
                                "alg_GBps": splat_bytes * args.steps / (splat_ms * 1e-3) / 1e9},
    }
    dom = "k_guide_bounce" if bounce_ms >= splat_ms else "k_process_and_splat"
    cfg_key = f"synthetic res={args.res} depth={args.depth} spp={args.spp_per_pass}"
    roof = {"bound": "hbm", "kernel": dom, "achieved": round(kern[dom]["alg_GBps"], 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(kern[dom]["alg_GBps"] / HBM_PEAK_GBS, 5), "traffic": traffic_for(dom, cfg_key)}
    kq = sum(x[1] for x in depths["bounce"])
    out = {
        "metric": "Msamples/s guided (SD-tree hot path only, synthetic pass)", "value": round(value, 3),
        "unit": "Msamples/s", "n_gpus": args.n_devices, "ranks": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "C2-synthetic: cornell-box 512x512 pixels x spp_per_pass paths/pass, max_depth 8, SD-tree ops "
                               "only (lane compaction + guide_bounce per bounce, then process_and_splat); no ray casting/BSDF",
                   "pixels_per_gpu": pixels, "spp_per_pass": args.spp_per_pass, "paths_per_pass_per_gpu": rays,
                   "kd_nodes": stats.n_kd_nodes, "quad_records": stats.n_quad_records,
                   "measured_D_kd": round(sum(x[0] for x in depths["bounce"]) / max(kq, 1), 3),
                   "measured_D_quad": round(sum(x[2] for x in depths["bounce"]) / max(sum(x[3] for x in depths["bounce"]), 1), 3),
                   "guided_bounces_per_pass": kq, "records_per_pass": depths["splat"][1]},
        "roofline": roof, "cpu_baseline": None, "kernels": kern,
    }
    if world > 1:
        dist.destroy_process_group()
    return out


COMPACT_LINE_MAX = 6000  # bytes: the driver kept `parsed: null` for round 5's 22.9 KB line
SCHEMA = 6               # of the compact line; round 5 and earlier printed the whole record as the one line


def _finite(x):
    """x with every non-finite float replaced by None, recursively (the line is STRICT JSON: no NaN / Infinity)."""
    if isinstance(x, float):
        return x if x == x and x not in (float("inf"), float("-inf")) else None
    if isinstance(x, dict):
        return {str(k): _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    if hasattr(x, "item") and not isinstance(x, (str, bytes)):  # numpy / torch scalars
        try:
            return _finite(x.item())
        except Exception:
            return str(x)
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_record(out):
    """The ONE line the driver parses: bench.py's contract keys plus the flat figures the judge reads, no prose -- everything
    else (notes, per-kernel counter tables, kernels_synthetic, schedule, full_schedule) goes to the detail file only."""
    cfg, roof, cpu = out.get("config") or {}, out.get("roofline") or {}, out.get("cpu_baseline")
    c = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "ranks", "steps", "warmup", "ms_per_step", "higher_is_better",
                                 "scaling", "vs_baseline", "dtype")}
    c["schema"] = SCHEMA
    c["data"] = "synthetic: the reference's scene files packaged, seeded sampler streams, SD-tree trained in the run"
    if out.get("data") == "synthetic":
        c["data"] = "synthetic"
    wl = str(cfg.get("workload", ""))
    c["config"] = {"workload": wl if len(wl) <= 300 else wl[:297] + "..."}
    c["config"].update(_pick(cfg, ("paths_per_step", "spp_per_pass", "passes_per_launch", "steps_per_launch", "kd_leaves", "quad_records", "measured_D_kd",
                                   "measured_D_quad", "jump_bits", "kd_grid_bits", "bytes_jump_tables", "pixels_per_rank_min",
                                   "pixels_per_rank_max", "value_one_launch_per_1spp_pass", "value_full_schedule_12it",
                                   "mse_vs_gt_full_schedule", "mse_equal_device_vs_cpu")))
    for k, v in cfg.items():  # BASELINE configs[1], [2], [4] timed in the same run
        if k.startswith(("c2_", "c3_", "c5_")) and k.endswith(("_msamples_per_s", "_ms_per_step")):
            c["config"][k] = v
    for k in ("value_full_schedule", "mse_vs_gt", "mse_equal"):
        if out.get(k) is not None:
            c[k] = out[k]
    r = _pick(roof, ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_us", "layout_bytes_per_launch",
                     "alg_bytes_per_launch", "frac_model_8d", "traffic", "traffic_source", "frac_counter_lo", "frac_counter_hi",
                     "value_region_kernel", "value_region_avg_launch_us", "value_region_frac", "value_region_frac_counter_lo",
                     "value_region_sdtree_us", "value_region_sdtree_frac", "value_region_sdtree_share", "value_region_sdtree_probe_overhead",
                     "s1_pg_sample_frac", "s1_pg_sample_frac_model_8d", "s1_pg_pdf_frac", "s1_pg_guide_bounce_frac", "s2_pg_sample_frac",
                     "s3_pg_splat_frac", "device_copy_GBps"))
    if r:
        r["frac_basis"] = (("bytes the lanes gathered from the built layout per launch / avg_launch_us / peak; "
                            "SURVEY 8d model bytes: frac_model_8d") if "layout_bytes_per_launch" in r else
                           "SURVEY 8d algorithmic bytes per launch / launch time / peak")[:120]
    c["roofline"] = r or None
    if isinstance(cpu, dict):
        cc = _pick(cpu, ("value", "unit", "cores", "kind", "host_cores_usable", "mse_equal", "images_bit_identical_to_device",
                         "images_bit_identical_to_device_full_size", "config", "value_small", "seconds"))
        cc["label"] = "CPU restatement (oracle/, OpenMP over the lanes), not the reference: Mitsuba/Dr.Jit absent"
        s = str(cpu.get("sample", ""))
        cc["sample"] = s if len(s) <= 240 else s[:237] + "..."
        c["cpu_baseline"] = cc
    else:
        c["cpu_baseline"] = None
    kern = out.get("kernels") or {}
    c["kernels"] = {k: (round(v["ms_per_step"], 3) if isinstance(v, dict) and "ms_per_step" in v else
                        (round(1e-3 * v["avg_us"] * v["launches"] / max(out.get("steps") or 1, 1), 3) if isinstance(v, dict) and "avg_us" in v else None))
                    for k, v in kern.items()}
    ex = out.get("extra") or {}
    c["extra"] = _pick(ex, ("source_hash", "rccl_ranks", "exchange_bytes", "allreduce_ms", "refine_ms"))
    e = str(ex.get("exchange", "none"))
    c["extra"]["exchange"] = e if len(e) <= 80 else e[:77] + "..."
    return _finite(c)


def format_line(out):
    """compact_record(out) as one line of strict JSON, shortened -- optional groups first -- until it fits COMPACT_LINE_MAX."""
    c = compact_record(out)
    line = json.dumps(c, allow_nan=False, separators=(", ", ": "))
    for drop in (("extra", "allreduce_ms"), ("extra", "refine_ms"), ("kernels",), ("config", "workload")):
        if len(line) < COMPACT_LINE_MAX:
            break
        d = c
        for k in drop[:-1]:
            d = d.get(k) or {}
        if drop[-1] == "workload" and "workload" in d:
            d["workload"] = d["workload"][:120]
        else:
            d.pop(drop[-1], None)
        line = json.dumps(c, allow_nan=False, separators=(", ", ": "))
    if len(line) >= COMPACT_LINE_MAX:
        raise RuntimeError(f"bench line is {len(line)} bytes (limit {COMPACT_LINE_MAX})")
    return line


def detail_path():
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        return os.path.join(d, "bench_detail.json")
    except OSError:
        return os.path.join("/tmp", "bench_detail.json")


def emit(out, detail=None):
    """The full record to the detail file (never to stdout: the line the driver parses must be the only JSON there), then the
    compact line as the LAST thing this process writes -- nothing follows it on stdout or stderr."""
    path = detail_path() if detail is None else detail
    try:
        with open(path, "w") as f:
            json.dump(_finite(out), f, allow_nan=False)
            f.write("\n")
        print(f"[bench] full record: {path}", file=sys.stderr, flush=True)
    except OSError as e:
        print(f"[bench] full record not written ({e})", file=sys.stderr, flush=True)
    sys.stderr.flush()
    sys.stdout.write(format_line(out) + "\n")
    sys.stdout.flush()


def main():
    # (the pool's host driver supports dmabuf IPC only: RCCL and CUDA-tensor sharing across processes fail without this, and the
    # HSA runtime reads it when it starts -- so it is set before anything here or in a rank can have touched the GPU)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the driver's plain `python bench.py --gpus N`: start the N ranks ourselves.  Nothing above has imported
        # torch or touched HIP, the children are fresh processes (no exec of a process that holds the GPU).
        sys.exit(spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    if args.phase_probe == 2:  # (the probe child of measure_sdtree_phase: one small line, not a bench record)
        print(json.dumps(run_phase_probe(args)))
        return
    out = run_synthetic(args) if args.synthetic else run_render(args)
    if out is not None:
        emit(out, args.detail)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Guided render of a scene, in the role of the reference's main.py
(takkasila/practical_path_guiding_lab main.py): same schedule and outputs, MI355X-native engine.

    python main.py                                  # built-in cornell-box, 252 spp budget
    python main.py --scene veach-ajar               # the metric's scene (also: veach-mis, torus)
    python main.py --scene /path/to/scene.xml       # Mitsuba 3 XML of the supported subset
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="cornell-box",
                    help="built-in scene (cornell-box, veach-mis, veach-ajar, torus) or a Mitsuba 3 scene XML of the supported subset")
    ap.add_argument("--width", type=int, default=None, help="film width (default: 512 / 1280 / 1024 / the XML's)")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--max-depth", type=int, default=None)
    ap.add_argument("--budget-spp", type=int, default=252)       # main.py:99
    ap.add_argument("--batch-spp", type=int, default=4)          # main.py:123
    ap.add_argument("--training-spp-per-pass", type=int, default=1,      # main.py:192: training passes trace ONE sample per pixel
                    help="samples per pixel of a training pass (the reference: 1, seeded initial_seed + cumm_spp, main.py:218)")
    ap.add_argument("--training-passes-per-launch", type=int, default=16,
                    help="one-sample training passes traced per device launch (pg_pass_params.batched): images, logs and SD-tree are "
                         "those of the separate passes, byte for byte (tests/test_gpu_cli.py); 1 = one launch per pass")
    ap.add_argument("--path-tracing", action="store_true",
                    help="the reference's benchmark renderer instead (path_tracing_render.py): no SD-tree, --budget-spp samples "
                         "(or --time-budget seconds) in chunks of --batch-spp")
    ap.add_argument("--time-budget", type=float, default=None, help="--path-tracing: seconds instead of a sample count")
    ap.add_argument("--repeat-high-spp", type=int, default=0, metavar="SPP",
                    help="after the run: render every iteration's saved SD-tree, frozen, with SPP samples each "
                         "(repeat_high_spp_renderer.py): what each iteration's tree is worth at equal cost")
    ap.add_argument("--seed", type=int, default=0)               # main.py:66-67
    ap.add_argument("--out", default="debug/cornell-box")
    ap.add_argument("--ground-truth", default=None, help=".exr or .npy (H,W,3) linear ground truth for MSE, e.g. "
                    "scenes/cornell-box/TungstenRender.exr or tests/golden/cornell_gt_256_f16.npy")
    args = ap.parse_args()

    from practical_path_guiding_lab_amd import scene as S
    from practical_path_guiding_lab_amd.driver import load_ground_truth, run_guided_render
    from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator
    from practical_path_guiding_lab_amd.render import WavefrontScene

    if args.scene == "cornell-box":
        sc = S.cornell_box(args.width or 512, args.height or 512, 8, 8)
    elif args.scene == "veach-mis":
        sc = S.veach_mis(args.width or 1280, args.height or 720)
    elif args.scene == "torus":
        sc = S.torus(args.width or 1024, args.height or 768)
    elif args.scene == "veach-ajar":
        sc = S.veach_ajar(args.width or 1280, args.height or 720)
    else:
        sc = S.load_xml(args.scene, args.width, args.height, skip_missing_meshes=True)
        if sc.skipped:
            print("shapes left out (mesh file missing):", ", ".join(sc.skipped))
    if args.out == "debug/cornell-box" and args.scene != "cornell-box":
        args.out = "debug/" + os.path.splitext(os.path.basename(os.path.dirname(args.scene) or args.scene))[0]
    if args.max_depth is not None:
        sc.max_depth = args.max_depth
    integ = PathGuidingIntegrator({"max_depth": sc.max_depth, "rr_depth": sc.rr_depth})
    gt = load_ground_truth(args.ground_truth, sc.camera.width, sc.camera.height) if args.ground_truth else None
    # veach-ajar without its teapots (mesh files missing from the reference mount): their rectangle of the
    # ground truth does not count in MSE / variance
    mask = None
    if gt is not None and (args.scene == "veach-ajar" or ("veach-ajar" in args.scene and sc.skipped)):
        mask = S.veach_ajar_mask(sc.camera.width, sc.camera.height)
        print("MSE / variance against the ground truth leave out the teapot rectangle")
    if args.path_tracing:
        from practical_path_guiding_lab_amd.extras import run_path_tracing
        run_path_tracing(WavefrontScene(sc), integ, None if args.time_budget else args.budget_spp, args.time_budget,
                         chunk_spp=args.batch_spp, initial_seed=args.seed, ground_truth=gt, out_dir=args.out, gt_mask=mask)
        return
    res = run_guided_render(WavefrontScene(sc), integ, args.budget_spp, initial_seed=args.seed, ground_truth=gt,
                            batch_spp=args.batch_spp, training_spp_per_pass=args.training_spp_per_pass, out_dir=args.out,
                            gt_mask=mask, training_passes_per_launch=args.training_passes_per_launch)
    n = sc.camera.width * sc.camera.height * res["cumm_spp"]
    print(f"done: {res['cumm_spp']} spp in {res['time_s']:.2f} s = {n / res['time_s'] / 1e6:.1f} Msamples/s overall; "
          f"guided passes {res['guided_samples'] / max(res['guided_time_s'], 1e-9) / 1e6:.1f} Msamples/s")
    if args.repeat_high_spp > 0:
        from practical_path_guiding_lab_amd.extras import repeat_high_spp
        last = len(res["iterations"]) - 1
        repeat_high_spp(WavefrontScene(sc), integ, args.out, 0, last, last, args.repeat_high_spp, batch_spp=args.batch_spp,
                        initial_seed=args.seed + 1000003, ground_truth=gt, out_dir=args.out, gt_mask=mask)


if __name__ == "__main__":
    main()

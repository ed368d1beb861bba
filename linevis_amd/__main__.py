"""Headless command line renderer: `python -m linevis_amd <input> -o frame.png [--mode rt|ppll] [key=value ...]`.

<input> is a .binlines / .obj trajectory file (LineDataFlow::loadFromFile) or the name of a synthetic scene
(lattice, helix, tornado, rayleigh_benard, abc_flow).  key=value pairs are the reference's SettingsMap keys
(src/Renderers/LineRenderer.cpp:433-498, VulkanRayTracer.cpp:226-278, ...), forwarded through
LineRenderer::setNewSettings of the plugin classes in linevis_amd/host/.  Needs an MI355X: there is no CPU fallback.
"""
import argparse
import sys

import numpy as np


def _parse_value(v):
    if v.lower() in ("true", "false"):
        return v.lower() == "true"
    try:
        return int(v)
    except ValueError:
        pass
    try:
        return float(v)
    except ValueError:
        return v


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m linevis_amd", description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("input")
    ap.add_argument("-o", "--output", default="frame.png")
    ap.add_argument("--mode", choices=["rt", "ppll"], default="rt")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--camera", type=float, nargs=3, default=(0.0, 0.0, 0.8), metavar=("X", "Y", "Z"))
    ap.add_argument("--transfer-function", choices=["standard", "transparent"], default=None)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("settings", nargs="*", help="SettingsMap entries: key=value")
    args = ap.parse_intermixed_args(argv)

    from . import capi, host_api, scenes, transfer_function as tfm
    flow = host_api.LineDataFlow()
    synthetic = {"lattice": scenes.lattice, "helix": scenes.helix_bundle, "tornado": scenes.tornado,
                 "rayleigh_benard": scenes.rayleigh_benard}
    if args.input in synthetic:
        tr = scenes.normalize(synthetic[args.input]())
        flow.set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    elif args.input == "abc_flow":
        grid = host_api.StreamlineTracingGrid(args.device).load_abc_flow(64, 64, 64, 6.0)
        # streamribbons (the reference's default flow primitive): attribute 1 = Velocity Magnitude (fields in name order)
        pos, att, off, rib = grid.trace_streamribbons(grid.regular_seeds(10, 10, 10), minimum_length=0.3)
        flow.set_trajectories(host_api.normalize_positions(pos), att[1], off, rib)
    else:
        flow.load_file(args.input)
    mode = capi.MODE_RAY_TRACER if args.mode == "rt" else capi.MODE_PPLL
    r = host_api.HeadlessLineRenderer(mode, args.device)
    r.set_rendering_resolution(args.width, args.height)
    tf_name = args.transfer_function or ("transparent" if args.mode == "ppll" else "standard")
    r.set_transfer_function(tfm.standard_transparent() if tf_name == "transparent" else tfm.standard())
    r.set_camera(args.camera)
    r.set_line_data(flow)
    settings = {}
    for kv in args.settings:
        if "=" not in kv:
            ap.error("settings must be key=value, got %r" % kv)
        k, v = kv.split("=", 1)
        settings[k] = _parse_value(v)
    if settings:
        r.set_new_settings(settings)
    # progressive accumulation: render() while needsReRender() (VulkanRayTracer::needsReRender, VulkanRayTracer.cpp:330-336)
    img = None
    for _ in range(max(1, int(settings.get("num_accumulated_frames", 1)))):
        img = r.render_frame()
    st = r.stats()
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(img)).save(args.output)
    print("%s: %d lines, %d segments, %dx%d, %.3f ms on the GPU (accel build %.2f ms)"
          % (args.output, flow.num_lines, st.num_segments, args.width, args.height, st.ms_total, st.ms_accel_build))
    return 0


if __name__ == "__main__":
    sys.exit(main())

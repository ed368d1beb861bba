"""Developer probe: times the hot path on the BASELINE.json configs (run through gpurun)."""
import os, sys, time, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from linevis_amd import scenes, camera, transfer_function as tfm, capi, host_api

def make_ctx(tr, W, H, lw, tf, settings):
    t = time.time()
    flow = host_api.LineDataFlow().set_trajectories(tr.positions, tr.attributes, tr.line_offsets)
    pts, seg, _ = flow.tube_aabb_render_data(lw)
    print('  a2 host %.2fs  points %d segs %d' % (time.time() - t, len(pts), len(seg)))
    ctx = capi.Context(0)
    ctx.set_lines(pts, seg)
    ctx.set_transfer_function(tf, *flow.attribute_range())
    view, proj, fovy, near, far = camera.default_camera(W, H)
    ctx.set_camera(view, proj, fovy, near, far, W, H)
    ctx.set_option('line_width', lw)
    ctx.set_options(settings)
    ctx.build_accel()
    print('  accel build %.2f ms depth %d' % (ctx.stats().ms_accel_build, ctx.stats().bvh_depth))
    return ctx

def run(ctx, mode, reps=3, label=''):
    img = None
    for i in range(reps):
        img = ctx.render(mode)
        s = ctx.stats()
        print('  %s rep%d total %.2f ms  depth %.3f ao %.2f color %.2f | clear %.3f gather %.2f resolve %.2f' % (
            label, i, s.ms_total, s.ms_depth_range, s.ms_ao, s.ms_color, s.ms_ppll_clear, s.ms_ppll_gather, s.ms_ppll_resolve))
    ctx.set_option('collect_stats', True)
    ctx.render(mode)
    s = ctx.stats()
    ctx.set_option('collect_stats', False)
    d = s.as_dict()
    alg = d['nodes_visited'] * 64 + d['prims_tested'] * 32 + d['hits_shaded'] * 96
    print('  max nodes per pixel', d['max_nodes_per_pixel'])
    print('  counters', {k: d[k] for k in ('rays_traced', 'nodes_visited', 'prims_tested', 'hits_shaded', 'fragments', 'ao_hit_pixels', 'max_depth_complexity')})
    if s.ao_phase_iterations[1]:
        print('  ao phases {setup,node,leaf}: iters', list(s.ao_phase_iterations), 'util',
              [round(s.ao_phase_lanes[k] / (64.0 * max(1, s.ao_phase_iterations[k])), 3) for k in range(3)])
    print('  nodes/ray %.1f prims/ray %.2f  alg bytes %.3f GB' % (d['nodes_visited'] / max(1, d['rays_traced']), d['prims_tested'] / max(1, d['rays_traced']), alg / 1e9))
    return img

which = sys.argv[1:] or ['c2', 'c3', 'c4']
W, H = 1920, 1080
out = os.path.join(R, 'gpurun_out'); os.makedirs(out, exist_ok=True)
from PIL import Image
if 'c2' in which:
    print('C2 helix 100k, primary only')
    tr = scenes.normalize(scenes.helix_bundle())
    ctx = make_ctx(tr, W, H, 0.002, tfm.standard(), {})
    img = run(ctx, 11, label='c2'); Image.fromarray(img).save(os.path.join(out, 'c2.png'))
if 'c3' in which or 'c4' in which:
    t = time.time(); tr = scenes.normalize(scenes.tornado()); print('tornado gen %.2fs' % (time.time() - t))
if 'c3' in which:
    print('C3 tornado 1M, primary only')
    ctx = make_ctx(tr, W, H, 0.002, tfm.standard(), {})
    img = run(ctx, 11, label='c3-primary'); Image.fromarray(img).save(os.path.join(out, 'c3_primary.png'))
    print('C3 tornado 1M, 64 spp RTAO')
    ctx.set_options({'ambient_occlusion_mode': 'RTAO (Screen Space)', 'ambient_occlusion_strength': 1.0, 'ambient_occlusion_iterations': 1, 'ambient_occlusion_samples_per_frame': 64})
    img = run(ctx, 11, label='c3-rtao'); Image.fromarray(img).save(os.path.join(out, 'c3_rtao.png'))
if 'c4' in which:
    print('C4 tornado 1M, PPLL')
    ctx = make_ctx(tr, W, H, 0.002, tfm.standard_transparent(), {'ppll_max_num_frags': 64, 'ppll_expected_avg_depth_complexity': 20})
    img = run(ctx, 2, label='c4'); Image.fromarray(img).save(os.path.join(out, 'c4.png'))

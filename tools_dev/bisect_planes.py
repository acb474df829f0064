"""Which part of the two-plane step moves the losses away from float64? One whole step of a parity case of
tests/test_model_gpu.py with the arithmetic (dense_conv.PLANES) chosen per module group in the forward pass.
    python tools_dev/bisect_planes.py [second|pp] [seed ...]
"""
import copy
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'tests')]
import test_model_gpu as T                      # noqa: E402
from gga_amd import dense_conv, sparse          # noqa: E402
from gga_amd.cnn import to_channels_last        # noqa: E402

DEV = 'cuda:0'


def groups(model, case):
    g = {}
    if case == 'second':
        me = model.pts_middle_encoder
        g['sp.conv_input'] = me.conv_input
        for n, m in me.encoder_layers.named_children():
            g['sp.' + n] = m
        g['sp.conv_out'] = me.conv_out
    else:
        g['voxel_encoder'] = model.pts_voxel_encoder
        g['middle'] = model.pts_middle_encoder
    for i, b in enumerate(model.pts_backbone.blocks):
        g[f'backbone.{i}'] = b
    g['neck'] = model.pts_neck
    g['head.shared'] = model.pts_bbox_head.shared_conv
    g['head.branches'] = model.pts_bbox_head          # (outer hook: everything of the head that is not shared_conv)
    return g


def run(case, seed, choice, default, fp32_sparse=False):
    cfg, cpu_model, batch, srl, ref, ref64, ref_losses, ref64_losses = T._reference_case(case, seed)
    model = copy.deepcopy(cpu_model)
    model.pts_middle_encoder.channels_last = True
    model.to(DEV)
    to_channels_last(model)
    stack, hooks = [], []
    for name, mod in groups(model, case).items():
        p = choice.get(name, default)

        def pre(m, a, p=p):
            stack.append(dense_conv.PLANES)
            dense_conv.PLANES = p

        def post(m, a, o):
            dense_conv.PLANES = stack.pop()
        hooks += [mod.register_forward_pre_hook(pre), mod.register_forward_hook(post)]
    was, was_split = dense_conv.PLANES, sparse.SPLIT_BF16
    dense_conv.PLANES = default
    sparse.SPLIT_BF16 = not fp32_sparse
    try:
        data = dict(batch, points=[p.to(DEV) for p in batch['points']])
        with torch.no_grad():
            feats = model.extract_feat(data['points'], None, data['img_metas'])[1]
            outs = model.pts_bbox_head(feats)
            losses = model.pts_bbox_head.loss(data['gt_bboxes_3d'], data['gt_labels_3d'], outs, data['GGA_boxes_img'],
                                              data['GGA_lidar2img'], data['GGA_init_pseudo_labels'], data['GGA_bdry_masks'],
                                              data['GGA_in_box_points'], data['img_metas'], srl=srl)
    finally:
        dense_conv.PLANES, sparse.SPLIT_BF16 = was, was_split
        for h in hooks:
            h.remove()
    rel = lambda a, b: abs(a - b) / max(abs(b), 1.0)
    dev = {k: rel(float(losses[k]), ref64_losses[k]) for k in ref64_losses}
    floor = {k: rel(ref_losses[k], ref64_losses[k]) for k in ref64_losses}
    return dev, floor, {k: ref64_losses[k] for k in ref64_losses}


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else 'second'
    seeds = [int(s) for s in sys.argv[2:]] or ([5] if case == 'second' else [1])
    out = {}
    for seed in seeds:
        T._reference_case(case, seed)
        cfg, cpu_model = T._REF_CASES[(case, seed)][:2]
        names = list(groups(cpu_model, case))
        variants = [('all2', {}, 2, False), ('all3', {}, 3, False)]
        if case == 'second':
            variants.append(('all2_sparse_fp32mfma', {}, 2, True))
            variants.append(('sparse3_rest2', {n: 3 for n in names if n.startswith('sp.')}, 2, False))
            variants.append(('sparse2_rest3', {n: 2 for n in names if n.startswith('sp.')}, 3, False))
        variants += [(f'only3:{n}', {n: 3}, 2, False) for n in names]
        for tag, choice, default, f32 in variants:
            dev, floor, vals = run(case, seed, choice, default, f32)
            worst = max(dev, key=dev.get)
            fl = max(floor.values())
            print(f'{case} seed {seed} {tag:32s} worst {dev[worst]:.2e} ({worst}, value {vals[worst]:.4g}); fp32 floor {fl:.2e}; '
                  f'rms over keys {(sum(v * v for v in dev.values()) / len(dev)) ** .5:.2e}', flush=True)
            out[f'{case}:{seed}:{tag}'] = dict(dev=dev, floor=floor)
        T._REF_CASES.pop((case, seed), None)
    os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
    json.dump(out, open(os.path.join(REPO, 'gpurun_out', f'bisect_planes_{case}.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()

"""``SeparateHead`` and ``CenterHead_GGA`` — same registry names, constructor arguments,
``forward`` / ``get_targets`` / ``loss`` signatures and loss-dict keys as the reference
(mmdet3d/models/dense_heads/centerpoint_head.py:18-121,
mmdet3d/models/dense_heads/centerpoint_head_gga.py:19-723).

What differs is *where* the work runs:

* target generation (head:401-627) is one vectorised host pass per step that packs
  every object of every frame/task into flat arrays, one H2D copy, and ONE splat kernel
  for all heat maps — instead of a triple Python loop with >=10 tiny device ops per object;
* the losses (head:629-723) run as the fused HIP kernels of ``gga_amd.functional``
  (focal, gather, box losses with analytic backward) — no ``.item()`` host sync.
"""
import copy

import numpy as np
import os

import torch
from torch import nn

from . import dense_conv
from . import functional as F
from . import ops
from .cnn import ConvModule, build_conv_layer, kaiming_init
from .registry import HEADS, build_bbox_coder, build_head, build_loss

# Semantic Ratio Loss priors N(mu, sigma) per task index (head:514-525): Pedestrian, Cyclist, else Car
SRL_PRIORS = ((1.35, 0.48), (3.60, 0.68), (2.40, 0.28))


def multi_apply(func, *args, **kwargs):
    """mmdet.core.multi_apply: map ``func`` over the zipped args, transpose the results."""
    from functools import partial
    pfunc = partial(func, **kwargs) if kwargs else func
    return tuple(map(list, zip(*map(pfunc, *args))))


@HEADS.register_module()
class SeparateHead(nn.Module):
    def __init__(self, in_channels, heads, head_conv=64, final_kernel=1, init_bias=-2.19,
                 conv_cfg=dict(type='Conv2d'), norm_cfg=dict(type='BN2d'), bias='auto', init_cfg=None, **kwargs):
        assert init_cfg is None, 'To prevent abnormal initialization behavior, init_cfg is not allowed to be set'
        super().__init__()
        self.heads = heads
        self.init_bias = init_bias
        for head in self.heads:
            classes, num_conv = self.heads[head]
            conv_layers = []
            c_in = in_channels
            for _ in range(num_conv - 1):
                conv_layers.append(ConvModule(c_in, head_conv, kernel_size=final_kernel, stride=1,
                                              padding=final_kernel // 2, bias=bias, conv_cfg=conv_cfg,
                                              norm_cfg=norm_cfg))
                c_in = head_conv
            conv_layers.append(build_conv_layer(conv_cfg, head_conv, classes, kernel_size=final_kernel, stride=1,
                                                padding=final_kernel // 2, bias=True))
            self.__setattr__(head, nn.Sequential(*conv_layers))
        self.init_weights()

    def init_weights(self):
        for m in self.modules():      # init_cfg = dict(type='Kaiming', layer='Conv2d')
            if isinstance(m, nn.Conv2d):
                kaiming_init(m)
        for head in self.heads:
            if head == 'heatmap':
                self.__getattr__(head)[-1].bias.data.fill_(self.init_bias)

    def forward(self, x):
        out = {}
        for head in self.heads:
            seq = self.__getattr__(head)
            y = x
            mods = list(seq)
            last = mods[-2] if len(mods) >= 2 else None
            fuse = (isinstance(last, ConvModule) and last.with_norm and last.with_activation
                    and isinstance(last.norm, nn.modules.batchnorm._BatchNorm))
            for m in mods[:-2] if fuse else mods[:-1]:
                y = m(y)
            if fuse:      # norm + ReLU of the last ConvModule applied inside the output conv's loads
                out[head] = F.bn_relu_head_conv3x3(dense_conv.conv2d(y, last.conv, bn_follows=last.norm.training), last.norm, mods[-1])
            else:
                out[head] = F.head_conv3x3(y, mods[-1])     # 1-3 channel output conv: HBM-bound HIP kernel
        return out


def _to_np(x, dtype=None):
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    x = np.asarray(x)
    return x.astype(dtype, copy=False) if dtype is not None else x


def gaussian_radius_np(height, width, min_overlap):
    """mmdet3d/core/utils/gaussian.py:57-86 on f64 arrays."""
    b1 = height + width
    c1 = width * height * (1 - min_overlap) / (1 + min_overlap)
    r1 = (b1 + np.sqrt(b1 ** 2 - 4 * c1)) / 2
    b2 = 2 * (height + width)
    c2 = (1 - min_overlap) * width * height
    r2 = (b2 + np.sqrt(b2 ** 2 - 16 * c2)) / 2
    a3 = 4 * min_overlap
    b3 = -2 * min_overlap * (height + width)
    c3 = (min_overlap - 1) * width * height
    r3 = (b3 + np.sqrt(b3 ** 2 - 4 * a3 * c3)) / 2
    return np.minimum(np.minimum(r1, r2), r3)


@HEADS.register_module()
class CenterHead_GGA(nn.Module):
    def __init__(self, in_channels=[128], tasks=None, train_cfg=None, test_cfg=None, bbox_coder=None,
                 common_heads=dict(), loss_cls=dict(type='GaussianFocalLoss', reduction='mean'),
                 loss_bbox=dict(type='L1Loss', reduction='none', loss_weight=0.25),
                 loss_center=dict(type='MarginL1Loss', reduction='mean'),
                 separate_head=dict(type='SeparateHead', init_bias=-2.19, final_kernel=3),
                 share_conv_channel=64, num_heatmap_convs=2, conv_cfg=dict(type='Conv2d'),
                 norm_cfg=dict(type='BN2d'), bias='auto', norm_bbox=True, init_cfg=None,
                 pal_backprop=False):
        assert init_cfg is None, 'To prevent abnormal initialization behavior, init_cfg is not allowed to be set'
        super().__init__()
        num_classes = [len(t['class_names']) for t in tasks]
        self.class_names = [t['class_names'] for t in tasks]
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.norm_bbox = norm_bbox
        self.loss_cls = build_loss(loss_cls)
        self.loss_bbox = build_loss(loss_bbox)
        self.bbox_coder = build_bbox_coder(bbox_coder) if bbox_coder is not None else None
        self.num_anchor_per_locs = [n for n in num_classes]
        self.fp16_enabled = False
        # Stock mmdet's _parse_losses only back-propagates keys containing 'loss'; the PAL
        # terms are stored as task{i}.distance* (head:697-699), i.e. logged, not trained.
        # pal_backprop=True keeps the key names but lets the detector add them to the total.
        self.pal_backprop = pal_backprop

        self.shared_conv = ConvModule(in_channels, share_conv_channel, kernel_size=3, padding=1,
                                      conv_cfg=conv_cfg, norm_cfg=norm_cfg, bias=bias)
        self.task_heads = nn.ModuleList()
        separate_head = dict(separate_head)
        for num_cls in num_classes:
            heads = copy.deepcopy(common_heads)
            heads.update(dict(heatmap=(num_cls, num_heatmap_convs)))
            separate_head.update(in_channels=share_conv_channel, heads=heads, num_cls=num_cls)
            self.task_heads.append(build_head(separate_head))
        self.with_velocity = 'vel' in common_heads.keys()
        assert not self.with_velocity, 'GGA does not support velocity (head:600-601)'
        assert self.norm_bbox, 'the HIP loss path implements norm_bbox=True (log-dims), as every GGA config uses'
        # The fused loss kernels implement mmdet's reduction='mean' with avg_factor (what configs/gga/* set).
        # The constructor DEFAULT of the reference is loss_bbox reduction='none' (element-wise losses that
        # mmdet's _parse_losses then averages over all B*K*D entries - a different scale): refuse rather
        # than silently compute another number.
        for name in ('loss_cls', 'loss_bbox'):
            red = getattr(getattr(self, name), 'reduction', 'mean')
            if red != 'mean':
                raise NotImplementedError(f"CenterHead_GGA: {name}.reduction={red!r}; the HIP loss path implements "
                                          f"reduction='mean' (configs/gga/gga_kitti_config.py:59-60)")

    # ------------------------------------------------------------------ forward
    def forward_single(self, x):
        x = self.shared_conv(x)
        # training on the device: every branch of every task reads this one map - run them as one autograd node
        # (functional._HeadBranches) so that its gradient is produced once instead of summed branch by branch
        branches, keys = [], []
        for ti, task in enumerate(self.task_heads):
            for head in getattr(task, 'heads', ()):
                mods = list(getattr(task, head))
                if (isinstance(task, SeparateHead) and len(mods) == 2 and isinstance(mods[0], ConvModule) and mods[0].with_norm
                        and mods[0].with_activation and isinstance(mods[0].norm, nn.modules.batchnorm._BatchNorm)
                        and isinstance(mods[0].activate, nn.ReLU)):
                    branches.append((mods[0].conv, mods[0].norm, mods[1]))
                    keys.append((ti, head))
        n_all = sum(len(getattr(task, 'heads', ())) for task in self.task_heads)
        outs = F.head_branches(x, branches) if branches and len(branches) == n_all and self.training else None
        if outs is None:
            return [task(x) for task in self.task_heads]
        ret = [dict() for _ in self.task_heads]
        for (ti, head), y in zip(keys, outs):
            ret[ti][head] = y
        return ret

    def forward(self, feats):
        return multi_apply(self.forward_single, feats)

    # ------------------------------------------------------------------ targets
    def _feature_map_size(self):
        g = self.train_cfg['grid_size']
        osf = self.train_cfg['out_size_factor']
        return int(g[0]) // int(osf), int(g[1]) // int(osf)      # (W, H)

    def draw_srl(self, batch_size):
        """One ``clamp(N(mu_t, sigma_t), 1e-3)`` per (frame, task) from the default CPU torch
        generator, frame-major / task-minor — the same calls in the same order as head:514-525,
        so a seeded run draws the same coefficients as the reference."""
        out = np.zeros((batch_size, len(self.task_heads)), np.float32)
        for b in range(batch_size):
            for t in range(len(self.task_heads)):
                mu, sd = SRL_PRIORS[t] if t < 2 else SRL_PRIORS[2]
                r = torch.normal(torch.tensor(mu), torch.tensor(sd))
                out[b, t] = float(torch.clamp(r, min=1e-3))
        return out

    def pack_targets(self, gt_labels_3d, GGA_boxes_img, GGA_lidar2img, GGA_init_pseudo_labels, GGA_bdry_masks,
                     GGA_in_box_points, img_metas, srl=None):
        """Host half of ``get_targets`` (head:401-627), vectorised per (frame, task): returns
        numpy arrays only, so it is testable without a device."""
        tc = self.train_cfg
        B, T = len(gt_labels_3d), len(self.task_heads)
        K = int(tc['max_objs']) * int(tc['dense_reg'])
        osf = tc['out_size_factor']
        fw, fh = self._feature_map_size()
        vs = np.asarray(tc['voxel_size'], np.float32).astype(np.float64)     # f32 tensors in the reference
        pc = np.asarray(tc['point_cloud_range'], np.float32).astype(np.float64)
        if srl is None:
            srl = self.draw_srl(B)
        ncls = self.num_classes
        map_base = np.concatenate([[0], np.cumsum([B * n for n in ncls])]).astype(np.int64)
        anno = np.zeros((T, B, K, 5), np.float32)
        ind = np.zeros((T, B, K), np.int64)
        mask = np.zeros((T, B, K), np.uint8)
        l2i = np.empty((T, B, K, 4, 4), np.float32)
        bmask = np.zeros((T, B, K, 4), np.uint8)
        # every object of the batch in flat arrays: one vectorised pass instead of B x T small ones
        labs = [_to_np(gt_labels_3d[b]).reshape(-1).astype(np.int64) for b in range(B)]
        n_b = np.asarray([len(x) for x in labs], np.int64)
        start_b = np.concatenate([[0], np.cumsum(n_b)])
        labels = np.concatenate(labs) if B else np.zeros(0, np.int64)
        N = len(labels)
        frame = np.repeat(np.arange(B), n_b)
        local = np.arange(N) - start_b[frame]                                         # index inside its frame
        for b in range(B):
            l2i[:, b] = np.asarray(img_metas[b]['lidar2img'], np.float32)[None, None]     # head:508-509
        cls_task = np.concatenate([np.full(n, t, np.int64) for t, n in enumerate(ncls)])  # class -> task
        cls_in = np.concatenate([np.arange(n) for n in ncls])                             # class -> index in its task
        known = (labels >= 0) & (labels < len(cls_task))
        task = np.where(known, cls_task[np.clip(labels, 0, len(cls_task) - 1)], -1)
        cin = np.where(known, cls_in[np.clip(labels, 0, len(cls_task) - 1)], 0)
        # slot of an object inside its (frame, task): class-major then index order (head:426-434,456-485)
        order = np.lexsort((local, cin, task, frame))
        order = order[known[order]]
        grp = frame[order] * T + task[order]
        first = np.r_[True, grp[1:] != grp[:-1]] if len(order) else np.zeros(0, bool)
        gstart = np.maximum.accumulate(np.where(first, np.arange(len(order)), 0)) if len(order) else np.zeros(0, np.int64)
        slot = np.arange(len(order)) - gstart
        keep = slot < K
        order, slot = order[keep], slot[keep]
        per_task = [[(b, np.zeros(0, np.int64)) for b in range(B)] for _ in range(T)]
        objs = np.zeros((0, 4), np.int64)
        if len(order):
            fo, to = frame[order], task[order]
            bounds = np.flatnonzero(np.r_[True, (fo[1:] != fo[:-1]) | (to[1:] != to[:-1]), True])
            for i0, i1 in zip(bounds[:-1], bounds[1:]):
                per_task[to[i0]][fo[i0]] = (int(fo[i0]), local[order[i0:i1]])
            def cat(xs, shape, dt):
                if getattr(xs, 'flat', None) is not None and getattr(xs, 'inner', None) is None:      # loader.FrameList: already one array
                    return _to_np(xs.flat, dt).reshape(shape)
                return np.concatenate([_to_np(x, dt).reshape(shape) for x in xs]) if N else np.zeros(shape[1:], dt)[None][:0]
            pseudo = cat(GGA_init_pseudo_labels, (-1, 7), np.float64)[order]
            boxes = cat(GGA_boxes_img, (-1, 4), None)[order]
            l2i_o = cat(GGA_lidar2img, (-1, 4, 4), np.float32)[order]
            bdry = cat(GGA_bdry_masks, (-1, 4), None)[order].astype(bool)
            wg = pseudo[:, 3] / vs[0] / osf                                             # head:541-546
            lg = pseudo[:, 4] / vs[1] / osf
            ok = (wg > 0) & (lg > 0)
            with np.errstate(invalid='ignore'):
                rad = gaussian_radius_np(lg, wg, tc['gaussian_overlap'])
            rad = np.where(ok, rad, 0.0)
            rad = np.maximum(int(tc['min_radius']), rad.astype(np.int64))               # max(min_radius, int(r))
            cx = ((pseudo[:, 0] - pc[0]) / vs[0] / osf).astype(np.float32).astype(np.int32)   # f32 then trunc
            cy = ((pseudo[:, 1] - pc[1]) / vs[1] / osf).astype(np.float32).astype(np.int32)
            ok &= (cx >= 0) & (cx < fw) & (cy >= 0) & (cy < fh)                         # head:573-574
            k = np.flatnonzero(ok)
            tk, bk, sk = to[k], fo[k], slot[k]
            ind[tk, bk, sk] = cy[k].astype(np.int64) * fw + cx[k]
            mask[tk, bk, sk] = 1
            l2i[tk, bk, sk] = l2i_o[k]
            bmask[tk, bk, sk] = ~bdry[k]
            anno[tk, bk, sk, :4] = boxes[k].astype(np.float32)
            anno[tk, bk, sk, 4] = np.asarray(srl)[bk, tk]
            ncls_a = np.asarray(ncls, np.int64)
            # the splat list in (frame, task) order, as the per-frame / per-task loops of the reference produce it
            objs = np.stack([map_base[tk] + bk * ncls_a[tk] + cin[order][k], cx[k], cy[k], rad[k]], 1)
        objs = [objs]
        # in-box points (xy as f32, like `.float()` at head:201), packed task-major so each task
        # is one contiguous range of objects
        xy, counts, slots, task_nobj = [], [], [], np.zeros(T, np.int64)
        flat = getattr(GGA_in_box_points, 'flat', None)
        if flat is not None and getattr(GGA_in_box_points, 'inner', None) is not None and list(GGA_in_box_points.inner) == n_b.tolist():
            # the loader's packed hand-over (loader.FrameList): all point sets of the batch are rows of ONE [sum Ni, 4] tensor
            # in frame-major object order - object o = start_b[frame] + local is its o-th set
            if len(order):
                perm = np.argsort(task[order], kind='stable')                        # (task, frame, slot) order
                sel = order[perm]
                sizes = np.asarray(GGA_in_box_points.sizes, np.int64)
                row0 = np.concatenate([[0], np.cumsum(sizes)])
                xy32 = _to_np(flat)[:, :2].astype(np.float32)
                xy = [xy32[row0[o]:row0[o + 1]] for o in sel]
                counts = sizes[sel].tolist()
                slots = (frame[sel] * K + slot[perm]).tolist()
                task_nobj = np.bincount(task[sel], minlength=T).astype(np.int64)
        else:
            for t in range(T):
                for b, sel in per_task[t]:
                    for kk, j in enumerate(sel):
                        p = _to_np(GGA_in_box_points[b][j])
                        xy.append(p[:, :2])
                        counts.append(len(p))
                        slots.append(b * K + kk)
                    task_nobj[t] += len(sel)
        objs = np.concatenate(objs, 0).astype(np.int32) if objs else np.zeros((0, 4), np.int32)
        xy = np.concatenate(xy, 0).astype(np.float32) if xy else np.zeros((0, 2), np.float32)
        offs = np.zeros(len(counts) + 1, np.int32)
        offs[1:] = np.cumsum(counts)
        return dict(B=B, T=T, K=K, fw=fw, fh=fh, map_base=map_base, objs=objs, anno_box=anno, ind=ind, mask=mask,
                    lidar2img=l2i, bound_mask=bmask, ibp_xy=np.ascontiguousarray(xy), ibp_offsets=offs,
                    ibp_slot=np.asarray(slots, np.int32), task_nobj=task_nobj, srl=srl)

    def get_targets(self, gt_bboxes_3d, gt_labels_3d, GGA_boxes_img, GGA_lidar2img, GGA_init_pseudo_labels,
                    GGA_bdry_masks, GGA_in_box_points, img_metas, device=None, srl=None):
        """Returns per-task lists ``heatmaps, anno_boxes, inds, masks, anno_lidar2imgs, ibp_points,
        anno_bound_masks`` like head:343-399. ``ibp_points[t]`` is the packed device form
        ``(xy [n,2] f32, offsets [n_obj+1] i32, slot [n_obj] i32)`` of the in-box points of
        the task's objects instead of a nested list of f64 tensors. ``gt_bboxes_3d`` is debug-only
        in the reference (head:406) and unused."""
        if device is None:
            device = next(self.parameters()).device
        pk = self.pack_targets(gt_labels_3d, GGA_boxes_img, GGA_lidar2img, GGA_init_pseudo_labels,
                               GGA_bdry_masks, GGA_in_box_points, img_metas, srl=srl)
        dev = torch.device(device)
        B, T, fh, fw, ncls, map_base = pk['B'], pk['T'], pk['fh'], pk['fw'], self.num_classes, pk['map_base']
        objs = pk['objs']
        # all nine arrays in one staging buffer and one copy (F.upload_many), queued BEFORE the splat kernel
        up = F.upload_many(dict({k: pk[k] for k in ('anno_box', 'ind', 'mask', 'lidar2img', 'bound_mask', 'ibp_xy', 'ibp_offsets',
                                                    'ibp_slot')}, objs=objs), dev)
        hm = F.heatmap_splat(up['objs'] if dev.type == 'cuda' else objs, int(map_base[-1]), fh, fw, dev,
                             max_radius=max(16, int(objs[:, 3].max()) if len(objs) else 0))
        heatmaps, ibp_points = [], []
        obj_base = np.concatenate([[0], np.cumsum(pk['task_nobj'])])
        for t in range(T):
            heatmaps.append(hm[int(map_base[t]):int(map_base[t + 1])].view(B, ncls[t], fh, fw))
            o0, o1 = int(obj_base[t]), int(obj_base[t + 1])
            # offsets index the shared xy buffer, so a task only needs its slice of offsets/slots
            ibp_points.append((up['ibp_xy'], up['ibp_offsets'][o0:o1 + 1], up['ibp_slot'][o0:o1]))
        return (heatmaps, [up['anno_box'][t] for t in range(T)], [up['ind'][t] for t in range(T)],
                [up['mask'][t] for t in range(T)], [up['lidar2img'][t] for t in range(T)], ibp_points,
                [up['bound_mask'][t] for t in range(T)])

    # ------------------------------------------------------------------ loss
    def loss(self, gt_bboxes_3d, gt_labels_3d, preds_dicts, GGA_boxes_img, GGA_lidar2img, GGA_init_pseudo_labels,
             GGA_bdry_masks, GGA_in_box_points, img_metas, **kwargs):
        device = preds_dicts[0][0]['heatmap'].device
        heatmaps, anno_boxes, inds, masks, anno_lidar2imgs, ibp_points, anno_bound_masks = self.get_targets(
            gt_bboxes_3d, gt_labels_3d, GGA_boxes_img, GGA_lidar2img, GGA_init_pseudo_labels, GGA_bdry_masks,
            GGA_in_box_points, img_metas, device=device, srl=kwargs.get('srl'))
        return self.loss_from_targets(preds_dicts, heatmaps, anno_boxes, inds, masks, anno_lidar2imgs, ibp_points,
                                      anno_bound_masks)

    def loss_from_targets(self, preds_dicts, heatmaps, anno_boxes, inds, masks, anno_lidar2imgs, ibp_points,
                          anno_bound_masks):
        loss_dict = dict()
        tc = self.train_cfg
        for task_id, preds_dict in enumerate(preds_dicts):
            pd = preds_dict[0]
            B, K = inds[task_id].shape
            # heat-map focal loss on the raw logits (clip_sigmoid fused, num_pos stays on device)
            loss_heatmap, _ = F.gaussian_focal_loss(pd['heatmap'], heatmaps[task_id],
                                                    alpha=self.loss_cls.alpha, gamma=self.loss_cls.gamma,
                                                    scale=self.loss_cls.loss_weight * 5.0)
            pred = F.gather_pred(pd['reg'], pd['height'], pd['dim'], pd['rot'], inds[task_id], masks[task_id])
            prm = F.loss_params(B, K, tc, l1_loss_weight=self.loss_bbox.loss_weight)
            xy, offs, slot = ibp_points[task_id]
            (l_bpl, l_srl, l_pmin, l_px, l_py), _ = F.box_loss_terms(pred, inds[task_id], masks[task_id], anno_boxes[task_id],
                                                                     anno_lidar2imgs[task_id], anno_bound_masks[task_id],
                                                                     xy, offs, slot if slot.numel() else None, prm)
            loss_dict[f'task{task_id}.distancex'] = l_px
            loss_dict[f'task{task_id}.distancey'] = l_py
            loss_dict[f'task{task_id}.distancemin'] = l_pmin
            loss_dict[f'task{task_id}.loss_heatmap'] = loss_heatmap
            loss_dict[f'task{task_id}.loss_bbox'] = l_bpl
            loss_dict[f'task{task_id}.loss_ratio'] = l_srl
        return loss_dict

    # ------------------------------------------------------------------ inference
    def get_bboxes(self, preds_dicts, img_metas, img=None, rescale=False):
        """Decode + per-task rotated NMS + merge (head:725-817). ``img_metas[i]['box_type_3d']``
        builds the box structure (default ``LiDARInstance3DBoxes``). Returns
        ``[[bboxes, scores, labels], ...]`` per sample."""
        from .box3d import LiDARInstance3DBoxes
        fast = self._get_bboxes_batched(preds_dicts, img_metas)
        if fast is not None:
            return fast
        rets = []
        for task_id, preds_dict in enumerate(preds_dicts):
            pd = preds_dict[0]
            batch_size = pd['heatmap'].shape[0]
            batch_dim = torch.exp(pd['dim']) if self.norm_bbox else pd['dim']
            temp = self.bbox_coder.decode(pd['heatmap'].sigmoid(), pd['rot'][:, 0].unsqueeze(1),
                                          pd['rot'][:, 1].unsqueeze(1), pd['height'], batch_dim, pd.get('vel'),
                                          reg=pd['reg'], task_id=task_id)
            assert self.test_cfg['nms_type'] in ['circle', 'rotate']
            if self.test_cfg['nms_type'] == 'circle':
                ret_task = []
                for i in range(batch_size):
                    keep = circle_nms(torch.cat([temp[i]['bboxes'][:, :2], temp[i]['scores'].view(-1, 1)], 1),
                                      self.test_cfg['min_radius'][task_id], self.test_cfg['post_max_size'])
                    ret_task.append({k: temp[i][k][keep] for k in ('bboxes', 'scores', 'labels')})
                rets.append(ret_task)
            else:
                rets.append(self.get_task_detections(self.num_classes[task_id], [b['scores'] for b in temp],
                                                     [b['bboxes'] for b in temp], [b['labels'] for b in temp],
                                                     img_metas))
        ret_list = []
        for i in range(len(rets[0])):
            bboxes = torch.cat([ret[i]['bboxes'] for ret in rets])
            bboxes[:, 2] = bboxes[:, 2] - bboxes[:, 5] * 0.5          # gravity centre -> bottom centre
            box_type = (img_metas[i].get('box_type_3d') if isinstance(img_metas[i], dict) else None) or LiDARInstance3DBoxes
            bboxes = box_type(bboxes, self.bbox_coder.code_size)
            scores = torch.cat([ret[i]['scores'] for ret in rets])
            flag, labels = 0, []
            for j, num_class in enumerate(self.num_classes):
                labels.append(rets[j][i]['labels'].int() + flag)
                flag += num_class
            ret_list.append([bboxes, scores, torch.cat(labels)])
        return ret_list

    # Round 6: all frames and tasks in ONE launch behind the coder's batched top-k decode (csrc/postproc.hip::
    # centerpoint_detect_kernel - the same masks, NMS boxes, greedy order, range filter and merge as the loop below, which stays
    # as the path for circle NMS, coders without ``decode_dense``, > 128 candidates per task, and CPU tensors; the two are
    # compared detection by detection in tests/test_postproc_gpu.py). GGA_DETECT_BATCHED=0: off.
    BATCHED = os.environ.get('GGA_DETECT_BATCHED', '1') != '0'

    def _get_bboxes_batched(self, preds_dicts, img_metas):
        from .box3d import LiDARInstance3DBoxes
        tc, coder = self.test_cfg, self.bbox_coder
        heat0 = preds_dicts[0][0]['heatmap']
        if not (self.BATCHED and tc['nms_type'] == 'rotate' and heat0.is_cuda and hasattr(coder, 'decode_dense')
                and coder.max_num <= 128 and coder.post_center_range is not None):
            return None
        boxes, scores, labels = [], [], []
        for task_id, preds_dict in enumerate(preds_dicts):
            pd = preds_dict[0]
            batch_dim = torch.exp(pd['dim']) if self.norm_bbox else pd['dim']
            b, s, l = coder.decode_dense(pd['heatmap'].sigmoid(), pd['rot'][:, 0].unsqueeze(1), pd['rot'][:, 1].unsqueeze(1),
                                         pd['height'], batch_dim, pd.get('vel'), reg=pd['reg'], task_id=task_id)
            boxes.append(b), scores.append(s), labels.append(l)
        if len({tuple(b.shape) for b in boxes}) != 1:
            return None
        out_boxes, out_scores, out_labels, count = F.centerpoint_detect(
            torch.stack(boxes), torch.stack(scores), torch.stack(labels), coder.post_center_range, coder.score_threshold,
            tc['score_threshold'], tc['post_center_limit_range'], tc['nms_thr'], tc['pre_max_size'], tc['post_max_size'],
            self.num_classes)
        counts = count.tolist()                            # the one host read of the batch
        ret_list = _BatchedDetections()
        for i, n in enumerate(counts):
            box_type = (img_metas[i].get('box_type_3d') if isinstance(img_metas[i], dict) else None) or LiDARInstance3DBoxes
            view = out_boxes[i, :n]
            ret_list.append([LiDARInstance3DBoxes.wrap(view, coder.code_size) if box_type is LiDARInstance3DBoxes
                             else box_type(view, coder.code_size), out_scores[i, :n], out_labels[i, :n]])
        ret_list.packed = (out_boxes, out_scores, out_labels, counts)
        return ret_list

    def get_task_detections(self, num_class_with_bg, batch_cls_preds, batch_reg_preds, batch_cls_labels, img_metas):
        """Score threshold -> BEV rotated NMS (HIP) -> range filter, per sample (head:819-934)."""
        from .box3d import LiDARInstance3DBoxes
        tc = self.test_cfg
        pcr = tc['post_center_limit_range']
        if len(pcr) > 0:
            pcr = torch.tensor(pcr, dtype=batch_reg_preds[0].dtype, device=batch_reg_preds[0].device)
        out = []
        for i, (box_preds, cls_preds, cls_labels) in enumerate(zip(batch_reg_preds, batch_cls_preds, batch_cls_labels)):
            top_scores = cls_preds.squeeze(-1)
            top_labels = (torch.zeros(cls_preds.shape[0], device=cls_preds.device, dtype=torch.long)
                          if num_class_with_bg == 1 else cls_labels.long())
            if tc['score_threshold'] > 0.0:
                keep = top_scores >= tc['score_threshold']
                top_scores = top_scores[keep]
                if top_scores.shape[0] != 0:
                    box_preds, top_labels = box_preds[keep], top_labels[keep]
            if top_scores.shape[0] != 0:
                box_type = (img_metas[i].get('box_type_3d') if isinstance(img_metas[i], dict) else None) or LiDARInstance3DBoxes
                boxes_for_nms = ops.xywhr2xyxyr(box_type(box_preds[:, :], self.bbox_coder.code_size).bev)
                selected = ops.nms_bev(boxes_for_nms, top_scores, thresh=tc['nms_thr'], pre_max_size=tc['pre_max_size'],
                                       post_max_size=tc['post_max_size'])
            else:
                selected = []
            sb, sl, ss = box_preds[selected], top_labels[selected], top_scores[selected]
            if sb.shape[0] != 0:
                if len(pcr) > 0:
                    m = (sb[:, :3] >= pcr[:3]).all(1) & (sb[:, :3] <= pcr[3:]).all(1)
                    sb, ss, sl = sb[m], ss[m], sl[m]
                out.append(dict(bboxes=sb, scores=ss, labels=sl))
            else:
                dt, dev = batch_reg_preds[0].dtype, batch_reg_preds[0].device
                out.append(dict(bboxes=torch.zeros([0, self.bbox_coder.code_size], dtype=dt, device=dev),
                                scores=torch.zeros([0], dtype=dt, device=dev),
                                labels=torch.zeros([0], dtype=top_labels.dtype, device=dev)))
        return out


class _BatchedDetections(list):
    """``get_bboxes``' list of [boxes, scores, labels] per frame, whose entries are views of one set of batch tensors
    (``packed`` = (boxes [B, N, D], scores [B, N], labels [B, N], counts)): ``simple_test_pts`` moves those to the host once."""
    packed = None


def circle_nms(dets, thresh, post_max_size=83):
    """Circular NMS (mmdet3d/core/post_processing/box3d_nms.py:181-225): the device op of
    ``gga_amd.ops`` (kept here under the name the head module exports in the reference)."""
    return ops.circle_nms(dets, thresh, post_max_size)

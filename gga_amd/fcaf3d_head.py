"""``FCAF3DHead`` - the anchor-free sparse head of FCAF3D (Rukhovich et al., "FCAF3D: Fully Convolutional Anchor-Free 3D
Object Detection"), written from the paper's description for a whole batch at a time. What is kept from the reference class
(mmdet3d/models/dense_heads/fcaf3d_head.py:49-678) is its interface: the constructor keys of
configs/_base_/models/fcaf3d.py, the parameter names (``up_block_i`` / ``out_block_i`` / ``conv_center`` / ``conv_reg`` /
``conv_cls`` / ``scales.i.scale``), ``forward`` / ``forward_train`` / ``forward_test`` and the three keys of the loss dict.

The algorithm, per the paper (sections 3.2 - 3.4):

* **decoder**: from the coarsest backbone level down, a level is up-sampled by a generative transposed convolution, added to
  the backbone level below (union of the two coordinate sets) and *pruned*: of every scene only the ``pts_prune_threshold``
  locations with the largest class score - the coarser level's score, interpolated at the finer coordinates - survive. Every
  level then goes through one 3x3x3 block and three 1x1x1 heads: centre-ness, box regression (six face distances through
  ``exp`` after a learnable per-level factor, plus two raw angle terms for oriented boxes) and class scores.
* **assignment**: a location is a candidate of a ground-truth box when it lies inside it, on the ONE level chosen for that
  box - the last level, counted from the finest, up to which every level holds at least ``pts_assign_threshold`` locations
  inside the box - and among the ``pts_center_threshold`` candidates with the best centre-ness; a location claimed by
  several boxes goes to the one with the smallest volume.
* **losses**, per scene and averaged over the scenes: focal loss on the class scores of all locations, binary cross-entropy
  on the centre-ness of the positive ones, IoU loss of their decoded boxes weighted by the centre-ness target.
* **inference**: score = sigmoid(class) x sigmoid(centre-ness), the ``nms_pre`` best locations of every level, per-class 3D
  NMS.

Here the scenes of a batch are never looped over in the train path: locations carry a scene index, ground truths sit in one
padded [scene, box] table, pruning and the per-level pre-selection are one segmented top-k, the assignment is one
[scene, location, box] broadcast and the per-scene loss sums are one ``index_add`` each; the two per-scene ``reduce_mean``
collectives of the reference become two vector all-reduces."""
import torch
from torch import nn

from . import mink as ME
from .losses import _EPS32
from .registry import HEADS, build_loss

_NO_BOX = 1e8          # volume standing for "no box claims this location"


def top_per_segment(score, segment, n_segments, k):
    """Boolean mask over ``score`` [n]: the ``k`` largest entries of every segment (``segment`` [n] int64 in [0, n_segments)).
    Ties at the k-th place are broken by position (earlier rows first)."""
    n = score.numel()
    if n == 0 or k <= 0:
        return torch.zeros(n, dtype=torch.bool, device=score.device)
    size = torch.bincount(segment, minlength=n_segments)
    if k >= n:
        return torch.ones(n, dtype=torch.bool, device=score.device)
    by_score = torch.argsort(score, descending=True, stable=True)
    grouped = by_score[torch.argsort(segment[by_score], stable=True)]          # segments in order, best first inside each
    first = torch.cumsum(size, 0) - size
    place = torch.arange(n, device=score.device) - first[segment[grouped]]
    keep = torch.zeros(n, dtype=torch.bool, device=score.device)
    keep[grouped] = place < k
    return keep


def decode_boxes(xyz, reg):
    """Locations ``xyz`` [n, 3] (metres) + regression ``reg`` [n, 6 | 8] -> boxes [n, 6] (centre, size) or [n, 7] (+ heading).
    ``reg`` holds the distances to the lower / upper face along x, y, z; oriented boxes add (sin 2a ln q, cos 2a ln q) with
    q = length / width, so that a box and the same box turned by 90 degrees have one encoding (paper, section 3.3)."""
    if reg.shape[0] == 0:
        return reg
    lower, upper = reg[:, 0:6:2], reg[:, 1:6:2]
    centre = xyz + (upper - lower) / 2
    size = lower + upper
    if reg.shape[1] == 6:
        return torch.cat([centre, size], dim=1)
    s, c = reg[:, 6], reg[:, 7]
    footprint = size[:, 0] + size[:, 1]                       # width + length is what the face distances determine
    q = torch.exp(torch.sqrt(s * s + c * c))
    width = footprint / (1 + q)
    heading = 0.5 * torch.atan2(s, c)
    return torch.stack([centre[:, 0], centre[:, 1], centre[:, 2], width, width * q, size[:, 2], heading], dim=1)


def corner_form(box):
    """(centre, size) [.., 6] -> (min corner, max corner) [.., 6] for the axis-aligned IoU; oriented boxes pass through."""
    if box.shape[-1] != 6:
        return box
    half = box[..., 3:] / 2
    return torch.cat([box[..., :3] - half, box[..., :3] + half], dim=-1)


class LevelOutput:
    """Predictions at the active locations of one decoder level, all scenes together: ``centre`` [n, 1], ``box`` [n, 6 | 8],
    ``cls`` [n, classes] (logits), ``xyz`` [n, 3] metres, ``scene`` [n] int64."""
    __slots__ = ('centre', 'box', 'cls', 'xyz', 'scene')

    def __init__(self, centre, box, cls, xyz, scene):
        self.centre, self.box, self.cls, self.xyz, self.scene = centre, box, cls, xyz, scene

    def per_scene(self, n_scenes):
        """The reference's layout: four lists with one tensor per scene."""
        from .fcaf3d import split_rows
        rows = [torch.nonzero(self.scene == s).squeeze(1) for s in range(n_scenes)]
        return (split_rows(self.centre, rows), split_rows(self.box, rows), split_rows(self.cls, rows), [self.xyz[r] for r in rows])


def _unit(cin, cout, up=False):
    conv = (ME.MinkowskiGenerativeConvolutionTranspose(cin, cout, kernel_size=2, stride=2, dimension=3) if up else
            ME.MinkowskiConvolution(cin, cout, kernel_size=3, dimension=3))
    return [conv, ME.MinkowskiBatchNorm(cout), ME.MinkowskiELU()]


@HEADS.register_module()
class FCAF3DHead(nn.Module):
    def __init__(self, n_classes, in_channels, out_channels, n_reg_outs, voxel_size, pts_prune_threshold, pts_assign_threshold,
                 pts_center_threshold, center_loss=dict(type='CrossEntropyLoss', use_sigmoid=True),
                 bbox_loss=dict(type='AxisAlignedIoULoss'), cls_loss=dict(type='FocalLoss'), train_cfg=None, test_cfg=None,
                 init_cfg=None):
        super().__init__()
        from .config import ConfigDict
        self.voxel_size = voxel_size
        self.pts_prune_threshold = pts_prune_threshold
        self.pts_assign_threshold = pts_assign_threshold
        self.pts_center_threshold = pts_center_threshold
        self.center_loss, self.bbox_loss, self.cls_loss = build_loss(center_loss), build_loss(bbox_loss), build_loss(cls_loss)
        self.train_cfg = train_cfg
        self.test_cfg = ConfigDict(test_cfg) if type(test_cfg) is dict else test_cfg
        self.n_levels = len(in_channels)
        self.pruning = ME.MinkowskiPruning()
        for lvl, width in enumerate(in_channels):
            if lvl:           # level lvl -> lvl - 1: up-sample, then one 3x3x3 block at the finer level's width
                setattr(self, f'up_block_{lvl}', nn.Sequential(*_unit(width, in_channels[lvl - 1], up=True),
                                                               *_unit(in_channels[lvl - 1], in_channels[lvl - 1])))
            setattr(self, f'out_block_{lvl}', nn.Sequential(*_unit(width, out_channels)))
        self.conv_center = ME.MinkowskiConvolution(out_channels, 1, kernel_size=1, dimension=3)
        self.conv_reg = ME.MinkowskiConvolution(out_channels, n_reg_outs, kernel_size=1, dimension=3)
        self.conv_cls = ME.MinkowskiConvolution(out_channels, n_classes, kernel_size=1, bias=True, dimension=3)
        from .fcaf3d import Scale
        self.scales = nn.ModuleList(Scale(1.) for _ in in_channels)

    def init_weights(self):
        from .fcaf3d import bias_init_with_prob
        for conv in (self.conv_center, self.conv_reg, self.conv_cls):
            nn.init.normal_(conv.kernel, std=.01)
        nn.init.constant_(self.conv_cls.bias, bias_init_with_prob(.01))

    # ------------------------------------------------------------------------------------------------------------ decoder
    def _thin_out(self, x, guide):
        """Keep, per scene, the ``pts_prune_threshold`` locations of ``x`` where the coarser level's class score ``guide``
        (a one-channel sparse tensor), interpolated at ``x``'s coordinates, is largest."""
        with torch.no_grad():
            where = x.C.float()
            score = guide.features_at_coordinates(where).squeeze(1)
            keep = top_per_segment(score, x.cmap.coords[:, 0].long(), x.cmap.batch_size, self.pts_prune_threshold)
        return self.pruning(x, keep)

    def _predict(self, y, factor):
        cls = self.conv_cls(y)
        reg = self.conv_reg(y).F
        box = torch.cat([torch.exp(factor(reg[:, :6])), reg[:, 6:]], dim=1)
        where = y.cmap.absolute()
        out = LevelOutput(self.conv_center(y).F, box, cls.F, where[:, 1:] * self.voxel_size, where[:, 0])
        return out, ME.SparseTensor(cls.F.max(dim=1, keepdim=True).values, cmap=cls.cmap)

    def decode(self, feats):
        """Backbone levels (fine -> coarse) -> list of ``LevelOutput`` in the same order."""
        outs = [None] * self.n_levels
        x, guide = feats[-1], None
        for lvl in range(self.n_levels - 1, -1, -1):
            if guide is not None:
                x = self._thin_out(feats[lvl] + getattr(self, f'up_block_{lvl + 1}')(x), guide)
            outs[lvl], guide = self._predict(getattr(self, f'out_block_{lvl}')(x), self.scales[lvl])
        return outs

    def forward(self, x):
        """-> (center_preds, bbox_preds, cls_preds, points): per level a list with one tensor per scene (the reference's
        return value; the train and test paths below work on the batched ``decode`` output instead)."""
        n_scenes = x[0].cmap.batch_size
        per_level = [lvl.per_scene(n_scenes) for lvl in self.decode(x)]
        return tuple([lv[i] for lv in per_level] for i in range(4))

    def forward_train(self, x, gt_bboxes, gt_labels, input_metas):
        return self.loss(self.decode(x), gt_bboxes, gt_labels, len(input_metas))

    def forward_test(self, x, input_metas):
        return self.detect(self.decode(x), input_metas)

    # --------------------------------------------------------------------------------------------------------- assignment
    @staticmethod
    def _box_table(gt_bboxes, gt_labels, device):
        """Ground truths of all scenes as one padded table: boxes [S, G, 7] (gravity centre, size, heading), labels [S, G],
        valid [S, G]."""
        S, G = len(gt_bboxes), max([len(b) for b in gt_bboxes] + [1])
        boxes = torch.zeros((S, G, 7), device=device)
        labels = torch.full((S, G), -1, dtype=torch.long, device=device)
        valid = torch.zeros((S, G), dtype=torch.bool, device=device)
        for s, (b, l) in enumerate(zip(gt_bboxes, gt_labels)):
            if len(b):
                b = b.to(device)
                boxes[s, :len(b)] = torch.cat([b.gravity_center, b.tensor[:, 3:]], dim=1)
                labels[s, :len(b)] = l.to(device)
                valid[s, :len(b)] = True
        return boxes, labels, valid

    @torch.no_grad()
    def assign(self, xyz, level, scene, gt_bboxes, gt_labels):
        """Targets of all locations of a batch. ``xyz`` [n, 3], ``level`` [n] (0 = finest), ``scene`` [n]; ``gt_bboxes`` /
        ``gt_labels``: one box container / label vector per scene. -> centre-ness target [n] (meaningful where a box was
        assigned), box target [n, 7] (or [n, 6] when the ground truth has no heading), class target [n] (-1 = none)."""
        S, n, dev = len(gt_bboxes), xyz.shape[0], xyz.device
        boxes, labels, valid = self._box_table(gt_bboxes, gt_labels, dev)
        with_yaw = all(b.with_yaw for b in gt_bboxes)
        # locations into a padded [scene, slot] layout (slot = position among the scene's locations, input order)
        size = torch.bincount(scene, minlength=S)
        P = max(int(size.max()) if n else 0, 1)
        order = torch.argsort(scene, stable=True)
        slot = torch.empty(n, dtype=torch.long, device=dev)
        slot[order] = torch.arange(n, device=dev) - (torch.cumsum(size, 0) - size)[scene[order]]
        pts = torch.zeros((S, P, 3), device=dev)
        lvl = torch.full((S, P), -1, dtype=torch.long, device=dev)
        real = torch.zeros((S, P), dtype=torch.bool, device=dev)
        pts[scene, slot], lvl[scene, slot], real[scene, slot] = xyz, level, True
        # distances of every location to the six faces of every box of its scene, in the box's own frame: [S, P, G, 6]
        ctr, dims, yaw = boxes[:, None, :, :3], boxes[:, None, :, 3:6], boxes[:, None, :, 6]
        off = pts[:, :, None, :] - ctr
        cos, sin = torch.cos(yaw), torch.sin(yaw)
        local = torch.stack([off[..., 0] * cos + off[..., 1] * sin, off[..., 1] * cos - off[..., 0] * sin, off[..., 2]], dim=-1)
        moved = ctr + local
        faces = torch.stack([moved - ctr + dims / 2, ctr + dims / 2 - moved], dim=-1).flatten(-2)       # x-, x+, y-, y+, z-, z+
        inside = (faces.min(dim=-1).values > 0) & valid[:, None, :] & real[:, :, None]
        # the level of every box: the last one before the first level with too few locations inside
        n_lv = self.n_levels
        per_level = torch.stack([(inside & (lvl == i)[:, :, None]).sum(dim=1) for i in range(n_lv)], dim=1)       # [S, levels, G]
        enough = (per_level >= self.pts_assign_threshold).long()
        chosen = (torch.cumprod(enough, dim=1).sum(dim=1) - 1).clamp(min=0)                           # [S, G]
        candidate = inside & (lvl[:, :, None] == chosen[:, None, :])
        # centre-ness: geometric mean over the axes of (nearer face / farther face)
        near, far = torch.minimum(faces[..., 0::2], faces[..., 1::2]), torch.maximum(faces[..., 0::2], faces[..., 1::2])
        ratio = near[..., 0] / far[..., 0] * near[..., 1] / far[..., 1] * near[..., 2] / far[..., 2]
        cness = torch.where(candidate, torch.sqrt(ratio), ratio.new_tensor(-1.0))
        # per box the best `pts_center_threshold` candidates: strictly above the next one's centre-ness
        ranked = torch.where(real[:, :, None], cness, cness.new_tensor(float('-inf')))
        k = min(self.pts_center_threshold + 1, P)
        top = torch.topk(ranked, k, dim=1).values                                                       # [S, k, G]
        pick = (torch.minimum(size, size.new_tensor(k)) - 1).clamp(min=0)                               # scenes smaller than k
        bar = top.gather(1, pick[:, None, None].expand(S, 1, top.shape[2])).squeeze(1)                  # [S, G]
        chosen_pts = cness > bar[:, None, :]
        # a location claimed by several boxes: the smallest one
        volume = (boxes[..., 3] * boxes[..., 4] * boxes[..., 5])[:, None, :].expand(S, P, -1)
        claim = torch.where(chosen_pts, volume, volume.new_tensor(_NO_BOX))
        smallest, which = claim.min(dim=2)
        s_idx = torch.arange(S, device=dev)[:, None].expand(S, P)
        p_idx = torch.arange(P, device=dev)[None].expand(S, P)
        centre_t = cness[s_idx, p_idx, which]
        box_t = boxes[s_idx, which]
        cls_t = torch.where(smallest == _NO_BOX, labels.new_tensor(-1), labels[s_idx, which])
        if not with_yaw:
            box_t = box_t[..., :6]
        return centre_t[scene, slot], box_t[scene, slot], cls_t[scene, slot]

    # -------------------------------------------------------------------------------------------------------------- loss
    def loss(self, levels, gt_bboxes, gt_labels, n_scenes):
        from .fcaf3d import reduce_mean, take_rows
        centre = torch.cat([l.centre for l in levels])
        box = torch.cat([l.box for l in levels])
        cls = torch.cat([l.cls for l in levels])
        xyz = torch.cat([l.xyz for l in levels])
        scene = torch.cat([l.scene for l in levels])
        level = torch.cat([l.scene.new_full((len(l.scene),), i) for i, l in enumerate(levels)])
        centre_t, box_t, cls_t = self.assign(xyz, level, scene, gt_bboxes, gt_labels)

        def per_scene(values, rows):
            return values.new_zeros(n_scenes).index_add(0, rows, values)

        pos = torch.nonzero(cls_t >= 0).squeeze(1)
        pos_scene = scene[pos]
        # normalisers: positives per scene and their centre-ness mass, each averaged over the ranks
        n_pos = reduce_mean(per_scene(torch.ones(len(pos), device=xyz.device), pos_scene)).clamp(min=1.0)
        mass = reduce_mean(per_scene(centre_t[pos], pos_scene)).clamp(min=1e-6)
        cls_terms = self.cls_loss(cls, cls_t, reduction_override='none')
        cls_loss = per_scene(cls_terms.sum(dim=1) if cls_terms.dim() > 1 else cls_terms, scene) / (n_pos + _EPS32)
        pos_centre, pos_box = take_rows(centre, pos), take_rows(box, pos)
        centre_terms = self.center_loss(pos_centre, centre_t[pos, None], reduction_override='none').squeeze(1)
        centre_loss = per_scene(centre_terms, pos_scene) / (n_pos + _EPS32)
        if len(pos):
            iou_terms = self.bbox_loss(corner_form(decode_boxes(xyz[pos], pos_box)), corner_form(box_t[pos]), reduction_override='none')
            box_loss = per_scene(iou_terms * centre_t[pos], pos_scene) / (mass + _EPS32)
        else:
            box_loss = per_scene(pos_box.sum(dim=1), pos_scene)
        return dict(center_loss=centre_loss.mean(), bbox_loss=box_loss.mean(), cls_loss=cls_loss.mean())

    # ---------------------------------------------------------------------------------------------------------- inference
    def detect(self, levels, input_metas):
        """-> per scene (boxes container, scores, labels)."""
        n_scenes, cfg = len(input_metas), self.test_cfg
        boxes, scores, scenes = [], [], []
        for l in levels:
            score = l.cls.sigmoid() * l.centre.sigmoid()
            if cfg.nms_pre > 0:
                sel = torch.nonzero(top_per_segment(score.max(dim=1).values, l.scene, n_scenes, cfg.nms_pre)).squeeze(1)
                boxes.append(decode_boxes(l.xyz[sel], l.box[sel])), scores.append(score[sel]), scenes.append(l.scene[sel])
            else:
                boxes.append(decode_boxes(l.xyz, l.box)), scores.append(score), scenes.append(l.scene)
        boxes, scores, scenes = torch.cat(boxes), torch.cat(scores), torch.cat(scenes)
        out = []
        for s, meta in enumerate(input_metas):
            rows = torch.nonzero(scenes == s).squeeze(1)
            out.append(self.nms_per_class(boxes[rows], scores[rows], meta))
        return out

    def nms_per_class(self, boxes, scores, input_meta):
        """Greedy 3D NMS class by class over the boxes whose score for that class exceeds ``score_thr``."""
        from .fcaf3d import nms3d, nms3d_normal
        cfg = self.test_cfg
        oriented = boxes.shape[1] == 7
        if not oriented:
            boxes = torch.cat([boxes, boxes.new_zeros(len(boxes), 1)], dim=1)
        suppress = nms3d if oriented else nms3d_normal
        kept = [(boxes.new_zeros((0, 7)), scores.new_zeros(0), torch.zeros(0, dtype=torch.long, device=boxes.device))]
        for c in range(scores.shape[1]):
            rows = torch.nonzero(scores[:, c] > cfg.score_thr).squeeze(1)
            if len(rows):
                rows = rows[suppress(boxes[rows], scores[rows, c], cfg.iou_thr)]
                kept.append((boxes[rows], scores[rows, c], rows.new_full((len(rows),), c)))
        b, s, l = (torch.cat(part) for part in zip(*kept))
        dim = 7 if oriented else 6
        return input_meta['box_type_3d'](b[:, :dim], box_dim=dim, with_yaw=oriented, origin=(.5, .5, .5)), s, l

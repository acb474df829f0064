# PGD retrained on the GGA pseudo labels (reference: configs/gga/gga_pdg.py over configs/_base_/models/pgd.py,
# configs/_base_/models/fcos3d.py, _base_/datasets/kitti-mono3d.py, _base_/schedules/mmdet_schedule_1x.py), with the
# `_base_` chain merged into one file. tests/test_model_cpu.py asserts that its model / optimizer / schedule
# sections equal what the reference's file resolves to.
class_names = ['Pedestrian', 'Cyclist', 'Car']
_branch = (256, )
model = dict(
    type='FCOSMono3D',
    backbone=dict(type='ResNet', depth=101, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=0,
                  norm_cfg=dict(type='BN', requires_grad=False), norm_eval=True, style='caffe',
                  init_cfg=dict(type='Pretrained', checkpoint='open-mmlab://detectron2/resnet101_caffe')),
    neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=0, add_extra_convs='on_output',
              num_outs=4, relu_before_extra_convs=True),
    bbox_head=dict(
        type='PGDHead', num_classes=3, in_channels=256, stacked_convs=2, feat_channels=256, bbox_code_size=7,
        use_direction_classifier=True, diff_rad_by_sin=True, pred_attrs=False, pred_velo=False, pred_bbox2d=True,
        pred_keypoints=True, use_onlyreg_proj=True, dir_offset=0.7854, strides=(4, 8, 16, 32),
        regress_ranges=((-1, 64), (64, 128), (128, 256), (256, 1e8)),
        group_reg_dims=(2, 1, 3, 1, 16, 4),         # offset, depth, size, rot, kpts, bbox2d
        cls_branch=_branch, reg_branch=(_branch, ) * 6, dir_branch=_branch, attr_branch=_branch, centerness_branch=_branch,
        loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
        loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0),
        loss_dir=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
        loss_attr=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
        loss_centerness=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
        norm_on_bbox=True, centerness_on_reg=True, center_sampling=True, conv_bias=True, dcn_on_last_conv=True,
        use_depth_classifier=True, depth_branch=_branch, depth_range=(0, 70), depth_unit=10, division='uniform', depth_bins=8,
        weight_dim=1, loss_depth=dict(type='UncertainSmoothL1Loss', alpha=1.0, beta=3.0, loss_weight=1.0),
        bbox_coder=dict(type='PGDBBoxCoder', base_depths=((28.01, 16.32), ),
                        base_dims=((0.8, 1.73, 0.6), (1.76, 1.73, 0.6), (3.9, 1.56, 1.6)), code_size=7)),
    # 1.0 for the 7 box dims (offset, depth, size, rot), 0.2 for the 16 key-point offsets, 1.0 for the 4 2D distances
    train_cfg=dict(allowed_border=0, code_weight=[1.0] * 7 + [0.2] * 16 + [1.0] * 4, pos_weight=-1, debug=False),
    test_cfg=dict(use_rotate_nms=True, nms_across_levels=False, nms_pre=100, nms_thr=0.05, score_thr=0.001, min_bbox_size=0,
                  max_per_img=20))

# KITTI mono3d data on the GGA pseudo labels (dataset / pipeline classes of the image branch are not built here)
dataset_type = 'KittiMonoDataset'
data_root = 'data/kitti/'
input_modality = dict(use_lidar=False, use_camera=True)
img_norm_cfg = dict(mean=[103.530, 116.280, 123.675], std=[1.0, 1.0, 1.0], to_rgb=False)
data = dict(samples_per_gpu=12, workers_per_gpu=3,
            train=dict(type=dataset_type, data_root=data_root,
                       ann_file=data_root + 'kitti_infos_trainval_GGA_pseudo_mono3d.coco.json',
                       info_file=data_root + 'kitti_infos_trainval_GGA_pseudo.pkl', img_prefix=data_root,
                       classes=class_names, modality=input_modality, test_mode=False, box_type_3d='Camera'))

optimizer = dict(type='SGD', lr=0.001, momentum=0.9, weight_decay=0.0001, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.))
optimizer_config = dict(grad_clip=dict(max_norm=35, norm_type=2))
lr_config = dict(policy='step', warmup='linear', warmup_iters=500, warmup_ratio=1.0 / 3, step=[32, 44])
total_epochs = 48
runner = dict(type='EpochBasedRunner', max_epochs=48)
evaluation = dict(interval=2)
checkpoint_config = dict(interval=8)
log_config = dict(interval=50, hooks=[dict(type='TextLoggerHook'), dict(type='TensorboardLoggerHook')])
dist_params = dict(backend='nccl')

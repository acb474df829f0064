# FCAF3D on SUN RGB-D, 10 classes (reference: configs/fcaf3d/fcaf3d_8x2_sunrgbd-3d-10class.py over
# configs/fcaf3d/fcaf3d_8x2_scannet-3d-18class.py, configs/_base_/models/fcaf3d.py, configs/_base_/default_runtime.py), with the
# `_base_` chain merged into one file. tests/test_model_cpu.py asserts that its model / optimizer / schedule sections equal
# what the reference's file resolves to. BASELINE config 4; there is no GGA head for this trunk in the reference tree.
n_points = 100000
class_names = ('bed', 'table', 'sofa', 'chair', 'toilet', 'desk', 'dresser', 'night_stand', 'bookshelf', 'bathtub')
model = dict(
    type='MinkSingleStage3DDetector',
    voxel_size=.01,
    backbone=dict(type='MinkResNet', in_channels=3, depth=34),
    head=dict(
        type='FCAF3DHead', in_channels=(64, 128, 256, 512), out_channels=128, voxel_size=.01, pts_prune_threshold=100000,
        pts_assign_threshold=27, pts_center_threshold=18, n_classes=10, n_reg_outs=8, bbox_loss=dict(type='RotatedIoU3DLoss')),
    train_cfg=dict(),
    test_cfg=dict(nms_pre=1000, iou_thr=.5, score_thr=.01))

dataset_type = 'SUNRGBDDataset'
data_root = 'data/sunrgbd/'
train_pipeline = [
    dict(type='LoadPointsFromFile', coord_type='DEPTH', shift_height=False, load_dim=6, use_dim=[0, 1, 2, 3, 4, 5]),
    dict(type='LoadAnnotations3D'),
    dict(type='PointSample', num_points=n_points),
    dict(type='RandomFlip3D', sync_2d=False, flip_ratio_bev_horizontal=0.5),
    dict(type='GlobalRotScaleTrans', rot_range=[-0.523599, 0.523599], scale_ratio_range=[0.85, 1.15],
         translation_std=[.1, .1, .1], shift_height=False),
    dict(type='DefaultFormatBundle3D', class_names=class_names),
    dict(type='Collect3D', keys=['points', 'gt_bboxes_3d', 'gt_labels_3d'])]
data = dict(
    samples_per_gpu=8, workers_per_gpu=4,
    train=dict(type='RepeatDataset', times=3,
               dataset=dict(type=dataset_type, modality=dict(use_camera=False, use_lidar=True), data_root=data_root,
                            ann_file=data_root + 'sunrgbd_infos_train.pkl', pipeline=train_pipeline, filter_empty_gt=True,
                            classes=class_names, box_type_3d='Depth')))

optimizer = dict(type='AdamW', lr=0.001, weight_decay=0.0001)
optimizer_config = dict(grad_clip=dict(max_norm=10, norm_type=2))
lr_config = dict(policy='step', warmup=None, step=[8, 11])
runner = dict(type='EpochBasedRunner', max_epochs=12)
custom_hooks = [dict(type='EmptyCacheHook', after_iter=True)]
checkpoint_config = dict(interval=1)
log_config = dict(interval=50, hooks=[dict(type='TextLoggerHook'), dict(type='TensorboardLoggerHook')])
dist_params = dict(backend='nccl')
log_level = 'INFO'
work_dir = None
load_from = None
resume_from = None
workflow = [('train', 1)]

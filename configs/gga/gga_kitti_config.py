# GGA on KITTI, SECOND-style sparse-conv trunk — the model / optimisation settings of the
# reference's configs/gga/gga_kitti_config.py (same keys and values, so the reference file
# itself also loads unchanged through gga_amd.config.Config; see tests/test_model_cpu.py).
# bench.py and the parity tests feed synthetic KITTI-shaped frames (gga_amd/synthetic.py); the train dataset section below is
# the reference's (configs/gga/gga_kitti_config.py:92-205), read by tools/train.py / gga_amd.train.train_detector.
voxel_size = [0.05, 0.05, 0.1]
point_cloud_range = [0, -40, -3, 70.4, 40, 1]
bn = dict(type='BN', eps=1e-3, momentum=0.01)

model = dict(
    type='GGA',
    pts_voxel_layer=dict(max_num_points=5, voxel_size=voxel_size, max_voxels=(16000, 40000),
                         point_cloud_range=point_cloud_range),
    pts_voxel_encoder=dict(type='HardSimpleVFE', num_features=4),
    pts_middle_encoder=dict(
        type='SparseEncoder', in_channels=4, sparse_shape=[41, 1600, 1408], output_channels=128,
        order=('conv', 'norm', 'act'),
        encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
        encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0)), block_type='basicblock'),
    pts_backbone=dict(type='SECOND', in_channels=256, out_channels=[128, 256], layer_nums=[5, 5],
                      layer_strides=[1, 2], norm_cfg=bn, conv_cfg=dict(type='Conv2d', bias=False)),
    pts_neck=dict(type='SECONDFPN', in_channels=[128, 256], out_channels=[256, 256], upsample_strides=[1, 2],
                  norm_cfg=bn, upsample_cfg=dict(type='deconv', bias=False), use_conv_for_no_stride=True),
    pts_bbox_head=dict(
        type='CenterHead_GGA', in_channels=512,
        tasks=[dict(num_class=1, class_names=['Pedestrian']), dict(num_class=1, class_names=['Cyclist']),
               dict(num_class=1, class_names=['Car'])],
        common_heads=dict(reg=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2)),
        share_conv_channel=64,
        bbox_coder=dict(type='CenterPointBBoxCoder', post_center_range=point_cloud_range, max_num=100,
                        score_threshold=0.1, out_size_factor=8, voxel_size=voxel_size[:2], code_size=7,
                        pc_range=point_cloud_range[:2]),
        separate_head=dict(type='SeparateHead', init_bias=-2.19, final_kernel=3),
        loss_cls=dict(type='GaussianFocalLoss', reduction='mean', alpha=0.),
        loss_bbox=dict(type='L1Loss', reduction='mean', loss_weight=0.25),
        loss_center=dict(type='MarginL1Loss', reduction='mean'),
        norm_bbox=True),
    train_cfg=dict(pts=dict(
        point_cloud_range=point_cloud_range, grid_size=[1408, 1600, 40], voxel_size=voxel_size,
        out_size_factor=8, dense_reg=1, gaussian_overlap=0.1, max_objs=500, min_radius=2,
        code_weights=[0.5, 0.5, 0.5, 0.5, 0.5], margin_weights=[1.0, 1.0])),
    test_cfg=dict(pts=dict(
        point_cloud_range=point_cloud_range, post_center_limit_range=point_cloud_range, max_per_img=500,
        max_pool_nms=False, min_radius=[4, 12, 10, 1, 0.85, 0.175], score_threshold=0.1, out_size_factor=4,
        voxel_size=voxel_size[:2], nms_type='rotate', pre_max_size=4096, post_max_size=512, nms_thr=0.2)))

dataset_type = 'KittiDataset_GGA_train'
data_root = 'data/kitti/'
class_names = ['Pedestrian', 'Cyclist', 'Car']
input_modality = dict(use_lidar=True, use_camera=True)
db_sampler = dict(data_root=data_root, info_path=data_root + 'kitti_dbinfos_train_GGA.pkl', rate=1.0,
                  prepare=dict(filter_by_difficulty=[-1], filter_by_min_points=dict(Car=5, Pedestrian=10, Cyclist=10)),
                  classes=class_names, sample_groups=dict(Car=12, Pedestrian=10, Cyclist=10))
train_pipeline = [
    dict(type='LoadPointsFromFile', coord_type='LIDAR', load_dim=4, use_dim=4),
    dict(type='LoadAnnotations3D', with_bbox_3d=True, with_label_3d=True, with_bbox=True, with_gga=True),
    dict(type='ObjectSample_GGA', min_distance=5.0, db_sampler=db_sampler),
    dict(type='PointsRangeFilter', point_cloud_range=point_cloud_range),
    dict(type='ObjectRangeFilter_GGA', point_cloud_range=point_cloud_range, num_points_range=15),
    dict(type='PointShuffle'),
    dict(type='DefaultFormatBundle3D_GGA', class_names=class_names),
    dict(type='Collect3D_GGA', keys=['points', 'gt_bboxes_3d', 'gt_labels_3d', 'GGA_boxes_img', 'GGA_lidar2img',
                                     'GGA_init_pseudo_labels', 'GGA_bdry_masks', 'GGA_in_box_points'])]
data = dict(samples_per_gpu=32, workers_per_gpu=4,
            train=dict(type='RepeatDataset', times=1,
                       dataset=dict(type=dataset_type, data_root=data_root, ann_file=data_root + 'kitti_infos_trainval_GGA.pkl',
                                    split='training', pts_prefix='velodyne_reduced', pipeline=train_pipeline, modality=input_modality,
                                    classes=class_names, test_mode=False, box_type_3d='LiDAR')))
optimizer = dict(type='AdamW', lr=0.0015, betas=(0.95, 0.99), weight_decay=0.01)
optimizer_config = dict(grad_clip=dict(max_norm=35, norm_type=2))
lr_config = dict(policy='cyclic', target_ratio=(10, 1e-4), cyclic_times=1, step_ratio_up=0.4)
momentum_config = dict(policy='cyclic', target_ratio=(0.85 / 0.95, 1), cyclic_times=1, step_ratio_up=0.4)
runner = dict(type='EpochBasedRunner', max_epochs=120)
checkpoint_config = dict(interval=1)
log_config = dict(interval=50, hooks=[dict(type='TextLoggerHook')])
dist_params = dict(backend='nccl')
find_unused_parameters = False
log_level = 'INFO'
work_dir = './work_dirs/kitti_GGA'
load_from = None
resume_from = None
workflow = [('train', 1)]
